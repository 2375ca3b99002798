#!/bin/bash
# Kernel trace of scripts/index_aux_ab.py (config 3: image alone, image + auxiliary index in both aux-block forms).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r04}
rm -rf gpurun_out/${R}_idxaux_trace
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_idxaux_trace -- python3 scripts/index_aux_ab.py > gpurun_out/${R}_idxaux_trace.log 2>&1
echo "rc=$?"
python3 - "$R" <<'PY' | tee gpurun_out/${R}_idxaux_kernel_stats.txt
import csv, glob, sys
R = sys.argv[1]
f = sorted(glob.glob(f"gpurun_out/{R}_idxaux_trace/*/*_kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:30]:
    n = r["Name"]
    n = n.split("(")[0][-78:]
    print(f"{n:80s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e6:8.3f} ms min {float(r['MinNs'])/1e6:8.3f} total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
