#!/bin/bash
# Kernel trace of the bench WITH its index legs (no files, no CPU legs): per-kernel averages of the index build.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r03}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_idx_trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-config5 --compare-pool 0 > gpurun_out/${R}_idx_trace.log 2>&1
echo "rc=$?"
python3 - "$R" <<'PY'
import csv, glob, sys
R = sys.argv[1]
f = sorted(glob.glob(f"gpurun_out/{R}_idx_trace/*/*_kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    n = r["Name"]
    n = n.split("(")[0][-70:]
    print(f"{n:72s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e6:8.3f} ms total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
