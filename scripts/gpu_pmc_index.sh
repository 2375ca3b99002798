#!/bin/bash
# PMC passes (counters only) for the index-build kernels: bench with the index build left in, one pass per counter set.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
export R=${1:-r02}
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${R}_pmcidx_$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > gpurun_out/${R}_pmcidx_$i.log 2>&1
  echo "pmc set $i rc=$?"
done
python3 - <<'PY'
import csv, glob, os
R = os.environ.get("R", "r02")
print("# per-launch means; rocprofv3 --pmc passes (counters only, one set per run) of `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e`, config 3")
for d in sorted(glob.glob(f"gpurun_out/{R}_pmcidx_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        per = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(t in k for t in ("k_pack_leaves_pairs", "k_pack_leaves_paths", "k_px_pairs", "k_px_permute", "k_rows_rank")):
                key = (k.split("(")[0][-40:], r["Counter_Name"])
                per.setdefault(key, {}).setdefault(r["Dispatch_Id"], 0.0)
                per[key][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, c), v in sorted(per.items()):
            vals = list(v.values())
            print(f"{k:42s} {c:32s} mean {sum(vals)/len(vals):.4g} over {len(vals)} launches")
PY
