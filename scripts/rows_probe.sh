#!/bin/bash
# k_rows_rank against its pieces (k_rows_rank_probe<2,true,MODE>: 1 = rank-only G pass with the pair scatter, 2 = the same
# without the scatter, 3 = the payload pass alone) at config 3: kernel times, then fabric / L2 counters per kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/rows_probe_trace
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rows_probe_trace -- python3 scripts/rows_probe.py > gpurun_out/rows_probe_trace.log 2>&1
echo "# k_rows_rank<2,true> and its pieces at config 3 (2.0e7 adjacency entries): per-launch kernel time, then per-launch counter means"
python3 - <<'PY'
import csv, glob, re
per = {}
for f in glob.glob("gpurun_out/rows_probe_trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_rows_rank" in n:
            m = re.search(r"k_rows_rank(_probe|_multi)?<[^>]*>", n)
            per.setdefault(m.group(0), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(per.items()):
    print(f"{k:40s} launches {len(v):2d}  min {min(v):.3f} ms  median {sorted(v)[len(v) // 2]:.3f} ms")
PY
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/rows_probe_pmc_$i
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/rows_probe_pmc_$i -- python3 scripts/rows_probe.py > gpurun_out/rows_probe_pmc_$i.log 2>&1
  python3 - "$i" <<'PY'
import csv, glob, re, sys
i = sys.argv[1]
per = {}
for f in glob.glob(f"gpurun_out/rows_probe_pmc_{i}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_rows_rank" in n:
            k = re.search(r"k_rows_rank(_probe|_multi)?<[^>]*>", n).group(0)
            per.setdefault((r["Counter_Name"], k), {}).setdefault(r["Dispatch_Id"], 0.0)
            per[(r["Counter_Name"], k)][r["Dispatch_Id"]] += float(r["Counter_Value"])
for (c, k), v in sorted(per.items()):
    vals = list(v.values())
    print(f"{c:36s} {k:40s} mean {sum(vals) / len(vals):.4g} over {len(vals)} launches")
PY
done
