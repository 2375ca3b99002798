#!/bin/bash
# Round 6, the index builds of both path lengths in one process (scripts/index_l3_bench.py --trace: l = 2 pair-major at config 3, l = 3
# triple-major at e = 2 and e = 8): kernel trace -> profiles/r06_index_kernel_stats.csv, then the leaf kernels' fabric counters (one
# rocprofv3 --pmc pass per counter set) -> profiles/r06_leaf_mem_pmc.txt.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/idx_r06_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/idx_r06_trace -- python3 scripts/index_l3_bench.py --trace > gpurun_out/idx_r06_trace.log 2>&1 || { echo "trace failed"; tail -5 gpurun_out/idx_r06_trace.log; exit 1; }
grep -v amdgpu.ids gpurun_out/idx_r06_trace.log | grep "l=\|per byte" 
python3 - <<'PY' > gpurun_out/r06_index_kernel_stats.csv
import csv, glob, collections
f = glob.glob("gpurun_out/idx_r06_trace/*/*_kernel_trace.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    n = "rocprim (radix sort / scan passes)" if "rocprim" in n else n.split("(")[0].replace("void ", "")
    agg.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("# rocprofv3 --kernel-trace --stats -- python3 scripts/index_l3_bench.py --trace (scripts/gpu_index_r06.sh): index builds of l = 2 (config 3, pair-major)")
print("# and l = 3 (G(70K, 600K) e = 2; G(40K, 220K) e = 8: triple-major), two counts x two builds each; per-launch times in ms")
print("Name,Calls,MinMs,AverageMs,MaxMs,TotalMs")
for n, v in sorted(agg.items(), key=lambda x: -sum(x[1])):
    if sum(v) >= 0.02:
        print(f'"{n}",{len(v)},{min(v):.4f},{sum(v)/len(v):.4f},{max(v):.4f},{sum(v):.3f}')
PY
head -30 gpurun_out/r06_index_kernel_stats.csv
{
echo "# leaf kernels of the index builds, fabric requests per launch (rocprofv3 --pmc, one pass per counter set over scripts/index_l3_bench.py --trace;"
echo "# scripts/gpu_index_r06.sh).  Bytes: 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B read; 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B) written."
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/idx_r06_pmc_$i
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/idx_r06_pmc_$i -- python3 scripts/index_l3_bench.py --trace > gpurun_out/idx_r06_pmc_$i.log 2>&1 || { echo "set $i failed"; tail -3 gpurun_out/idx_r06_pmc_$i.log; exit 1; }
  python3 - "$i" <<'PY'
import csv, glob, sys
i = sys.argv[1]
for f in glob.glob(f"gpurun_out/idx_r06_pmc_{i}/*/*_counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(f)):
        for k in ("k_pack_leaves_pairs", "k_tx_leaves<2>", "k_tx_leaves<8>", "k_tx_units", "k_tx_inner<8>", "k_tx_gather"):
            if k in r["Kernel_Name"]:
                per.setdefault((k, r["Counter_Name"]), {}).setdefault(int(r["Dispatch_Id"]), 0.0)
                per[(k, r["Counter_Name"])][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c), v in sorted(per.items(), key=lambda x: (x[0][0], x[0][1])):
        vals = list(v.values())
        print(f"{k:22s} {c:28s} mean {sum(vals)/len(vals):.5g} max {max(vals):.5g} over {len(vals)} launches")
PY
done
} > gpurun_out/r06_leaf_mem_pmc.txt
cat gpurun_out/r06_leaf_mem_pmc.txt
