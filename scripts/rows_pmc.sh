#!/bin/bash
# Counters of the count kernel (k_rows_rank) at config 3: one rocprofv3 --pmc pass per set over scripts/count_ab.py with ONE
# round of TWO iterations (counter collection serialises every kernel of the process: round 3 learnt that 30 iterations x 5
# cases cost 13 GPU-minutes; at most four counters of one hardware block per pass, or rocprofv3 refuses the set).
# Optional argument: the numbers of the sets to run (default all), e.g. `scripts/rows_pmc.sh 1 3 6`.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp GNNPE_AB_ROUNDS=1 GNNPE_AB_ITERS=2
mkdir -p gpurun_out
echo "# k_rows_rank<2,true>, config 3 (1M / 10M, 2.0e7 adjacency entries), per-launch means; rocprofv3 --pmc, one pass per counter set"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  if [ $# -gt 0 ] && ! echo " $* " | grep -q " $i "; then continue; fi
  rm -rf gpurun_out/rows_pmc_$i
  timeout 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/rows_pmc_$i -- python3 scripts/count_ab.py "A=1" > gpurun_out/rows_pmc_$i.log 2>&1
  python3 - "$i" <<'PY'
import csv, glob, sys
i = sys.argv[1]
for f in glob.glob(f"gpurun_out/rows_pmc_{i}/*/*_counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_rows_rank" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, v in sorted(per.items()):
        vals = list(v.values())
        print(f"{c:40s} mean {sum(vals)/len(vals):.5g} over {len(vals)} launches")
PY
done
