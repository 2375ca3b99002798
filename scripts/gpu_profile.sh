#!/bin/bash
# Round profile: kernel-trace stats of the default bench command + PMC passes (counters only) for the fill kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r02}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-index > gpurun_out/${R}_trace.log 2>&1
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${R}_pmc_$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-index > gpurun_out/${R}_pmc_$i.log 2>&1
  echo "pmc set $i rc=$?"
done
