#!/bin/bash
# Round profile: kernel-trace stats of the DEFAULT bench command (device legs only) + PMC passes (counters only, separate
# runs, few steps) for the fill kernel.  Summaries: scripts/summarize_profile.py <round>.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r03}
STEPS=${2:-10}
WARM=${3:-2}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-index --no-config5 > gpurun_out/${R}_trace.log 2>&1
echo "trace rc=$?"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${R}_pmc_$i -- python3 bench.py --steps 2 --warmup 1 --placements 1 --no-cpu-baseline --no-index --no-config5 > gpurun_out/${R}_pmc_$i.log 2>&1
  echo "pmc set $i rc=$?"
done
