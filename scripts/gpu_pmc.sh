#!/bin/bash
# PMC passes for the fill kernel of a given variant: separate rocprofv3 runs, counters only.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
V=${1:-3}
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_v${V}_$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --fill-variant $V > gpurun_out/pmc_v${V}_$i.log 2>&1
  echo "set $i rc=$?"
done
ls gpurun_out/pmc_v${V}_*/ 2>/dev/null | head
