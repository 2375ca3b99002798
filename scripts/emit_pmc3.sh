#!/bin/bash
# Counters of the three emit shapes (k_fill_ranked, k_fill_tiles, k_fill_tickets [+ k_fill_tile_jobs]) at config 3, same process,
# same buffers: one rocprofv3 --pmc pass per counter set over `scripts/emit_ab3.py --quick`.  Arguments: the set numbers (default all);
# EMIT_AB3_ARGS = extra arguments of the script (e.g. "--tpt 8 --occ 3 --shapes tickets").
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "# emit kernels at config 3 (2.0e8 paths, 18.4 GB algorithmic), per-launch means; rocprofv3 --pmc, one pass per counter set; args: $EMIT_AB3_ARGS"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  if [ $# -gt 0 ] && ! echo " $* " | grep -q " $i "; then continue; fi
  rm -rf gpurun_out/emit_pmc3_$i
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/emit_pmc3_$i -- python3 scripts/emit_ab3.py --quick $EMIT_AB3_ARGS > gpurun_out/emit_pmc3_$i.log 2>&1 || { echo "set $i failed"; tail -3 gpurun_out/emit_pmc3_$i.log; exit 1; }
  python3 - "$i" <<'PY'
import csv, glob, sys
i = sys.argv[1]
for f in glob.glob(f"gpurun_out/emit_pmc3_{i}/*/*_counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(f)):
        for k in ("k_fill_ranked", "k_fill_tiles", "k_fill_tickets", "k_fill_tile_jobs"):
            if k + "<" in r["Kernel_Name"]:
                per.setdefault((k, r["Counter_Name"]), {}).setdefault(int(r["Dispatch_Id"]), 0.0)
                per[(k, r["Counter_Name"])][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c), v in sorted(per.items(), key=lambda x: (x[0][1], x[0][0])):
        vals = list(v.values())
        print(f"{c:40s} {k:16s} mean {sum(vals)/len(vals):.5g} over {len(vals)} launches")
PY
done
