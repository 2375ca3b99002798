#!/bin/bash
# Memory-side PMC passes (counters only) of the pair-major leaf kernel at config 3: requests from the vector L1s, fabric
# reads, L2 hit rate, and where the wave's memory instructions queue.  One rocprofv3 --pmc pass per set over scripts/index_ab.py.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
export R=${1:-r03leafmem}
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum TCC_HIT_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf gpurun_out/${R}_$i
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${R}_$i -- python3 scripts/index_ab.py "" > gpurun_out/${R}_$i.log 2>&1
  echo "pmc set $i rc=$?"
done
python3 - <<'PY'
import csv, glob, os
R = os.environ["R"]
print("# per-launch means, k_pack_leaves_pairs at config 3 (5.13e6 leaves of 39 entries, image alone); rocprofv3 --pmc (counters only) of `python3 scripts/index_ab.py`")
for d in sorted(glob.glob(f"gpurun_out/{R}_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        per = {}
        for r in csv.DictReader(open(f)):
            if "k_pack_leaves_pairs" in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for c, v in sorted(per.items()):
            vals = list(v.values())
            print(f"{c:34s} mean {sum(vals)/len(vals):.5g} over {len(vals)} launches")
PY
