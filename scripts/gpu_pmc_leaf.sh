#!/bin/bash
# SQ-side PMC passes (counters only) of the pair-major leaf kernel: instruction mix, issue/wait cycles, LDS conflicts.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
export R=${1:-r02leaf}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${R}_$i -- python3 scripts/index_ab.py "" > gpurun_out/${R}_$i.log 2>&1
  echo "pmc set $i rc=$?"
done
python3 - <<'PY'
import csv, glob, os
R = os.environ["R"]
print("# per-launch means, k_pack_leaves_pairs at config 3 (5.26e6 leaves); rocprofv3 --pmc (counters only) of `python3 scripts/index_ab.py`")
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
