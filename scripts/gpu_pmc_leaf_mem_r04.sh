#!/bin/bash
# Memory-side PMC passes (counters only) of the pair-major leaf kernel at config 3, image alone and with the auxiliary rows
# (both aux-block forms): one rocprofv3 --pmc pass per counter set over scripts/index_aux_ab.py; per-launch means by instantiation.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
export R=${1:-r04leafmem}
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum TCC_HIT_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf gpurun_out/${R}_$i
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${R}_$i -- python3 scripts/index_aux_ab.py > gpurun_out/${R}_$i.log 2>&1
  echo "pmc set $i rc=$?"
done
python3 - <<'PY' | tee gpurun_out/${R}_summary.txt
import csv, glob, os, re
R = os.environ["R"]
print("# per-launch means by instantiation of k_pack_leaves_pairs<E, PACKED, AUX, NL> at config 3 (5.13e6 leaves of 39 entries; AUX 0 = image alone,")
print("# 2 = with the auxiliary rows, {degree, label} inside the records' id bits, 1 = as 8 bytes behind every record); rocprofv3 --pmc (counters only)")
print("# passes of `python3 scripts/index_aux_ab.py` (scripts/gpu_pmc_leaf_mem_r04.sh)")
for d in sorted(glob.glob(f"gpurun_out/{R}_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        per = {}
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_pack_leaves_pairs<([^>]*)>", r["Kernel_Name"])
            if m:
                per.setdefault((m.group(1), r["Counter_Name"]), {}).setdefault(r["Dispatch_Id"], 0.0)
                per[(m.group(1), r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (k, c), v in sorted(per.items()):
            vals = list(v.values())
            print(f"<{k:14s}> {c:34s} mean {sum(vals)/len(vals):.5g} over {len(vals)} launches")
PY
