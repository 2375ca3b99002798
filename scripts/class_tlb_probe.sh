#!/bin/bash
# Round 6: the emit kernel into eight independent allocations, once plain (times) and once per counter set under rocprofv3 --pmc
# (address translation: UTCL1 hits / misses per dispatch, UTCL2 busy cycles) -> profiles/r06_class_tlb.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS"; do
  i=$((i+1))
  rm -rf gpurun_out/tlb_pmc_$i
  timeout -k 10 280 rocprofv3 --pmc $set --output-format csv -d gpurun_out/tlb_pmc_$i -- python3 scripts/class_tlb_probe.py 8 > gpurun_out/tlb_pmc_$i.log 2>&1 || { echo "set $i failed"; tail -3 gpurun_out/tlb_pmc_$i.log; exit 1; }
  echo "## counter set $i: $set"
  grep "^buffer" gpurun_out/tlb_pmc_$i.log
  python3 - "$i" <<'PY'
import csv, glob, sys
i = sys.argv[1]
for f in glob.glob(f"gpurun_out/tlb_pmc_{i}/*/*_counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_fill_ranked" in r["Kernel_Name"]:
            per.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ds = sorted(per)  # (dispatch 0 of the kernel is the count's own calibration-free first fill? no: four per buffer, in buffer order)
    ds = ds[-32:]
    for b in range(len(ds) // 4):
        grp = ds[4 * b + 1:4 * b + 4]
        names = sorted(per[grp[0]])
        print(f"buffer {b}: " + "  ".join(f"{n} {sum(per[d][n] for d in grp) / len(grp):.4g}" for n in names))
PY
done
