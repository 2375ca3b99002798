#!/bin/bash
# counters of the emit kernel per output placement (scripts/placement_pmc.py): one rocprofv3 --pmc pass per set
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/place_pmc_$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/place_pmc_$i -- python3 scripts/placement_pmc.py > gpurun_out/place_pmc_$i.log 2>&1
  echo "set $i rc=$?"; grep "fill ms" gpurun_out/place_pmc_$i.log
  python3 - "$i" <<'PY'
import csv, glob, sys
i = sys.argv[1]
for f in glob.glob(f"gpurun_out/place_pmc_{i}/*/*_counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_fill_ranked" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for c, v in sorted(per.items()):
        print(f"{c:40s}", [f"{v[k]:.4g}" for k in sorted(v)])
PY
done
