cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
i=2
for set in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/tlb_pmc_$i
  timeout -k 10 280 rocprofv3 --pmc $set --output-format csv -d gpurun_out/tlb_pmc_$i -- python3 scripts/class_tlb_probe.py 8 > gpurun_out/tlb_pmc_$i.log 2>&1 || { echo "set $i failed"; tail -3 gpurun_out/tlb_pmc_$i.log; continue; }
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
    ds = sorted(per)
    print("k_fill_ranked dispatches:", len(ds))
    ds = ds[-32:]
    for b in range(len(ds) // 4):
        grp = ds[4 * b + 1:4 * b + 4]
        names = sorted(per[grp[0]])
        print(f"buffer {b}: " + "  ".join(f"{n.replace('TCC_EA0_','').replace('_sum','')} {sum(per[d][n] for d in grp) / len(grp):.4g}" for n in names))
PY
done
