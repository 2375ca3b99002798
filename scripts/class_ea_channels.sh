cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/tlb_pmc_6
timeout -k 10 280 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL --output-format csv -d gpurun_out/tlb_pmc_6 -- python3 scripts/class_tlb_probe.py 8 > gpurun_out/tlb_pmc_6.log 2>&1 || { echo failed; tail -5 gpurun_out/tlb_pmc_6.log; exit 1; }
grep "^buffer" gpurun_out/tlb_pmc_6.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tlb_pmc_6/*/*_counter_collection.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "k_fill_ranked" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
ds = sorted(per)[-32:]
print("rows per dispatch and counter:", {k: len(v) for k, v in per[ds[0]].items()})
for b in range(8):
    d = ds[4 * b + 2]
    for name, vals in per[d].items():
        v = sorted(vals)
        print(f"buffer {b} {name}: n {len(v)} sum {sum(v):.4g} min {v[0]:.4g} median {v[len(v)//2]:.4g} max {v[-1]:.4g}")
PY
