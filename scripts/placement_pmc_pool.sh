#!/bin/bash
# Write-side counters of EVERY emit-kernel launch of a default bench run (output pool with 8 candidates): the pool's probes
# per candidate, the timed steps into the kept buffer, the plain-allocation steps.  One rocprofv3 --pmc pass.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
R=${1:-r03}
rm -rf gpurun_out/${R}_placement_pmc
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum --output-format csv -d gpurun_out/${R}_placement_pmc -- python3 bench.py --steps 3 --warmup 1 --placements 8 --no-cpu-baseline --no-index --no-config5 > gpurun_out/${R}_placement_pmc.log 2>&1
echo "rc=$?"
python3 - "$R" <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
f = sorted(glob.glob(f"gpurun_out/{R}_placement_pmc/*/*_counter_collection.csv"))[-1]
per = {}
for r in csv.DictReader(open(f)):
    if "k_fill_ranked" in r["Kernel_Name"]:
        d = per.setdefault(int(r["Dispatch_Id"]), dict(grid=int(r.get("Grid_Size", 0) or 0)))
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(per)
gmax = max(per[i]["grid"] for i in ids)
big = [i for i in ids if per[i]["grid"] == gmax]
line = [l for l in open(f"gpurun_out/{R}_placement_pmc.log") if l.startswith('{"metric"')]
pool = json.loads(line[-1])["roofline"]["output_pool"] if line else {}
plan = [(f"pool probe, candidate {k}", 3) for k in range(len(pool.get("candidates_fill_ms") or range(8)))] + [("warm-up step", 1), ("timed steps (kept buffer)", 3), ("plain allocation (1 warm + 3)", 4), ("phase step", 1)]
print("# rocprofv3 --pmc TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum -- python3 bench.py --steps 3 --warmup 1 --placements 8 --no-cpu-baseline --no-index --no-config5")
print("# every launch of k_fill_ranked<2,true,64> at config 3, in launch order; probe times of the same run (under the profiler):", pool.get("candidates_fill_ms"), "kept:", pool.get("kept"))
print("launches,what,DRAM_CREDIT_STALL_mean,WRREQ_STALL_mean,WRREQ_mean")
at = 0
for label, n in plan:
    chunk = big[at:at + n]
    at += n
    if not chunk:
        continue
    m = lambda c: sum(per[i].get(c, 0.0) for i in chunk) / len(chunk)
    print(f"{len(chunk)},{label},{m('TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum'):.4g},{m('TCC_EA0_WRREQ_STALL_sum'):.4g},{m('TCC_EA0_WRREQ_sum'):.5g}")
PY
