"""Condense the rocprofv3 output of scripts/gpu_profile.sh (gpurun_out/<round>_trace, <round>_pmc_*) into the two
files kept under profiles/: <round>_kernel_stats.csv (verbatim --stats table) and <round>_pmc_fill.json (per-launch
counter means of the fill kernel + the derived HBM traffic bracket bench.py reports as roofline.traffic).

    python scripts/summarize_profile.py r01
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
kernel = "k_fill_ranked"


def newest(pattern):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", pattern))
    return max(files, key=os.path.getmtime) if files else None


stats = newest(f"{rnd}_trace/*/*_kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv"))
counters = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{rnd}_pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = newest(os.path.relpath(d, os.path.join(ROOT, "gpurun_out")) + "/*/*_counter_collection.csv")
    if not f:
        continue
    per = {}
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for name, by_dispatch in per.items():
        v = list(by_dispatch.values())
        counters[name] = dict(launches=len(v), mean=sum(v) / len(v), min=min(v), max=max(v))
out = dict(kernel=f"{kernel}<2>",
           command="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-index (one rocprofv3 --pmc pass per counter set)",
           counters=counters)
if "WRITE_SIZE" in counters:
    wb = counters["WRITE_SIZE"]["mean"] * 1024
    d = dict(write_bytes=wb)
    if "TCC_EA0_RDREQ_sum" in counters:
        # fabric read requests by size: what the L2 actually fetched (FETCH_SIZE tallies every request at 64 bytes,
        # MI355X_MICROARCH.md section HBM; here nearly all of them are 128-byte line fills)
        r128 = counters.get("TCC_EA0_RDREQ_128B_sum", {}).get("mean", 0.0)
        r64 = counters.get("TCC_EA0_RDREQ_64B_sum", {}).get("mean", 0.0)
        r32 = counters.get("TCC_EA0_RDREQ_32B_sum", {}).get("mean", 0.0)
        rb = r128 * 128 + r64 * 64 + r32 * 32
        d.update(read_bytes=rb, traffic_bytes=wb + rb,
                 note="WRITE_SIZE is in KiB; read bytes = TCC_EA0_RDREQ_{32,64,128}B x their sizes (one number, no bracket)")
    if "FETCH_SIZE" in counters:
        d["fetch_size_raw_bytes"] = counters["FETCH_SIZE"]["mean"] * 1024
    out["derived"] = d
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_fill.json"), "w"), indent=1)
print("stats from", stats)
for k, v in counters.items():
    print(f"{k:36s} launches={v['launches']} mean={v['mean']:.4g}")
print(out.get("derived"))
