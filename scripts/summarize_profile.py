"""Condense the rocprofv3 output of scripts/gpu_profile.sh (gpurun_out/<round>_trace, <round>_pmc_*) into the files kept
under profiles/:
  <round>_kernel_stats.csv  per kernel: calls / average / total, with the launches of the emit kernel split by WHAT launched
                            them (sizing pass, the output pool's probes per candidate, warm-up, the timed steps, the plain-
                            allocation steps, the phase step) -- the timed-step row is the one bench.py's roofline.launch_ms
                            must agree with, and the file alone reproduces the fraction;
  <round>_pmc_fill.json     per-launch counter means of the fill kernel + the derived HBM traffic.

    python scripts/summarize_profile.py r03 [steps=10] [warmup=2] [placements=8]
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
placements = int(sys.argv[4]) if len(sys.argv) > 4 else 8
kernel = "k_fill_ranked"
HBM = 8000.0
PATHS, BPP = 200031576, 92  # config 3: paths per launch, SURVEY 8(d) bytes per path


def newest(pattern):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", pattern))
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    n = name.split("(")[0]
    if "rocprim" in n or "hipcub" in n:
        for key in ("radix_sort_onesweep", "radix_sort_block_sort", "merge_sort", "scan", "transform", "partition", "reduce", "histogram"):
            if key in n:
                return "rocprim " + key
        return "rocprim"
    return n.replace("void ", "")[-72:]


trace = newest(f"{rnd}_trace/*/*_kernel_trace.csv")
if trace:
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    groups = {}
    fills = [r for r in rows if kernel in r["Kernel_Name"]]
    # launch order of the emit kernel in `bench.py --steps S --warmup W` (config 3 on one GPU, device legs only):
    # 1 sizing pass is count-only; pool: 3 launches per candidate; W warm-up; S timed; plain allocation: 1 warm + max(3, S // 2);
    # 1 phase step.  (k_fill_ranked launches of other sizes -- config 2 -- are told apart by their grid.)
    big = [r for r in fills if int(r["Grid_Size_X"]) == max(int(x["Grid_Size_X"]) for x in fills)]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    plan = [(f"pool probe, candidate {k}", 3) for k in range(placements)] if placements > 1 else []
    plan += [("warm-up steps", warm), ("TIMED STEPS", steps)]
    if placements > 1:
        plan += [("plain-allocation steps (1 warm + timed)", 1 + max(3, steps // 2))]
    plan += [("phase step", 1)]
    at = 0
    lines = []
    for label, n in plan:
        chunk = big[at:at + n]
        at += n
        if chunk:
            d = [dur(r) for r in chunk]
            lines.append((f"{kernel}<2,true,64> [{label}]", len(d), sum(d) / len(d), sum(d)))
    rest = big[at:]
    if rest:
        d = [dur(r) for r in rest]
        lines.append((f"{kernel}<2,true,64> [other launches at config 3]", len(d), sum(d) / len(d), sum(d)))
    small = [r for r in fills if r not in big]
    if small:
        d = [dur(r) for r in small]
        lines.append((f"{kernel}<2,true,64> [config 2 leg]", len(d), sum(d) / len(d), sum(d)))
    for r in rows:
        if kernel in r["Kernel_Name"]:
            continue
        groups.setdefault(short(r["Kernel_Name"]), []).append(dur(r))
    for k, d in groups.items():
        lines.append((k, len(d), sum(d) / len(d), sum(d)))
    timed = [l for l in lines if "TIMED STEPS" in l[0]]
    with open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps} --warmup {warm} --no-cpu-baseline --no-index --no-config5 (scripts/gpu_profile.sh)\n")
        f.write("# one row per kernel; the emit kernel's launches are split by what launched them (scripts/summarize_profile.py)\n")
        if timed:
            ms = timed[0][2]
            f.write(f"# timed steps: {ms:.3f} ms per launch -> {PATHS} paths x {BPP} B / {ms:.3f} ms = {PATHS * BPP / ms / 1e6:.0f} GB/s = "
                    f"{PATHS * BPP / ms / 1e6 / HBM:.3f} of the {HBM / 1000:.0f} TB/s spec\n")
        f.write("Name,Calls,AverageMs,TotalMs\n")
        for name, n, avg, tot in sorted(lines, key=lambda l: -l[3]):
            f.write(f'"{name}",{n},{avg:.4f},{tot:.3f}\n')
    print("kernel stats from", trace)
    for l in sorted(lines, key=lambda l: -l[3])[:16]:
        print(f"  {l[0]:70s} calls {l[1]:4d} avg {l[2]:8.3f} ms")

counters = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{rnd}_pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = newest(os.path.relpath(d, os.path.join(ROOT, "gpurun_out")) + "/*/*_counter_collection.csv")
    if not f:
        continue
    per, grid = {}, {}
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            grid[r["Dispatch_Id"]] = int(r.get("Grid_Size", 0) or 0)
    gmax = max(grid.values()) if grid else 0
    for name, by_dispatch in per.items():
        v = [x for k, x in by_dispatch.items() if grid.get(k, gmax) == gmax]  # config-3 launches only
        counters[name] = dict(launches=len(v), mean=sum(v) / len(v), min=min(v), max=max(v))
out = dict(kernel=f"{kernel}<2,true,64>",
           command="python3 bench.py --steps 2 --warmup 1 --placements 1 --no-cpu-baseline --no-index --no-config5 (one rocprofv3 --pmc pass per counter set)",
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
if counters:
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_fill.json"), "w"), indent=1)
for k, v in counters.items():
    print(f"{k:36s} launches={v['launches']} mean={v['mean']:.4g}")
print(out.get("derived"))
