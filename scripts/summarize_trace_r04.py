"""profiles/<round>_kernel_stats.csv from the kernel trace of the round-4 bench command (scripts/gpu_profile_r04.sh):
    rocprofv3 --kernel-trace --stats -- python3 bench.py --steps S --warmup W --compare-pool 0 --no-cpu-baseline --no-index --no-config5
One row per kernel; the launches of the two emit kernels (k_fill_ranked = start-vertex waves, k_fill_tiles = output tiles) at
config-3 size are split by WHAT launched them, in launch order: the pool's shape calibration (3 + 3), the bench's own
calibration call (3 + 3), W warm-up steps, S TIMED STEPS, the phase step.  The timed-step row is the one bench.py's
roofline.launch_ms must agree with.     python scripts/summarize_trace_r04.py r04 [S=10] [W=2]"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
HBM, PATHS, BPP = 8000.0, 200031576, 92
files = glob.glob(os.path.join(ROOT, "gpurun_out", f"{rnd}_trace/*/*_kernel_trace.csv"))
trace = max(files, key=os.path.getmtime)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
def short(name):
    n = name.split("(")[0].replace("void ", "")
    if "rocprim" in n or "hipcub" in n:
        for key in ("radix_sort_onesweep", "radix_sort_block_sort", "merge_sort", "scan", "transform", "partition", "reduce", "histogram"):
            if key in n:
                return "rocprim " + key
        return "rocprim"
    return n[-72:]
emit = [r for r in rows if "k_fill_ranked" in r["Kernel_Name"] or "k_fill_tiles" in r["Kernel_Name"]]
kind = lambda r: "k_fill_tiles" if "k_fill_tiles" in r["Kernel_Name"] else "k_fill_ranked"
gmax = {k: max([int(r["Grid_Size_X"]) for r in emit if kind(r) == k] or [0]) for k in ("k_fill_ranked", "k_fill_tiles")}
big = [r for r in emit if int(r["Grid_Size_X"]) == gmax[kind(r)]]
plan = [("pool calibration, start-vertex waves", 3), ("pool calibration, output tiles", 3), ("bench calibration, start-vertex waves", 3),
        ("bench calibration, output tiles", 3), ("warm-up steps", warm), ("TIMED STEPS", steps), ("phase step", 1)]
lines, at = [], 0
for label, n in plan:
    chunk = big[at:at + n]
    at += n
    if chunk:
        d = [dur(r) for r in chunk]
        names = sorted({kind(r) for r in chunk})
        lines.append((f"{'/'.join(names)}<2,true> [{label}]", len(d), sum(d) / len(d), sum(d)))
rest = big[at:]
if rest:
    d = [dur(r) for r in rest]
    lines.append((f"emit kernels [other launches at config 3]", len(d), sum(d) / len(d), sum(d)))
small = [r for r in emit if r not in big]
if small:
    d = [dur(r) for r in small]
    lines.append(("emit kernels [other sizes]", len(d), sum(d) / len(d), sum(d)))
groups = {}
for r in rows:
    if r in emit:
        continue
    groups.setdefault(short(r["Kernel_Name"]), []).append(dur(r))
for k, d in groups.items():
    lines.append((k, len(d), sum(d) / len(d), sum(d)))
timed = [l for l in lines if "TIMED STEPS" in l[0]]
cal = {l[0]: l[2] for l in lines if "calibration" in l[0]}
with open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps} --warmup {warm} --compare-pool 0 --no-cpu-baseline --no-index --no-config5 (scripts/gpu_profile_r04.sh)\n")
    f.write("# one row per kernel; the emit kernels' launches at config-3 size are split by what launched them (scripts/summarize_trace_r04.py)\n")
    if timed:
        ms = timed[0][2]
        f.write(f"# timed steps ({timed[0][0].split('<')[0]}): {ms:.3f} ms per launch -> {PATHS} paths x {BPP} B / {ms:.3f} ms = {PATHS * BPP / ms / 1e6:.0f} GB/s = "
                f"{PATHS * BPP / ms / 1e6 / HBM:.3f} of the {HBM / 1000:.0f} TB/s spec (one allocation as it came; the faster emit shape for it)\n")
    try:  # the bench's own event timing of the SAME run (the JSON line the traced command printed)
        import json
        line = [l for l in open(os.path.join(ROOT, "gpurun_out", f"{rnd}_trace.log")) if l.startswith('{"metric"')][-1]
        rf = json.loads(line)["roofline"]
        f.write(f"# bench.py's own timing of the same run: roofline.launch_ms {rf['launch_ms']:.3f} ms, frac {rf['frac']:.3f}, kernel {rf['kernel']}, "
                f"emit_shapes {rf['emit_shapes']['starts_ms']} / {rf['emit_shapes']['tiles_ms']} ms kept {rf['emit_shapes']['kept']}\n")
    except Exception as ex:  # the log is optional
        f.write(f"# (bench line of the traced run not found: {ex})\n")
    f.write("Name,Calls,AverageMs,TotalMs\n")
    for name, n, avg, tot in sorted(lines, key=lambda l: -l[3]):
        f.write(f'"{name}",{n},{avg:.4f},{tot:.3f}\n')
print(open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv")).read()[:4000])
