"""profiles/<round>_kernel_stats.csv from the kernel trace of the round-5 bench command (scripts/gpu_profile_r06.sh):
    rocprofv3 --kernel-trace --stats -- python3 bench.py --steps S --warmup W --compare-pool 0 --no-cpu-baseline --no-index --no-config5
One row per kernel.  The emit kernels' launches at config-3 size are split by WHAT launched them, from the end of the run backwards: the
phase step (last), the S TIMED STEPS, the W warm-up steps; everything before that is calibration (the pool's and the bench's own call of
gnnpe_emit_calibrate_device: shapes 1 / 4 / 2, three launches each), grouped by kernel and grid (k_fill_ranked: 1280 workgroups = five per
CU, 768 = three per CU).  The TIMED STEPS row is the one bench.py's roofline.launch_ms must agree with; the calibration rows are the
other shapes into the SAME buffer.     python scripts/summarize_trace_r06.py r05 [S=10] [W=2]"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
HBM, PATHS, BPP = 8000.0, 200031576, 92
files = glob.glob(os.path.join(ROOT, "gpurun_out", f"{rnd}_trace/*/*_kernel_trace.csv"))
trace = max(files, key=os.path.getmtime)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
KERNELS = ("k_fill_ranked", "k_fill_tiles", "k_fill_tickets", "k_fill_tile_jobs")


def kind(r):
    for k in KERNELS:
        if k + "<" in r["Kernel_Name"]:
            return k
    return None


def short(name):
    n = name.split("(")[0].replace("void ", "")
    if "rocprim" in n or "hipcub" in n:
        for key in ("radix_sort_onesweep", "radix_sort_block_sort", "merge_sort", "scan", "transform", "partition", "reduce", "histogram"):
            if key in n:
                return "rocprim " + key
        return "rocprim"
    return n[-72:]


emit = [r for r in rows if kind(r) in ("k_fill_ranked", "k_fill_tiles", "k_fill_tickets")]
# config-3 launches: the long ones (a launch over 2.0e8 paths takes milliseconds; everything else here is microseconds)
big = [r for r in emit if dur(r) > 1.5]
wg = lambda r: int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
label = lambda r: f"{kind(r)} ({wg(r)} workgroups)" if kind(r) == "k_fill_ranked" else kind(r)
lines = []
phase, timed, warmup, cal = big[-1:], big[-1 - steps:-1], big[-1 - steps - warm:-1 - steps], big[:-1 - steps - warm]
for name, chunk in (("TIMED STEPS", timed), ("warm-up steps", warmup), ("phase step", phase)):
    if chunk:
        d = [dur(r) for r in chunk]
        lines.append((f"{'/'.join(sorted({label(r) for r in chunk}))} [{name}]", len(d), sum(d) / len(d), sum(d)))
groups = {}
for r in cal:
    groups.setdefault(label(r), []).append(dur(r))
for k, d in groups.items():
    best = sorted(d)[:max(1, len(d) * 2 // 3)]  # (the first launch of every shape touches the pages / builds the tile table)
    lines.append((f"{k} [calibration launches into the same buffer; mean of the fastest two thirds {sum(best) / len(best):.3f} ms]", len(d), sum(d) / len(d), sum(d)))
small = [r for r in emit if r not in big]
if small:
    d = [dur(r) for r in small]
    lines.append(("emit kernels [other sizes]", len(d), sum(d) / len(d), sum(d)))
others = {}
for r in rows:
    if r in emit:
        continue
    others.setdefault(short(r["Kernel_Name"]), []).append(dur(r))
for k, d in others.items():
    lines.append((k, len(d), sum(d) / len(d), sum(d)))
with open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps {steps} --warmup {warm} --compare-pool 0 --no-cpu-baseline --no-index --no-config5 (scripts/gpu_profile_r06.sh)\n")
    f.write("# one row per kernel; the emit kernels' launches at config-3 size are split by what launched them (scripts/summarize_trace_r06.py)\n")
    if timed:
        ms = sum(dur(r) for r in timed) / len(timed)
        f.write(f"# timed steps ({label(timed[0])}): {ms:.3f} ms per launch -> {PATHS} paths x {BPP} B / {ms:.3f} ms = {PATHS * BPP / ms / 1e6:.0f} GB/s = "
                f"{PATHS * BPP / ms / 1e6 / HBM:.3f} of the {HBM / 1000:.0f} TB/s spec (one allocation as it came; the fastest of three emit launches for it)\n")
    try:  # the bench's own event timing of the SAME run (the JSON line the traced command printed)
        line = [l for l in open(os.path.join(ROOT, "gpurun_out", f"{rnd}_trace.log")) if l.startswith('{"metric"')][-1]
        d = json.loads(line)
        rf = d["roofline"]
        f.write(f"# bench.py's own timing of the same run: ms_per_step {d['ms_per_step']:.3f}, roofline.launch_ms {rf['launch_ms']:.3f} ms, frac {rf['frac']:.3f}, "
                f"kernel {rf['kernel']}, emit_shapes starts {rf['emit_shapes']['starts_ms']} / starts_low {rf['emit_shapes']['starts_low_ms']} / tiles "
                f"{rf['emit_shapes']['tiles_ms']} ms, kept {rf['emit_shapes']['kept']}\n")
    except Exception as ex:  # the log is optional
        f.write(f"# (bench line of the traced run not found: {ex})\n")
    f.write("Name,Calls,AverageMs,TotalMs\n")
    for name, n, avg, tot in sorted(lines, key=lambda l: -l[3]):
        f.write(f'"{name}",{n},{avg:.4f},{tot:.3f}\n')
print(open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.csv")).read()[:5000])
