#!/usr/bin/env python3
"""Round 6: the l = 3 index build (triple-major, gnnpe_index_deep.hip.h) timed per byte of image, beside the l = 2 pair-major
build of config 3 in the same process ("l = 3 index build <= 1.5 x the l = 2 build per byte").
usage: python scripts/index_l3_bench.py [--trace]      (--trace: fewer repeats, for rocprofv3 --kernel-trace)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth

reps = 2 if "--trace" in sys.argv else 4
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)


def run(name, n, m, l, e):
    g = synth.gnm_graph(n, m)
    sn = synth.degree_order(g["offsets"])
    eng = binding.Engine(0, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(64, e)); eng.vde(want=False)
    full, cached = [], []
    nbytes = total = 0
    for _ in range(reps):
        total = eng.count_paths(l)
        for out in (full, cached):  # the first build of a count sorts the units, the second reuses the order
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(); _, nbytes, hdr = eng.build_index_partition_device(0); ev1.record(); torch.cuda.synchronize()
            out.append((ev0.elapsed_time(ev1), (time.perf_counter() - t0) * 1e3))
    gb = nbytes / 1e9
    f, c = min(x[0] for x in full[1:]), min(x[0] for x in cached[1:])
    print(f"{name}: G({n}, {m}) l={l} e={e}: {total} paths, image {gb:.2f} GB; whole build {f:.2f} ms = {f / gb:.3f} ms/GB "
          f"(host clock {min(x[1] for x in full[1:]):.2f}), from the cached unit order {c:.2f} ms = {c / gb:.3f} ms/GB; first call {full[0][1]:.0f} ms", flush=True)
    eng.close()
    return f / gb, c / gb


base = run("l=2 pair-major (config 3)", 1_000_000, 10_000_000, 2, 2)
for name, n, m, e in (("l=3 triple-major, e=2", 70_000, 600_000, 2), ("l=3 triple-major, e=8", 40_000, 220_000, 8)):
    r = run(name, n, m, 3, e)
    print(f"    per byte against l = 2: whole build x{r[0] / base[0]:.2f}, cached order x{r[1] / base[1]:.2f}", flush=True)
