#!/usr/bin/env python3
"""Config 5 (power-law 4M / 64M, l = 3, e = 8) emission on the three sampled 2^26-path ranges of bench.py's config5 leg, three calls
each -- the command behind profiles/r06_deep_emit_kernel_stats.csv:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_deep -- python3 scripts/deep_emit_profile.py
Prints the calls' event times (the bench's figure: 352 algorithmic bytes per path against 8 TB/s)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import binding, synth  # noqa: E402

L, e = 4, 8
log2_chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 26
t0 = time.time()
g = synth.powerlaw_graph(4_000_000, 64_000_000, exponent=2.1, max_degree=3000, n_labels=64, seed=synth.SEED)
sn = synth.degree_order(g["offsets"])
print(f"graph {time.time() - t0:.0f} s", flush=True)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, e))
eng.vde(want=False)
total = eng.count_paths(3)
torch.cuda.synchronize()
t0 = time.time()
eng.vde(want=False)
assert eng.count_paths(3) == total
torch.cuda.synchronize()
print(f"paths {total}, vde + count {time.time() - t0:.3f} s", flush=True)
chunk = 1 << log2_chunk
ids = torch.empty((chunk, L), dtype=torch.int32, device=dev)
pde = torch.empty((chunk, L * e), dtype=torch.float64, device=dev)
bpp = 4 * L + 8 * e * L + 16 + 8 * e
for frac_at in (0.0, 0.37, 0.81):
    b = min(int(total * frac_at), total - chunk)
    ms = []
    for _ in range(4):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        eng.fill_paths_device(b, b + chunk, ids, pde, None)
        ev1.record()
        torch.cuda.synchronize()
        ms.append(ev0.elapsed_time(ev1))
    m = min(ms[1:])
    print(f"range at {frac_at}: first path {b}, calls {[round(x, 3) for x in ms]} ms, best {m:.3f} ms = {chunk * bpp / (m / 1e3) / 8e12:.3f} of spec; "
          f"checksum {eng.rows_checksum_device(chunk, L, ids, first_id=b):#x}", flush=True)
eng.close()
