#!/usr/bin/env python3
"""Round 6, the count phase at config 3: per-step time of vde + count (enqueued, no fill) on the engine's stream,
for the library named by GNNPE_LIB_PATH (default: the shipped one).  With the diagnostic build GNNPE_ROWS_PAIR8=1 makes the row
kernel scatter 8-byte pair records {block, count} instead of 16-byte {block, count, G} (the emit kernel cannot use those: timing
only).  Run by scripts/count_phase_r06.sh under rocprofv3 for kernel times and fabric request counters."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import binding, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2))
eng.vde(want=False)
pair8 = os.environ.get("GNNPE_ROWS_PAIR8") == "1"
total = eng.count_paths(2)
if not pair8:
    assert total == synth.expected_paths_l2(g["offsets"])
for _ in range(3):
    eng.vde(want=False)
    eng.count_paths_enqueue(2)
torch.cuda.synchronize()
ms = []
for _ in range(iters):
    ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    ev0.record()
    eng.vde(want=False)
    ev1.record()
    eng.count_paths_enqueue(2)
    ev2.record()
    torch.cuda.synchronize()
    ms.append((ev0.elapsed_time(ev1), ev1.elapsed_time(ev2)))
v = sorted(x[0] for x in ms)
c = sorted(x[1] for x in ms)
t0 = time.perf_counter()
for _ in range(iters):
    eng.vde(want=False)
    eng.count_paths_enqueue(2)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3 / iters
print(f"lib {os.path.basename(binding.LIB_PATH)} pair8={int(pair8)}: vde median {v[len(v) // 2]:.4f} ms, count median {c[len(c) // 2]:.4f} ms (min {c[0]:.4f}); "
      f"vde + count back to back {wall:.4f} ms per step over {iters} steps", flush=True)
eng.close()
