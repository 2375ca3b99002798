#!/usr/bin/env python3
"""Output pool at config 3: the emit kernel's time in every candidate allocation of the pool
(gnnpe_output_pool_*), next to plain torch allocations; then timed fills into the kept window.
    python scripts/pool_probe.py [candidates=12]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import binding, synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2))
eng.vde(want=False)
total = eng.count_paths(2)


def timed_fill(ids, pde, reps=5):
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.fill_paths_device(0, total, ids, pde, None)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return min(ms[1:]), float(np.median(ms[1:]))


plain = []
keep = []
for _ in range(4):
    ids = torch.empty((total, 3), dtype=torch.int32, device=dev)
    pde = torch.empty((total, 6), dtype=torch.float64, device=dev)
    keep.append((ids, pde))
    plain.append(timed_fill(ids, pde))
print("plain torch allocations, fill ms (min, median):", [(round(a, 3), round(b, 3)) for a, b in plain], flush=True)
print("their addresses:", [(hex(i.data_ptr()), hex(p.data_ptr())) for i, p in keep], flush=True)
del keep, ids, pde
torch.cuda.empty_cache()
t0 = time.perf_counter()
pool = binding.OutputPool(eng, total, 3, 6, candidates=K)
t1 = time.perf_counter()
rep = pool.report()
print(f"pool of {K} windows created in {t1 - t0:.3f} s:", rep, flush=True)
print("kept window: fill ms (min, median)", timed_fill(pool.ids, pool.pde), "at", hex(pool.pde), flush=True)
# bit-exactness of what lands in the pool against a plain buffer
ids = torch.empty((total, 3), dtype=torch.int32, device=dev)
eng.fill_paths_device(0, total, ids, None, None)
torch.cuda.synchronize()
print("ids in the pool equal a plain fill:", bool(torch.equal(pool.ids_tensor(dev)[:total], ids)))
pool.close()
eng.close()
