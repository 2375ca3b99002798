#!/usr/bin/env python3
"""Pool creation at config-3 size, step by step (GNNPE_POOL_DEBUG=1): stream probe without a count, then with the count."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import binding, synth  # noqa: E402

os.environ["GNNPE_POOL_DEBUG"] = "1"
dev = torch.device("cuda", 0)
stage = sys.argv[1] if len(sys.argv) > 1 else "stream"
n, m = (1_000_000, 10_000_000)
g = synth.gnm_graph(n, m)
sn = synth.degree_order(g["offsets"])
eng = binding.Engine(0)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2))
eng.vde(want=False)
total = synth.expected_paths_l2(g["offsets"])
if stage == "kernel":
    assert eng.count_paths(2) == total
print("stage", stage, "total", total, flush=True)
pool = binding.OutputPool(eng, total, 3, 6, candidates=3)
print(pool.report(), hex(pool.ids), hex(pool.pde), flush=True)
if stage == "stream":
    assert eng.count_paths(2) == total
eng.fill_paths_device(0, total, pool.ids, pool.pde, None)
eng.sync()
ids = pool.ids_tensor(dev)
print("middle sum", int(ids[:total, 1].to(torch.int64).sum()), flush=True)
pool.close()
eng.close()
print("done", stage)
