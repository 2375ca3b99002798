"""Placement probe under the profiler: one engine, six live output allocations, the emit kernel launched into each in
turn (twice).  Run plain for the event times, or under `rocprofv3 --pmc ...` to compare the counters of the fast and the
slow placements (dispatch order = candidate order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
total = eng.count_paths(2)
bufs = [(torch.empty((total, 3), dtype=torch.int32, device=dev), torch.empty((total, 6), dtype=torch.float64, device=dev)) for _ in range(6)]
for rep in range(2):
    ts = []
    for ids, pde in bufs:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1), 3))
    print("rep", rep, "fill ms per candidate:", ts, "pde ptr mod 2^30:", [hex(p.data_ptr() & ((1 << 30) - 1)) for _, p in bufs], flush=True)
eng.close()
