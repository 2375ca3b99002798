"""Does the emit kernel's time depend on WHERE its output buffers landed?  Same process, same records, fresh output
allocations each round (torch caching allocator emptied in between)."""
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
def timed(ids, pde, reps=5):
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for rnd in range(6):
    ids = torch.empty((total, 3), dtype=torch.int32, device=dev); pde = torch.empty((total, 6), dtype=torch.float64, device=dev)
    t = timed(ids, pde)
    eng.count_paths(2)  # rebuild the records in place (same buffers)
    t2 = timed(ids, pde)
    print(f"round {rnd}: ids@{ids.data_ptr():#x} pde@{pde.data_ptr():#x} fill {t:.3f} ms, after recount {t2:.3f} ms")
    del ids, pde; torch.cuda.empty_cache()
    if rnd == 2:
        pad = torch.empty(3 << 30, dtype=torch.uint8, device=dev)  # shift where the next allocations land
eng.close()
