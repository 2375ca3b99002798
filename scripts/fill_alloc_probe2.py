"""Placement probe, part 2: several output buffers alive at once (distinct physical memory), each timed."""
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
def timed(ids, pde, reps=4):
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
bufs = []
for k in range(8):
    ids = torch.empty((total, 3), dtype=torch.int32, device=dev); pde = torch.empty((total, 6), dtype=torch.float64, device=dev)
    bufs.append((ids, pde))
res = [timed(i, p) for i, p in bufs]
print("8 live output pairs:", [round(x, 3) for x in res])
# cross: ids of pair a with pde of pair b
print("ids0+pde_k:", [round(timed(bufs[0][0], bufs[k][1]), 3) for k in range(8)])
print("ids_k+pde0:", [round(timed(bufs[k][0], bufs[0][1]), 3) for k in range(8)])
big = torch.empty(total * 60 + 4096, dtype=torch.uint8, device=dev)
ids = big[: total * 12].view(torch.int32).view(total, 3); pde = big[total * 12 + (-(total * 12) % 256):][: total * 48].view(torch.float64).view(total, 6)
print("one allocation for both:", round(timed(ids, pde), 3))
eng.close()
