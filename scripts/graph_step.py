"""EXPERIMENT: the config-3 step (vde + enqueue-only count + capped emit) replayed from a HIP graph captured on the engine's stream
(torch.cuda.graph) against plain launches -- same process, same buffers.   python scripts/graph_step.py [n m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 10_000_000)
g = synth.gnm_graph(n, m)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
total = eng.count_paths(2)
ids = torch.empty((total, 3), dtype=torch.int32, device="cuda"); pde = torch.empty((total, 6), dtype=torch.float64, device="cuda")
def step():
    eng.vde(want=False)
    eng.count_paths_enqueue(2)
    eng.fill_paths_capped_device(total, ids, pde)
for _ in range(3): step()
torch.cuda.synchronize()
def timeit(fn, k=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
plain = [timeit(step) for _ in range(3)]
print(f"plain launches : {min(plain):.4f} ms per step", flush=True)
try:
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=stream):
        step()
    torch.cuda.synchronize()
    rep = [timeit(gr.replay) for _ in range(3)]
    print(f"graph replay   : {min(rep):.4f} ms per step", flush=True)
    assert eng.count_total() == total
except Exception as ex:
    print("capture failed:", repr(ex)[:500])
eng.close()
