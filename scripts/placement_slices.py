"""Where inside a slow output placement is the time lost?  Six live candidates; the slowest and the fastest are filled
slice by slice (16 slices of the path range = 16 consecutive pieces of the buffers) and timed per slice."""
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
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return round(best, 3)
bufs = [(torch.empty((total, 3), dtype=torch.int32, device=dev), torch.empty((total, 6), dtype=torch.float64, device=dev)) for _ in range(6)]
base = [timed(lambda: eng.fill_paths_device(0, total, i, p, None)) for i, p in bufs]
print("whole:", base, flush=True)
S = 16
cuts = [total * k // S for k in range(S + 1)]
for which, name in ((int(np.argmax(base)), "slowest"), (int(np.argmin(base)), "fastest")):
    ids, pde = bufs[which]
    per = [timed(lambda a=a, b=b: eng.fill_paths_device(a, b, ids[a:b], pde[a:b], None)) for a, b in zip(cuts[:-1], cuts[1:])]
    print(name, which, "per slice:", per, "sum", round(sum(per), 3), flush=True)
    # the same slices of the path range written into the FIRST sixteenth of the buffers (same work, one place)
    a0, b0 = cuts[0], cuts[1]
    per2 = [timed(lambda a=a, b=b: eng.fill_paths_device(a, b, ids[a0:a0 + (b - a)], pde[a0:a0 + (b - a)], None)) for a, b in zip(cuts[:-1], cuts[1:]) if b - a <= b0 - a0]
    print(name, which, "same slices into the buffer's first piece:", per2, flush=True)
eng.close()
