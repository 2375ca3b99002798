"""Is a slow output placement a property of the physical region, or of how the pde stream's addresses line up?  Six live
candidates (pde allocated 256 MiB too large); the slowest is timed again with its pde rows starting at several offsets
inside the same allocation."""
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
EXTRA = (256 << 20) // 48
def timed(ids, pde, reps=3):
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return round(best, 3)
bufs = [(torch.empty((total, 3), dtype=torch.int32, device=dev), torch.empty((total + EXTRA, 6), dtype=torch.float64, device=dev)) for _ in range(6)]
base = [timed(i, p[:total]) for i, p in bufs]
print("offset 0:", base, flush=True)
for which, name in ((int(np.argmax(base)), "slowest"), (int(np.argmin(base)), "fastest")):
    ids, pde = bufs[which]
    out = {}
    for off_bytes in (4 << 10, 64 << 10, 1 << 20, 2 << 20, 8 << 20, 32 << 20, 128 << 20, 255 << 20):
        r = (off_bytes + 47) // 48
        out[f"{r * 48 / (1 << 20):.3f}MiB"] = timed(ids, pde[r:r + total])
    print(name, "candidate", which, "with its pde rows shifted:", out, flush=True)
eng.close()
