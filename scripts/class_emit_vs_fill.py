"""Same buffers, two writers: K torch allocations of one emit output ([pde rows | id rows], 12.0 GB at config 3) alive at once;
per buffer the emit kernel's time (reads its records, writes both streams), torch's fill_ over the whole buffer, and the
library's plain streaming probe pattern emulated by a strided copy -- do they rank the buffers alike?
    python scripts/class_emit_vs_fill.py [K=6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
total = eng.count_paths(2)
MiB2 = 2 << 20
pde_b = (total * 48 + MiB2 - 1) // MiB2 * MiB2
ids_b = (total * 12 + MiB2 - 1) // MiB2 * MiB2
bufs = [torch.empty(pde_b + ids_b, dtype=torch.uint8, device="cuda") for _ in range(K)]
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for rnd in range(2):
    for k, b in enumerate(bufs):
        pde = b[:pde_b].view(torch.float64)[: total * 6].view(total, 6)
        ids = b[pde_b:].view(torch.int32)[: total * 3].view(total, 3)
        t_emit = timed(lambda: eng.fill_paths_device(0, total, ids, pde, None))
        w64 = b.view(torch.int64)
        t_fill = timed(lambda: w64.fill_(7))
        print(f"round {rnd} buffer {k} at {b.data_ptr():#x}: emit {t_emit:.3f} ms   torch fill_ {t_fill:.3f} ms = {b.numel() / t_fill / 1e6:.0f} GB/s", flush=True)
eng.close()
