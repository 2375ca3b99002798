#!/usr/bin/env python3
"""Round 6: does the RELATIVE placement of the two output streams inside one allocation move the emit kernel's time?  (The allocation classes
show as DRAM write-credit stalls, profiles/r06_class_counters.txt: if they came from the id rows and the embedding rows meeting in the same
banks, shifting one stream against the other inside the same pages would change them.)  K allocations of 12 GB + 64 MiB; ids at the start,
pde behind them at a series of extra offsets; the product's emit kernel (shape 1), best of three after a first touch.
usage: python scripts/class_phase_probe.py [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
total = eng.count_paths(2)
eng.set_emit_shape(1)
dev = torch.device("cuda", 0)
ids_b, pde_b = total * 12, total * 48
ids_r = (ids_b + 4095) // 4096 * 4096
slack = 64 << 20
offs = [0, 256, 1024, 4096, 16384, 65536, 1 << 20, (2 << 20) + 4096, 8 << 20, (32 << 20) + 512]
for bi in range(K):
    buf = torch.empty(ids_r + pde_b + slack, dtype=torch.uint8, device=dev)
    base = buf.data_ptr()
    res = []
    for off in offs + [0]:
        ts = []
        for rep in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.fill_paths_device(0, total, base, base + ids_r + off, None); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append(min(ts[1:]))
    print(f"allocation {bi} at {base:#x}: " + "  ".join(f"+{o}: {t:.3f}" for o, t in zip(offs + [0], res)) + " ms", flush=True)
    del buf
    torch.cuda.empty_cache()
eng.close()
