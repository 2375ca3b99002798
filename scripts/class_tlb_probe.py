#!/usr/bin/env python3
"""Round 6: do the allocation classes show in the address-translation counters?  K independent 12 GB output buffers in one
process, the product's emit kernel (shape 1) four times into each; the script prints every buffer's best time, rocprofv3 --pmc records the
per-dispatch counters (scripts/class_tlb_probe.sh joins the two).  usage: python scripts/class_tlb_probe.py [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
total = eng.count_paths(2)
eng.set_emit_shape(1)
dev = torch.device("cuda", 0)
bufs = [(torch.empty((total, 3), dtype=torch.int32, device=dev), torch.empty((total, 6), dtype=torch.float64, device=dev)) for _ in range(K)]
for bi, (ids, pde) in enumerate(bufs):
    ts = []
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"buffer {bi} ids at {ids.data_ptr():#x}: best of the last three {min(ts[1:]):.3f} ms (first touch {ts[0]:.3f})", flush=True)
eng.close()
