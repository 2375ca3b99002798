#!/usr/bin/env python3
"""Config 3: count, then the pair-major index build of partition 0 (first build of the count: pair order + leaves + upper levels),
three times -- under `rocprofv3 --kernel-trace` the timeline of one build (kernel, duration, start offset): where the build's
wall-clock goes beside its kernels.   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/idx_tl -- python3 scripts/index_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
for _ in range(4):
    eng.count_paths(2)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); eng.build_index_partition_device(0); ev1.record(); torch.cuda.synchronize()
    print("build %.3f ms" % ev0.elapsed_time(ev1), flush=True)
eng.close()
