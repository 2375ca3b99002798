"""Index build of TWO builds of the library at config 3 (the tree's against scripts/ab_old_libgnnpe_hip.so, a build of the previous
commit), one process: the leaf kernel's time goes with its image buffer's allocation, so the two engines are built twice, in
both orders, and the image addresses are printed -- compare runs whose address agrees.  Whole build = count's first build
(pair order + leaves + upper levels); cached = a further build from the cached pair order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
NEW = binding.LIB_PATH
OLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_old_libgnnpe_hip.so")
def engine(path):
    binding._lib = None
    binding.LIB_PATH = path
    eng = binding.Engine(0, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
    return eng
def timed(f):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); r = f(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), r
for order in (("new", "old"), ("old", "new"), ("new", "old")):
    engs = [(name, engine(NEW if name == "new" else OLD)) for name in order]
    for aux in (False, True):
        build = (lambda e: e.build_index_partition_aux_device(0)) if aux else (lambda e: e.build_index_partition_device(0))
        for name, eng in engs:
            eng.count_paths(2); build(eng); torch.cuda.synchronize()
        for rnd in range(2):
            for name, eng in engs:
                whole, cached = [], []
                for _ in range(3):
                    eng.count_paths(2); torch.cuda.synchronize()
                    t, r = timed(lambda: build(eng)); whole.append(t)
                    cached.append(min(timed(lambda: build(eng))[0] for _ in range(3)))
                print(f"order {'/'.join(order)} {'image+aux' if aux else 'image    '} [{name}] image @ {r[0]:#x}: whole build {min(whole):.3f} ms, cached pair order {min(cached):.3f} ms", flush=True)
    for name, eng in engs:
        eng.close()
