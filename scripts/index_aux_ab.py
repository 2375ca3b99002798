"""Same-process timing of the pair-major index build at config 3, image alone against image + auxiliary index with the
aux row blocks in both forms (GNNPE_AUX_WIDE is read once per count by build_raux, so every case counts again).
Times are builds from the cached pair order (bench.py: next_partition_ms / image_and_aux_index_ms)."""
import os, sys
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
keep = {}
for rnd in range(2):
    for case in ["image", "aux compact", "aux wide"]:
        os.environ["GNNPE_AUX_WIDE"] = "1" if case.startswith("aux wide") else "0"
        eng.count_paths(2)
        build = (lambda: eng.build_index_partition_device(0)) if case.startswith("image") else (lambda: eng.build_index_partition_aux_device(0))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); build(); e1.record(); torch.cuda.synchronize()
        first = e0.elapsed_time(e1)
        ts = []
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); build(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        if case in ("aux compact", "aux wide") and rnd == 0:
            r = eng.build_index_partition_aux_device(0, fetch=True)
            keep[case] = (r[3].copy(), r[4].copy(), r[5].copy())
        print(f"round {rnd} [{case:24s}]: first build of the count {first:.3f} ms, from the cached pair order min {min(ts):.3f} median {sorted(ts)[2]:.3f} ms", flush=True)
a = keep["aux compact"]
for other in ("aux wide",):
    print(f"auxiliary arrays of [{other}] identical to [aux compact]:", all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, keep[other])))
eng.close()
