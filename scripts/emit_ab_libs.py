"""Same-process A/B of the start-vertex emit kernel of TWO builds of the library (the tree's against scripts/ab_old_libgnnpe_hip.so,
a build of the previous commit): both engines count the same graph and emit into the same output buffers, alternately."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _diag  # noqa: F401  (the diagnostic build: this script's knobs live there)
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
os.environ["GNNPE_EMIT"] = "starts"
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
def engine():
    eng = binding.Engine(0, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
    return eng, eng.count_paths(2)
new, total = engine()
binding._lib = None
binding.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_old_libgnnpe_hip.so")
old, total2 = engine()
assert total == total2
dev = torch.device("cuda:0")
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bufs = [(torch.empty((total, 3), dtype=torch.int32, device=dev), torch.empty((total, 6), dtype=torch.float64, device=dev)) for _ in range(nbuf)]
B = 92
for rnd in range(2):
    for bi, (ids, pde) in enumerate(bufs):
        res = {}
        for name, eng in (("old", old), ("new", new)):
            eng.fill_paths_device(0, total, ids, pde, None); torch.cuda.synchronize()
            ts = []
            for _ in range(8):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res[name] = min(ts)
        print(f"round {rnd} buf {bi}: old {res['old']:.3f} ms ({total * B / res['old'] / 1e-3 / 8e12:.3f})  new {res['new']:.3f} ms ({total * B / res['new'] / 1e-3 / 8e12:.3f})  new/old {res['new'] / res['old']:.3f}", flush=True)
