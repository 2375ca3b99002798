"""Same-process A/B of emit-kernel experiments (GNNPE_FILL_EXP, temporary): same records, same output buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _diag  # noqa: F401  (the diagnostic build: this script's knobs live there)
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
cases = sys.argv[1:] or ["0", "1", "2", "3"]
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
total = eng.count_paths(2)
dev = torch.device("cuda:0")
ids = torch.empty((total, 3), dtype=torch.int32, device=dev); pde = torch.empty((total, 6), dtype=torch.float64, device=dev)
ref = None
for rnd in range(3):
    for case in cases:
        os.environ["GNNPE_FILL_EXP"] = case
        eng.fill_paths_device(0, total, ids, pde, None); torch.cuda.synchronize()
        chk = (int(ids[:, 2].long().sum()), float(pde[:, 5].sum()))
        if ref is None: ref = chk
        assert chk == ref, (case, chk, ref)
        ts = []
        for _ in range(8):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"round {rnd} exp {case}: fill min {min(ts):.3f} median {sorted(ts)[4]:.3f} ms", flush=True)
eng.close()
