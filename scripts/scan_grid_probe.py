import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _diag  # noqa: F401  (the diagnostic build: this script's knobs live there)
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
g = synth.gnm_graph(1_000_000, 10_000_000); sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
tot = eng.count_paths(2)
for grid in ("", "256", "512", "1024", "2048", "3907"):
    if grid: os.environ["GNNPE_START_SCAN_GRID"] = grid
    ts = []
    for _ in range(6):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); eng.count_paths_enqueue(2); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    assert eng.count_total() == tot
    print(f"grid {grid or 'default':8s}: count phase min {min(ts):.3f} ms", flush=True)
