"""Same-process timing of the count phase (k_rows_rank + scan + start records) at config 3.  The cases are environment
settings read per call; the knobs this was written for (rows in flight per wave, grid, knock-outs of the gathers / pair
scatter / record stores) were temporary and are gone from the library -- DESIGN.md section 3 has the results.
    python scripts/count_ab.py "GNNPE_ROWS_ILP=1" "GNNPE_ROWS_ILP=2" "GNNPE_ROWS_ILP=4,GNNPE_ROWS_GRID=2048" """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _diag  # noqa: F401  (the diagnostic build: this script's knobs live there)
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
cases = sys.argv[1:] or ["GNNPE_ROWS_ILP=1", "GNNPE_ROWS_ILP=2", "GNNPE_ROWS_ILP=4"]
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
want = synth.expected_paths_l2(g["offsets"])
KEYS = ["GNNPE_ROWS_ILP", "GNNPE_ROWS_GRID", "GNNPE_ROWS_KNOCK", "GNNPE_ROWS_EXP"]
ROUNDS, ITERS = int(os.environ.get("GNNPE_AB_ROUNDS", "3")), int(os.environ.get("GNNPE_AB_ITERS", "10"))  # keep both small under --pmc
for rnd in range(ROUNDS):
    for case in cases:
        for k in KEYS: os.environ.pop(k, None)
        for kv in case.split(","):
            k, v = kv.split("="); os.environ[k] = v
        got = eng.count_paths(2)
        assert got == want or 'KNOCK' in case, (got, want)
        ts = []
        for _ in range(ITERS):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); eng.count_paths(2); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"round {rnd} {case}: count phase min {min(ts):.3f} median {sorted(ts)[len(ts) // 2]:.3f} ms", flush=True)
eng.close()
