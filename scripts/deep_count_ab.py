"""Same-process timing of the l = 3 count on the config-5 graph under environment settings the library reads per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
cases = sys.argv[1:] or [""]
e = 8
g = synth.powerlaw_graph(4_000_000, 64_000_000, exponent=2.1, max_degree=3000, n_labels=64, seed=1)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, e)); eng.vde(want=False)
total = eng.count_paths(3)
KEYS = set(kv.split("=")[0] for c in cases for kv in c.split(",") if kv)
for rnd in range(2):
    for case in cases:
        for k in KEYS: os.environ.pop(k, None)
        for kv in case.split(","):
            if kv:
                k, v = kv.split("="); os.environ[k] = v
        ts = []
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            t = eng.count_paths(3); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            assert t == total
        print(f"round {rnd} [{case or 'default':32s}] count_paths(3): min {min(ts)*1e3:.1f} ms  ({total} paths)", flush=True)
eng.close()
