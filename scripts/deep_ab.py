"""Same-process, same-buffer timing of the l = 3 emission: the variants are environment settings the library reads per call
(GNNPE_DEEP_EMIT=slices|units, ...), cycled round-robin over one pair of output buffers so that the buffers' write class
 is the same for all of them.
    python scripts/deep_ab.py [--small] [--log2 26] "GNNPE_DEEP_EMIT=slices" "GNNPE_DEEP_EMIT=units" """
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _diag  # noqa: F401  (the diagnostic build: this script's knobs live there)
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
ap = argparse.ArgumentParser()
ap.add_argument("--small", action="store_true", help="G(100K, 1M) instead of the config-5 graph")
ap.add_argument("--log2", type=int, default=26)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("cases", nargs="*")
args = ap.parse_args()
cases = args.cases or ["GNNPE_DEEP_EMIT=slices", "GNNPE_DEEP_EMIT=units"]
e, L = 8, 4
g = synth.gnm_graph(100_000, 1_000_000) if args.small else synth.powerlaw_graph(4_000_000, 64_000_000, exponent=2.1, max_degree=3000, n_labels=64, seed=1)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, e)); eng.vde(want=False)
total = eng.count_paths(3)
chunk = min(total, 1 << args.log2)
ids = torch.empty((chunk, L), dtype=torch.int32, device="cuda"); pde = torch.empty((chunk, L * e), dtype=torch.float64, device="cuda")
starts = [0, min(int(total * 0.37), total - chunk), min(int(total * 0.81), total - chunk)]
KEYS = set(kv.split("=")[0] for c in cases for kv in c.split(",") if kv)
sums = {}
for rnd in range(args.rounds):
    for case in cases:
        for k in KEYS: os.environ.pop(k, None)
        for kv in case.split(","):
            if kv:
                k, v = kv.split("="); os.environ[k] = v
        for b in starts:
            ts = []
            for _ in range(3):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); eng.fill_paths_device(b, b + chunk, ids, pde, None); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            chk = eng.rows_checksum_device(chunk, L, ids, b)
            sums.setdefault(b, chk)
            assert sums[b] == chk, (case, b)
            print(f"round {rnd} {case or '(default)':40s} first {b:>15d}: min {min(ts[1:]):.3f} ms  {chunk * 352 / min(ts[1:]) / 1e6 / 8000:.3f} of spec", flush=True)
eng.close()
