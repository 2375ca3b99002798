"""l = 3 emission at BASELINE config 5 (power-law 4M / 64M, e = 8): the slices of the emitting launch taken in order from a ticket counter
(default) against w, w + waves, ... (GNNPE_DEEP_TICKETS=0), same process, same output buffers, 2^26-path ranges at three places of the order;
rows compared by checksum.   usage: deep_ticket_ab.py [n m]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4_000_000, 64_000_000)
L, e = 4, 8
g = synth.powerlaw_graph(n, m, exponent=2.1, max_degree=3000, n_labels=64, seed=synth.SEED)
sn = synth.degree_order(g["offsets"])
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, e)); eng.vde(want=False)
total = eng.count_paths(3)
chunk = 1 << 26
bpp = 4 * L + 8 * e * L + 16 + 8 * e
bufs = [(torch.empty((chunk, L), dtype=torch.int32, device=dev), torch.empty((chunk, L * e), dtype=torch.float64, device=dev)) for _ in range(2)]
print(f"paths {total}", flush=True)
for bi, (ids, pde) in enumerate(bufs):
    for frac_at in (0.0, 0.37, 0.81):
        b = min(int(total * frac_at), total - chunk)
        sums = {}
        for mode in ("0", "1", "0", "1"):
            os.environ["GNNPE_DEEP_TICKETS"] = mode
            ts = []
            for _ in range(3):
                a = torch.cuda.Event(enable_timing=True); z = torch.cuda.Event(enable_timing=True)
                a.record(); eng.fill_paths_device(b, b + chunk, ids, pde, None); z.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(z))
            sums.setdefault(mode, eng.rows_checksum_device(chunk, L, ids, first_id=b))
            print(f"buffer {bi} range at {frac_at:.2f}: tickets={mode}  min {min(ts[1:]):.3f} ms  frac {chunk * bpp / min(ts[1:]) / 1e-3 / 8e12:.3f}", flush=True)
        assert sums["0"] == sums["1"], "rows differ"
eng.close()
