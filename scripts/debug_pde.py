import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
n, m = int(sys.argv[1]), int(sys.argv[2])
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = synth.gnm_graph(n, m)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
eng.set_order(sn, synth.block_membership(n, 8), 8)
eng.set_label_table(binding.host_label_table(64, 2))
eng.set_fill_variant(variant)
x, nx, vde = eng.vde()
total = eng.count_paths(2)
dev = torch.device("cuda:0")
ids = torch.empty((total, 3), dtype=torch.int32, device=dev)
pde = torch.full((total, 6), -1.0, dtype=torch.float64, device=dev)
eng.fill_paths_device(0, total, ids, pde, None)
torch.cuda.synchronize()
vde_t = torch.from_numpy(vde).to(dev)
dv, dx = eng.vde_device_ptr()
print("total", total, "variant", variant)
for k in range(3):
    col = ids[:, k].long()
    bad = (pde[:, 2*k:2*k+2] != vde_t[col]).any(dim=1)
    nb = int(bad.sum())
    print("k", k, "mismatching rows", nb)
    if nb:
        idx = torch.nonzero(bad)[:, 0]
        print("  first", idx[:10].tolist(), "last", idx[-5:].tolist())
        print("  tiles(512) of first", (idx[:10] // 512).tolist(), "pos in tile", (idx[:10] % 512).tolist())
        i0 = int(idx[0])
        print("  row", i0, ids[i0].tolist(), pde[i0].tolist(), "expect", vde_t[col[i0]].tolist())
        print("  untouched(-1):", int((pde[idx, 2*k] == -1.0).sum()))
