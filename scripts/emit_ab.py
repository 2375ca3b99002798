"""Same-process A/B of the two emit shapes (GNNPE_EMIT=starts|tiles): same count, same output buffers, outputs compared
bit for bit.  usage: emit_ab.py [n m] [--bufs K]  (K independent output allocations, every shape timed into each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _diag  # noqa: F401  (the diagnostic build: this script's knobs live there)
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and not sys.argv[i - 1].startswith("--")]
n, m = (int(args[0]), int(args[1])) if len(args) >= 2 else (1_000_000, 10_000_000)
tile_shapes = sys.argv[sys.argv.index("--shapes") + 1].split(",") if "--shapes" in sys.argv else ["1", "2"]
nbuf = int(sys.argv[sys.argv.index("--bufs") + 1]) if "--bufs" in sys.argv else 3
e = int(sys.argv[sys.argv.index("--e") + 1]) if "--e" in sys.argv else 2
g = synth.gnm_graph(n, m)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, e)); eng.vde(want=False)
total = eng.count_paths(2)
dev = torch.device("cuda:0")
print(f"n={n} m={m} e={e} paths={total}", flush=True)
bufs = [(torch.empty((total, 3), dtype=torch.int32, device=dev), torch.empty((total, 3 * e), dtype=torch.float64, device=dev)) for _ in range(nbuf)]
B = 4 * 3 + 8 * 3 * e + 16 + 8 * e
if "--stamps" in sys.argv:  # diagnostic build: cycles per phase and wave (stderr), full launches with and without the stores
    ids, pde = bufs[0]
    os.environ["GNNPE_EMIT"] = "tiles"
    for tsh in tile_shapes:
        for xf in (16, 17, 24):
            os.environ["GNNPE_TILE_SHAPE"] = tsh; os.environ["GNNPE_TILE_EXP"] = str(xf)
            print(f"shape {tsh} exp {xf}:", flush=True)
            for _ in range(2):
                eng.fill_paths_device(0, total, ids, pde, None); torch.cuda.synchronize()
    eng.close(); sys.exit(0)
if "--quick" in sys.argv:  # counter passes: two launches of each shape, nothing else
    ids, pde = bufs[0]
    for shape in ("starts", "tiles"):
        os.environ["GNNPE_EMIT"] = shape
        for _ in range(2):
            eng.fill_paths_device(0, total, ids, pde, None); torch.cuda.synchronize()
    eng.close(); sys.exit(0)
# parity: starts into buffer 0, tiles into buffer 1 (or the same buffer twice)
os.environ["GNNPE_EMIT"] = "starts"
ids0, pde0 = bufs[0]
eng.fill_paths_device(0, total, ids0, pde0, None); torch.cuda.synchronize()
ref_ids, ref_pde = ids0.clone(), pde0.clone()
ids0.zero_(); pde0.zero_()
os.environ["GNNPE_EMIT"] = "tiles"
for tsh in tile_shapes:
    if ":" in tsh and (int(tsh.split(":")[1]) & 3): continue  # knock-outs produce wrong rows by design
    os.environ["GNNPE_TILE_SHAPE"] = tsh.split(":")[0]
    os.environ["GNNPE_TILE_EXP"] = tsh.split(":")[1] if ":" in tsh else "0"
    ids0.zero_(); pde0.zero_()
    eng.fill_paths_device(0, total, ids0, pde0, None); torch.cuda.synchronize()
    print("kernel:", eng.emit_kernel_name(), tsh, flush=True)
    assert torch.equal(ids0, ref_ids), "ids differ"
    assert torch.equal(pde0.view(torch.int64), ref_pde.view(torch.int64)), "pde differs"
# chunked: an unaligned range
if total > 1000:
    lo, hi = 77, min(total, 77 + 100_003)
    ci = torch.zeros((hi - lo, 3), dtype=torch.int32, device=dev); cp = torch.zeros((hi - lo, 3 * e), dtype=torch.float64, device=dev)
    eng.fill_paths_device(lo, hi, ci, cp, None); torch.cuda.synchronize()
    assert torch.equal(ci, ref_ids[lo:hi]) and torch.equal(cp.view(torch.int64), ref_pde[lo:hi].view(torch.int64)), "chunk differs"
print("parity: tiles == starts bit for bit (full + unaligned chunk)", flush=True)
del ref_ids, ref_pde
for rnd in range(2):
    for bi, (ids, pde) in enumerate(bufs):
        for shape in ["starts"] + ["tiles:" + x for x in tile_shapes]:
            os.environ["GNNPE_EMIT"] = shape.split(":")[0]
            if ":" in shape:
                os.environ["GNNPE_TILE_SHAPE"] = shape.split(":")[1]
                os.environ["GNNPE_TILE_EXP"] = shape.split(":")[2] if shape.count(":") > 1 else "0"
            eng.fill_paths_device(0, total, ids, pde, None); torch.cuda.synchronize()
            ts = []
            for _ in range(8):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            print(f"round {rnd} buf {bi} {shape:14s}: min {ts[0]:.3f} median {ts[4]:.3f} ms  frac(min) {total * B / ts[0] / 1e-3 / 8e12:.3f}", flush=True)
eng.close()
