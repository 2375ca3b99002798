"""Same-process A/B of the THREE emit shapes (GNNPE_EMIT=starts|tiles|tickets): same count, same output buffers, outputs
compared bit for bit with the start-vertex shape; then every shape timed into K independent allocations of the output.
usage: emit_ab3.py [n m] [--bufs K] [--e E] [--tpt 1,2,4,8] [--occ 0,4,3] [--check-only]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _diag  # noqa: F401  (the diagnostic build: this script's knobs live there)
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth


def opt(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and not sys.argv[i - 1].startswith("--")]
n, m = (int(args[0]), int(args[1])) if len(args) >= 2 else (1_000_000, 10_000_000)
nbuf = int(opt("--bufs", "3"))
e = int(opt("--e", "2"))
tpts = [int(x) for x in opt("--tpt", "4").split(",")]
occs = [int(x) for x in opt("--occ", "0").split(",")]
nhs = [int(x) for x in opt("--heads", "64").split(",")]
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)


def check(n_, m_, e_, order, seed=5):
    g = synth.gnm_graph(n_, m_, n_labels=7, seed=seed)
    sn = synth.degree_order(g["offsets"]) if order == "degree" else np.random.default_rng(seed).permutation(n_).astype(np.uint32)
    eng = binding.Engine(0, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(n_, np.uint32), 1)
    eng.set_label_table(binding.host_label_table(7, e_)); eng.vde(want=False)
    total = eng.count_paths(2)
    ids = torch.zeros((total + 64, 3), dtype=torch.int32, device=dev); pde = torch.zeros((total + 64, 3 * e_), dtype=torch.float64, device=dev)
    os.environ["GNNPE_EMIT"] = "starts"
    eng.fill_paths_device(0, total, ids, pde, None); torch.cuda.synchronize()
    ref_i, ref_p = ids.clone(), pde.clone()
    os.environ["GNNPE_EMIT"] = "tickets"
    for tpt in (1, 3, 4):
        os.environ["GNNPE_TICKET_TILES"] = str(tpt)
        ids.fill_(-1); pde.fill_(-1.0)
        eng.fill_paths_device(0, total, ids, pde, None); torch.cuda.synchronize()
        assert eng.emit_kernel_name() == ("k_fill_tickets" if e_ <= 2 else "k_fill_tiles"), eng.emit_kernel_name()  # e > 2: the one-shot tile kernel
        bad = (ids[:total] != ref_i[:total]).any(dim=1).nonzero()
        assert bad.numel() == 0, (n_, m_, e_, order, tpt, "ids differ at rows", bad[:8].flatten().tolist(), "of", total)
        assert torch.equal(pde[:total].view(torch.int64), ref_p[:total].view(torch.int64)), (n_, m_, e_, order, tpt, "pde differs")
        assert bool((ids[total:] == -1).all()) and bool((pde[total:] == -1.0).all()), "wrote past the total"
        # ids only / pde only
        ids.fill_(-1)
        eng.fill_paths_device(0, total, ids, None, None); torch.cuda.synchronize()
        assert torch.equal(ids[:total], ref_i[:total])
        pde.fill_(-1.0)
        eng.fill_paths_device(0, total, None, pde, None); torch.cuda.synchronize()
        assert torch.equal(pde[:total].view(torch.int64), ref_p[:total].view(torch.int64))
        for lo, hi in ((0, 1), (1, 64), (63, 65), (64, 128), (77, min(total, 77 + 100_003)), (total - 65, total), (total - 1, total)):
            if lo < 0 or hi > total or hi <= lo: continue
            ci = torch.full((hi - lo + 8, 3), -1, dtype=torch.int32, device=dev); cp = torch.full((hi - lo + 8, 3 * e_), -1.0, dtype=torch.float64, device=dev)
            eng.fill_paths_device(lo, hi, ci, cp, None); torch.cuda.synchronize()
            assert torch.equal(ci[:hi - lo], ref_i[lo:hi]) and torch.equal(cp[:hi - lo].view(torch.int64), ref_p[lo:hi].view(torch.int64)), ("chunk", lo, hi)
            assert bool((ci[hi - lo:] == -1).all()) and bool((cp[hi - lo:] == -1.0).all()), ("chunk overrun", lo, hi)
        # capped enqueue-only fill
        for cap in (total + 50, total, max(total - 777, 1)):
            ci = torch.full((cap, 3), -1, dtype=torch.int32, device=dev); cp = torch.full((cap, 3 * e_), -1.0, dtype=torch.float64, device=dev)
            eng.count_paths_enqueue(2); eng.fill_paths_capped_device(cap, ci, cp); torch.cuda.synchronize()
            k = min(cap, total)
            assert torch.equal(ci[:k], ref_i[:k]) and torch.equal(cp[:k].view(torch.int64), ref_p[:k].view(torch.int64)), ("capped", cap)
            assert bool((ci[k:] == -1).all()), ("capped overrun", cap)
    os.environ.pop("GNNPE_TICKET_TILES")
    eng.close()
    print(f"ok: n={n_} m={m_} e={e_} order={order} paths={total}", flush=True)


if "--quick" in sys.argv:  # for traces / counter passes: three launches of every shape into one buffer, nothing else
    g = synth.gnm_graph(n, m)
    sn = synth.degree_order(g["offsets"])
    eng = binding.Engine(0, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(64, e)); eng.vde(want=False)
    total = eng.count_paths(2)
    ids = torch.empty((total, 3), dtype=torch.int32, device=dev); pde = torch.empty((total, 3 * e), dtype=torch.float64, device=dev)
    os.environ["GNNPE_TICKET_TILES"] = str(tpts[0]); os.environ["GNNPE_TICKET_OCC"] = str(occs[0]); os.environ["GNNPE_TICKET_HEADS"] = str(nhs[0])
    for shape in opt("--shapes", "starts,tiles,tickets").split(","):
        os.environ["GNNPE_EMIT"] = shape
        for _ in range(3):
            eng.fill_paths_device(0, total, ids, pde, None); torch.cuda.synchronize()
    if "--knockouts" in sys.argv:  # diagnostic library (make DIAG=1): the ticket kernel without its stores / record loads
        os.environ["GNNPE_EMIT"] = "tickets"
        for xf in (0, 1, 2, 3, 16, 17, 18, 19):
            os.environ["GNNPE_TICKET_EXP"] = str(xf)
            ts = []
            for _ in range(5):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"tickets knock-out {xf} (1 = no stores, 2 = no record loads): min {min(ts):.3f} ms", flush=True)
    eng.close(); sys.exit(0)

for (n_, m_) in (() if "--no-check" in sys.argv else ((300, 1500), (900, 9000), (3000, 45000), (100_000, 1_000_000))):
    for e_ in ((1, 2, 3, 4, 8) if n_ < 10000 else (2,)):
        for order in ("degree", "random"):
            check(n_, m_, e_, order)
if "--check-only" in sys.argv:
    sys.exit(0)

g = synth.gnm_graph(n, m)
sn = synth.degree_order(g["offsets"])
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, e)); eng.vde(want=False)
total = eng.count_paths(2)
print(f"n={n} m={m} e={e} paths={total}", flush=True)
bufs = [(torch.empty((total, 3), dtype=torch.int32, device=dev), torch.empty((total, 3 * e), dtype=torch.float64, device=dev)) for _ in range(nbuf)]
B = 4 * 3 + 8 * 3 * e + 16 + 8 * e
os.environ["GNNPE_EMIT"] = "starts"
ids0, pde0 = bufs[0]
eng.fill_paths_device(0, total, ids0, pde0, None); torch.cuda.synchronize()
ref_ids, ref_pde = ids0.clone(), pde0.clone()
for shape in ("tiles", "tickets"):
    os.environ["GNNPE_EMIT"] = shape
    ids0.zero_(); pde0.zero_()
    eng.fill_paths_device(0, total, ids0, pde0, None); torch.cuda.synchronize()
    assert torch.equal(ids0, ref_ids) and torch.equal(pde0.view(torch.int64), ref_pde.view(torch.int64)), shape
print("parity at full size: tiles == tickets == starts bit for bit", flush=True)
del ref_ids, ref_pde
pads = [int(x) for x in opt("--pads", "").split(",") if x]  # k_fill_ranked with other occupancies (GNNPE_FILL_LDS_PAD: dynamic LDS nobody touches)
rtk = [int(x) for x in opt("--ranked-tickets", "").split(",") if x]  # k_fill_ranked taking its start vertices from this many ticket heads
oneshot = [int(x) for x in opt("--oneshot", "").split(",") if x]  # one wave per start vertex in launch order (GNNPE_FILL_ONESHOT) at these LDS pads (-1: the default pad)
variants = ([("starts", None, None), ("starts_low", None, None)] + ([("starts", -6, None)] if "--resident" in sys.argv else []) + [("starts", -5, p) for p in oneshot] + ([(sh, -4, r) for r in (64, 256) for sh in ("starts", "starts_low")] if "--rows" in sys.argv else []) + [("starts", -1, p) for p in pads] + [("starts", -2, h) for h in rtk] +
            [("starts", -3, (h, p)) for h in rtk for p in pads] + [("tiles", None, None)]) + [("tickets", t, (o, h)) for t in tpts for o in occs for h in nhs]
for rnd in range(2):
    for bi, (ids, pde) in enumerate(bufs):
        for shape, tpt, occ in variants:
            os.environ["GNNPE_EMIT"] = shape
            os.environ.pop("GNNPE_FILL_LDS_PAD", None)
            os.environ.pop("GNNPE_RANKED_TICKETS", None)
            os.environ.pop("GNNPE_FILL_ROWS", None)
            os.environ.pop("GNNPE_FILL_ONESHOT", None)
            if tpt == -6:  # shape 1 as the resident ticket grid of rounds 4-5 (the default is one-shot since round 6)
                os.environ["GNNPE_FILL_ONESHOT"] = "0"
            if tpt == -5:
                os.environ["GNNPE_FILL_ONESHOT"] = "1"
                if occ >= 0:
                    os.environ["GNNPE_FILL_LDS_PAD"] = str(occ)
            if tpt == -4:
                os.environ["GNNPE_FILL_ROWS"] = str(occ)
            if tpt in (-4, -5, -6):
                pass
            elif tpt == -3:
                os.environ["GNNPE_RANKED_TICKETS"] = str(occ[0]); os.environ["GNNPE_FILL_LDS_PAD"] = str(occ[1])
            elif tpt == -2:
                os.environ["GNNPE_RANKED_TICKETS"] = str(occ)
            elif tpt == -1:
                os.environ["GNNPE_FILL_LDS_PAD"] = str(occ)
            elif tpt is not None:
                os.environ["GNNPE_TICKET_TILES"] = str(tpt); os.environ["GNNPE_TICKET_OCC"] = str(occ[0]); os.environ["GNNPE_TICKET_HEADS"] = str(occ[1])
            eng.fill_paths_device(0, total, ids, pde, None); torch.cuda.synchronize()
            ts = []
            for _ in range(8):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            tag = shape if tpt is None else "starts resident grid" if tpt == -6 else f"starts one-shot pad={occ}" if tpt == -5 else f"{shape} rows{occ}" if tpt == -4 else f"starts lds_pad={occ}" if tpt == -1 else f"starts tickets heads={occ}" if tpt == -2 else f"starts tickets heads={occ[0]} pad={occ[1]}" if tpt == -3 else f"tickets tpt={tpt} occ={occ[0]} heads={occ[1]}"
            print(f"round {rnd} buf {bi} {tag:30s}: min {ts[0]:.3f} median {ts[4]:.3f} ms  frac(min) {total * B / ts[0] / 1e-3 / 8e12:.3f}", flush=True)
eng.close()
