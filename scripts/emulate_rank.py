"""Per-rank compute cost of the N-rank slab build, emulated on ONE GPU: rank r's owned rows are loaded, its halo
rows are appended from the full graph (what the all-to-all-v would deliver), then the step's kernels are timed.
Communication is not included.  Usage: emulate_rank.py N [r ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
from gnnpe_amd.dist import owned_rows, plan_slabs
N = int(sys.argv[1]); ranks = [int(x) for x in sys.argv[2:]] or [0, N // 2, N - 1]
g = synth.gnm_graph(1_000_000, 10_000_000)
n = g["n"]; sn = synth.degree_order(g["offsets"]); mem = synth.block_membership(n, N)
bounds = plan_slabs(g["offsets"], sn, N, g["nbrs"])
offs = g["offsets"].astype(np.int64); dev = torch.device("cuda:0")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
table = binding.host_label_table(64, 2)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for r in ranks:
    rows, roff, rnbr = owned_rows(g, sn, bounds, r)
    eng = binding.Engine(0, stream=stream.cuda_stream)
    eng.load_rows(n, g["labels"], rows, roff, rnbr, nbr_capacity=len(g["nbrs"]) + int(roff[-1]))
    eng.set_order(sn, mem, N); eng.set_slab(int(bounds[r]), int(bounds[r + 1])); eng.set_label_table(table)
    need = torch.zeros(n, dtype=torch.int32, device=dev)
    res = {}
    for rep in range(3):
        eng.rows_drop_halo()
        t0 = ev()
        counts = eng.halo_need(bounds, need, n)
        t1 = ev()
        k = int(counts.sum())
        ids = need[:k].cpu().numpy().view(np.uint32).astype(np.int64)
        deg = (offs[ids + 1] - offs[ids])
        idx = np.repeat(offs[ids] - np.concatenate([[0], np.cumsum(deg)[:-1]]), deg) + np.arange(int(deg.sum()))
        hn = torch.from_numpy(g["nbrs"][idx].view(np.int32)).to(dev); hd = torch.from_numpy(deg.astype(np.int32)).to(dev)
        torch.cuda.synchronize()
        t2 = ev(); eng.rows_append(k, need[:k], hd, hn, int(deg.sum())); t3 = ev()
        # full vde: emulate the all-gather by computing everything on a second context is overkill; own rows only + time it
        eng.vde(want=False); t4 = ev()
        total = eng.count_paths(2); t5 = ev()
        out_ids = torch.empty((total, 3), dtype=torch.int32, device=dev); out_pde = torch.empty((total, 6), dtype=torch.float64, device=dev)
        t6 = ev(); eng.fill_paths_device(0, total, out_ids, out_pde, None); t7 = ev()
        torch.cuda.synchronize()
        res = dict(slab=int(bounds[r + 1] - bounds[r]), paths=total, halo_rows=k, halo_entries=int(deg.sum()), halo_need_ms=t0.elapsed_time(t1),
                   rows_append_ms=t2.elapsed_time(t3), vde_ms=t3.elapsed_time(t4), count_ms=t4.elapsed_time(t5), fill_ms=t6.elapsed_time(t7))
        del out_ids, out_pde
    print(f"N={N} rank {r}:", {k2: (round(v, 3) if isinstance(v, float) else v) for k2, v in res.items()})
    eng.close()
