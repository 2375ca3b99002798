"""Per-rank compute cost of the N-rank slab build, emulated on ONE GPU (the round's box has one): rank r's owned rows
are loaded, its halo rows are appended from the full graph (what the one-time all-to-all-v delivers, truncated to the
slab's rank range), then the per-step kernels are timed: vde of the owned rows, count, fill.  Communication (the
per-step all-gather of n x e doubles) is not included.  The single-GPU step of the same build is timed first, so the
table is a modelled compute speed-up  t(N=1) / max_r t_r(N).

    python scripts/emulate_rank.py N [--weights 1,3.9,1.3] [--no-truncate] [r ...]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gnnpe_amd  # noqa: F401
from gnnpe_amd import binding, synth
from gnnpe_amd.dist import STEP_COST_WEIGHTS, owned_rows, plan_slabs

ap = argparse.ArgumentParser()
ap.add_argument("N", type=int)
ap.add_argument("ranks", type=int, nargs="*")
ap.add_argument("--weights", type=str, default=",".join(str(x) for x in STEP_COST_WEIGHTS), help="w_paths,w_owned,w_held of dist.plan_slabs")
ap.add_argument("--no-truncate", action="store_true")
ap.add_argument("--out", default=None)
args = ap.parse_args()
N = args.N
ranks = args.ranks or list(range(N))
g = synth.gnm_graph(1_000_000, 10_000_000)
n = g["n"]
sn = synth.degree_order(g["offsets"])
mem = synth.block_membership(n, N)
weights = tuple(float(x) for x in args.weights.split(","))
bounds = plan_slabs(g["offsets"], sn, N, g["nbrs"], weights=weights)
offs = g["offsets"].astype(np.int64)
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
table = binding.host_label_table(64, 2)


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def time_step(eng, reps=5):
    best = None
    for _ in range(reps):
        t0 = ev()
        eng.vde(want=False)
        t1 = ev()
        total = eng.count_paths(2)
        t2 = ev()
        out_ids = torch.empty((total, 3), dtype=torch.int32, device=dev)
        out_pde = torch.empty((total, 6), dtype=torch.float64, device=dev)
        t3 = ev()
        eng.fill_paths_device(0, total, out_ids, out_pde, None)
        t4 = ev()
        torch.cuda.synchronize()
        cur = dict(paths=total, vde_ms=t0.elapsed_time(t1), count_ms=t1.elapsed_time(t2), fill_ms=t3.elapsed_time(t4))
        cur["step_ms"] = cur["vde_ms"] + cur["count_ms"] + cur["fill_ms"]
        if best is None or cur["step_ms"] < best["step_ms"]:
            best = cur
        del out_ids, out_pde
    return best


eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
eng.set_order(sn, mem, N)
eng.set_label_table(table)
single = time_step(eng)
eng.close()
print("N=1:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in single.items()})
rows_out = []
for r in ranks:
    rows, roff, rnbr = owned_rows(g, sn, bounds, r)
    eng = binding.Engine(0, stream=stream.cuda_stream)
    eng.load_rows(n, g["labels"], rows, roff, rnbr, nbr_capacity=len(g["nbrs"]))
    eng.set_order(sn, mem, N)
    eng.set_slab(int(bounds[r]), int(bounds[r + 1]))
    eng.set_label_table(table)
    need = torch.zeros(n, dtype=torch.int32, device=dev)
    counts = eng.halo_need(bounds, need, n)
    k = int(counts.sum())
    ids = need[:k].cpu().numpy().view(np.uint32).astype(np.int64)
    deg = offs[ids + 1] - offs[ids]
    idx = np.repeat(offs[ids] - np.concatenate([[0], np.cumsum(deg)[:-1]]), deg) + np.arange(int(deg.sum()))
    hn = torch.from_numpy(g["nbrs"][idx].view(np.int32)).to(dev)
    hd = torch.from_numpy(deg.astype(np.int32)).to(dev)
    t0 = ev()
    eng.rows_append(k, need[:k], hd, hn, int(deg.sum()), 0 if args.no_truncate else int(bounds[r]))
    t1 = ev()
    torch.cuda.synchronize()
    res = time_step(eng)
    rk = np.empty(n, np.int64)
    rk[sn] = np.arange(n)
    kept = int((rk[g["nbrs"][idx].astype(np.int64)] >= (0 if args.no_truncate else int(bounds[r]))).sum())
    res.update(rank=r, slab=int(bounds[r + 1] - bounds[r]), owned_entries=int(roff[-1]), halo_rows=k, halo_entries_sent=int(deg.sum()),
               held_entries=int(roff[-1]) + kept,
               halo_install_once_ms=t0.elapsed_time(t1))
    rows_out.append(res)
    print(f"N={N} rank {r}:", {k2: (round(v, 3) if isinstance(v, float) else v) for k2, v in res.items()})
    eng.close()
worst = max(x["step_ms"] for x in rows_out)
summary = dict(N=N, weights=weights, truncate=not args.no_truncate, single_gpu=single, ranks=rows_out,
               max_rank_step_ms=worst, modelled_compute_speedup=single["step_ms"] / worst)
print(f"modelled compute speed-up at N={N}: {single['step_ms']:.3f} / {worst:.3f} = {single['step_ms'] / worst:.2f}x")
if args.out:
    json.dump(summary, open(args.out, "w"), indent=1)
