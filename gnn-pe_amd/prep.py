#!/usr/bin/env python3
"""Prep step without pymetis: what the reference's gnnpe.py / gnnpge.py leave on
disk for `main -m offline` -- the partition directories (gnnpe.py:60-64), the processing order
(ascending degree, ties by id, gnnpe.py:71-72) and `membership.txt` (gnnpe.py:74-76) -- computed from the
`.graph` text file itself.  The reference partitions with METIS (pymetis.part_graph, gnnpe.py:66-69),
which is not available here; any partition is valid for the offline/online pipeline (the answer count
does not depend on it), so the VALUES of membership.txt are not pinned to METIS' (they cannot be: SURVEY 8(c)).
What a partitioner is for -- balanced parts with few cut edges -- is what the methods below are measured on
(tests/test_host_cli.py: balance and edge cut on a planted-partition graph):
  blocks   contiguous id blocks  floor(id * p / n)
  bfs      breadth-first growth of p regions of ~n/p vertices each (keeps neighbours together)
  lp       bfs regions refined by balanced label propagation: vertices move to the part holding most of their
           neighbours while no part exceeds (1 + slack) n / p (default)

    python gnn-pe_amd/prep.py -f <dataset dir>/ -d <graph> -p <partitions> [--variant gnn-pe|gnn-pge] [--method lp]
"""
import argparse
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import binding, synth  # noqa: E402


def bfs_partition(offsets, nbrs, p):
    """Grow p regions breadth-first from the lowest unassigned id until each holds ~n/p vertices."""
    n = len(offsets) - 1
    part = np.full(n, -1, np.int64)
    target = -(-n // p)
    cur, filled, nxt = 0, 0, 0
    from collections import deque
    q = deque()
    for _ in range(n):
        while not q:
            while nxt < n and part[nxt] >= 0:
                nxt += 1
            if nxt >= n:
                break
            q.append(nxt)
            part[nxt] = cur
            filled += 1
        if not q:
            break
        v = q.popleft()
        for u in nbrs[offsets[v]:offsets[v + 1]]:
            if part[u] < 0:
                if filled >= target and cur < p - 1:
                    cur, filled = cur + 1, 0
                    q.clear()
                    break
                part[u] = cur
                filled += 1
                q.append(u)
        if filled >= target and cur < p - 1:
            cur, filled = cur + 1, 0
            q.clear()
    part[part < 0] = p - 1
    return part.astype(np.uint32)


def edge_cut(offsets, nbrs, part):
    """Number of undirected edges whose endpoints lie in different parts."""
    src = np.repeat(np.arange(len(offsets) - 1), np.diff(offsets.astype(np.int64)))
    return int((part[src] != part[nbrs]).sum()) // 2


def refine_lp(offsets, nbrs, part, p, sweeps=12, slack=0.03, seed=2022):
    """Balanced label propagation (vectorised): per sweep, half of the vertices (random, so that neighbours rarely move
    together) may move to the part that holds most of their neighbours; moves are granted in order of gain while the
    target part stays within (1 + slack) n / p.  Moves of one sweep are simultaneous, so two neighbours can still swap
    sides (hence the random half); the result is what tests/test_host_cli.py measures.  Stops when a sweep moves nothing."""
    n = len(offsets) - 1
    if n == 0 or p <= 1:
        return part.astype(np.uint32)
    rng = np.random.default_rng(seed)
    part = part.astype(np.int64).copy()
    deg = np.diff(offsets.astype(np.int64))
    src = np.repeat(np.arange(n), deg)
    cap = int(np.ceil((1.0 + slack) * n / p))
    rows = np.arange(n)
    for _ in range(sweeps):
        cnt = np.bincount(src * p + part[nbrs], minlength=n * p).reshape(n, p)
        best = cnt.argmax(1)
        gain = cnt[rows, best] - cnt[rows, part]
        cand = np.flatnonzero((gain > 0) & (rng.random(n) < 0.5))
        if len(cand) == 0:
            break
        cand = cand[np.argsort(-gain[cand], kind="stable")]
        sizes = np.bincount(part, minlength=p)
        moved = 0
        for t in range(p):
            free = cap - int(sizes[t])
            if free <= 0:
                continue
            take = cand[best[cand] == t][:free]
            part[take] = t
            moved += len(take)
        if moved == 0:
            break
    return part.astype(np.uint32)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-f", "--file", dest="dataset", required=True)
    ap.add_argument("-d", "--data", dest="graph", required=True)
    ap.add_argument("-p", "--partition", dest="p", type=int, default=5)
    ap.add_argument("--variant", choices=["gnn-pe", "gnn-pge"], default="gnn-pe")
    ap.add_argument("--method", choices=["blocks", "bfs", "lp"], default="lp")
    args = ap.parse_args(argv)
    g = binding.host_load_graph(args.graph)  # the library's own loader (host/graph_loader.cpp); needs no GPU
    n = g["n"]
    base = os.path.join(args.dataset, args.variant)
    shutil.rmtree(base, ignore_errors=True)  # gnnpe.py:60 deletes the old tree
    for i in range(args.p):
        os.makedirs(os.path.join(base, "partitions", f"partition-{i}"))
    if args.method == "blocks":
        mem = synth.block_membership(n, args.p)
    else:
        mem = bfs_partition(g["offsets"], g["nbrs"], args.p)
        if args.method == "lp":
            mem = refine_lp(g["offsets"], g["nbrs"], mem, args.p)
    order = synth.degree_order(g["offsets"])
    synth.write_membership(os.path.join(base, "membership.txt"), order, mem)
    sizes = np.bincount(mem, minlength=args.p)
    print(f"{args.variant}: {n} vertices -> {args.p} partitions ({args.method}), sizes {sizes.tolist()}, "
          f"cut edges {edge_cut(g['offsets'], g['nbrs'], mem)} of {g['m']}")


if __name__ == "__main__":
    main()
