"""Deterministic synthetic inputs for the offline path (SURVEY.md 8(d) "Synthetic inputs").

The reference ships one sample graph (Test/) and no generator; BASELINE.json's configs 2-5 are
synthetic.  This module is the build's own generator plus the prep-step file layout the
reference's `gnnpe.py` produces (mkdirs `gnnpe.py:60-64`, degree sort + `membership.txt`
`gnnpe.py:71-76`).  METIS is not available, so the partition column is contiguous id blocks.

Everything here is host-side numpy; nothing touches the GPU.
"""
import os

import numpy as np

SEED = 2022  # the reference's own seed (gnnpe.py:14)


def _csr_from_edges(n, eu, ev):
    """Undirected edge list -> CSR with ascending neighbour lists (what graph.cpp:211-233 builds from `e u v` lines)."""
    # one sort of the combined key src << 32 | dst = the (src, dst) lexicographic order (shifts: a division of 1.3e8 keys by n costs
    # more than the sort)
    eu = np.asarray(eu).astype(np.uint64)
    ev = np.asarray(ev).astype(np.uint64)
    key = np.concatenate([(eu << np.uint64(32)) | ev, (ev << np.uint64(32)) | eu])
    key.sort()
    offs = np.searchsorted(key, np.arange(n + 1, dtype=np.uint64) << np.uint64(32)).astype(np.uint32)  # first key of every source
    return offs, key.astype(np.uint32)  # (the low 32 bits)


def _searchsorted_left(cdf, r):
    """np.searchsorted(cdf, r) for MANY unsorted r in [0, 1) (6.5e7 needles into 4e6 ascending values: a binary search per needle
    is 22 cache misses).  A grid of 2^k cells over [0, 1) holds each cell's first candidate -- found with sorted needles, which is
    fast -- and a needle then walks forward from its cell's candidate (cells hold a couple of values).  r * 2^k and c / 2^k are exact
    in binary floating point, so the result equals np.searchsorted's bit for bit."""
    k = int(min(26, max(8, np.ceil(np.log2(max(len(cdf), 2))) + 4)))
    cells = 1 << k
    table = np.searchsorted(cdf, np.arange(cells, dtype=np.float64) / cells).astype(np.int64)
    idx = table[(r * cells).astype(np.int64)]
    todo = np.flatnonzero(cdf[idx] < r)
    while len(todo):
        idx[todo] += 1
        todo = todo[cdf[idx[todo]] < r[todo]]
    return idx


_MEMO = {}  # the last few generated graphs of this process (arrays read-only): a test session asks for G(1M, 10M) eight times


def _memo(key, make):
    g = _MEMO.get(key)
    if g is None:
        g = make()
        for v in g.values():
            if isinstance(v, np.ndarray):
                v.setflags(write=False)
        while len(_MEMO) >= 3:
            _MEMO.pop(next(iter(_MEMO)))
        _MEMO[key] = g
    return dict(g)


def gnm_graph(n, m, n_labels=64, seed=SEED):
    return _memo(("gnm", n, m, n_labels, seed), lambda: _gnm_graph(n, m, n_labels, seed))


def _gnm_graph(n, m, n_labels, seed):
    """G(n,m): exactly m distinct undirected edges without self-loops, uniform labels.

    Returns dict(n, m, offsets u32[n+1], nbrs u32[2m], labels u32[n], eu, ev (u<v, sorted))."""
    rng = np.random.default_rng(seed)
    keys = np.zeros(0, np.int64)
    need = m
    while True:
        k = int(need * 1.05) + 1024
        u = rng.integers(0, n, size=k, dtype=np.int64)
        v = rng.integers(0, n, size=k, dtype=np.int64)
        ok = u != v
        lo = np.minimum(u, v)[ok]
        hi = np.maximum(u, v)[ok]
        cand = np.concatenate([keys, lo * n + hi])
        # keep draw order so that the first m distinct edges are the ones selected: cand without the later occurrences of a value.
        # (np.unique(cand, return_index=True) says the same through a stable argsort of everything -- most of a G(1.5M, 60M)'s
        # generation time; repeats are a few thousand of 6e7 draws, so only THEIR positions are looked up)
        srt = np.sort(cand)
        dup = np.unique(srt[1:][srt[1:] == srt[:-1]])
        del srt
        if len(dup):
            at = np.searchsorted(dup, cand)
            at[at == len(dup)] = 0
            pos = np.flatnonzero(dup[at] == cand)  # every occurrence of a repeated value, ascending
            vals = cand[pos]
            o = np.argsort(vals, kind="stable")
            later = np.ones(len(o), bool)
            later[0] = False
            later[1:] = vals[o][1:] == vals[o][:-1]
            keep = np.ones(len(cand), bool)
            keep[pos[o[later]]] = False
            cand = cand[keep]
        keys = cand
        if len(keys) >= m:
            keys = keys[:m]
            break
        need = m - len(keys)
    keys = np.sort(keys)
    eu = (keys // n).astype(np.uint32)
    ev = (keys % n).astype(np.uint32)
    labels = rng.integers(0, n_labels, size=n, dtype=np.int64).astype(np.uint32)
    offs, nbrs = _csr_from_edges(n, eu, ev)
    return dict(n=n, m=m, offsets=offs, nbrs=nbrs, labels=labels, eu=eu, ev=ev)


def multigraph(n, m, n_dup=0, n_loops=0, n_labels=8, seed=SEED):
    """A NON-simple input: G(n, m) plus `n_dup` repeated `e` lines (one edge many times, some with the endpoints swapped) and
    `n_loops` self-loop lines (for refusal tests: the reference's loader has no defined result for them), the lines shuffled.  The
    dict is what the reference's loader makes of the duplicate lines (graph.cpp:211-233: every line adds an entry to both endpoints'
    lists; lists sorted): m = number of `e` lines, offsets / nbrs with the repeats, eu / ev in file order.  `write_graph_file`
    writes it (a self-loop line counts two in its vertex' degree field)."""
    g = gnm_graph(n, m, n_labels=n_labels, seed=seed)
    rng = np.random.default_rng(seed + 7)
    eu, ev = g["eu"].astype(np.int64), g["ev"].astype(np.int64)
    if n_dup and m:
        pick = rng.integers(0, m, size=n_dup)
        pick[: n_dup // 3] = pick[0]  # one edge many times
        du, dv = eu[pick], ev[pick]
        swap = rng.random(n_dup) < 0.5
        eu = np.concatenate([eu, np.where(swap, dv, du)])
        ev = np.concatenate([ev, np.where(swap, du, dv)])
    if n_loops:
        lv = rng.integers(0, n, size=n_loops)
        lv[: n_loops // 4] = lv[0]  # one vertex with several loops
        eu = np.concatenate([eu, lv])
        ev = np.concatenate([ev, lv])
    perm = rng.permutation(len(eu))
    eu, ev = eu[perm], ev[perm]
    offs, nbrs = _csr_from_edges(n, eu, ev)
    return dict(n=n, m=len(eu), offsets=offs, nbrs=nbrs, labels=g["labels"], eu=eu.astype(np.uint32), ev=ev.astype(np.uint32))


def powerlaw_graph(n, m, exponent=2.1, max_degree=2000, n_labels=64, seed=SEED):
    return _memo(("powerlaw", n, m, exponent, max_degree, n_labels, seed), lambda: _powerlaw_graph(n, m, exponent, max_degree, n_labels, seed))


def _powerlaw_graph(n, m, exponent, max_degree, n_labels, seed):
    """Chung-Lu style power-law graph with a degree cap (config 5 stress input).

    Endpoints are drawn proportionally to weights w_i ~ i^(-1/(exponent-1)) truncated so the
    expected degree stays <= max_degree; duplicates/self-loops are dropped, so the final edge
    count is <= m (returned in the dict)."""
    rng = np.random.default_rng(seed)
    w = (np.arange(1, n + 1, dtype=np.float64)) ** (-1.0 / (exponent - 1.0))
    w *= (2.0 * m) / w.sum()
    w = np.minimum(w, max_degree)
    p = w / w.sum()
    cdf = np.cumsum(p)
    cdf[-1] = 1.0
    k = int(m * 1.02)
    u = _searchsorted_left(cdf, rng.random(k))
    v = _searchsorted_left(cdf, rng.random(k))
    perm = rng.permutation(n).astype(np.int64)  # decouple id from weight rank
    u = perm[u]
    v = perm[v]
    ok = u != v
    lo = np.minimum(u, v)[ok].astype(np.uint64)
    hi = np.maximum(u, v)[ok].astype(np.uint64)
    keys = np.unique((lo << np.uint64(32)) | hi)[:m]  # (lo, hi) ascending, like lo * n + hi
    eu = (keys >> np.uint64(32)).astype(np.uint32)
    ev = (keys & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    labels = rng.integers(0, n_labels, size=n, dtype=np.int64).astype(np.uint32)
    offs, nbrs = _csr_from_edges(n, eu, ev)
    return dict(n=n, m=len(keys), offsets=offs, nbrs=nbrs, labels=labels, eu=eu, ev=ev)


def degree_order(offsets):
    """Processing order of `gnnpe.py:71-72`: ascending degree, ties by id (stable)."""
    deg = np.diff(offsets.astype(np.int64))
    return np.argsort(deg, kind="stable").astype(np.uint32)


def block_membership(n, p):
    """Partition column used when METIS is unavailable: contiguous id blocks floor(id*p/n)."""
    return (np.arange(n, dtype=np.int64) * p // max(n, 1)).astype(np.uint32)


def expected_paths_l2(offsets):
    """P for l=2 on a simple graph: sum_v C(deg v, 2)."""
    deg = np.diff(offsets.astype(np.int64))
    return int((deg * (deg - 1) // 2).sum())


def _render_lines(prefix, cols):
    """b"<prefix> c0 c1 ...\n" for every row of the non-negative integer columns `cols`, as one bytes object
    (vectorised decimal rendering: np.savetxt needs ~2 us per number)."""
    n = len(cols[0])
    width = 10  # ids and counts are < 2^32
    pieces, keep = [], []
    if prefix:
        pieces.append(np.full((n, 1), ord(prefix), np.uint8))
        keep.append(np.ones((n, 1), bool))
    for ci, c in enumerate(cols):
        c = np.asarray(c, np.int64)
        if prefix or ci:
            pieces.append(np.full((n, 1), ord(" "), np.uint8))
            keep.append(np.ones((n, 1), bool))
        pw = 10 ** np.arange(width - 1, -1, -1, dtype=np.int64)
        digits = (c[:, None] // pw[None, :]) % 10
        pieces.append((digits + ord("0")).astype(np.uint8))
        lead = np.cumsum(digits != 0, axis=1) == 0   # leading zeros
        lead[:, -1] = False                           # the value 0 keeps its last digit
        keep.append(~lead)
    pieces.append(np.full((n, 1), ord("\n"), np.uint8))
    keep.append(np.ones((n, 1), bool))
    return np.concatenate(pieces, axis=1)[np.concatenate(keep, axis=1)].tobytes()


def write_graph_file(path, g):
    """Text `.graph` in the format graph.cpp:172-219 parses: `t n m`, `v id label degree`
    (ascending id, true degree), `e u v` (u<v, sorted)."""
    n, m = g["n"], g["m"]
    deg = np.diff(g["offsets"].astype(np.int64))
    step = 1 << 21
    with open(path, "wb") as f:
        f.write(f"t {n} {m}\n".encode())
        for a in range(0, n, step):
            z = slice(a, min(a + step, n))
            f.write(_render_lines("v", [np.arange(z.start, z.stop), g["labels"][z], deg[z]]))
        for a in range(0, m, step):
            z = slice(a, min(a + step, m))
            f.write(_render_lines("e", [g["eu"][z], g["ev"][z]]))


def write_membership(path, sorted_nodes, membership):
    """`membership.txt` as `gnnpe.py:74-76` writes it: line i = "<vertex> <partition>"."""
    sn = np.asarray(sorted_nodes, np.int64)
    with open(path, "wb") as f:
        f.write(_render_lines("", [sn, np.asarray(membership, np.int64)[sn]]))


def make_dataset_dir(root, p):
    """Directory layout of `gnnpe.py:60-64`: <root>/gnn-pe/partitions/partition-i/."""
    for i in range(p):
        os.makedirs(os.path.join(root, "gnn-pe", "partitions", f"partition-{i}"), exist_ok=True)
    return os.path.join(root, "gnn-pe")
