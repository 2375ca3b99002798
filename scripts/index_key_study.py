"""Host-side study of the bulk-load sort key of R6 (numpy only, no GPU): how many leaves / level-1 nodes would the
reference's online traversal (GNN-PE/include/custom.h:441-478) open for queries drawn from the data, for different
key constructions?  A node is opened when the query's label embedding lies inside the node's label MBR and the
query's pde is dominated by the node's upper corner.

    python scripts/index_key_study.py [--vertices 100000 --edges 1000000 --labels 64]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import binding, synth  # noqa: E402


def paths_l2(g, rank):
    """all 3-vertex paths (s, b, c), rank[s] < rank[c]"""
    off, nb = g["offsets"].astype(np.int64), g["nbrs"]
    deg = np.diff(off)
    n = len(deg)
    mids = np.repeat(np.arange(n), deg * deg)
    base = np.repeat(np.cumsum(deg * deg) - deg * deg, deg * deg)
    k = np.arange(len(mids)) - base
    d = deg[mids]
    a = nb[off[mids] + k // d]
    c = nb[off[mids] + k % d]
    keep = rank[a] < rank[c]
    return np.stack([a[keep], mids[keep], c[keep]], 1).astype(np.uint32)


def zorder(q, bits):
    n, D = q.shape
    key = np.zeros(n, np.uint64)
    for t in range(bits):
        for k in range(D):
            key |= ((q[:, k] >> np.uint64(t)) & np.uint64(1)) << np.uint64(t * D + (D - 1 - k))
    return key


def uniform_q(pts, bits):
    lo, hi = pts.min(0), pts.max(0)
    qmax = (1 << bits) - 1
    return np.minimum((pts - lo) / (hi - lo) * qmax, qmax).astype(np.uint64)


def equidepth_q(vde, ids, bits):
    """per-vertex, per-component rank quantiser"""
    n, e = vde.shape
    q = np.zeros((n, e), np.uint64)
    for k in range(e):
        r = np.empty(n, np.int64)
        r[np.argsort(vde[:, k], kind="stable")] = np.arange(n)
        q[:, k] = (r << bits) // n
    return q[ids].reshape(len(ids), -1)


def evaluate(order, pts, lab, F, queries):
    """mean number of leaves and of level-1 nodes opened per query"""
    P, D = pts.shape
    sp, sl = pts[order], lab[order]
    out = []
    for fan in (F, F * F):
        nl = -(-P // fan)
        pad = nl * fan - P
        hi = np.concatenate([sp, np.full((pad, D), -np.inf)]).reshape(nl, fan, D).max(1)
        llo = np.concatenate([sl, np.full((pad, D), np.inf)]).reshape(nl, fan, D).min(1)
        lhi = np.concatenate([sl, np.full((pad, D), -np.inf)]).reshape(nl, fan, D).max(1)
        tot = 0
        for qi in queries:
            ok = np.all(hi >= pts[qi], 1) & np.all(llo <= lab[qi], 1) & np.all(lhi >= lab[qi], 1)
            tot += int(ok.sum())
        out.append(tot / len(queries))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vertices", type=int, default=100_000)
    ap.add_argument("--edges", type=int, default=1_000_000)
    ap.add_argument("--labels", type=int, default=64)
    ap.add_argument("--e", type=int, default=2)
    ap.add_argument("--queries", type=int, default=200)
    args = ap.parse_args()
    g = synth.gnm_graph(args.vertices, args.edges, n_labels=args.labels)
    sn = synth.degree_order(g["offsets"])
    rank = np.empty(args.vertices, np.int64)
    rank[sn] = np.arange(args.vertices)
    ids = paths_l2(g, rank)
    P, e = len(ids), args.e
    table = binding.host_label_table(args.labels, e)
    x = table[g["labels"]]
    vde = x.copy()
    src = np.repeat(np.arange(args.vertices), np.diff(g["offsets"].astype(np.int64)))
    np.add.at(vde, src, x[g["nbrs"]])
    pts = vde[ids].reshape(P, 3 * e)
    lab = x[ids].reshape(P, 3 * e)
    D = 3 * e
    F = (4096 - 5) // (16 * D + 4) - 2
    rng = np.random.default_rng(1)
    queries = rng.integers(0, P, args.queries)
    print(f"{P} paths, D={D}, F={F}, {-(-P // F)} leaves")
    lab_id = (g["labels"][ids[:, 0]].astype(np.uint64) * args.labels + g["labels"][ids[:, 1]]) * args.labels + g["labels"][ids[:, 2]]
    cases = {"path-id order (no sort)": np.arange(P)}
    for bits in (10, 5):
        cases[f"uniform {bits}b z-order"] = np.argsort(zorder(uniform_q(pts, bits), bits), kind="stable")
    for bits in (10, 6, 5, 4):
        cases[f"equi-depth {bits}b z-order"] = np.argsort(zorder(equidepth_q(vde, ids, bits), bits), kind="stable")
    for bits in (6, 4):
        z = zorder(equidepth_q(vde, ids, bits), bits)
        cases[f"labels, then equi-depth {bits}b z-order"] = np.lexsort((z, lab_id))
    cases["labels only (ties in path-id order)"] = np.argsort(lab_id, kind="stable")
    for bits in (2, 1):
        z = zorder(equidepth_q(vde, ids, bits), bits)
        cases[f"labels, then equi-depth {bits}b z-order"] = np.lexsort((z, lab_id))
    z = zorder(uniform_q(pts, 10), 10)
    cases["labels, then uniform 10b z-order"] = np.lexsort((z, lab_id))
    for name, order in cases.items():
        t0 = time.time()
        lv, l1 = evaluate(order, pts, lab, F, queries)
        print(f"{name:42s} leaves opened {lv:10.1f}   level-1 nodes opened {l1:8.1f}   ({time.time() - t0:.1f}s)")


if __name__ == "__main__":
    main()
