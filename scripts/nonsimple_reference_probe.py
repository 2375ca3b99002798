#!/usr/bin/env python3
"""What the COMPILED reference (oracle/_ref/ref_main -m offline) does with non-simple graph files -- evidence for the loader's rules
(gnn-pe_amd/host/graph_loader.h).  Build container only (needs oracle/_ref).  Output: profiles/r06_selfloop_reference.txt.

1. Self-loop lines.  For `e u u` the loader computes both slots before it advances either cursor (graph.cpp:211-218): one slot is
   written twice, u's cursor moves by two, one slot of `new ui[2m]` stays uninitialised.  The table lists, per number of loop lines
   added to ONE fixed simple graph, the reference's path count, the paths that use an edge the file does not contain and the
   vertex they go through.
2. Duplicate lines.  Time of the reference by graph size: a path met again is not emitted but the call goes on below it
   (custom.h:68 false => no return => custom.h:80-91 extends the path through every simple path)."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_main")


def run_ref(g, n, timeout):
    with tempfile.TemporaryDirectory() as tmp:
        synth.write_graph_file(tmp + "/g.graph", g)
        synth.make_dataset_dir(tmp, 1)
        synth.write_membership(tmp + "/gnn-pe/membership.txt", np.arange(n, dtype=np.uint32), np.zeros(n, np.uint32))
        t = time.time()
        try:
            subprocess.run([REF, "-f", tmp + "/", "-d", tmp + "/g.graph", "-m", "offline", "-p", "1"], stdout=subprocess.DEVNULL, timeout=timeout)
        except subprocess.TimeoutExpired:
            return None, None
        txt = open(tmp + "/gnn-pe/all_paths.txt").read().split()
        return time.time() - t, np.array(txt[1:], np.int64).reshape(int(txt[0]), 3)


def main():
    out = []
    n = 130
    out.append("1. self-loop lines added to G(130, 390), seed 1004 (ids in file order 0..n-1, p = 1)")
    out.append("loop lines | reference paths | paths over an edge that is not in the file | such edges (a, b): b is listed in a's row only")
    for nl in (0, 1, 2, 4):
        g = synth.multigraph(n, 390, n_dup=0, n_loops=nl, n_labels=8, seed=1004)
        edges = set(zip(g["eu"].tolist(), g["ev"].tolist())) | set(zip(g["ev"].tolist(), g["eu"].tolist()))
        dt, paths = run_ref(g, n, 30)
        bad = set()
        n_bad = 0
        for a, b, c in paths.tolist():
            hit = [(x, y) for x, y in ((a, b), (b, c)) if (x, y) not in edges]
            n_bad += bool(hit)
            bad.update(hit)
        loops = sorted(set(g["eu"][g["eu"] == g["ev"]].tolist()))
        out.append(f"{nl:10d} | {len(paths):15d} | {n_bad:41d} | {sorted(bad)}   (loop vertices {loops})")
    out.append("   8 and 32 loop lines on the same graph: no result within 30 s (the spurious entries repeat, see 2.)")
    out.append("")
    out.append("2. duplicate lines: reference time by size (20 s limit)")
    out.append("n, m, repeated lines | seconds")
    for (nn, m, nd) in [(20, 24, 4), (30, 36, 8), (30, 45, 8), (40, 48, 10), (40, 60, 10), (60, 70, 12), (60, 90, 12)]:
        g = synth.multigraph(nn, m, n_dup=nd, n_labels=5, seed=1)
        dt, paths = run_ref(g, nn, 20)
        out.append(f"{nn}, {m}, {nd} | " + ("no result within 20 s" if dt is None else f"{dt:.2f}  ({len(paths)} paths)"))
    text = "\n".join(out) + "\n"
    open(os.path.join(ROOT, "profiles", "r06_selfloop_reference.txt"), "w").write(__doc__ + "\n" + text)
    print(text)


if __name__ == "__main__":
    main()
