#!/usr/bin/env python3
"""Golden fixtures for NON-simple graph files (duplicate `e` lines) from the COMPILED REFERENCE.

The reference loads such a file as it is (graph.cpp:211-218: no de-duplication); its `degree` and the neighbour sum of gen_vde
count the repeats (graph.h:154-156, custom.h:527-534) and its hash set drops a path met again (custom.h:68-77).  This script writes
20 small random multigraphs (gnn-pe_amd/synth.py:multigraph -- few and many repeated lines, one edge repeated many times, swapped
endpoints), runs oracle/_ref/ref_main -m offline and oracle/_ref/ref_dump on each and stores INPUTS and OUTPUTS as data in
tests/golden/multigraph.npz:

  c<i>_n, _labels, _eu, _ev          the file's vertices and its `e` lines in file order (the test re-renders the same text)
  c<i>_order, _member, _p            membership.txt (random processing order, random partition)
  c<i>_paths                         reference all_paths.txt rows (uint32 P x 3)
  c<i>_part<j>                       reference partition_paths.txt ids of partition j
  c<i>_x, _nx, _vde, _degree         reference gen_vde dump (e = 2)
  c<i>_pde, _pde_label, _pdeg        reference gen_pde dump: embeddings and degree columns of every path
  c<i>_stdout                        the reference's two printGraphMetaData lines

Runs only where /root/reference exists (the build container).  No reference source text is stored, only data.
Re-run: python tests/golden/make_golden_multigraph.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import synth  # noqa: E402

REF_MAIN = os.path.join(ROOT, "oracle", "_ref", "ref_main")
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")


def cases():
    """(n, m, n_dup, n_labels, p, seed): 20 graphs.  They stay small and sparse: when the reference's DFS meets a path that is
    already in its hash set, the test at custom.h:68 is false and the call does not return -- it goes on extending the path
    (custom.h:80-91) through EVERY simple path below it, emitting nothing (depth never equals path_length again).  Same output,
    exponential time: G(60, 90) with 12 repeated lines does not finish in 20 s (profiles/r06_selfloop_reference.txt).
    No self-loops: the reference has no defined result for them (same file)."""
    out = []
    for i in range(20):
        n = [12, 16, 24, 32, 40][i % 5]
        m = n + (n // 10) * (i % 3)
        n_dup = 1 + (i * 7) % 11
        out.append((n, m, n_dup, [3, 8, 17][i % 3], [1, 2, 3, 5][i % 4], 1000 + i))
    return out


def read_dump(vde_bin, pde_bin):
    b = open(vde_bin, "rb").read()
    n, e = np.frombuffer(b, np.uint32, 2)
    arr = np.frombuffer(b, np.float64, 3 * n * e, 8).reshape(3, n, e)
    lab_deg = np.frombuffer(b, np.uint32, 2 * n, 8 + 3 * n * e * 8).reshape(2, n)
    b = open(pde_bin, "rb").read()
    P = int(np.frombuffer(b, np.uint64, 1)[0])
    L, e2 = np.frombuffer(b, np.uint32, 2, 8)
    rec = np.dtype([("vids", np.uint32, (L,)), ("labels", np.uint32, (L,)), ("degrees", np.uint32, (L,)),
                    ("pde", np.float64, (e2 * L,)), ("pde_label", np.float64, (e2 * L,))])
    rows = np.frombuffer(b, rec, P, 16)
    return arr, lab_deg[1], rows


def main():
    assert os.path.exists(REF_MAIN) and os.path.exists(REF_DUMP), "build oracle/_ref first (make -C oracle)"
    out = {"n_cases": np.int64(len(cases()))}
    for ci, (n, m, n_dup, n_labels, p, seed) in enumerate(cases()):
        g = synth.multigraph(n, m, n_dup=n_dup, n_labels=n_labels, seed=seed)
        rng = np.random.default_rng(seed)
        order = rng.permutation(n).astype(np.uint32)
        member = rng.integers(0, p, size=n).astype(np.uint32)
        with tempfile.TemporaryDirectory() as tmp:
            gp = os.path.join(tmp, "g.graph")
            synth.write_graph_file(gp, g)
            synth.make_dataset_dir(tmp, p)
            synth.write_membership(os.path.join(tmp, "gnn-pe", "membership.txt"), order, member)
            stdout = subprocess.check_output([REF_MAIN, "-f", tmp + "/", "-d", gp, "-m", "offline", "-p", str(p)], timeout=120)
            ap = os.path.join(tmp, "gnn-pe", "all_paths.txt")
            subprocess.check_call([REF_DUMP, gp, "2", os.path.join(tmp, "vde.bin"), ap, os.path.join(tmp, "pde.bin")])
            txt = open(ap).read().split()
            P = int(txt[0])
            paths = np.array(txt[1:], np.uint32).reshape(P, 3)
            arr, degree, rows = read_dump(os.path.join(tmp, "vde.bin"), os.path.join(tmp, "pde.bin"))
            assert np.array_equal(rows["vids"], paths)
            pre = f"c{ci}_"
            out.update({pre + "n": np.int64(n), pre + "labels": g["labels"], pre + "eu": g["eu"], pre + "ev": g["ev"],
                        pre + "order": order, pre + "member": member, pre + "p": np.int64(p), pre + "paths": paths,
                        pre + "x": arr[0], pre + "nx": arr[1], pre + "vde": arr[2], pre + "degree": degree,
                        pre + "pde": rows["pde"], pre + "pde_label": rows["pde_label"], pre + "pdeg": rows["degrees"],
                        pre + "stdout": np.frombuffer(stdout, np.uint8)})
            for j in range(p):
                t = open(os.path.join(tmp, "gnn-pe", "partitions", f"partition-{j}", "partition_paths.txt")).read().split()
                ids = np.array(t[1:], np.uint32)
                assert len(ids) == int(t[0])
                out[pre + f"part{j}"] = ids
            simple = len(np.unique(np.minimum(g["eu"], g["ev"]).astype(np.int64) * n + np.maximum(g["eu"], g["ev"])))
            print(f"case {ci}: n={n} lines={g['m']} distinct={simple} p={p} paths={P}")
    np.savez_compressed(os.path.join(HERE, "multigraph.npz"), **out)
    print("wrote", os.path.join(HERE, "multigraph.npz"), os.path.getsize(os.path.join(HERE, "multigraph.npz")), "bytes")


if __name__ == "__main__":
    main()
