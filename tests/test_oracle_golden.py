"""Pin the CPU oracle (oracle/gnnpe_oracle.c) to golden vectors produced by the compiled
reference (tests/golden/make_golden.py).  CPU only."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, small_cases


def _md5(b):
    return hashlib.md5(b).hexdigest()


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(GOLDEN, "test_graph", "golden.json")))


def test_loader_metadata(test_graph):
    # printGraphMetaData of the reference on Test/: |V| 3112 |E| 12519 |Sigma| 71, max deg 168, max label freq 622
    m = test_graph["meta"]
    assert (m["n"], m["m"], m["labels_count"], m["max_degree"], m["max_label_freq"]) == (3112, 12519, 71, 168, 622)
    offs, nbrs = test_graph["offsets"], test_graph["nbrs"]
    assert offs[-1] == 2 * m["m"]
    for v in (0, 1, 2, 1745):
        seg = nbrs[offs[v]:offs[v + 1]]
        assert np.all(np.diff(seg.astype(np.int64)) > 0)


@pytest.mark.parametrize("which", ["closed", "dfs_hash"])
def test_all_paths_bytes_match_reference(oracle, test_graph, gold, which):
    fn = oracle.enumerate_closed if which == "closed" else oracle.enumerate_dfs_hash
    paths = fn(test_graph["offsets"], test_graph["nbrs"], test_graph["sorted_nodes"], 3)
    assert paths.shape == (gold["p1"]["header"], 3) == (415545, 3)
    txt = oracle.format_all_paths(paths)
    assert len(txt) == gold["p1"]["all_paths_bytes"]
    assert _md5(txt) == gold["p1"]["all_paths_md5"] == gold["p2"]["all_paths_md5"]
    ref = gzip.open(os.path.join(GOLDEN, "test_graph", "all_paths.txt.gz")).read()
    assert txt == ref
    assert txt.split(b"\n")[1:5] == [r.encode() for r in gold["p1"]["first_rows"]]


def test_partition_paths_match_reference(oracle, test_graph, gold, tmp_path):
    paths = oracle.enumerate_closed(test_graph["offsets"], test_graph["nbrs"], test_graph["sorted_nodes"], 3)
    n = test_graph["meta"]["n"]
    p = str(tmp_path / "pp.txt")
    oracle.write_partition_paths(p, paths, np.zeros(n, np.uint32), 0)
    assert _md5(open(p, "rb").read()) == gold["p1"]["partition_paths_md5"][0]
    mem = (np.arange(n) % 2).astype(np.uint32)
    for pid in range(2):
        oracle.write_partition_paths(p, paths, mem, pid)
        b = open(p, "rb").read()
        assert _md5(b) == gold["p2"]["partition_paths_md5"][pid]
        ids = [int(x) for x in b.split()[1:5]]
        assert ids == gold["p2"]["partition_first_ids"][pid]
        assert int(b.split()[0]) == gold["p2"]["partition_sizes"][pid]


def test_writer_file_equals_formatter(oracle, tmp_path):
    paths = np.array([[0, 10, 4294967295], [7, 8, 9]], np.uint32)
    p = str(tmp_path / "a.txt")
    oracle.write_all_paths(p, paths)
    assert open(p, "rb").read() == oracle.format_all_paths(paths) == b"2\n0 10 4294967295 \n7 8 9 \n"


def test_label_table_bit_exact(oracle):
    z = np.load(os.path.join(GOLDEN, "label_table.npz"))
    for e in (1, 2, 3, 8):
        ref = z[f"e{e}"]
        got = oracle.label_table(ref.shape[0], e)
        assert np.array_equal(got, ref), f"gen_vde_x differs at e={e}"
    # SURVEY 8(c)(3) known answers
    assert oracle.gen_vde_x(0, 2).tolist() == [0.41252546269145579, 0.58747453730854426]
    assert oracle.gen_vde_x(4, 2).tolist() == [0.83910125293091886, 0.16089874706908117]


@pytest.mark.parametrize("e", [2, 8])
def test_gen_vde_bit_exact(oracle, test_graph, e):
    z = np.load(os.path.join(GOLDEN, "test_graph", f"vde_e{e}.npz"))
    x, nx, vde = oracle.gen_vde(test_graph["offsets"], test_graph["nbrs"], test_graph["labels"], e)
    assert np.array_equal(z["label"], test_graph["labels"])
    assert np.array_equal(z["degree"], np.diff(test_graph["offsets"]))
    assert np.array_equal(x, z["x"]) and np.array_equal(nx, z["nx"]) and np.array_equal(vde, z["vde"])
    if e == 2:
        assert vde[1].tolist() == [2.7618965678486669, 7.2381034321513331]


def test_gen_pde_sample_bit_exact(oracle, test_graph):
    z = np.load(os.path.join(GOLDEN, "test_graph", "pde_sample_e2.npz"))
    paths = oracle.enumerate_closed(test_graph["offsets"], test_graph["nbrs"], test_graph["sorted_nodes"], 3)
    x, nx, vde = oracle.gen_vde(test_graph["offsets"], test_graph["nbrs"], test_graph["labels"], 2)
    pde, pdl, pl, pd = oracle.gen_pde(paths, 2, test_graph["offsets"], test_graph["labels"], x, vde)
    idx = z["index"]
    assert np.array_equal(paths[idx], z["vids"])
    assert np.array_equal(pl[idx], z["labels"]) and np.array_equal(pd[idx], z["degrees"])
    assert np.array_equal(pde[idx], z["pde"]) and np.array_equal(pdl[idx], z["pde_label"])


@pytest.mark.parametrize("ci", range(6))
def test_small_graphs_match_reference(oracle, ci):
    c = small_cases()[ci]
    for fn in (oracle.enumerate_closed, oracle.enumerate_dfs_hash):
        paths = fn(c["offsets"], c["nbrs"], c["sorted_nodes"], 3)
        assert np.array_equal(paths, c["paths"].reshape(-1, 3))
    txt = oracle.format_all_paths(paths)
    assert _md5(txt) == bytes(c["all_paths_md5"]).decode()
    starts = paths[:, 0]
    for pid in range(3):
        ids = np.nonzero(c["membership"][starts] == pid)[0]
        assert np.array_equal(ids.astype(np.uint64), c[f"part{pid}"])
    x, nx, vde = oracle.gen_vde(c["offsets"], c["nbrs"], c["labels"], 2)
    assert np.array_equal(vde, c["vde"]) and np.array_equal(nx, c["nx"])
    counts = oracle.count_per_start(c["offsets"], c["nbrs"], c["sorted_nodes"], 3)
    assert counts.sum() == len(paths)
    # per-start counts agree with the emitted rows
    rank = np.empty(len(c["sorted_nodes"]), np.int64)
    rank[c["sorted_nodes"]] = np.arange(len(rank))
    assert np.array_equal(np.bincount(rank[starts], minlength=len(rank)), counts.astype(np.int64))


def test_closed_form_count_formula(oracle, test_graph):
    # P = sum_v C(deg v, 2) on a simple graph
    from gnnpe_amd import synth
    assert synth.expected_paths_l2(test_graph["offsets"]) == 415545


def test_edge_cases(oracle):
    # empty graph, isolated vertices, a single edge, a triangle
    offs = np.zeros(5, np.uint32)
    assert oracle.enumerate_closed(offs, np.zeros(0, np.uint32), np.arange(4, dtype=np.uint32), 3).shape == (0, 3)
    assert oracle.format_all_paths(np.zeros((0, 3), np.uint32)) == b"0\n"
    # triangle 0-1-2: 3 paths, each unordered pair of neighbours around a middle vertex once
    offs = np.array([0, 2, 4, 6], np.uint32)
    nbrs = np.array([1, 2, 0, 2, 0, 1], np.uint32)
    sn = np.array([2, 0, 1], np.uint32)
    a = oracle.enumerate_closed(offs, nbrs, sn, 3)
    b = oracle.enumerate_dfs_hash(offs, nbrs, sn, 3)
    assert np.array_equal(a, b) and a.tolist() == [[2, 0, 1], [2, 1, 0], [0, 2, 1]]
    # l=3 extension (4-vertex paths): closed form == hash-set DFS on a small graph
    from gnnpe_amd import synth
    g = synth.gnm_graph(30, 70, n_labels=3, seed=5)
    sn = synth.degree_order(g["offsets"])
    a = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 4)
    b = oracle.enumerate_dfs_hash(g["offsets"], g["nbrs"], sn, 4)
    assert len(a) > 0 and np.array_equal(a, b)


@pytest.mark.parametrize("kind", ["gnm", "dense", "powerlaw", "triangle", "star"])
def test_p4_closed_form_equals_the_fixed_depth_dfs(oracle, kind):
    """The independent count used for config 5 at full size (tests/test_gpu_slabs_full.py): number of 4-vertex simple
    paths = sum_E (du - 1)(dv - 1) - 3 T, against the oracle's restatement of the reference DFS (custom.h:66-92) with the
    depth fixed to 4 vertices -- hash-set form and closed form -- on graphs small enough to enumerate."""
    from gnnpe_amd import synth
    if kind == "gnm":
        g = synth.gnm_graph(600, 3000, n_labels=7, seed=31)
    elif kind == "dense":  # many triangles
        g = synth.gnm_graph(60, 900, n_labels=3, seed=2)
    elif kind == "powerlaw":
        g = synth.powerlaw_graph(3000, 20000, exponent=2.1, max_degree=200, seed=5)
    elif kind == "triangle":
        g = dict(n=3, offsets=np.array([0, 2, 4, 6], np.uint32), nbrs=np.array([1, 2, 0, 2, 0, 1], np.uint32))
    else:  # a star has no 4-vertex path at all
        g = dict(n=6, offsets=np.array([0, 5, 6, 7, 8, 9, 10], np.uint32), nbrs=np.array([1, 2, 3, 4, 5, 0, 0, 0, 0, 0], np.uint32))
    sn = synth.degree_order(g["offsets"])
    tri, p4 = oracle.count_p4(g["offsets"], g["nbrs"])
    assert p4 == len(oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 4))
    if kind in ("dense", "triangle", "star"):
        assert p4 == len(oracle.enumerate_dfs_hash(g["offsets"], g["nbrs"], sn, 4))
    # triangles against a dense count
    n = g["n"]
    A = np.zeros((n, n))
    A[np.repeat(np.arange(n), np.diff(g["offsets"].astype(np.int64))), g["nbrs"]] = 1.0
    assert tri == int(round(np.trace(A @ A @ A) / 6.0))
    if kind == "triangle":
        assert (tri, p4) == (1, 0)


def test_all_core_port_equals_the_sequential_restatement(oracle):
    """bench.py's all-core CPU baseline (closed form, OpenMP) must produce exactly what the sequential oracle does."""
    from gnnpe_amd import synth
    g = synth.gnm_graph(3000, 20000, n_labels=11, seed=4)
    rng = np.random.default_rng(4)
    sn = rng.permutation(3000).astype(np.uint32)
    for e in (2, 5):
        P, vde, so, ids, pde = oracle.offline_parallel(g["offsets"], g["nbrs"], g["labels"], sn, e, threads=4)
        want = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
        x, nx, ovde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], e)
        assert P == len(want) and np.array_equal(ids, want)
        assert np.array_equal(vde, ovde)
        assert np.array_equal(pde, ovde[want].reshape(P, 3 * e))
        assert np.array_equal(np.diff(so), oracle.count_per_start(g["offsets"], g["nbrs"], sn, 3))
