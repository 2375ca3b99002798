"""SURVEY 8(f) row 4 -- the online FILTER (data side).  Golden vectors: query plans and candidate sets dumped from
the COMPILED reference by oracle/ref_online.cpp (tests/golden/make_golden_online.py) for its sample query and four
more query graphs cut out of the data graph.  CPU: the oracle's leaf-test restatement and the host query planner
against them.  GPU: the engine's filter against them, and the reference's own refinement on the engine's candidates."""
import json
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

ONLINE = os.path.join(GOLDEN, "online")
QUERIES = ["q0", "q1", "q2", "q3", "q4"]


def load_dump(name):
    b = open(os.path.join(ONLINE, f"{name}.bin"), "rb").read()
    nq, nqp, L, e = struct.unpack_from("<4I", b, 0)
    off = 16
    plan = dict(vids=[], labels=[], degrees=[], pde=[], pde_label=[])
    for _ in range(nqp):
        for key, dt, cnt in (("vids", np.uint32, L), ("labels", np.uint32, L), ("degrees", np.uint32, L),
                             ("pde", np.float64, e * L), ("pde_label", np.float64, e * L)):
            plan[key].append(np.frombuffer(b, dt, cnt, off))
            off += cnt * np.dtype(dt).itemsize
    plan = {k: np.array(v) for k, v in plan.items()}
    cand = []
    for _ in range(nq):
        c, = struct.unpack_from("<I", b, off)
        off += 4
        cand.append(np.frombuffer(b, np.uint32, c, off))
        off += 4 * c
    assert off == len(b)
    return nq, plan, cand


@pytest.fixture(scope="module")
def data_side(oracle, test_graph):
    g = test_graph
    paths = oracle.enumerate_closed(g["offsets"], g["nbrs"], g["sorted_nodes"], 3)
    x, nx, vde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
    return g, paths, vde


@pytest.mark.parametrize("name", QUERIES)
def test_oracle_leaf_test_equals_the_reference_traversal(oracle, data_side, name):
    """the R-tree traversal only prunes: its candidate sets equal the leaf test applied to every data path"""
    g, paths, vde = data_side
    nq, plan, want = load_dump(name)
    got = oracle.filter_candidates(paths, g["offsets"], g["labels"], vde, plan["vids"], plan["labels"], plan["degrees"],
                                   plan["pde"], nq)
    for u in range(nq):
        assert np.array_equal(got[u], want[u]), (name, u)


@pytest.mark.parametrize("name", QUERIES)
def test_host_query_plan_equals_the_reference_plan(name):
    """dfs_query + gen_vde + gen_query_pde restated in host C++ (same std::sort): plan identical to the dump, bit for bit"""
    from gnnpe_amd import binding
    nq, want, _ = load_dump(name)
    plan = binding.host_query_plan(os.path.join(ONLINE, f"{name}.graph"), 2)
    assert plan["n_vertices"] == nq
    for k in ("vids", "labels", "degrees", "pde"):
        assert np.array_equal(plan[k], want[k]), (name, k)


def test_host_query_plan_errors():
    from gnnpe_amd import binding
    with pytest.raises(FileNotFoundError):
        binding.host_query_plan("/nonexistent/q.graph", 2)


def _engine(binding, g, e):
    eng = binding.Engine(0)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(g["sorted_nodes"], g["membership"], 1)
    eng.set_label_table(binding.host_label_table(int(g["labels"].max()) + 1, e))
    eng.vde(want=False)
    eng.count_paths(2)
    return eng


@pytest.mark.gpu
@pytest.mark.parametrize("name", QUERIES)
def test_gpu_filter_equals_reference_candidates_and_refinement(oracle, test_graph, tmp_path, name):
    """engine filter (enumerate + leaf test on the GPU, no index, no files) == the reference's candidate sets;
    the reference's own refinement on the engine's candidates prints the reference's answer count"""
    from gnnpe_amd import binding
    from oracle import bitmap_to_sets, ref_online_path
    nq, _, want = load_dump(name)
    qpath = os.path.join(ONLINE, f"{name}.graph")
    plan = binding.host_query_plan(qpath, 2)
    eng = _engine(binding, test_graph, 2)
    bm, ms = eng.filter_candidates(plan)
    got = bitmap_to_sets(bm, eng.n)
    for u in range(nq):
        assert np.array_equal(got[u], want[u]), (name, u)
    assert ms > 0
    eng.close()
    if not os.path.exists(ref_online_path()):
        pytest.skip("oracle/_ref/ref_online not built")
    cand = str(tmp_path / "cand.bin")
    with open(cand, "wb") as f:
        f.write(struct.pack("<I", nq))
        for u in range(nq):
            f.write(struct.pack("<I", len(got[u])))
            f.write(got[u].astype("<u4").tobytes())
    graph = os.path.join(GOLDEN, "test_graph", "data_graph.graph")
    out = subprocess.check_output([ref_online_path(), str(tmp_path) + "/", graph, qpath, "1", "refine", cand], text=True)
    answers = json.load(open(os.path.join(ONLINE, "answers.json")))
    assert int(re.search(r"Answer Number: (\d+)", out).group(1)) == answers[name]


@pytest.mark.gpu
def test_gpu_filter_random_graphs_against_the_oracle(oracle):
    """other graphs, orders, embedding widths and epsilon edge: engine bitmap == oracle leaf test over every path"""
    from gnnpe_amd import binding, synth
    from oracle import bitmap_to_sets
    rng = np.random.default_rng(8)
    for e, n, m, nl in ((2, 800, 4000, 3), (3, 500, 3500, 2), (8, 300, 1500, 4)):
        g = synth.gnm_graph(n, m, n_labels=nl, seed=int(rng.integers(1 << 30)))
        g["sorted_nodes"] = rng.permutation(n).astype(np.uint32)
        g["membership"] = np.zeros(n, np.uint32)
        eng = _engine(binding, g, e)
        x, nx, vde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], e)
        paths = oracle.enumerate_closed(g["offsets"], g["nbrs"], g["sorted_nodes"], 3)
        deg = np.diff(g["offsets"].astype(np.int64))
        # query paths = sampled data paths (they match themselves), some with degrees / pde nudged across the thresholds
        pick = paths[rng.integers(0, len(paths), 40)]
        qv = np.arange(120, dtype=np.uint32).reshape(40, 3) % 17
        ql = g["labels"][pick]
        qd = deg[pick].astype(np.uint32)
        qd[::5] += 1  # one degree too many: the path itself no longer qualifies
        qp = vde[pick].reshape(40, 3 * e).copy()
        qp[1::4] += 5e-7  # inside epsilon: still dominated
        qp[2::4] += 2e-6  # outside epsilon
        plan = dict(n_vertices=17, vids=qv, labels=ql, degrees=qd, pde=qp)
        want = oracle.filter_candidates(paths, g["offsets"], g["labels"], vde, qv, ql, qd, qp, 17)
        assert sum(len(w) for w in want) > 0
        bm, _ = eng.filter_candidates(plan)
        got = bitmap_to_sets(bm, n)
        for u in range(17):
            assert np.array_equal(got[u], want[u]), (e, u)
        eng.close()
        # the filter needs the order and vde only -- no count, no emitted paths
        eng = binding.Engine(0)
        eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
        eng.set_order(g["sorted_nodes"], g["membership"], 1)
        eng.set_label_table(binding.host_label_table(int(g["labels"].max()) + 1, e))
        with pytest.raises(binding.GnnpeError):
            eng.filter_candidates(plan)  # vde missing
        eng.vde(want=False)
        got = bitmap_to_sets(eng.filter_candidates(plan)[0], n)
        assert all(np.array_equal(got[u], want[u]) for u in range(17))
        eng.close()


@pytest.mark.gpu
def test_cli_filter_mode_and_reference_refinement(tmp_path):
    """`gnnpe_main -m filter`: data graph + membership.txt + query graph -> candidates.bin; the reference's own
    refinement on that file prints the reference's answer (no all_paths.txt, no index.dat involved)"""
    from gnnpe_amd import synth
    from oracle import ref_online_path
    cli = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")
    graph = os.path.join(GOLDEN, "test_graph", "data_graph.graph")
    deg = np.array([int(l.split()[3]) for l in open(graph) if l.startswith("v")])
    tmp = str(tmp_path)
    synth.make_dataset_dir(tmp, 2)
    synth.write_membership(os.path.join(tmp, "gnn-pe", "membership.txt"), np.argsort(deg, kind="stable").astype(np.uint32),
                           (np.arange(len(deg)) % 2).astype(np.uint32))
    answers = json.load(open(os.path.join(ONLINE, "answers.json")))
    for name in ("q0", "q3"):
        qpath = os.path.join(ONLINE, f"{name}.graph")
        r = subprocess.run([cli, "-f", tmp + "/", "-d", graph, "-q", qpath, "-m", "filter", "-p", "2", "--timing"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        nq, plan, want = load_dump(name)
        assert r.stdout.splitlines()[-1] == str(len(plan["vids"]))  # the plan size, as gen_query_pde prints it
        b = open(os.path.join(tmp, "gnn-pe", "candidates.bin"), "rb").read()
        off = 4
        assert struct.unpack_from("<I", b, 0)[0] == nq
        for u in range(nq):
            c, = struct.unpack_from("<I", b, off)
            assert np.array_equal(np.frombuffer(b, np.uint32, c, off + 4), want[u])
            off += 4 + 4 * c
        if os.path.exists(ref_online_path()):
            out = subprocess.check_output([ref_online_path(), tmp + "/", graph, qpath, "2", "refine",
                                           os.path.join(tmp, "gnn-pe", "candidates.bin")], text=True)
            assert int(re.search(r"Answer Number: (\d+)", out).group(1)) == answers[name]


def _sets_to_bitmap(sets, n):
    bm = np.zeros((len(sets), (n + 31) // 32), np.uint32)
    for u, ids in enumerate(sets):
        ids = np.asarray(ids, np.int64)
        np.bitwise_or.at(bm[u], ids >> 5, (np.uint32(1) << (ids & 31).astype(np.uint32)))
    return bm


@pytest.mark.parametrize("name", QUERIES)
def test_host_refinement_reproduces_the_reference_answers(test_graph, name):
    """candidate sets dumped from the reference -> the product's own refinement -> the reference's answer count"""
    from gnnpe_amd import binding
    nq, _, cand = load_dump(name)
    answers = json.load(open(os.path.join(ONLINE, "answers.json")))
    qpath = os.path.join(ONLINE, f"{name}.graph")
    bm = _sets_to_bitmap(cand, len(test_graph["labels"]))
    assert binding.host_refine(test_graph, qpath, bm) == answers[name]
    assert binding.host_refine(test_graph, qpath, bm, limit=7) == min(7, answers[name])  # the reference's -n


def test_host_refinement_equals_reference_refinement_on_random_cases(oracle, tmp_path):
    """random graphs, random connected queries, candidate sets from the oracle filter (sometimes thinned so that the
    start-vertex rule matters): product refinement == oracle/_ref/ref_online ... refine"""
    from gnnpe_amd import binding, synth
    from oracle import ref_online_path
    sys_path = os.path.join(ROOT, "tests", "golden")
    import sys
    sys.path.insert(0, sys_path)
    from make_golden_online import cut_query
    if not os.path.exists(ref_online_path()):
        pytest.skip("oracle/_ref/ref_online not built")
    rng = np.random.default_rng(12)
    seen_answers = 0
    for trial in range(6):
        n = 400
        g = synth.gnm_graph(n, 2400, n_labels=3, seed=100 + trial)
        gp = str(tmp_path / f"g{trial}.graph")
        synth.write_graph_file(gp, g)
        qtext = cut_query(g["offsets"].astype(np.int64), g["nbrs"], g["labels"], int(rng.integers(3, 7)), rng)
        qp = str(tmp_path / f"q{trial}.graph")
        open(qp, "w").write(qtext)
        plan = binding.host_query_plan(qp, 2)
        sn = synth.degree_order(g["offsets"])
        paths = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
        x, nx, vde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
        cand = oracle.filter_candidates(paths, g["offsets"], g["labels"], vde, plan["vids"], plan["labels"],
                                        plan["degrees"], plan["pde"], plan["n_vertices"])
        if trial % 2:  # thin the sets: only the start vertex' set may change the count
            cand = [c[rng.random(len(c)) < 0.6] for c in cand]
        cf = str(tmp_path / f"c{trial}.bin")
        with open(cf, "wb") as f:
            f.write(struct.pack("<I", len(cand)))
            for c in cand:
                f.write(struct.pack("<I", len(c)) + np.asarray(c, "<u4").tobytes())
        out = subprocess.check_output([ref_online_path(), str(tmp_path) + "/", gp, qp, "1", "refine", cf], text=True)
        want = int(re.search(r"Answer Number: (\d+)", out).group(1))
        got = binding.host_refine(g, qp, _sets_to_bitmap(cand, n))
        assert got == want, (trial, got, want)
        seen_answers += want
    assert seen_answers > 0


@pytest.mark.gpu
def test_cli_online_mode_prints_the_reference_answer_line(tmp_path):
    """`gnnpe_main -m online`: GPU filter + host refinement from the data graph and membership.txt alone"""
    from gnnpe_amd import synth
    cli = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")
    graph = os.path.join(GOLDEN, "test_graph", "data_graph.graph")
    deg = np.array([int(l.split()[3]) for l in open(graph) if l.startswith("v")])
    tmp = str(tmp_path)
    synth.make_dataset_dir(tmp, 1)
    synth.write_membership(os.path.join(tmp, "gnn-pe", "membership.txt"), np.argsort(deg, kind="stable").astype(np.uint32),
                           np.zeros(len(deg), np.uint32))
    answers = json.load(open(os.path.join(ONLINE, "answers.json")))
    for name in QUERIES:
        r = subprocess.run([cli, "-f", tmp + "/", "-d", graph, "-q", os.path.join(ONLINE, f"{name}.graph"), "-m", "online",
                            "-p", "1"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        m = re.search(r"Answer Number: (\d+) Query Time \(ms\): ([0-9.e+-]+)", r.stdout)
        assert m and int(m.group(1)) == answers[name]
    r = subprocess.run([cli, "-f", tmp + "/", "-d", graph, "-q", os.path.join(ONLINE, "q0.graph"), "-m", "online", "-p", "1",
                        "-n", "1000"], capture_output=True, text=True)
    assert "Answer Number: 1000 " in r.stdout


@pytest.mark.gpu
def test_gpu_refinement_equals_host_refinement_and_reference_answers(oracle, test_graph, tmp_path):
    """gnnpe_refine (one thread per start candidate x neighbour slot) == host/refine.cpp == the reference's answers,
    with and without an answer limit, on the golden queries and on random graph / query / thinned-candidate cases"""
    from gnnpe_amd import binding, synth
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_online import cut_query
    answers = json.load(open(os.path.join(ONLINE, "answers.json")))
    eng = binding.Engine(0)
    eng.load_csr(test_graph["offsets"], test_graph["nbrs"], test_graph["labels"])
    for name in QUERIES:
        nq, _, cand = load_dump(name)
        bm = _sets_to_bitmap(cand, len(test_graph["labels"]))
        qpath = os.path.join(ONLINE, f"{name}.graph")
        got, ms = eng.refine(qpath, bm)
        assert got == answers[name] and ms > 0
        assert eng.refine(qpath, bm, limit=7)[0] == min(7, answers[name])
    eng.close()
    rng = np.random.default_rng(21)
    total = 0
    for trial in range(8):
        n = 600
        g = synth.gnm_graph(n, int(rng.integers(2000, 6000)), n_labels=int(rng.integers(1, 4)), seed=200 + trial)
        qp = str(tmp_path / f"q{trial}.graph")
        open(qp, "w").write(cut_query(g["offsets"].astype(np.int64), g["nbrs"], g["labels"], int(rng.integers(1, 8)), rng))
        nq = int(open(qp).readline().split()[1])
        cand = [np.flatnonzero(rng.random(n) < 0.5).astype(np.uint32) for _ in range(nq)]
        bm = _sets_to_bitmap(cand, n)
        eng = binding.Engine(0)
        eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
        want = binding.host_refine(g, qp, bm)
        assert eng.refine(qp, bm)[0] == want, trial
        assert eng.refine(qp, bm, limit=3)[0] == min(3, want)
        total += want
        eng.close()
    assert total > 0
