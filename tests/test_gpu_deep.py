"""GPU tests of l=3 (4-vertex paths, BASELINE config 5).  The reference cannot run l != 2, so
parity here is UNPINNED: the checker is the oracle's restatement of the reference DFS with the depth fixed
(hash-set form and closed form, proven equal to each other on CPU in test_oracle_golden.py), plus the
properties the enumeration must have at any size."""
import numpy as np
import pytest

from conftest import small_cases
from gnnpe_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def binding():
    from gnnpe_amd import binding as b
    b.load()
    return b


def _engine(binding, g, sn, mem, p, e):
    eng = binding.Engine(0)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(sn, mem, p)
    nl = int(g["labels"].max()) + 1 if len(g["labels"]) else 1
    eng.set_label_table(binding.host_label_table(nl, e))
    return eng


def _checksum(ids, first_id=0):
    """numpy restatement of k_rows_checksum"""
    with np.errstate(over="ignore"):
        C1, C2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9)
        h = (np.arange(len(ids), dtype=np.uint64) + np.uint64(first_id)) * C1
        for k in range(ids.shape[1]):
            h ^= ids[:, k].astype(np.uint64) + C1 + (h << np.uint64(6)) + (h >> np.uint64(2))
            h *= C2
        return int((h ^ (h >> np.uint64(31))).sum(dtype=np.uint64))


@pytest.mark.parametrize("ci", range(6))
def test_small_graphs_l3_match_the_fixed_depth_dfs(binding, oracle, ci):
    import torch
    g = small_cases()[ci]
    sn = g["sorted_nodes"]
    e = 2
    eng = _engine(binding, g, sn, np.zeros(len(sn), np.uint32), 1, e)
    x, nx, vde = eng.vde()
    total, per_start = eng.count_paths(3, per_start=True)
    want = oracle.enumerate_dfs_hash(g["offsets"], g["nbrs"], sn, 4)  # the reference's dfs with the depth fixed
    assert total == len(want)
    assert total == oracle.count_p4(g["offsets"], g["nbrs"])[1]  # and the closed form that checks config 5 at full size
    assert np.array_equal(per_start, oracle.count_per_start(g["offsets"], g["nbrs"], sn, 4))
    ids, pde, pdl = eng.fill_paths(pde=True, pde_label=True)
    assert ids.shape == (total, 4) and np.array_equal(ids, want)
    opde, opdl, _, _ = oracle.gen_pde(want, e, g["offsets"], g["labels"], x, vde)
    assert np.array_equal(pde, opde) and np.array_equal(pdl, opdl)
    # arbitrary chunk boundaries give the same rows
    if total > 10:
        cuts = [0, 1, total // 3, total // 3 + 1, total - 1, total]
        got = np.concatenate([eng.fill_paths(a, b, pde=False)[0] for a, b in zip(cuts[:-1], cuts[1:])])
        assert np.array_equal(got, want)
    # checksum of chunks adds up to the checksum of the whole
    if total:
        t = torch.from_numpy(ids.view(np.int32)).cuda()
        whole = eng.rows_checksum_device(total, 4, t, 0)
        assert whole == _checksum(want)
        h = total // 2
        parts = eng.rows_checksum_device(h, 4, t[:h], 0) + eng.rows_checksum_device(total - h, 4, t[h:], h)
        assert parts % (1 << 64) == whole
    eng.close()


@pytest.mark.parametrize("e", [1, 5, 8])
def test_l3_wide_embeddings_and_partitions(binding, oracle, e):
    import torch
    g = synth.gnm_graph(400, 1600, n_labels=5, seed=11)
    sn = synth.degree_order(g["offsets"])
    mem = synth.block_membership(g["n"], 3)
    eng = _engine(binding, g, sn, mem, 3, e)
    x, nx, vde = eng.vde()
    total = eng.count_paths(3)
    want = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 4)
    assert total == len(want)
    ids, pde, pdl = eng.fill_paths(pde=True, pde_label=True)
    assert np.array_equal(ids, want)
    assert np.array_equal(pde, vde[want].reshape(total, 4 * e))
    assert np.array_equal(pdl, x[want].reshape(total, 4 * e))
    part = torch.empty(total, dtype=torch.int32, device="cuda")
    eng.path_partitions_device(0, total, part)
    eng.sync()
    assert np.array_equal(part.cpu().numpy().astype(np.uint32), mem[want[:, 0]])
    eng.close()


def test_l3_hub_rows_beyond_one_wave(binding, oracle):
    """degrees far above 64: several c-batches per pair and candidate rows longer than a wave"""
    g = synth.powerlaw_graph(1500, 9000, exponent=2.0, max_degree=400, n_labels=4, seed=3)
    assert np.diff(g["offsets"].astype(np.int64)).max() > 128
    rng = np.random.default_rng(5)
    sn = rng.permutation(g["n"]).astype(np.uint32)  # arbitrary processing order
    eng = _engine(binding, g, sn, np.zeros(g["n"], np.uint32), 1, 2)
    eng.vde(want=False)
    total, per_start = eng.count_paths(3, per_start=True)
    assert np.array_equal(per_start, oracle.count_per_start(g["offsets"], g["nbrs"], sn, 4))
    want = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 4)
    assert total == len(want)
    chunk = 1 << 18
    for b in range(0, total, chunk):
        ids, _, _ = eng.fill_paths(b, min(total, b + chunk), pde=False)
        assert np.array_equal(ids, want[b:b + chunk]), b
    eng.close()


def _star_with_chords(hub_degree, chords, seed):
    """vertex 0 joined to 1..hub_degree, plus random edges among the leaves: a middle vertex with more start vertices than the
    count kernel's LDS holds thresholds (kHistChunk = 512 per wave, kCoopChunk = 4 608 per workgroup), and a start vertex that is one of its own batch's third vertices"""
    rng = np.random.default_rng(seed)
    n = hub_degree + 1
    edges = {(0, v) for v in range(1, n)}
    while len(edges) < hub_degree + chords:
        a, b = rng.integers(1, n, 2)
        if a != b:
            edges.add((min(a, b), max(a, b)))
    ea = np.array(sorted(edges), dtype=np.int64)
    src = np.concatenate([ea[:, 0], ea[:, 1]])
    dst = np.concatenate([ea[:, 1], ea[:, 0]])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    offsets = np.zeros(n + 1, np.uint32)
    np.add.at(offsets, src + 1, 1)
    offsets = np.cumsum(offsets, dtype=np.uint64).astype(np.uint32)
    return dict(n=n, offsets=offsets, nbrs=dst.astype(np.uint32), labels=rng.integers(0, 4, n).astype(np.uint32))


@pytest.mark.parametrize("count", ["hist", "merge"])
def test_l3_both_count_kernels_against_the_checker(binding, oracle, monkeypatch, count):
    """k_deep3_count_hist / _coop (rows streamed into an LDS histogram over the thresholds; the default) and k_deep3_count_rows
    (one pointer per lane walked through row c) give every start vertex the checker's count and every path its slot: a
    power-law graph, a flat one, an arbitrary processing order, middle vertices with more neighbours than a wave's LDS holds
    thresholds (512: the workgroup form takes them) and than a workgroup's does (4 608: two passes)."""
    monkeypatch.setenv("GNNPE_DEEP_COUNT", count)
    cases = [(synth.powerlaw_graph(1500, 9000, exponent=2.0, max_degree=400, n_labels=4, seed=3), "degree"),
             (synth.gnm_graph(700, 4200, n_labels=5, seed=8), "random"),
             (_star_with_chords(2300, 2500, seed=2), "degree"),
             (_star_with_chords(1030, 900, seed=6), "random"),
             (_star_with_chords(5000, 1500, seed=4), "degree")]
    for g, how in cases:
        sn = synth.degree_order(g["offsets"]) if how == "degree" else np.random.default_rng(1).permutation(g["n"]).astype(np.uint32)
        eng = _engine(binding, g, sn, np.zeros(g["n"], np.uint32), 1, 2)
        eng.vde(want=False)
        total, per_start = eng.count_paths(3, per_start=True)
        assert np.array_equal(per_start, oracle.count_per_start(g["offsets"], g["nbrs"], sn, 4))
        want = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 4)
        assert total == len(want)
        chunk = 1 << 19
        for b in range(0, total, chunk):
            ids, _, _ = eng.fill_paths(b, min(total, b + chunk), pde=False)
            assert np.array_equal(ids, want[b:b + chunk]), b
        eng.close()


@pytest.mark.parametrize("mode", ["slices", "units"])
@pytest.mark.parametrize("e", [2, 3, 4, 8])
def test_l3_both_emit_paths_agree_with_the_checker(binding, oracle, monkeypatch, mode, e):
    """The library picks slices (one wave per 2048 candidates) on graphs with hub rows and one workgroup per unit elsewhere;
    both are forced here on a hub graph and on a flat one, every embedding-width specialisation of the row writer, whole
    and in ragged chunks that cut units and slices."""
    monkeypatch.setenv("GNNPE_DEEP_EMIT", mode)
    for g in (synth.powerlaw_graph(1200, 7000, exponent=2.0, max_degree=300, n_labels=6, seed=9),
              synth.gnm_graph(500, 2500, n_labels=6, seed=4)):
        sn = synth.degree_order(g["offsets"])
        eng = _engine(binding, g, sn, np.zeros(g["n"], np.uint32), 1, e)
        x, nx, vde = eng.vde()
        total = eng.count_paths(3)
        want = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 4)
        assert total == len(want)
        ids, pde, pdl = eng.fill_paths(pde=True, pde_label=True)
        assert np.array_equal(ids, want)
        assert np.array_equal(pde, vde[want].reshape(total, 4 * e))
        assert np.array_equal(pdl, x[want].reshape(total, 4 * e))
        cuts = [0, 1, 63, 64, 4097, total // 3, total // 2 + 7, total - 1, total]
        for a, b in zip(cuts[:-1], cuts[1:]):
            i2, p2, _ = eng.fill_paths(a, b, pde=True)
            assert np.array_equal(i2, want[a:b]), (a, b)
            assert np.array_equal(p2, vde[want[a:b]].reshape(b - a, 4 * e)), (a, b)
        eng.close()


def test_l3_properties_at_scale(binding):
    """size-independent properties on a graph the CPU checker would take minutes for: simple paths, real edges,
    rank[last] > rank[first], lexicographic order inside a start, counts consistent with the l=2 run"""
    import torch
    g = synth.gnm_graph(20000, 120000, n_labels=16, seed=21)
    sn = synth.degree_order(g["offsets"])
    rank = np.empty(g["n"], np.int64)
    rank[sn] = np.arange(g["n"])
    eng = _engine(binding, g, sn, np.zeros(g["n"], np.uint32), 1, 2)
    eng.vde(want=False)
    total, per_start = eng.count_paths(3, per_start=True)
    assert int(per_start.sum()) == total
    # total = number of 4-vertex simple paths (each undirected path once): sum over middle edges (b,c) of
    # (deg b - 1)(deg c - 1) minus 3 x triangles; checked through the emitted rows instead of a formula
    ids = torch.empty((total, 4), dtype=torch.int32, device="cuda")
    eng.fill_paths_device(0, total, ids, None, None)
    eng.sync()
    v = ids.cpu().numpy().astype(np.int64)
    assert np.all(rank[v[:, 3]] > rank[v[:, 0]])
    for a, b in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)):
        assert np.all(v[:, a] != v[:, b])
    off, nb = g["offsets"].astype(np.int64), g["nbrs"].astype(np.int64)
    edge_keys = np.repeat(np.arange(g["n"]), np.diff(off)) * g["n"] + nb
    for a, b in ((0, 1), (1, 2), (2, 3)):
        assert np.all(np.isin(v[:, a] * g["n"] + v[:, b], edge_keys))
    # emission order: starts in processing order; inside a start rows ascend lexicographically
    assert np.all(np.diff(rank[v[:, 0]]) >= 0)
    key = (v[:, 1] * g["n"] + v[:, 2]) * g["n"] + v[:, 3]
    same = v[1:, 0] == v[:-1, 0]
    assert np.all(np.diff(key)[same] > 0)
    # every row once: (s,b,c,d) unique and its reverse absent
    fwd = ((v[:, 0] * g["n"] + v[:, 1]) * g["n"] + v[:, 2]) * g["n"] + v[:, 3]
    rev = ((v[:, 3] * g["n"] + v[:, 2]) * g["n"] + v[:, 1]) * g["n"] + v[:, 0]
    assert len(np.unique(fwd)) == total and not np.isin(rev, fwd).any()
    # independent count: ordered simple 4-vertex walks / 2
    deg = np.diff(off)
    src = np.repeat(np.arange(g["n"]), deg)
    tri = 0
    nbr_sets = [set(nb[off[u]:off[u + 1]].tolist()) for u in range(g["n"])]
    for u, w in zip(src.tolist(), nb.tolist()):
        if u < w:
            tri += len(nbr_sets[u] & nbr_sets[w])
    # tri = sum over edges of common neighbours (3 x triangles): a path loses one choice per (middle edge, common neighbour)
    expect = int(((deg[src] - 1) * (deg[nb] - 1)).sum()) // 2 - tri
    assert total == expect
    eng.close()


def test_l3_counts_and_slots_beyond_32_bits(binding):
    """maximum sizes: more than 2^32 paths.  The count must equal the closed form over middle edges
    sum_{b,c} [(deg b - 1)(deg c - 1) - |N(b) & N(c)|], and rows emitted around slot 2^32 must be valid and
    consistent whatever the chunk boundaries are (64-bit offsets everywhere)."""
    import torch
    n, m = 4000, 800_000
    g = synth.gnm_graph(n, m, n_labels=8, seed=9)
    sn = synth.degree_order(g["offsets"])
    rank = np.empty(n, np.int64)
    rank[sn] = np.arange(n)
    off, nb = g["offsets"].astype(np.int64), g["nbrs"].astype(np.int64)
    deg = np.diff(off)
    src = np.repeat(np.arange(n), deg)
    A = np.zeros((n, n))
    A[src, nb] = 1.0
    common = float(((A @ A) * A).sum()) / 2.0  # sum over undirected edges of common neighbours (exact in f64)
    expect = int(((deg[src] - 1) * (deg[nb] - 1)).sum()) // 2 - int(round(common))
    assert expect > 1 << 32
    eng = _engine(binding, g, sn, np.zeros(n, np.uint32), 1, 2)
    eng.vde(want=False)
    total, per_start = eng.count_paths(3, per_start=True)
    assert total == expect and int(per_start.sum(dtype=np.uint64)) == total
    lo = (1 << 32) - 1500
    ids = torch.empty((3000, 4), dtype=torch.int32, device="cuda")
    eng.fill_paths_device(lo, lo + 3000, ids, None, None)
    eng.sync()
    v = ids.cpu().numpy().astype(np.int64)
    assert np.all(rank[v[:, 3]] > rank[v[:, 0]])
    for a, b in ((0, 1), (1, 2), (2, 3)):
        assert np.all(A[v[:, a], v[:, b]] == 1.0)
    for a, b in ((0, 2), (0, 3), (1, 3)):
        assert np.all(v[:, a] != v[:, b])
    # the same rows from two chunks split exactly at 2^32
    a = torch.empty((1500, 4), dtype=torch.int32, device="cuda")
    b = torch.empty((1500, 4), dtype=torch.int32, device="cuda")
    eng.fill_paths_device(lo, 1 << 32, a, None, None)
    eng.fill_paths_device(1 << 32, lo + 3000, b, None, None)
    eng.sync()
    assert torch.equal(torch.cat([a, b]), ids)
    # start vertex of slot 2^32 according to the per-start counts
    cs = np.cumsum(per_start.astype(np.uint64))
    i = int(np.searchsorted(cs, np.uint64(1 << 32), side="right"))
    assert v[1500, 0] == sn[i]
    eng.close()
