"""The oracle's l = 3 helpers for config-5 sizes (orc_count_per_start_l3, orc_enumerate_starts) against the oracle's own plain
closed-form DFS (cf_rec) on graphs that DFS can walk.  l = 3 has no reference that runs: parity unpinned; what is
pinned here is only that the fast count is the DFS's count."""
import numpy as np
import pytest

from gnnpe_amd import synth


@pytest.mark.parametrize("kind", ["gnm", "powerlaw", "dense", "star"])
def test_fast_l3_counts_are_the_dfs_counts(oracle, kind):
    rng = np.random.default_rng(5)
    if kind == "gnm":
        g = synth.gnm_graph(400, 2400, n_labels=4, seed=2)
    elif kind == "powerlaw":
        g = synth.powerlaw_graph(900, 5000, exponent=2.0, max_degree=250, n_labels=4, seed=3)
    elif kind == "dense":
        g = synth.gnm_graph(60, 1200, n_labels=3, seed=4)
    else:  # one hub, a few chords: rows much longer than their neighbours' (the search branch)
        n = 700
        eu = np.concatenate([np.zeros(n - 1, np.int64), rng.integers(1, n, 40)])
        ev = np.concatenate([np.arange(1, n, dtype=np.int64), rng.integers(1, n, 40)])
        ok = eu != ev
        key = np.unique(np.minimum(eu, ev)[ok] * n + np.maximum(eu, ev)[ok])
        offs, nbrs = synth._csr_from_edges(n, key // n, key % n)
        g = dict(n=n, offsets=offs, nbrs=nbrs)
    for order in (synth.degree_order(g["offsets"]), rng.permutation(g["n"]).astype(np.uint32)):
        want = oracle.count_per_start(g["offsets"], g["nbrs"], order, 4)
        got = oracle.count_per_start_l3(g["offsets"], g["nbrs"], order)
        assert np.array_equal(got, want)
        assert int(want.sum()) == oracle.count_p4(g["offsets"], g["nbrs"])[1]
        # the per-start enumerator writes the DFS's rows of the chosen start vertices
        rows = oracle.enumerate_closed(g["offsets"], g["nbrs"], order, 4)
        base = np.concatenate([[0], np.cumsum(want)]).astype(np.int64)
        for first, cnt in ((0, 7), (g["n"] // 2, 21), (g["n"] - 9, 9)):
            part = oracle.enumerate_starts(g["offsets"], g["nbrs"], order, 4, first, want[first:first + cnt])
            assert np.array_equal(part, rows[base[first]:base[first + cnt]])
        with pytest.raises(ValueError):
            bad = want[:5].copy()
            bad[2] += 1
            oracle.enumerate_starts(g["offsets"], g["nbrs"], order, 4, 0, bad)
