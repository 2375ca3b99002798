"""Host-side slab planning (gnn-pe_amd/dist.py:plan_slabs): bounds are a partition of the processing order for any rank
count and weights; equal-path planning balances the estimated paths; the step-cost weights move work away from the last
slabs (the high-degree vertices own many adjacency entries and emit few paths each)."""
import numpy as np
import pytest

from gnnpe_amd import synth
from gnnpe_amd.dist import STEP_COST_WEIGHTS, _start_weights, owned_rows, plan_slabs


@pytest.fixture(scope="module")
def graph():
    g = synth.gnm_graph(20000, 200000, seed=11)
    return g, synth.degree_order(g["offsets"])


@pytest.mark.parametrize("ranks", [1, 2, 3, 8, 17])
@pytest.mark.parametrize("weights", [(1.0, 0.0, 0.0), STEP_COST_WEIGHTS, (1.0, 10.0, 0.0), (0.0, 0.0, 1.0)])
def test_bounds_partition_the_order(graph, ranks, weights):
    g, sn = graph
    b = plan_slabs(g["offsets"], sn, ranks, g["nbrs"], weights=weights)
    assert b.dtype == np.uint32 and len(b) == ranks + 1 and b[0] == 0 and b[-1] == g["n"]
    assert np.all(np.diff(b.astype(np.int64)) >= 0)
    rows = [owned_rows(g, sn, b, r)[0] for r in range(ranks)]
    assert np.array_equal(np.concatenate(rows), sn)  # every start vertex in exactly one slab, order kept


def test_equal_paths_and_cost_model(graph):
    g, sn = graph
    w, _ = _start_weights(g["offsets"], sn, g["nbrs"])
    eq = plan_slabs(g["offsets"], sn, 8, g["nbrs"])
    per = np.array([w[eq[r]:eq[r + 1]].sum() for r in range(8)])
    assert per.max() / per.mean() < 1.02
    cm = plan_slabs(g["offsets"], sn, 8, g["nbrs"], weights=STEP_COST_WEIGHTS)
    per_cm = np.array([w[cm[r]:cm[r + 1]].sum() for r in range(8)])
    assert per_cm[-1] < per[-1] and per_cm[0] > per_cm[-1]  # the last slab gives paths away
    deg = np.diff(g["offsets"].astype(np.int64))[sn.astype(np.int64)]
    own = np.array([deg[cm[r]:cm[r + 1]].sum() for r in range(8)])
    own_eq = np.array([deg[eq[r]:eq[r + 1]].sum() for r in range(8)])
    assert own[-1] < own_eq[-1]                              # ... and owns fewer adjacency entries than before


def test_degenerate_inputs():
    g = synth.gnm_graph(50, 0, seed=1)
    sn = synth.degree_order(g["offsets"])
    b = plan_slabs(g["offsets"], sn, 4, g["nbrs"], weights=STEP_COST_WEIGHTS)
    assert b[0] == 0 and b[-1] == 50 and np.all(np.diff(b.astype(np.int64)) >= 0)
    e = plan_slabs(np.zeros(1, np.uint32), np.zeros(0, np.uint32), 3, np.zeros(0, np.uint32))
    assert list(e) == [0, 0, 0, 0]
