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
