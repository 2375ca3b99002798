"""Non-simple graph files (duplicate `e` lines), CPU side: the oracle against the compiled reference's outputs
(tests/golden/multigraph.npz, made by tests/golden/make_golden_multigraph.py) and the host loader's two views of such a file.
The reference stores the repeats (graph.cpp:211-218), counts them in `degree` and in gen_vde's neighbour sum (graph.h:154-156,
custom.h:527-534) and drops the repeated path in its hash set (custom.h:68-77)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from gnnpe_amd import binding, synth


def multigraph_cases():
    z = np.load(os.path.join(GOLDEN, "multigraph.npz"))
    out = []
    for ci in range(int(z["n_cases"])):
        pre = f"c{ci}_"
        c = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
        n = int(c["n"])
        offs, nbrs = synth._csr_from_edges(n, c["eu"].astype(np.int64), c["ev"].astype(np.int64))
        c["g"] = dict(n=n, m=len(c["eu"]), offsets=offs, nbrs=nbrs, labels=c["labels"], eu=c["eu"], ev=c["ev"])
        out.append(c)
    return out


def test_oracle_equals_the_reference_on_files_with_duplicate_lines(oracle):
    for c in multigraph_cases():
        g = c["g"]
        assert len(binding.simple_rows(g["offsets"], g["nbrs"])[1]) < len(g["nbrs"])  # every case has a repeated line
        # the reference's DFS + hash set on the rows AS STORED
        paths = oracle.enumerate_dfs_hash(g["offsets"], g["nbrs"], c["order"], 3)
        assert np.array_equal(paths, c["paths"])
        # ... is the closed form on the de-duplicated rows (what the engine enumerates)
        so, sn = binding.simple_rows(g["offsets"], g["nbrs"])
        assert np.array_equal(oracle.enumerate_closed(so, sn, c["order"], 3), c["paths"])
        x, nx, vde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
        assert np.array_equal(x, c["x"]) and np.array_equal(nx, c["nx"]) and np.array_equal(vde, c["vde"])
        assert np.array_equal(np.diff(g["offsets"].astype(np.int64)), c["degree"])  # graph.h:154-156: the stored row's length
        pde, pdl, _, pdeg = oracle.gen_pde(paths, 2, g["offsets"], g["labels"], x, vde)
        assert np.array_equal(pde, c["pde"]) and np.array_equal(pdl, c["pde_label"]) and np.array_equal(pdeg, c["pdeg"])
        for j in range(int(c["p"])):
            assert np.array_equal(np.nonzero(c["member"][paths[:, 0]] == j)[0], c[f"part{j}"])


def test_host_loader_keeps_the_repeats_and_offers_the_simple_rows(tmp_path, oracle):
    for c in multigraph_cases()[:6]:
        g = c["g"]
        gp = str(tmp_path / "g.graph")
        synth.write_graph_file(gp, g)
        h = binding.host_load_graph(gp, strict=False)
        assert np.array_equal(h["offsets"], g["offsets"]) and np.array_equal(h["nbrs"], g["nbrs"])
        so, sn = binding.simple_rows(g["offsets"], g["nbrs"])
        assert np.array_equal(h["simple_offsets"], so) and np.array_equal(h["simple_nbrs"], sn)
        offs, nbrs, _, meta = oracle.load_graph(gp)  # the restated loader holds the same rows
        assert np.array_equal(offs, g["offsets"]) and np.array_equal(nbrs, g["nbrs"])
        assert (h["labels_count"], h["max_degree"]) == (meta["labels_count"], meta["max_degree"])
        with pytest.raises(binding.GnnpeError, match="duplicate edge"):
            binding.host_load_graph(gp)  # strict: rounds 1-5
    # a simple file: no second view
    g = synth.gnm_graph(50, 120, n_labels=4, seed=5)
    gp = str(tmp_path / "s.graph")
    synth.write_graph_file(gp, g)
    h = binding.host_load_graph(gp, strict=False)
    assert h["simple_offsets"] is None and np.array_equal(h["nbrs"], g["nbrs"])


def test_self_loop_lines_are_refused_in_every_mode(tmp_path):
    """For `e u u` the reference writes one slot twice and leaves the next one uninitialised (graph.cpp:211-218); what it then
    enumerates depends on that memory (profiles/r06_selfloop_reference.txt).  Nothing to be identical to: refused."""
    g = synth.multigraph(30, 40, n_dup=0, n_loops=2, n_labels=4, seed=9)
    gp = str(tmp_path / "loop.graph")
    synth.write_graph_file(gp, g)
    for strict in (True, False):
        with pytest.raises(binding.GnnpeError, match="self-loop at vertex"):
            binding.host_load_graph(gp, strict=strict)
