"""Output pool (gnnpe_output_pool_*, gnn-pe_amd/csrc/gnnpe_pool.hip): the emitted rows in pool memory are the rows a
plain buffer receives, whichever candidate allocation the pool kept; the report lists every candidate."""
import numpy as np
import pytest
import torch

from gnnpe_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def binding():
    from gnnpe_amd import binding as b
    b.load()
    return b


@pytest.mark.parametrize("candidates,with_count", [(1, True), (5, True), (3, False)])
def test_pool_rows_equal_plain_rows(binding, oracle, candidates, with_count, monkeypatch):
    monkeypatch.setenv("GNNPE_TESTING", "pool_min_probe_bytes=0")  # the draw is for multi-GiB outputs; let a small graph exercise it
    g = synth.gnm_graph(20000, 160000, n_labels=16, seed=3)
    sn = synth.degree_order(g["offsets"])
    eng = binding.Engine(0)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(16, 2))
    x, nx, vde = eng.vde()
    want = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    total = len(want)
    if with_count:
        assert eng.count_paths(2) == total
    pool = binding.OutputPool(eng, total + 11, 3, 6, candidates=candidates)
    rep = pool.report()
    assert len(rep["candidates_ms"]) == candidates and 0 <= rep["kept"] < candidates
    assert rep["probe"] == ("emit kernel" if with_count and candidates > 1 else "streaming write") or candidates == 1
    if candidates > 1:
        assert all(ms > 0 for ms in rep["candidates_ms"]) and rep["candidates_ms"][rep["kept"]] == min(rep["candidates_ms"])
    if not with_count:
        assert eng.count_paths(2) == total
    dev = torch.device("cuda", 0)
    eng.fill_paths_device(0, total, pool.ids, pool.pde, None)
    eng.sync()
    ids, pde = pool.ids_tensor(dev), pool.pde_tensor(dev)
    assert ids.shape == (total + 11, 3) and pde.shape == (total + 11, 6)
    assert np.array_equal(ids[:total].cpu().numpy().view(np.uint32), want)
    assert np.array_equal(pde[:total].cpu().numpy(), vde[want].reshape(total, 6))
    if candidates > 1:  # the enqueue-only step writes into pool memory as well
        ids.zero_()
        eng.vde(want=False)
        eng.count_paths_enqueue(2)
        eng.fill_paths_capped_device(pool.rows_cap, pool.ids, pool.pde)
        assert eng.count_total() == total
        assert np.array_equal(pool.pde_tensor(dev)[:total].cpu().numpy(), vde[want].reshape(total, 6))
        assert np.array_equal(ids[:total].cpu().numpy().view(np.uint32), want) and bool((ids[total:] == 0).all())
    with pytest.raises(binding.GnnpeError):  # a tensor still shares the pool's memory
        pool.close()
    del ids, pde
    pool.close()
    eng.close()


def test_small_outputs_are_not_probed(binding):
    eng = binding.Engine(0)
    pool = binding.OutputPool(eng, 1000, 3, 6, candidates=5)
    assert len(pool.report()["candidates_ms"]) == 1 and pool.ids and pool.pde
    pool.close()
    eng.close()


def test_pool_rejects_bad_arguments(binding):
    eng = binding.Engine(0)
    with pytest.raises(binding.GnnpeError):
        binding.OutputPool(eng, 1000, 3, 6, candidates=0)
    with pytest.raises(binding.GnnpeError):
        binding.OutputPool(eng, 1000, 0, 6, candidates=2)
    eng.close()


def test_pool_probes_with_the_l3_emission_too(binding, oracle, monkeypatch):
    """4-vertex paths: the pool's probe is whichever emit path the context's count selects (here the l = 3 one)"""
    monkeypatch.setenv("GNNPE_TESTING", "pool_min_probe_bytes=0")
    g = synth.gnm_graph(600, 3000, n_labels=5, seed=8)
    sn = synth.degree_order(g["offsets"])
    eng = binding.Engine(0)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(5, 2))
    x, nx, vde = eng.vde()
    want = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 4)
    assert eng.count_paths(3) == len(want)
    pool = binding.OutputPool(eng, len(want), 4, 8, candidates=3)
    assert pool.report()["probe"] == "emit kernel"
    eng.fill_paths_device(0, len(want), pool.ids, pool.pde, None)
    eng.sync()
    dev = torch.device("cuda", 0)
    assert np.array_equal(pool.ids_tensor(dev).cpu().numpy().view(np.uint32), want)
    assert np.array_equal(pool.pde_tensor(dev).cpu().numpy(), vde[want].reshape(len(want), 8))
    with pytest.raises(binding.GnnpeError):
        binding.OutputPool(eng, 1 << 41, 4, 8, candidates=1)
    pool.close()
    eng.close()
