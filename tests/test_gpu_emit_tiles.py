"""The output-tile-driven emit shape (k_fill_tiles, gnnpe_set_emit_shape(ctx, 2)) against the oracle, bit for bit, and
against the start-vertex shape: every embedding width, both tile heights, chunk boundaries around tile edges, processing
orders that put long runs of empty pairs behind one tile (several strips per tile), the capped enqueue-only fill, BASELINE
config 2.  Reference: custom.h:66-92 (dfs) + 546-572 (gen_pde)."""
import os

import numpy as np
import pytest

from conftest import small_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def binding():
    from gnnpe_amd import binding as b
    b.load()
    return b


@pytest.fixture(autouse=True)
def _no_env_override(monkeypatch):
    monkeypatch.delenv("GNNPE_EMIT", raising=False)  # (read when a context is created)


def _engine(binding, g, sn, mem, p, e, shape=2):
    eng = binding.Engine(0)
    eng.set_emit_shape(shape)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(sn, mem, p)
    eng.set_label_table(binding.host_label_table(int(g["labels"].max()) + 1 if len(g["labels"]) else 1, e))
    return eng


@pytest.mark.parametrize("ci", range(6))
def test_small_graphs(binding, ci):
    c = small_cases()[ci]
    g = dict(offsets=c["offsets"], nbrs=c["nbrs"], labels=c["labels"])
    eng = _engine(binding, g, c["sorted_nodes"], c["membership"], 3, 2)
    x, nx, vde = eng.vde()
    total = eng.count_paths(2)
    ids, pde, pdl = eng.fill_paths(pde_label=True)
    ref = c["paths"].reshape(-1, 3)
    assert total == len(ref) and np.array_equal(ids, ref)
    assert np.array_equal(pde, vde[ref].reshape(len(ref), 6)) and np.array_equal(pdl, x[ref].reshape(len(ref), 6))
    hubs = int(np.diff(c["offsets"].astype(np.int64)).max()) > 64 if len(c["offsets"]) > 1 else False
    if total:
        assert eng.emit_kernel_name() == ("k_fill_ranked" if hubs else "k_fill_tiles")
    eng.close()


@pytest.mark.parametrize("e", [1, 2, 3, 4, 8])
def test_embedding_widths_and_chunks(binding, oracle, monkeypatch, e):
    from gnnpe_amd import synth
    g = synth.gnm_graph(900, 9000, n_labels=7, seed=70 + e)
    rng = np.random.default_rng(e)
    sn = rng.permutation(900).astype(np.uint32)  # a random order: the last start vertices hold runs of empty pairs
    eng = _engine(binding, g, sn, np.zeros(900, np.uint32), 1, e)
    x, nx, vde = eng.vde()
    total = eng.count_paths(2)
    ref = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    assert total == len(ref) > 20000
    ids, pde, pdl = eng.fill_paths(pde_label=True)
    assert eng.emit_kernel_name() == "k_fill_tiles"
    assert np.array_equal(ids, ref)
    assert np.array_equal(pde, vde[ref].reshape(len(ref), 3 * e)) and np.array_equal(pdl, x[ref].reshape(len(ref), 3 * e))
    for b_, e_ in [(0, 1), (0, 64), (1, 64), (63, 65), (64, 128), (127, 129), (100, 1000), (128, 129), (total - 1, total),
                   (total - 65, total), (5, total - 5), (4096, 4096)]:
        ci, cp, _ = eng.fill_paths(b_, e_)
        assert np.array_equal(ci, ref[b_:e_]), (b_, e_)
        assert np.array_equal(cp, vde[ref[b_:e_]].reshape(e_ - b_, 3 * e)), (b_, e_)
    only_ids, _, _ = eng.fill_paths(0, total, pde=False)
    assert np.array_equal(only_ids, ref)
    eng.close()


def test_many_empty_pairs_behind_one_tile(binding, oracle):
    """A processing order that ends on the high-degree vertices' neighbours: hundreds of consecutive pairs without a path
    (more than one strip of 64) between two rows of one tile."""
    from gnnpe_amd import synth
    g = synth.gnm_graph(3000, 45000, n_labels=5, seed=9)
    deg = np.diff(g["offsets"].astype(np.int64))
    assert deg.max() <= 64
    sn = np.argsort(-deg, kind="stable").astype(np.uint32)  # DEscending degree: late starts are small, early ones keep nearly all
    sn = np.concatenate([sn[1500:], sn[:1500]]).astype(np.uint32)  # ... and the big ones last: almost every pair of theirs is empty
    eng = _engine(binding, g, sn, np.zeros(3000, np.uint32), 1, 2)
    x, nx, vde = eng.vde()
    total = eng.count_paths(2)
    ref = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    assert total == len(ref)
    ids, pde, _ = eng.fill_paths()
    assert eng.emit_kernel_name() == "k_fill_tiles"
    assert np.array_equal(ids, ref) and np.array_equal(pde, vde[ref].reshape(len(ref), 6))
    eng.close()


def test_capped_enqueue_only_fill(binding, oracle):
    """The enqueue-only step (count without a read-back, fill clipped on the device) through the tile shape: the launch
    covers the buffer's capacity and the tiles past the count leave it untouched."""
    import torch
    from gnnpe_amd import synth
    g = synth.gnm_graph(2000, 20000, n_labels=4, seed=3)
    sn = synth.degree_order(g["offsets"])
    eng = _engine(binding, g, sn, np.zeros(2000, np.uint32), 1, 2)
    x, nx, vde = eng.vde()
    ref = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    dev = torch.device("cuda:0")
    for cap in (len(ref) + 1000, len(ref), len(ref) - 777):
        ids = torch.full((cap, 3), -1, dtype=torch.int32, device=dev)
        pde = torch.full((cap, 6), -1.0, dtype=torch.float64, device=dev)
        eng.count_paths_enqueue(2)
        eng.fill_paths_capped_device(cap, ids, pde)
        eng.sync()
        k = min(cap, len(ref))
        assert eng.emit_kernel_name() == "k_fill_tiles"
        assert np.array_equal(ids[:k].cpu().numpy().view(np.uint32), ref[:k])
        assert np.array_equal(pde[:k].cpu().numpy(), vde[ref[:k]].reshape(k, 6))
        assert bool((ids[k:] == -1).all()) and bool((pde[k:] == -1.0).all())
        assert eng.count_total() == len(ref)
    eng.close()


def test_config2_100k_1m_equals_the_oracle_and_the_start_shape(binding, oracle):
    from gnnpe_amd import synth
    g = synth.gnm_graph(100_000, 1_000_000)
    sn = synth.degree_order(g["offsets"])
    ox, onx, ovde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
    ref = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    out = {}
    for shape in (1, 2):
        eng = _engine(binding, g, sn, synth.block_membership(g["n"], 4), 4, 2, shape)
        eng.vde(want=False)
        assert eng.count_paths(2) == len(ref)
        ids, pde, _ = eng.fill_paths()
        assert eng.emit_kernel_name() == ("k_fill_ranked", "k_fill_tiles")[shape - 1]
        assert np.array_equal(ids, ref) and np.array_equal(pde, ovde[ref].reshape(len(ref), 6))
        out[shape] = (ids, pde)
        eng.close()
    assert np.array_equal(out[1][0], out[2][0]) and np.array_equal(out[1][1].view(np.uint64), out[2][1].view(np.uint64))
    # shape 0: whatever gnnpe_emit_calibrate_device measured faster into THESE buffers; other buffers keep the default
    import torch
    dev = torch.device("cuda:0")
    eng = _engine(binding, g, sn, synth.block_membership(g["n"], 4), 4, 2, 0)
    eng.vde(want=False)
    total = eng.count_paths(2)
    ids = torch.empty((total, 3), dtype=torch.int32, device=dev)
    pde = torch.empty((total, 6), dtype=torch.float64, device=dev)
    cal = eng.emit_calibrate_device(ids, pde)
    times = {"starts": cal["starts_ms"], "starts_low": cal["starts_low_ms"], "tiles": cal["tiles_ms"]}
    assert all(t > 0 for t in times.values()) and cal["kept"] in times
    assert times[cal["kept"]] == min(times.values())
    with pytest.raises(binding.GnnpeError):  # the buffers' capacity is checked: a count that does not fit is refused
        eng.emit_calibrate_device(ids, pde, rows_cap=total - 1)
    ids.zero_()
    eng.fill_paths_device(0, total, ids, pde, None)
    eng.sync()
    assert eng.emit_kernel_name() == eng.EMIT_SHAPE_KERNELS[cal["kept_shape"]]
    assert np.array_equal(ids.cpu().numpy().view(np.uint32), ref) and np.array_equal(pde.cpu().numpy(), ovde[ref].reshape(len(ref), 6))
    other = torch.empty((total, 3), dtype=torch.int32, device=dev)
    eng.fill_paths_device(0, total, other, None, None)
    eng.sync()
    assert eng.emit_kernel_name() == "k_fill_ranked" and torch.equal(other, ids)
    eng.close()


def test_start_vertices_from_ticket_counters(binding, oracle):
    """k_fill_ranked takes its start vertices in order from 16 ticket counters (the number of heads and the static assignment
    w, w + waves, ... are knobs of diagnostic builds: GNNPE_RANKED_TICKETS): the reference's rows on a graph with hub rows
    (degree > 64: streamed in id order by the same kernel), at both occupancies, whole and in chunks, and on a graph smaller
    than the grid (fewer waves than heads)."""
    from gnnpe_amd import synth
    g = synth.powerlaw_graph(3000, 30000, exponent=2.1, max_degree=300, n_labels=5, seed=11)
    assert int(np.diff(g["offsets"].astype(np.int64)).max()) > 64
    sn = synth.degree_order(g["offsets"])
    ref = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    for shape in (1, 4):
        eng = _engine(binding, g, sn, np.zeros(g["n"], np.uint32), 1, 2, shape)
        x, nx, vde = eng.vde()
        total = eng.count_paths(2)
        assert total == len(ref) > 100000
        ids, pde, _ = eng.fill_paths()
        assert eng.emit_kernel_name() == "k_fill_ranked"
        assert np.array_equal(ids, ref) and np.array_equal(pde, vde[ref].reshape(len(ref), 6))
        for b_, e_ in [(0, 1), (63, 65), (1000, 50000), (total - 3, total)]:
            ci, cp, _ = eng.fill_paths(b_, e_)
            assert np.array_equal(ci, ref[b_:e_]) and np.array_equal(cp, vde[ref[b_:e_]].reshape(e_ - b_, 6)), (shape, b_, e_)
        eng.close()
    # a graph smaller than the grid: fewer waves than heads
    g = synth.gnm_graph(40, 120, n_labels=3, seed=2)
    sn = synth.degree_order(g["offsets"])
    ref = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    eng = _engine(binding, g, sn, np.zeros(40, np.uint32), 1, 2, 1)
    eng.vde(want=False)
    assert eng.count_paths(2) == len(ref)
    ids, _, _ = eng.fill_paths(pde=False)
    assert np.array_equal(ids, ref)
    eng.close()


def test_the_step_is_four_launches_after_the_first():
    """Round 6: in a steady-state step (vde, count, fill on the same graph / order / slab) k_vde writes the count kernel's per-vertex
    records itself -- no k_pack_vinfo, no k_x_from_labels launch -- and the clears ride in the neighbouring kernels.  GNNPE_DEBUG=1
    (read when the context is created) says which path a count took: the first count packs, every later one finds the records."""
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import numpy as np, torch, gnnpe_amd\n"
        "from gnnpe_amd import binding, synth\n"
        "g = synth.gnm_graph(5000, 40000, n_labels=8, seed=4)\n"
        "eng = binding.Engine(0)\n"
        "eng.load_csr(g['offsets'], g['nbrs'], g['labels']); eng.set_order(synth.degree_order(g['offsets']), np.zeros(5000, np.uint32), 1)\n"
        "eng.set_label_table(binding.host_label_table(8, 2))\n"
        "tot = []\n"
        "for step in range(3):\n"
        "    eng.vde(want=False); tot.append(eng.count_paths(2)); ids, pde, _ = eng.fill_paths()\n"
        "x, nx, vde = eng.vde()\n"
        "assert len(set(tot)) == 1 and np.array_equal(pde, vde[ids].reshape(len(ids), 6)) and np.array_equal(x, binding.host_label_table(8, 2)[g['labels']])\n"
        "eng.count_paths(2)\n"
        "eng.set_slab(100, 4000); eng.vde(want=False); eng.count_paths(2)  # a new slab: the pair offsets are rebuilt, so is the packing\n"
        "eng.vde(want=False); eng.count_paths(2)\n"
        "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, GNNPE_DEBUG="1"), cwd=ROOT, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    how = [ln.split(": ")[1] for ln in r.stderr.splitlines() if ln.startswith("[count] vertex records")]
    # three steps; a count behind the vde that fetched the arrays; the two counts around the new slab
    assert how == ["k_pack_vinfo", "written by k_vde", "written by k_vde", "written by k_vde", "k_pack_vinfo", "written by k_vde"], how
