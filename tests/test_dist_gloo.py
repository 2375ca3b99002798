"""world_size-2/3 gloo tests of the slab-partitioned build (gnn-pe_amd/dist.py): the real driver
code runs its all-to-all-v halo exchange and all-gathers over gloo on CPU, with the local kernels
replaced by an oracle-backed engine (tests/fake_engine.py).  The concatenated per-rank outputs must
equal the single-rank reference output for every world size (SURVEY 8(e) invariant)."""
import os
import pickle
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gnnpe_amd import synth
from gnnpe_amd.dist import SlabBuild, owned_rows, plan_slabs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _query_plan(g):
    """a plan made of a few data paths (they match themselves) -- same on every rank"""
    from oracle import Oracle
    orc = Oracle()
    sn = synth.degree_order(g["offsets"])
    paths = orc.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    x, nx, vde = orc.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
    pick = paths[np.random.default_rng(3).integers(0, len(paths), 12)]
    deg = np.diff(g["offsets"].astype(np.int64))
    return dict(n_vertices=9, vids=(np.arange(36, dtype=np.uint32).reshape(12, 3) % 9), labels=g["labels"][pick],
                degrees=deg[pick].astype(np.uint32), pde=vde[pick].reshape(12, 6))


def _worker(rank, world, port, out_dir, bounds, l=2, force=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_engine import FakeEngine
    from oracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = synth.gnm_graph(600, 3000, n_labels=7, seed=31)
    sn = synth.degree_order(g["offsets"])
    rows, roff, rnbr = owned_rows(g, sn, bounds, rank)
    eng = FakeEngine(Oracle(), g["n"], g["labels"], rows, roff, rnbr, sn, 2)
    eng.set_slab(int(bounds[rank]), int(bounds[rank + 1]))
    sb = SlabBuild(eng, g["n"], 2, bounds, rank, world, torch.device("cpu"), nbr_capacity=2 * g["m"],
                   owned_entries=int(roff[-1]), l=l, force_collectives=force)
    assert sb.dist_on
    L = l + 1
    res = []
    for rep in range(2):  # the step is repeatable
        total, base = sb.step()
        ids = torch.zeros((max(total, 1), L), dtype=torch.int32)
        pde = torch.zeros((max(total, 1), 2 * L), dtype=torch.float64)
        if rep == 1 and l == 2:
            # the step with no host round trip before the fill: outputs sized by the earlier pass, totals collected after
            sb.step_enqueue(ids, pde, ids.shape[0])
            base2 = sb.count_end()
            total2 = sb.local_total
        else:
            total2, base2 = sb.step(ids, pde)
        assert (total2, base2) == (total, base)
        res.append(dict(total=total, base=base, global_total=sb.global_total, ids=ids[:total].numpy(),
                        pde=pde[:total].numpy(), stats=dict(sb.stats)))
    if l == 2:  # online filter across ranks: local leaf tests, bitmaps OR-ed
        eng.set_degrees(np.diff(g["offsets"].astype(np.int64)))
        res[-1]["filter"] = sb.filter(_query_plan(g))
    with open(os.path.join(out_dir, f"r{rank}.pkl"), "wb") as f:
        pickle.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "planned"), (3, "planned"), (2, "empty_slab")])
def test_slab_build_equals_single_rank(oracle, tmp_path, world, kind):
    g = synth.gnm_graph(600, 3000, n_labels=7, seed=31)
    sn = synth.degree_order(g["offsets"])
    bounds = plan_slabs(g["offsets"], sn, world)
    if kind == "empty_slab":
        bounds = np.array([0, 0, g["n"]], np.uint32)
    assert bounds[0] == 0 and bounds[-1] == g["n"] and np.all(np.diff(bounds.astype(np.int64)) >= 0)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), bounds), nprocs=world, join=True)
    ref_ids = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    x, nx, vde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
    res = [pickle.load(open(tmp_path / f"r{r}.pkl", "rb")) for r in range(world)]
    for rep in range(2):
        parts = [res[r][rep] for r in range(world)]
        assert [p["base"] for p in parts] == list(np.cumsum([0] + [p["total"] for p in parts[:-1]]))
        assert all(p["global_total"] == len(ref_ids) for p in parts)
        ids = np.concatenate([p["ids"] for p in parts]).astype(np.uint32)
        pde = np.concatenate([p["pde"] for p in parts])
        assert np.array_equal(ids, ref_ids)
        assert np.array_equal(pde, vde[ref_ids].reshape(len(ref_ids), 6))
    # online filter: every rank ends with the same OR-ed bitmap = the single-rank leaf test over all paths
    plan = _query_plan(g)
    want = oracle.filter_candidates(ref_ids, g["offsets"], g["labels"], vde, plan["vids"], plan["labels"], plan["degrees"],
                                    plan["pde"], 9)
    from oracle import bitmap_to_sets
    for r in range(world):
        got = bitmap_to_sets(res[r][-1]["filter"], g["n"])
        assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert sum(len(w) for w in want) > 0
    if kind == "planned":  # partitioning must actually move rows between ranks
        assert all(res[r][0]["stats"]["halo_rows"] > 0 for r in range(world))


def test_one_rank_group_runs_the_collective_step(oracle, tmp_path):
    """world_size 1 with force_collectives: the N > 1 step (halo plan, all-to-all-v, vde all-gather, totals all-gather)
    over a 1-rank group -- what the single-GPU box runs over RCCL (tests/test_gpu_rccl.py), here over gloo."""
    g = synth.gnm_graph(600, 3000, n_labels=7, seed=31)
    sn = synth.degree_order(g["offsets"])
    bounds = np.array([0, g["n"]], np.uint32)
    mp.spawn(_worker, args=(1, _free_port(), str(tmp_path), bounds, 2, True), nprocs=1, join=True)
    ref_ids = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    x, nx, vde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
    res = pickle.load(open(tmp_path / "r0.pkl", "rb"))
    for rep in res:
        assert rep["base"] == 0 and rep["total"] == rep["global_total"] == len(ref_ids)
        assert np.array_equal(rep["ids"].astype(np.uint32), ref_ids)
        assert np.array_equal(rep["pde"], vde[ref_ids].reshape(len(ref_ids), 6))


def test_slab_build_l3_needs_the_second_hop(oracle, tmp_path):
    """4-vertex paths: the rows two hops from the slab travel in a second exchange round; without it the
    per-rank counts fall short of the single-rank enumeration."""
    world = 2
    g = synth.gnm_graph(600, 3000, n_labels=7, seed=31)
    sn = synth.degree_order(g["offsets"])
    bounds = plan_slabs(g["offsets"], sn, world)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), bounds, 3), nprocs=world, join=True)
    ref_ids = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 4)
    x, nx, vde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
    res = [pickle.load(open(tmp_path / f"r{r}.pkl", "rb")) for r in range(world)]
    parts = [res[r][1] for r in range(world)]
    ids = np.concatenate([p["ids"] for p in parts]).astype(np.uint32)
    assert np.array_equal(ids, ref_ids)
    assert np.array_equal(np.concatenate([p["pde"] for p in parts]), vde[ref_ids].reshape(len(ref_ids), 8))
    assert all(p["global_total"] == len(ref_ids) for p in parts)


def test_plan_slabs_balances_and_covers():
    g = synth.gnm_graph(5000, 40000, seed=5)
    sn = synth.degree_order(g["offsets"])
    for R in (1, 2, 4, 8):
        b = plan_slabs(g["offsets"], sn, R)
        assert len(b) == R + 1 and b[0] == 0 and b[-1] == g["n"]
        assert np.all(np.diff(b.astype(np.int64)) >= 0)
    rows, roff, rnbr = owned_rows(g, sn, plan_slabs(g["offsets"], sn, 4), 2)
    offs = g["offsets"].astype(np.int64)
    for k in (0, len(rows) // 2, len(rows) - 1):
        v = int(rows[k])
        assert np.array_equal(rnbr[int(roff[k]):int(roff[k + 1])], g["nbrs"][offs[v]:offs[v + 1]])


@pytest.mark.parametrize("world,l", [(8, 2), (3, 3)])
def test_slab_build_over_thread_ranks(oracle, world, l):
    """dist.ThreadRanks: the ranks as threads of one process (what the single-GPU box uses for 8 slab ranks) run
    the same exchange code and produce the single-rank output."""
    import sys
    import threading
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_engine import FakeEngine
    from gnnpe_amd.dist import ThreadRanks
    g = synth.gnm_graph(600, 3000, n_labels=7, seed=31)
    sn = synth.degree_order(g["offsets"])
    bounds = plan_slabs(g["offsets"], sn, world)
    tr = ThreadRanks(world, timeout=120.0)
    L = l + 1
    res, errors = [None] * world, []

    def body(rank):
        try:
            rows, roff, rnbr = owned_rows(g, sn, bounds, rank)
            eng = FakeEngine(oracle, g["n"], g["labels"], rows, roff, rnbr, sn, 2)
            eng.set_slab(int(bounds[rank]), int(bounds[rank + 1]))
            sb = SlabBuild(eng, g["n"], 2, bounds, rank, world, torch.device("cpu"), nbr_capacity=2 * g["m"],
                           owned_entries=int(roff[-1]), l=l, comm=tr.comm(rank))
            total, base = sb.step()
            ids = torch.zeros((max(total, 1), L), dtype=torch.int32)
            pde = torch.zeros((max(total, 1), 2 * L), dtype=torch.float64)
            assert sb.step(ids, pde) == (total, base)
            out = dict(total=total, base=base, global_total=sb.global_total, ids=ids[:total].numpy(), pde=pde[:total].numpy())
            if l == 2:
                eng.set_degrees(np.diff(g["offsets"].astype(np.int64)))
                out["filter"] = sb.filter(_query_plan(g))
            res[rank] = out
        except BaseException as ex:  # noqa: BLE001
            errors.append((rank, repr(ex)))
            tr.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    ref_ids = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, L)
    x, nx, vde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
    assert [p["base"] for p in res] == list(np.cumsum([0] + [p["total"] for p in res[:-1]]))
    assert all(p["global_total"] == len(ref_ids) for p in res)
    assert np.array_equal(np.concatenate([p["ids"] for p in res]).astype(np.uint32), ref_ids)
    assert np.array_equal(np.concatenate([p["pde"] for p in res]), vde[ref_ids].reshape(len(ref_ids), 2 * L))
    if l == 2:
        assert all(np.array_equal(res[0]["filter"], p["filter"]) for p in res[1:])
