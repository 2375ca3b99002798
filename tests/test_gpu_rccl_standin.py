"""The N >= 2 RCCL branch of `gnnpe_main --gpus N` (gnn-pe_amd/host/slab_offline.cpp), executed.

The boxes of this pool have one GPU and RCCL refuses the same device twice in a communicator, so until round 5 the code that
issues ncclGetUniqueId / ncclCommInitRank / ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd / ncclCommAbort / ncclCommDestroy
had only ever run with a rank as its own peer.  tests/fake_rccl/libfake_rccl.so implements those entry points (declared by the
real rccl.h) over hipMemcpyAsync between the rank threads' buffers, with NCCL's matching rule and stream order, and checks what
a real run would hang or corrupt memory on (byte counts of the two sides of a pair, unmatched sends).  GNNPE_RCCL_LIB names it;
`--same-device --transport rccl` then walks the real schedule -- counts, displacements, group nesting, the failure lock -- with
2, 4 and 8 ranks.  The files must be the one-context run's.  Reference work being split: GNN-PE/src/main.cpp:87-96."""
import hashlib
import json
import os
import shutil
import subprocess
import time

import numpy as np
import pytest

from conftest import ROOT
from gnnpe_amd import synth

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")
FAKE = os.path.join(FAKE_DIR, "libfake_rccl.so")


@pytest.fixture(scope="module")
def standin():
    if not os.path.exists(FAKE):
        subprocess.check_call(["make", "-C", FAKE_DIR])
    return FAKE


def _md5(path):
    """Digest of a file: xxh3-128 where the module is there (the config-4 test digests 12 GB: md5 runs at 0.6 GB/s), md5 otherwise --
    only ever compared with digests of the same function."""
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except ImportError:
        h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def _run(tmp_path, name, gp, sn, mem, p, extra, standin=None, index=False):
    d = str(tmp_path / name)
    os.makedirs(d)
    synth.make_dataset_dir(d, p)
    synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), sn, mem)
    env = dict(os.environ)
    log = str(tmp_path / f"{name}.rccl.log")
    if standin:
        env.update(GNNPE_RCCL_LIB=standin, FAKE_RCCL_LOG=log)
    r = subprocess.run([CLI, "-f", d + "/", "-d", gp, "-m", "offline", "-p", str(p), "--timing"] + (["--index"] if index else []) + extra,
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    t = json.loads(r.stderr.strip().splitlines()[-1])
    stats = [json.loads(l) for l in open(log)] if standin and os.path.exists(log) else []
    return d, t, stats


def _text_md5s(d, p):
    return [_md5(os.path.join(d, "gnn-pe", "all_paths.txt"))] + \
        [_md5(os.path.join(d, "gnn-pe", "partitions", f"partition-{i}", "partition_paths.txt")) for i in range(p)]


def _check_schedule(t, stats, n):
    """The run went through the stand-in, as an n-rank communicator, and every send met a recv of its size."""
    assert t["gpus"] == n and t["transport"] == "rccl"
    assert len(stats) == 1, stats
    s = stats[0]
    assert s["ranks"] == n and not s["aborted"] and s["mismatches"] == 0
    assert s["sends"] == s["recvs"] > n * n and s["groups"] >= 4 * n  # plan (3), lists, vde: one group per rank and exchange
    assert s["self_pairs"] >= n  # a rank's piece for itself takes the same route
    return s


def test_config2_files_through_the_rccl_schedule_with_2_4_8_ranks(tmp_path, oracle, standin):
    """BASELINE config 2 (100K / 1M, 2.0e7 paths) with p = 8 and --index: text files by md5 and every index.dat through the
    oracle's validator, one context against 2, 4 and 8 rank threads whose exchanges are ncclSend / ncclRecv groups."""
    g = synth.gnm_graph(100_000, 1_000_000)
    sn = synth.degree_order(g["offsets"])
    p = 8
    mem = synth.block_membership(g["n"], p)
    gp = str(tmp_path / "g.graph")
    synth.write_graph_file(gp, g)
    d1, t1, _ = _run(tmp_path, "one", gp, sn, mem, p, [], index=True)
    want = _text_md5s(d1, p)
    cnt = []
    for i in range(p):
        info = oracle.index_validate(open(os.path.join(d1, "gnn-pe", "partitions", f"partition-{i}", "index.dat"), "rb").read())
        cnt.append(info["num_data"])
    assert sum(cnt) == t1["paths"] == synth.expected_paths_l2(g["offsets"])
    shutil.rmtree(d1)
    moved = {}
    for n in (2, 4, 8):
        d, t, stats = _run(tmp_path, f"n{n}", gp, sn, mem, p, ["--gpus", str(n), "--same-device", "--transport", "rccl"], standin, index=True)
        s = _check_schedule(t, stats, n)
        moved[n] = s["bytes"]
        assert _text_md5s(d, p) == want, n
        for i in range(p):  # built by rank i mod n from every rank's tuples (one all-to-all-v each)
            info = oracle.index_validate(open(os.path.join(d, "gnn-pe", "partitions", f"partition-{i}", "index.dat"), "rb").read())
            assert info["num_data"] == cnt[i] and info["root_is_data"] == 0, (n, i)
            if i in (0, p - 1):
                assert np.array_equal(np.sort(info["leaf_son"]), np.arange(cnt[i])), (n, i)
        shutil.rmtree(d)
    assert moved[2] < moved[4] < moved[8]  # more ranks, more of the graph is somebody's halo


def test_config4_text_files_through_the_rccl_schedule_with_8_ranks(tmp_path, standin):
    """BASELINE config 4 (1M / 10M, 2.0e8 paths, p = 8) at full size: 6.2 GB of text, md5 for md5 the one-context run's."""
    g = synth.gnm_graph(1_000_000, 10_000_000)
    sn = synth.degree_order(g["offsets"])
    p = 8
    mem = synth.block_membership(g["n"], p)
    gp = str(tmp_path / "g.graph")
    synth.write_graph_file(gp, g)
    d1, t1, _ = _run(tmp_path, "one", gp, sn, mem, p, [])
    want = _text_md5s(d1, p)
    shutil.rmtree(d1)
    d, t, stats = _run(tmp_path, "n8", gp, sn, mem, p, ["--gpus", "8", "--same-device", "--transport", "rccl"], standin)
    s = _check_schedule(t, stats, 8)
    assert t["paths"] == t1["paths"] == synth.expected_paths_l2(g["offsets"])
    assert _text_md5s(d, p) == want
    assert s["largest"] > 1 << 20 and s["bytes"] > 100 << 20  # the halo lists of a 1M / 10M graph, not a toy
    shutil.rmtree(d)


@pytest.mark.parametrize("fault", ["1:init", "1:halo", "3:emit", "2:index"])
def test_fault_in_one_rank_of_the_rccl_schedule_ends_the_process(tmp_path, standin, fault):
    """GNNPE_FAULT_RANK with four ranks inside ncclSend / ncclRecv groups: exit code 1 with that rank's message within 20 s (a
    peer inside a group is released by ncclCommAbort; peers blocked in ncclCommInitRank -- it returns only when ALL ranks have
    called it, and the one that died never will -- are ended by the main thread's _exit(1) 5 s after the failure; never a
    re-exec of a process that has touched the GPU)."""
    g = synth.gnm_graph(3000, 21000, n_labels=9, seed=17)
    sn = synth.degree_order(g["offsets"])
    gp = str(tmp_path / "g.graph")
    synth.write_graph_file(gp, g)
    d = str(tmp_path / "ds")
    os.makedirs(d)
    synth.make_dataset_dir(d, 2)
    synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), sn, synth.block_membership(3000, 2))
    args = [CLI, "-f", d + "/", "-d", gp, "-p", "2", "--gpus", "4", "--same-device", "--transport", "rccl", "--index"]
    env = dict(os.environ, GNNPE_RCCL_LIB=standin, GNNPE_FAULT_RANK=fault)
    t0 = time.time()
    r = subprocess.run(args, capture_output=True, text=True, timeout=120, env=env)
    took = time.time() - t0
    assert r.returncode == 1, (r.returncode, r.stderr[-1000:])
    assert f"rank {fault.split(':')[0]}: injected fault at stage {fault.split(':')[1]}" in r.stderr
    assert "terminate called" not in r.stderr and took < 20, took
