"""RCCL executed on the one GPU a box has: a 1-rank communicator whose exchanges (self ncclSend/ncclRecv inside a
group; all-gather / all-to-all-v of torch.distributed's backend "nccl") carry the N > 1 code of SURVEY 8(e) --
`gnnpe_main --gpus 1 --transport rccl` (host/slab_offline.cpp) and dist.SlabBuild(force_collectives=True).  These are
NOT skipped on a single-GPU box: they are what proves that librccl loads, ncclCommInitRank works from a rank thread /
process, and the halo / vde / tuple exchanges produce the single-GPU files (replaces GNN-PE/src/main.cpp:87-119 across
devices).  The >= 2-GPU variants live in test_gpu_cli.py / test_gpu_slabs_full.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from gnnpe_amd import synth

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")
WORKER = os.path.join(ROOT, "tests", "slab_worker.py")
M64 = (1 << 64) - 1


def _dataset(tmp_path, name, g, sn, mem, p):
    d = str(tmp_path / name)
    os.makedirs(d)
    synth.make_dataset_dir(d, p)
    synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), sn, mem)
    return d


@pytest.mark.parametrize("l", [2, 3])
def test_cli_one_rank_over_rccl_writes_the_single_gpu_files(tmp_path, oracle, l):
    g = synth.gnm_graph(30000, 300000, n_labels=9, seed=17) if l == 2 else synth.gnm_graph(3000, 15000, n_labels=9, seed=17)
    sn = synth.degree_order(g["offsets"])
    p = 4
    mem = synth.block_membership(g["n"], p)
    gp = str(tmp_path / "g.graph")
    synth.write_graph_file(gp, g)
    outs = {}
    for name, extra in (("plain", []), ("rccl", ["--gpus", "1", "--transport", "rccl"]), ("copy", ["--gpus", "1", "--transport", "copy"])):
        d = _dataset(tmp_path, name, g, sn, mem, p)
        r = subprocess.run([CLI, "-f", d + "/", "-d", gp, "-p", str(p), "-l", str(l), "--index", "--timing", "--chunk", "200000"] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        t = json.loads(r.stderr.strip().splitlines()[-1])
        if extra:
            assert t["transport"] == name and t["gpus"] == 1
            assert t["ranks"][0]["owned_entries"] == t["csr_entries"] == 2 * g["m"]
        outs[name] = d
    rel = ["gnn-pe/all_paths.txt"] + [f"gnn-pe/partitions/partition-{i}/partition_paths.txt" for i in range(p)]
    for f in rel:
        want = open(os.path.join(outs["plain"], f), "rb").read()
        assert len(want) > 16
        for name in ("rccl", "copy"):
            assert open(os.path.join(outs[name], f), "rb").read() == want, (name, f)
    # index.dat through the tuple exchange (rank pid mod N builds partition pid from the tuples every rank sent it)
    for i in range(p):
        a = open(os.path.join(outs["rccl"], f"gnn-pe/partitions/partition-{i}/index.dat"), "rb").read()
        b = open(os.path.join(outs["copy"], f"gnn-pe/partitions/partition-{i}/index.dat"), "rb").read()
        assert a == b, i
        if l == 2:
            paths = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
            cnt = int((mem[paths[:, 0]] == i).sum())
            info = oracle.index_validate(a)
            assert info["num_data"] == cnt and np.array_equal(np.sort(info["leaf_son"]), np.arange(cnt))


def test_slabbuild_one_rank_process_group_over_rccl(tmp_path):
    """dist.SlabBuild's N > 1 step over torch.distributed backend "nccl" with world_size 1: halo plan (three
    all-to-all-v), row exchange, vde all-gather, async all-gather of the totals, then the enqueue-only step (no read-back
    before the fill).  The rank's rows must be the plain single-GPU run's rows (order-sensitive checksum, total)."""
    g = synth.gnm_graph(200_000, 2_000_000)
    gp = str(tmp_path / "graph.npz")
    np.savez(gp, offsets=g["offsets"], nbrs=g["nbrs"], labels=g["labels"])
    res = {}
    for name, extra in (("plain", []), ("rccl", ["--force-rccl", "1"])):
        out = str(tmp_path / name)
        env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
        r = subprocess.run([sys.executable, WORKER, "--graph", gp, "--out", out, "-l", "2", "-e", "2"] + extra,
                           capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        res[name] = json.load(open(os.path.join(out, "rank0.json")))
    a, b = res["plain"], res["rccl"]
    assert b["backend"] == "nccl" and a["backend"] == "none"
    assert a["total"] == b["total"] == b["global_total"] == synth.expected_paths_l2(g["offsets"])
    assert a["checksum"] == b["checksum"] and a["middle_sum"] == b["middle_sum"]
    assert all(b["props"].values()), b["props"]


def test_bench_multi_gpu_path_over_a_one_rank_rccl_group(tmp_path):
    """bench.py's OWN N > 1 code (slab rows, halo plan, vde all-gather, enqueue-only count with the asynchronous all-gather of
    the totals, capped fill, RCCL all-reduce of the sanity numbers) over a 1-rank process group: GNNPE_BENCH_FORCE_RCCL=1.
    What the driver launches on 2/4/8 GPUs has then run end to end on this box, at a smaller graph."""
    env = dict(os.environ, GNNPE_BENCH_FORCE_RCCL="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--vertices", "200000", "--edges", "2000000", "--steps", "3",
                        "--warmup", "1", "--placements", "3", "--no-cpu-baseline", "--no-index"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    g = synth.gnm_graph(200_000, 2_000_000)
    assert d["n_gpus"] == 1 and d["config"]["collectives"] == "rccl" and d["config"]["paths"] == synth.expected_paths_l2(g["offsets"])
    assert d["value"] > 1e9 and d["halo"]["owned_entries"] == 2 * g["m"] and d["halo"]["halo_rows"] == 0
    assert "vde_and_allgather_ms" in d["phases_ms"]["per_step"]


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the way the driver starts N = 1): bench.py starts the two rank
    processes itself before touching HIP, relays rank 0's JSON line and leaves with the children's exit code.  On this
    single-GPU box both ranks share device 0 (GNNPE_BENCH_SAME_DEVICE=1: collectives staged over gloo, RCCL refuses two
    ranks on one device); on a multi-GPU node the same command runs over RCCL.  Replaces the split of main.cpp:87-96."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GNNPE_BENCH_SAME_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--vertices", "100000", "--edges", "1000000",
                        "--steps", "3", "--warmup", "1", "--no-index", "--no-cpu-baseline", "--no-config5"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    g = synth.gnm_graph(100_000, 1_000_000)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "slab2"
    assert d["config"]["paths"] == synth.expected_paths_l2(g["offsets"])
    assert d["sanity"].startswith("path count and middle-vertex checksum match")
    assert d["halo"]["halo_rows"] > 0 and d["value"] > 1e6  # (collectives staged through host memory here: no rate to speak of)
