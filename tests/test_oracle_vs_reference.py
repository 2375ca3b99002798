"""Live cross-checks of the oracle against the compiled reference binary (oracle/_ref/ref_main).
The binary is built by oracle/Makefile where /root/reference exists and travels (git-ignored)
to the GPU box; the tests skip when it is absent.  CPU only."""
import os
import re
import subprocess

import numpy as np
import pytest

from gnnpe_amd import synth
from oracle import ref_main_path

pytestmark = pytest.mark.skipif(not os.path.exists(ref_main_path()), reason="oracle/_ref/ref_main not built")


def _run_ref(tmp, g, sn, mem, p, mode, query=None):
    gp = os.path.join(tmp, "g.graph")
    if not os.path.exists(gp):
        synth.write_graph_file(gp, g)
        synth.make_dataset_dir(tmp, p)
        synth.write_membership(os.path.join(tmp, "gnn-pe", "membership.txt"), sn, mem)
    cmd = [ref_main_path(), "-f", tmp + "/", "-d", gp, "-m", mode, "-p", str(p)]
    if query:
        cmd += ["-q", query]
    return subprocess.check_output(cmd, text=True)


@pytest.mark.parametrize("seed,n,m", [(11, 300, 1500), (12, 1000, 3000)])
def test_random_graph_offline_bytes(oracle, tmp_path, seed, n, m):
    g = synth.gnm_graph(n, m, n_labels=9, seed=seed)
    rng = np.random.default_rng(seed)
    sn = rng.permutation(n).astype(np.uint32)
    mem = rng.integers(0, 4, size=n).astype(np.uint32)
    tmp = str(tmp_path)
    _run_ref(tmp, g, sn, mem, 4, "offline")
    # the oracle reads the same files through its own loader / membership reader
    offs, nbrs, labels, meta = oracle.load_graph(os.path.join(tmp, "g.graph"))
    assert np.array_equal(offs, g["offsets"]) and np.array_equal(nbrs, g["nbrs"]) and np.array_equal(labels, g["labels"])
    sn2, mem2 = oracle.read_membership(os.path.join(tmp, "gnn-pe", "membership.txt"), n)
    assert np.array_equal(sn2, sn) and np.array_equal(mem2, mem)
    paths = oracle.enumerate_closed(offs, nbrs, sn2, 3)
    assert oracle.format_all_paths(paths) == open(os.path.join(tmp, "gnn-pe", "all_paths.txt"), "rb").read()
    for pid in range(4):
        out = os.path.join(tmp, f"pp{pid}.txt")
        oracle.write_partition_paths(out, paths, mem2, pid)
        ref = os.path.join(tmp, "gnn-pe", "partitions", f"partition-{pid}", "partition_paths.txt")
        assert open(out, "rb").read() == open(ref, "rb").read()


def test_index_validator_on_reference_built_index(oracle, tmp_path, golden_dir):
    """The R6 decoder/validator accepts what the reference's own insert loop writes, and the decoded
    leaf multiset is {(i, pde of the i-th path of the partition)} (custom.h:240-248)."""
    g = synth.gnm_graph(400, 1800, n_labels=5, seed=21)
    sn = synth.degree_order(g["offsets"])
    mem = synth.block_membership(400, 2)
    tmp = str(tmp_path)
    _run_ref(tmp, g, sn, mem, 2, "offline")
    out = _run_ref(tmp, g, sn, mem, 2, "online", query=os.path.join(golden_dir, "test_graph", "query_graph.graph"))
    assert re.search(r"Answer Number: \d+", out)
    paths = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    x, nx, vde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
    pde, _, _, _ = oracle.gen_pde(paths, 2, g["offsets"], g["labels"], x, vde)
    for pid in range(2):
        img = open(os.path.join(tmp, "gnn-pe", "partitions", f"partition-{pid}", "index.dat"), "rb").read()
        d = oracle.index_validate(img)
        ids = np.nonzero(mem[paths[:, 0]] == pid)[0]
        assert d["num_data"] == len(ids) and d["dim"] == 6 and d["root_is_data"] == 0
        order = np.argsort(d["leaf_son"], kind="stable")
        assert np.array_equal(d["leaf_son"][order], np.arange(len(ids)))
        assert np.array_equal(d["leaf_pt"][order], pde[ids])


def test_index_validator_rejects_corruption(oracle, tmp_path, golden_dir):
    g = synth.gnm_graph(300, 1200, n_labels=5, seed=22)
    sn = synth.degree_order(g["offsets"])
    mem = np.zeros(300, np.uint32)
    tmp = str(tmp_path)
    _run_ref(tmp, g, sn, mem, 1, "offline")
    _run_ref(tmp, g, sn, mem, 1, "online", query=os.path.join(golden_dir, "test_graph", "query_graph.graph"))
    img = bytearray(open(os.path.join(tmp, "gnn-pe", "partitions", "partition-0", "index.dat"), "rb").read())
    oracle.index_validate(bytes(img))
    bad = bytearray(img)
    bad[24] = 1  # root_is_data
    with pytest.raises(ValueError):
        oracle.index_validate(bytes(bad))
    bad = bytearray(img)
    bad[4:8] = (int.from_bytes(img[4:8], "little") + 1).to_bytes(4, "little")  # block count mismatch
    with pytest.raises(ValueError):
        oracle.index_validate(bytes(bad))
    # shrink an internal MBR: find root block and raise its first lo bound
    root = int.from_bytes(img[25:29], "little")
    off = (root + 1) * 4096 + 5
    bad = bytearray(img)
    hi = np.frombuffer(bytes(img[off + 8:off + 16]), np.float64)[0]
    bad[off:off + 8] = np.float64(hi).tobytes()  # lo := hi  -> no longer encloses the child
    with pytest.raises(ValueError):
        oracle.index_validate(bytes(bad))
