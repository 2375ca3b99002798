"""SURVEY 8(f) row 3, second half -- what the reference's Partition constructor builds at every start of `-m online`:
the partition's copy of the paths (custom.h:205-216) and the auxiliary index of its R-tree (build_auxiliary_index,
custom.h:268-364).

Pinned by tests/golden/aux_index/ = dumps of the COMPILED reference's own constructor on trees its own insert loop
built (tests/golden/make_golden_aux.py): the oracle's restatement, the host loaders and the HIP pass over the same
index.dat must reproduce them bit for bit.  The GPU tests then run the CLI (`gnnpe_main --index --sidecars`) and, where
oracle/_ref/ref_online exists, hand OUR files to the reference's constructor and compare what it builds from them with
aux_index.bin and the partition loader."""
import gzip
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from gnnpe_amd import binding, synth

AUX = os.path.join(GOLDEN, "aux_index")
CLI = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")
sys.path.insert(0, GOLDEN)
from make_golden_aux import QUERY, read_aux_dump  # noqa: E402  (parser of the harness dump: test infrastructure)


def _golden(pid, prefix=""):
    img = gzip.open(os.path.join(AUX, f"{prefix}partition-{pid}.index.dat.gz"), "rb").read()
    return img, np.load(os.path.join(AUX, f"{prefix}partition-{pid}.npz"))


WIDTHS = [("", 2), ("e1-", 1)]  # fixture prefix, embedding width (e1-: nodes of up to 76 entries)


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64) if a.dtype == np.float64 else a


def _same(got, want, what):
    for name_g, name_w in (("key", "key"), ("degrees", "node_degrees"), ("label_mbr", "label_mbr")):
        assert np.array_equal(_bits(got[name_g]), _bits(want[name_w])), (what, name_g)


def _graph(oracle):
    offs, nbrs, labels, meta = oracle.load_graph(os.path.join(AUX, "graph.graph"))
    sn, mem = oracle.read_membership(os.path.join(AUX, "membership.txt"), meta["n"])
    return dict(offsets=offs, nbrs=nbrs, labels=labels, n=meta["n"]), sn, mem


@pytest.mark.parametrize("prefix,e", WIDTHS)
@pytest.mark.parametrize("pid", [0, 1])
def test_oracle_aux_index_matches_the_reference_constructor(oracle, pid, prefix, e):
    img, z = _golden(pid, prefix)
    key, deg, mbr = oracle.aux_index(img, 3, z["degrees"], z["pde_label"])
    _same(dict(key=key, degrees=deg, label_mbr=mbr), z, f"partition {pid}")
    # the fixture exercises what it should: a tree of height >= 2, non-trivial keys, one root with key 0
    assert len(key) > 100 and (key == 0).sum() == 1 and (key < 0).sum() == len(key) - 1
    ne = [struct.unpack_from("<i", img, (b + 1) * 4096 + 1)[0] for b in range(len(key))]
    assert max(ne) > 64 if e == 1 else max(ne) <= 64


def _write_sidecars(tmp_path, oracle, g, sn):
    paths = oracle.enumerate_closed(g["offsets"], g["nbrs"], sn, 3)
    x, nx, vde = oracle.gen_vde(g["offsets"], g["nbrs"], g["labels"], 2)
    pb, vb = str(tmp_path / "paths.bin"), str(tmp_path / "vde.bin")
    with open(pb, "wb") as f:
        f.write(b"GNNPEPTH" + struct.pack("<IIQ", 1, 3, len(paths)))
        f.write(np.ascontiguousarray(paths, np.uint32).tobytes())
    with open(vb, "wb") as f:
        f.write(struct.pack("<II", g["n"], 2))
        for a in (x, nx, vde):
            f.write(np.ascontiguousarray(a, np.float64).tobytes())
    return pb, vb, paths


@pytest.mark.parametrize("pid", [0, 1])
def test_partition_loader_matches_the_reference_partition_copy(oracle, tmp_path, pid):
    g, sn, mem = _graph(oracle)
    pb, vb, paths = _write_sidecars(tmp_path, oracle, g, sn)
    _, z = _golden(pid)
    pp = str(tmp_path / "partition_paths.txt")
    with open(pp, "w") as f:  # main.cpp:98-108
        f.write(f"{len(z['path_ids'])}\n" + "".join(f"{i}\n" for i in z["path_ids"]))
    deg = np.diff(g["offsets"].astype(np.int64)).astype(np.uint32)
    got = binding.host_load_partition_sidecar(pb, vb, pp, g["labels"], deg)
    assert np.array_equal(got["path_ids"], z["path_ids"])
    for name in ("vids", "labels", "degrees", "pde", "pde_label"):
        assert np.array_equal(_bits(got[name]), _bits(z[name])), name
    # fail loud: an id beyond the path list, a file that is not a partition_paths.txt
    with open(pp, "w") as f:
        f.write(f"1\n{len(paths)}\n")
    with pytest.raises(binding.GnnpeError, match="path id"):
        binding.host_load_partition_sidecar(pb, vb, pp, g["labels"], deg)
    with open(pp, "w") as f:
        f.write("3\n1\n2\n")
    with pytest.raises(binding.GnnpeError, match="not a partition_paths.txt"):
        binding.host_load_partition_sidecar(pb, vb, pp, g["labels"], deg)


def test_aux_file_loader_round_trip(tmp_path):
    _, z = _golden(0)
    N, L, D = len(z["key"]), 3, 6
    p = str(tmp_path / "aux_index.bin")
    with open(p, "wb") as f:
        f.write(b"GNNPEAUX" + struct.pack("<IIIIQ", 1, L, D, 0, N))
        f.write(z["key"].tobytes() + z["node_degrees"].tobytes() + z["label_mbr"].tobytes())
    got = binding.host_load_aux_index(p)
    assert (got["L"], got["D"]) == (L, D)
    _same(got, z, "loader")
    with open(p, "r+b") as f:
        f.truncate(os.path.getsize(p) - 8)
    with pytest.raises(binding.GnnpeError, match="not an aux_index.bin"):
        binding.host_load_aux_index(p)


# ---- GPU -------------------------------------------------------------------------------------------------------------
def _engine(g, sn, mem, p, e=2):
    eng = binding.Engine(0)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(sn, mem, p)
    eng.set_label_table(binding.host_label_table(int(g["labels"].max()) + 1, e))
    eng.vde(want=False)
    return eng


@pytest.mark.gpu
@pytest.mark.parametrize("prefix,e", WIDTHS)
@pytest.mark.parametrize("pid", [0, 1])
def test_hip_aux_of_the_reference_tree_matches_the_reference(oracle, pid, prefix, e):
    """The HIP pass over the index.dat the reference's insert loop wrote = the reference's own auxiliary index."""
    import torch
    g, sn, mem = _graph(oracle)
    eng = _engine(g, sn, mem, 2, e)
    img, z = _golden(pid, prefix)
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(np.frombuffer(img, np.uint8).copy()).to(dev)
    d_tup = torch.from_numpy(np.ascontiguousarray(z["vids"]).view(np.int32)).to(dev)
    got = eng.aux_index_device(d_img, len(img), len(z["vids"]), 3, d_tup)
    _same(got, z, f"partition {pid}")
    # damaged images are refused, not walked: a leaf entry pointing past the partition's paths
    bad = np.frombuffer(img, np.uint8).copy()
    nblk = struct.unpack_from("<i", img, 4)[0]
    leaf = next(b for b in range(nblk) if img[(b + 1) * 4096] == 0)
    D = 3 * e
    bad[(leaf + 1) * 4096 + 5 + 16 * D:(leaf + 1) * 4096 + 5 + 16 * D + 4] = np.frombuffer(struct.pack("<i", 1 << 30), np.uint8)
    with pytest.raises(binding.GnnpeError, match="outside the partition"):
        eng.aux_index_device(torch.from_numpy(bad).to(dev), len(img), len(z["vids"]), 3, d_tup)
    eng.close()


def _run_cli(d, graph, p):
    r = subprocess.run([CLI, "-f", d + "/", "-d", graph, "-p", str(p), "--index", "--sidecars"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def _check_cli_outputs(oracle, d, graph, g, p, reference=True):
    from oracle import ref_online_path
    deg = np.diff(g["offsets"].astype(np.int64)).astype(np.uint32)
    pb, vb = os.path.join(d, "gnn-pe", "paths.bin"), os.path.join(d, "gnn-pe", "vde.bin")
    ref = None
    if reference and os.path.exists(ref_online_path()):
        qp, dump = os.path.join(d, "q.graph"), os.path.join(d, "aux_dump.bin")
        open(qp, "w").write(QUERY)
        subprocess.check_call([ref_online_path(), d + "/", graph, qp, str(p), "aux", dump], stdout=subprocess.DEVNULL)
        ref = read_aux_dump(dump, p)
    for pid in range(p):
        pdir = os.path.join(d, "gnn-pe", "partitions", f"partition-{pid}")
        mine = binding.host_load_partition_sidecar(pb, vb, os.path.join(pdir, "partition_paths.txt"), g["labels"], deg)
        aux = binding.host_load_aux_index(os.path.join(pdir, "aux_index.bin"))
        img = open(os.path.join(pdir, "index.dat"), "rb").read()
        key, dg, mbr = oracle.aux_index(img, 3, mine["degrees"], mine["pde_label"])
        _same(aux, dict(key=key, node_degrees=dg, label_mbr=mbr), f"oracle, partition {pid}")
        assert len(aux["key"]) == struct.unpack_from("<i", img, 4)[0]
        if ref is not None:  # the reference's own constructor on OUR files
            _same(aux, ref[pid], f"reference, partition {pid}")
            for name in ("vids", "labels", "degrees", "pde", "pde_label"):
                assert np.array_equal(_bits(mine[name]), _bits(ref[pid][name])), (pid, name)
    return ref is not None


@pytest.mark.gpu
def test_cli_aux_index_through_the_reference_constructor(oracle, tmp_path):
    """`gnnpe_main --index --sidecars` on the fixture graph (two interleaved partitions): aux_index.bin equals the oracle's
    walk of our index.dat and what the reference's Partition constructor builds from our files."""
    g, sn, mem = _graph(oracle)
    d = str(tmp_path)
    synth.make_dataset_dir(d, 2)
    synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), sn, mem)
    graph = os.path.join(AUX, "graph.graph")
    _run_cli(d, graph, 2)
    _check_cli_outputs(oracle, d, graph, g, 2)


@pytest.mark.gpu
def test_cli_aux_index_on_the_test_graph_with_a_hub_row(oracle, test_graph, tmp_path):
    """The reference's sample graph (a degree-168 row: hub units in the pair-major build), p = 1."""
    g = test_graph
    d = str(tmp_path)
    synth.make_dataset_dir(d, 1)
    synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), g["sorted_nodes"], g["membership"])
    graph = os.path.join(GOLDEN, "test_graph", "data_graph.graph")
    _run_cli(d, graph, 1)
    _check_cli_outputs(oracle, d, graph, g, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["generic pass", "leaf kernel"])
def test_aux_index_of_a_large_partition_by_properties(oracle, source):
    """config 2 size (100K/1M, 2.0e7 paths, ~5.4e5 nodes): too large for the host walk in a test, so the arrays are
    checked through what must hold for any tree: the root's label MBR / degrees are the extremes over all paths, every
    node's arrays are dominated by its parent's (checked from the image's entries), keys are the negated upper-bound
    sums of the parents' entries."""
    import torch
    g = synth.gnm_graph(100_000, 1_000_000)
    sn = synth.degree_order(g["offsets"])
    eng = _engine(g, sn, np.zeros(g["n"], np.uint32), 1)
    total = eng.count_paths(2)
    dev = torch.device("cuda:0")
    ids = torch.empty((total, 3), dtype=torch.int32, device=dev)
    eng.fill_paths_device(0, total, ids, None, None)
    if source == "generic pass":
        img_ptr, nbytes, hdr = eng.build_index_partition_device(0)
        got = eng.aux_index_device(img_ptr, nbytes, total, 3, ids)
    else:  # the one-pass build: leaf rows by the pair-major leaf kernel, upper levels by k_aux_level
        img_ptr, nbytes, hdr, key, deg_, mbr, n_nodes = eng.build_index_partition_aux_device(0, fetch=True)
        got = dict(key=key, degrees=deg_, label_mbr=mbr)
    img = eng.copy_to_host(img_ptr, nbytes)
    N, root, D, esz = hdr[1], hdr[7], 6, 100
    assert len(got["key"]) == N
    x, nx, vde = eng.vde()
    deg = np.diff(g["offsets"].astype(np.int64))
    h_ids = ids.cpu().numpy().view(np.uint32)
    assert np.array_equal(got["degrees"][root], deg[h_ids].max(axis=0))
    pl = x[h_ids].reshape(total, D)
    assert np.array_equal(got["label_mbr"][root, 0::2], pl.min(axis=0)) and np.array_equal(got["label_mbr"][root, 1::2], pl.max(axis=0))
    assert got["key"][root] == 0.0
    blocks = img[4096:].reshape(N, 4096)
    level = blocks[:, 0].view(np.int8)
    ne = blocks[:, 1:5].copy().view(np.int32).reshape(-1)
    inner = np.nonzero(level > 0)[0]
    seen = np.zeros(N, bool)
    for b in inner:
        ent = blocks[b, 5:5 + ne[b] * esz].reshape(ne[b], esz)
        son = ent[:, 96:].copy().view(np.int32).reshape(-1)
        bounces = ent[:, :96].copy().view(np.float64).reshape(ne[b], 12)
        seen[son] = True
        k = np.zeros(ne[b])
        for j in range(D):
            k -= bounces[:, 2 * j + 1]
        assert np.array_equal(got["key"][son], k)
        assert (got["degrees"][son] <= got["degrees"][b]).all() and np.array_equal(got["degrees"][son].max(axis=0), got["degrees"][b])
        assert np.array_equal(got["label_mbr"][son][:, 0::2].min(axis=0), got["label_mbr"][b, 0::2])
        assert np.array_equal(got["label_mbr"][son][:, 1::2].max(axis=0), got["label_mbr"][b, 1::2])
    assert seen.sum() == N - 1 and not seen[root]
    # leaves: a sample against the paths they hold
    leaves = np.nonzero(level == 0)[0]
    for b in leaves[:: max(1, len(leaves) // 500)]:
        son = blocks[b, 5:5 + ne[b] * esz].reshape(ne[b], esz)[:, 96:].copy().view(np.int32).reshape(-1)
        assert np.array_equal(got["degrees"][b], deg[h_ids[son]].max(axis=0))
        assert np.array_equal(got["label_mbr"][b, 0::2], pl[son].min(axis=0)) and np.array_equal(got["label_mbr"][b, 1::2], pl[son].max(axis=0))
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["compact", "wide"])
@pytest.mark.parametrize("kind,e,p", [("gnm", 2, 1), ("gnm", 2, 3), ("gnm", 1, 2), ("gnm", 4, 2), ("powerlaw", 2, 2), ("test", 2, 1), ("gnm", 8, 1), ("biglabels", 2, 2), ("powerlaw", 8, 1), ("powerlaw", 4, 2), ("powerlaw", 1, 1)])
def test_leaf_kernel_aux_rows_equal_the_generic_pass(oracle, test_graph, monkeypatch, kind, e, p, form):
    """gnnpe_build_index_partition_aux_device: the auxiliary index the pair-major LEAF KERNEL computes while it assembles
    the leaves (+ the upper levels) must equal, bit for bit, the generic pass over the finished image with the partition's
    tuples (gnnpe_aux_index_device) -- which the tests above pin to the reference's constructor.  Hub rows (power-law, Test/),
    several partitions, every specialised embedding width; both forms of the aux row blocks (the {degree, label} word
    inside a record's id bits -- the default wherever it fits -- and as 8 bytes behind every record)."""
    import torch
    monkeypatch.setenv("GNNPE_AUX_WIDE", "1" if form == "wide" else "0")  # read once per count by build_raux
    if kind == "gnm":
        g = synth.gnm_graph(6000, 48000, n_labels=11, seed=5)
    elif kind == "powerlaw":
        g = synth.powerlaw_graph(8000, 40000, exponent=2.1, max_degree=300, n_labels=9, seed=6)
    elif kind == "biglabels":
        # degrees past 1 024 (11 bits) and labels up to 65 535 (16 bits): the {degree, label} word does not fit a record's 26
        # id bits, so the DATA chooses the 8-byte words whatever `form` asks for, and the label tables (131 072 entries) stay
        # in global memory instead of the leaf kernel's LDS
        g = synth.powerlaw_graph(8000, 80000, exponent=1.8, max_degree=3000, n_labels=9, seed=6)
        assert int(np.diff(g["offsets"].astype(np.int64)).max()) >= 1024
        g["labels"] = ((g["labels"].astype(np.uint64) * 8191 + np.arange(g["n"], dtype=np.uint64) * 7) % 65536).astype(np.uint32)
        g["labels"][0] = 65535
    else:
        g = dict(n=test_graph["meta"]["n"], offsets=test_graph["offsets"], nbrs=test_graph["nbrs"], labels=test_graph["labels"])
    sn = synth.degree_order(g["offsets"])
    mem = (np.arange(g["n"]) % p).astype(np.uint32)
    eng = binding.Engine(0)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(sn, mem, p)
    eng.set_label_table(binding.host_label_table(int(g["labels"].max()) + 1, e))
    eng.vde(want=False)
    total = eng.count_paths(2)
    dev = torch.device("cuda:0")
    ids = torch.empty((max(total, 1), 3), dtype=torch.int32, device=dev)
    eng.fill_paths_device(0, total, ids, None, None)
    part = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    eng.path_partitions_device(0, total, part)
    for pid in range(p):
        img_ptr, nbytes, hdr, key, deg, mbr, N = eng.build_index_partition_aux_device(pid, fetch=True)
        mine = ids[:total][part[:total] == pid].contiguous()
        assert hdr[3] == len(mine) and N == hdr[1]
        # consumer constraints of the image itself (leaves now hold up to capacity - 1 entries)
        info = oracle.index_validate(eng.copy_to_host(img_ptr, nbytes).tobytes())
        assert info["num_data"] == len(mine) and np.array_equal(np.sort(info["leaf_son"]), np.arange(len(mine)))
        want = eng.aux_index_device(img_ptr, nbytes, len(mine), 3, mine)
        assert np.array_equal(key.view(np.uint64), want["key"].view(np.uint64)), (pid, "key")
        assert np.array_equal(deg, want["degrees"]), (pid, "degrees")
        assert np.array_equal(mbr.view(np.uint64), want["label_mbr"].view(np.uint64)), (pid, "label_mbr")
    eng.close()
