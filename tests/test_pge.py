"""GNN-PGE offline ("next" row): oracle pinned to the compiled GNN-PGE reference (CPU),
then the HIP path against the oracle / golden vectors and the untouched GNN-PGE online binary (GPU)."""
import json
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from gnnpe_amd import synth
from oracle import ref_main_pge_path

CLI = os.path.join(ROOT, "gnn-pe_amd", "gnnpge_main")


def _decode_bin(path, e):
    b = open(path, "rb").read()
    n = struct.unpack_from("<I", b, 0)[0]
    D = 2 * e
    rec = np.dtype([("vid", "<u4"), ("label", "<u4"), ("degree", "<u4"), ("key", "<f8"), ("x", "<f8", (e,)),
                    ("nx", "<f8", (e,)), ("vde", "<f8", (e,)), ("pg", "<f8", (2 * D,)), ("plg", "<f8", (2 * D,))])
    assert len(b) == 4 + n * rec.itemsize
    return np.frombuffer(b, rec, n, 4)


def test_oracle_pge_groups_match_reference_dump(oracle, test_graph):
    z = np.load(os.path.join(GOLDEN, "pge_test_graph_e2.npz"))
    x, nx, vde = oracle.gen_vde(test_graph["offsets"], test_graph["nbrs"], test_graph["labels"], 2)
    pg, plg = oracle.pge_groups(test_graph["offsets"], test_graph["nbrs"], 2, x, vde)
    assert np.array_equal(z["vid"], np.arange(len(x))) and np.array_equal(z["label"], test_graph["labels"])
    assert np.array_equal(z["degree"], np.diff(test_graph["offsets"]))
    assert np.array_equal(x, z["x"]) and np.array_equal(nx, z["nx"]) and np.array_equal(vde, z["vde"])
    assert np.array_equal(pg, z["pg"]) and np.array_equal(plg, z["plg"])
    # isolated vertices: [vde, vde] then zeros (main.cpp:104-121); Test/ has 11 of them
    iso = np.nonzero(np.diff(test_graph["offsets"]) == 0)[0]
    assert len(iso) == 11 and np.all(pg[iso, 4:] == 0) and np.array_equal(pg[iso, 0], vde[iso, 0])


def test_oracle_pge_bin_layout(oracle, test_graph, tmp_path):
    gold = json.load(open(os.path.join(GOLDEN, "pge_golden.json")))
    x, nx, vde = oracle.gen_vde(test_graph["offsets"], test_graph["nbrs"], test_graph["labels"], 2)
    pg, plg = oracle.pge_groups(test_graph["offsets"], test_graph["nbrs"], 2, x, vde)
    p = str(tmp_path / "dv.bin")
    oracle.pge_write_bin(p, 2, test_graph["offsets"], test_graph["labels"], x, nx, vde, pg, plg)
    assert os.path.getsize(p) == gold["p1"]["bin_bytes"] == 609956
    rec = _decode_bin(p, 2)
    assert np.array_equal(rec["pg"], pg) and np.array_equal(rec["vid"], np.arange(len(x)))


@pytest.mark.gpu
@pytest.mark.parametrize("e", [2, 3, 8])
def test_gpu_pge_groups_bit_exact(oracle, test_graph, e):
    from gnnpe_amd import binding
    eng = binding.Engine(0)
    eng.load_csr(test_graph["offsets"], test_graph["nbrs"], test_graph["labels"])
    eng.set_label_table(binding.host_label_table(71, e))
    x, nx, vde = eng.vde()
    pg, plg = eng.pge_groups()
    ox, onx, ovde = oracle.gen_vde(test_graph["offsets"], test_graph["nbrs"], test_graph["labels"], e)
    opg, oplg = oracle.pge_groups(test_graph["offsets"], test_graph["nbrs"], e, ox, ovde)
    assert np.array_equal(pg, opg) and np.array_equal(plg, oplg)
    if e == 2:
        z = np.load(os.path.join(GOLDEN, "pge_test_graph_e2.npz"))
        assert np.array_equal(pg, z["pg"]) and np.array_equal(plg, z["plg"])
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("p", [1, 2])
def test_gpu_pge_cli_files_and_reference_online(tmp_path, oracle, p):
    gold = json.load(open(os.path.join(GOLDEN, "pge_golden.json")))[f"p{p}"]
    graph = os.path.join(GOLDEN, "test_graph", "data_graph.graph")
    deg = np.array([int(l.split()[3]) for l in open(graph) if l.startswith("v")])
    n = len(deg)
    sn = np.argsort(deg, kind="stable").astype(np.uint32)
    mem = np.zeros(n, np.uint32) if p == 1 else (np.arange(n) % 2).astype(np.uint32)
    tmp = str(tmp_path)
    for i in range(p):
        os.makedirs(os.path.join(tmp, "gnn-pge", "partitions", f"partition-{i}"))
    synth.write_membership(os.path.join(tmp, "gnn-pge", "membership.txt"), sn, mem)
    r = subprocess.run([CLI, "-f", tmp + "/", "-d", graph, "-m", "offline", "-p", str(p), "--timing"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(os.path.join(tmp, "gnn-pge", "data_vertices.bin")) == gold["bin_bytes"]
    rec = _decode_bin(os.path.join(tmp, "gnn-pge", "data_vertices.bin"), 2)
    z = np.load(os.path.join(GOLDEN, "pge_test_graph_e2.npz"))
    for k in ("vid", "label", "degree", "x", "nx", "vde", "pg", "plg"):  # every field but the uninitialised `key`
        assert np.array_equal(rec[k], z[k]), k
    for i in range(p):
        img = open(os.path.join(tmp, "gnn-pge", "partitions", f"partition-{i}", "index.dat"), "rb").read()
        hdr = struct.unpack_from("<iiiiii", img, 0)
        assert hdr[2] == 4 and hdr[3] == gold["index"][i]["num_data"]
        # structure: reuse the validator's block walk on rectangles (lo <= hi instead of lo == hi)
        part = sn[mem[sn] == i]
        bl, nb = hdr[0], hdr[1]
        sons, seen = [], set()
        root = struct.unpack_from("<i", img, 25)[0]
        assert img[24] == 0 and len(img) == (nb + 1) * bl

        def walk(blk, level):
            assert blk not in seen
            seen.add(blk)
            off = (blk + 1) * bl
            lv, ne = struct.unpack_from("<bi", img, off)
            assert lv == level and 1 <= ne <= 60
            for k in range(ne):
                ent = np.frombuffer(img, np.float64, 8, off + 5 + k * 68)
                son = struct.unpack_from("<i", img, off + 5 + k * 68 + 64)[0]
                if lv == 0:
                    sons.append(son)
                    assert np.array_equal(ent, z["pg"][part[son]])
                else:
                    walk(son, level - 1)
                    cb = (son + 1) * bl
                    cne = struct.unpack_from("<i", img, cb + 1)[0]
                    c = np.stack([np.frombuffer(img, np.float64, 8, cb + 5 + j * 68) for j in range(cne)])
                    assert np.all(ent[0::2] <= c[:, 0::2].min(0)) and np.all(ent[1::2] >= c[:, 1::2].max(0))
        walk(root, struct.unpack_from("<b", img, (root + 1) * bl)[0])
        assert seen == set(range(nb)) and sorted(sons) == list(range(len(part)))
    if not os.path.exists(ref_main_pge_path()):
        pytest.skip("oracle/_ref/ref_main_pge not built")
    out = subprocess.check_output([ref_main_pge_path(), "-f", tmp + "/", "-d", graph, "-q",
                                   os.path.join(GOLDEN, "test_graph", "query_graph.graph"), "-m", "online", "-p", str(p)],
                                  text=True)
    assert int(re.search(r"Answer Num: (\d+)", out).group(1)) == gold["answer_num"] == 221832
