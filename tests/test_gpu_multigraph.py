"""Non-simple graph files (duplicate `e` lines) through the HIP path: the same bytes and the same bits as the reference.

The reference stores the repeats (graph.cpp:211-218), counts them in `degree` and in gen_vde's neighbour sum (graph.h:154-156,
custom.h:527-534) and drops the repeated path in its hash set (custom.h:68-77).  Until round 5 `gnnpe_main` refused such a file
(still available as --strict) -- the one input class on which it differed from `main -m offline`.  Self-loop lines stay refused:
the reference's loader leaves a slot uninitialised for them (tests/test_multigraph.py, profiles/r06_selfloop_reference.txt).

Fixture: tests/golden/multigraph.npz = outputs of the compiled reference on 20 such files (make_golden_multigraph.py).  Where
oracle/_ref/ref_main exists (it travels to the GPU box) 20 MORE random files go through both binaries live."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from gnnpe_amd import binding, synth
from oracle import ref_dump_path, ref_main_path
from test_multigraph import multigraph_cases

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")


def _all_paths_text(paths):
    return (f"{len(paths)}\n" + "".join(f"{a} {b} {c} \n" for a, b, c in paths.tolist())).encode()  # main.cpp:111-118


def _ids_text(ids):
    return (f"{len(ids)}\n" + "".join(f"{i}\n" for i in ids.tolist())).encode()  # main.cpp:102-106


def test_engine_equals_the_reference_on_graphs_with_repeated_entries(oracle):
    """C-ABI: simple rows for the enumeration + the stored rows for gen_vde and the degree columns."""
    eng = binding.Engine(0)
    for ci, c in enumerate(multigraph_cases()):
        g, p = c["g"], int(c["p"])
        with pytest.raises(binding.GnnpeError, match="strictly ascending"):
            eng.load_csr(g["offsets"], g["nbrs"], g["labels"])  # gnnpe_load_csr itself takes simple rows only
        eng.load_multigraph(g["offsets"], g["nbrs"], g["labels"])
        eng.set_order(c["order"], c["member"], p)
        eng.set_label_table(binding.host_label_table(int(g["labels"].max()) + 1, 2))
        x, nx, vde = eng.vde()
        assert np.array_equal(x, c["x"]) and np.array_equal(nx, c["nx"]) and np.array_equal(vde, c["vde"]), ci
        assert eng.count_paths(2) == len(c["paths"])
        ids, pde, pdl = eng.fill_paths(pde_label=True)
        assert np.array_equal(ids, c["paths"]), ci
        assert np.array_equal(pde, c["pde"]) and np.array_equal(pdl, c["pde_label"]), ci
        # the degree columns (graph.h:154-156: the STORED row's length) reach the auxiliary index of every partition's tree
        for pid in range(p):
            sel = c[f"part{pid}"]
            if len(sel) == 0:
                continue
            img, nb, hdr, key, deg, mbr, n_nodes = eng.build_index_partition_aux_device(pid, fetch=True)
            image = bytes(eng.copy_to_host(img, nb))
            info = oracle.index_validate(image)
            assert info["num_data"] == len(sel)
            okey, odeg, ombr = oracle.aux_index(image, 3, c["pdeg"][sel], c["pde_label"][sel])
            assert np.array_equal(deg, odeg) and np.array_equal(mbr.view(np.uint64), ombr.view(np.uint64)), (ci, pid)
            assert np.array_equal(key.view(np.uint64), okey.view(np.uint64)), (ci, pid)
    # the two forms must belong together
    c = multigraph_cases()[0]
    g = c["g"]
    so, sn = binding.simple_rows(g["offsets"], g["nbrs"])
    eng.load_csr(so, sn, g["labels"])
    bad = g["nbrs"].copy()
    bad[0] = (bad[0] + 1) % g["n"]
    with pytest.raises(binding.GnnpeError, match="gnnpe_set_multigraph_rows: row"):
        eng.set_multigraph_rows(g["offsets"].astype(np.uint64), bad)
    eng.close()


def _dataset(d, c):
    os.makedirs(d, exist_ok=True)
    p = int(c["p"])
    synth.make_dataset_dir(d, p)
    synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), c["order"], c["member"])
    return p


@pytest.mark.parametrize("mode", ["one context", "two slabs"])
def test_cli_writes_the_reference_files_for_the_golden_multigraphs(tmp_path, mode):
    for ci, c in enumerate(multigraph_cases()):
        d = str(tmp_path / f"c{ci}")
        p = _dataset(d, c)
        gp = os.path.join(d, "g.graph")
        synth.write_graph_file(gp, c["g"])
        extra = ["--sidecars"] if mode == "one context" else ["--gpus", "2", "--same-device", "--transport", "copy"]
        r = subprocess.run([CLI, "-f", d + "/", "-d", gp, "-m", "offline", "-p", str(p)] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert r.stdout.encode() == bytes(c["stdout"]), ci  # printGraphMetaData: |E| = lines, Max Degree counts the repeats
        assert open(os.path.join(d, "gnn-pe", "all_paths.txt"), "rb").read() == _all_paths_text(c["paths"]), ci
        for j in range(p):
            got = open(os.path.join(d, "gnn-pe", "partitions", f"partition-{j}", "partition_paths.txt"), "rb").read()
            assert got == _ids_text(c[f"part{j}"]), (ci, j)
        if mode == "one context":
            b = open(os.path.join(d, "gnn-pe", "vde.bin"), "rb").read()
            arr = np.frombuffer(b, np.float64, 3 * c["g"]["n"] * 2, 8).reshape(3, -1, 2)
            assert np.array_equal(arr[0], c["x"]) and np.array_equal(arr[1], c["nx"]) and np.array_equal(arr[2], c["vde"]), ci
        if ci == 0:  # --strict: the refusal of rounds 1-5
            r = subprocess.run([CLI, "-f", d + "/", "-d", gp, "-m", "offline", "-p", str(p), "--strict"], capture_output=True, text=True)
            assert r.returncode != 0 and "duplicate edge" in r.stderr


def test_cli_refuses_self_loop_lines(tmp_path):
    g = synth.multigraph(30, 40, n_dup=2, n_loops=2, n_labels=4, seed=9)
    d = str(tmp_path)
    synth.make_dataset_dir(d, 1)
    synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), np.arange(30, dtype=np.uint32), np.zeros(30, np.uint32))
    gp = os.path.join(d, "g.graph")
    synth.write_graph_file(gp, g)
    r = subprocess.run([CLI, "-f", d + "/", "-d", gp, "-m", "offline", "-p", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "self-loop at vertex" in r.stderr and "graph.cpp:211-218" in r.stderr
    assert not os.path.exists(os.path.join(d, "gnn-pe", "all_paths.txt"))


def test_twenty_random_multigraphs_through_both_binaries(tmp_path):
    """Live: `ref_main -m offline` and `gnnpe_main` on the same 20 files; all_paths.txt and every partition_paths.txt byte for
    byte, vde bit for bit against ref_dump.  (Small sparse files: the reference's DFS walks every simple path below a repeated
    path -- custom.h:68 false, no return -- so its time is exponential in the size; ours is not.)"""
    if not (os.path.exists(ref_main_path()) and os.path.exists(ref_dump_path())):
        pytest.skip("oracle/_ref not built")
    for i in range(20):
        n = [14, 20, 28, 36][i % 4]
        m = n + (n // 8) * (i % 3)
        p = [1, 2, 4][i % 3]
        g = synth.multigraph(n, m, n_dup=2 + (5 * i) % 9, n_labels=[2, 6, 11][i % 3], seed=7000 + i)
        rng = np.random.default_rng(7000 + i)
        order, member = rng.permutation(n).astype(np.uint32), rng.integers(0, p, size=n).astype(np.uint32)
        gp = str(tmp_path / f"g{i}.graph")
        synth.write_graph_file(gp, g)
        dirs = {}
        for who in ("ref", "ours"):
            d = dirs[who] = str(tmp_path / f"{who}{i}")
            os.makedirs(d)
            synth.make_dataset_dir(d, p)
            synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), order, member)
        ref_out = subprocess.run([ref_main_path(), "-f", dirs["ref"] + "/", "-d", gp, "-m", "offline", "-p", str(p)], capture_output=True,
                                 timeout=120)
        assert ref_out.returncode == 0
        r = subprocess.run([CLI, "-f", dirs["ours"] + "/", "-d", gp, "-m", "offline", "-p", str(p), "--sidecars"], capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert r.stdout == ref_out.stdout
        rel = ["gnn-pe/all_paths.txt"] + [f"gnn-pe/partitions/partition-{j}/partition_paths.txt" for j in range(p)]
        for f in rel:
            assert open(os.path.join(dirs["ours"], f), "rb").read() == open(os.path.join(dirs["ref"], f), "rb").read(), (i, f)
        dump = str(tmp_path / f"vde{i}.bin")
        subprocess.check_call([ref_dump_path(), gp, "2", dump])
        ref_b, our_b = open(dump, "rb").read(), open(os.path.join(dirs["ours"], "gnn-pe", "vde.bin"), "rb").read()
        assert our_b == ref_b[: 8 + 3 * n * 2 * 8], i  # header n, e and x, nx, vde: the same bytes


@pytest.mark.parametrize("world", [1, 2])
def test_python_driver_writes_the_reference_files_for_golden_multigraphs(tmp_path, world):
    """gnn-pe_amd/offline.py (one process per GPU over torch.distributed; here one or two rank processes sharing device 0 with
    their collectives staged over gloo): the same two-step load -- simple rows, then the stored rows -- on every rank."""
    import socket
    import sys
    driver = os.path.join(ROOT, "gnn-pe_amd", "offline.py")
    for ci in (1, 7, 14):  # p = 2, 5, 3
        c = multigraph_cases()[ci]
        d = str(tmp_path / f"c{ci}")
        p = _dataset(d, c)
        gp = os.path.join(d, "g.graph")
        synth.write_graph_file(gp, c["g"])
        args = [driver, "-f", d + "/", "-d", gp, "-p", str(p)]
        env = dict(os.environ)
        if world == 1:
            cmd = [sys.executable] + args
        else:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            env["GNNPE_BENCH_SAME_DEVICE"] = "1"
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                   "--master-port", str(port)] + args
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        assert open(os.path.join(d, "gnn-pe", "all_paths.txt"), "rb").read() == _all_paths_text(c["paths"]), ci
        for j in range(p):
            got = open(os.path.join(d, "gnn-pe", "partitions", f"partition-{j}", "partition_paths.txt"), "rb").read()
            assert got == _ids_text(c[f"part{j}"]), (ci, j)
        if world == 1:
            r = subprocess.run([sys.executable] + args + ["--strict"], capture_output=True, text=True, env=env, timeout=600)
            assert r.returncode != 0 and "duplicate edge" in (r.stderr + r.stdout)
