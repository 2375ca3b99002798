"""SURVEY 8(f) row 3 -- the online side's data load from binary sidecars instead of the text re-parse.

gnnpe_host_load_path_sidecar must return exactly what the reference's gen_pde (custom.h:546-572) builds from
all_paths.txt: pinned here against strided rows of the compiled reference's own gen_pde dump
(tests/golden/test_graph/pde_sample_e2.npz, made by oracle/_ref/ref_dump) and the full golden path list.  The loader
is host code, so the fixture-driven test needs no GPU; the GPU test runs the writer side (`gnnpe_main --sidecars`)."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from gnnpe_amd import binding

TG = os.path.join(GOLDEN, "test_graph")
CLI = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")


def _golden_paths():
    rows = gzip.open(os.path.join(TG, "all_paths.txt.gz"), "rt").read().split("\n")
    P = int(rows[0])
    return np.array([r.split() for r in rows[1:1 + P]], np.uint32)


def _check_against_reference(got, ids):
    s = np.load(os.path.join(TG, "pde_sample_e2.npz"))
    idx = s["index"]
    assert np.array_equal(got["vids"], ids)
    for name in ("vids", "labels", "degrees"):
        assert np.array_equal(got[name][idx], s[name]), name
    for name in ("pde", "pde_label"):  # fp64, bit for bit
        assert np.array_equal(got[name][idx].view(np.uint64), s[name].view(np.uint64)), name


def test_loader_matches_the_reference_gen_pde(tmp_path):
    v = np.load(os.path.join(TG, "vde_e2.npz"))
    ids = _golden_paths()
    n, e = v["x"].shape
    pb, vb = str(tmp_path / "paths.bin"), str(tmp_path / "vde.bin")
    with open(pb, "wb") as f:
        f.write(b"GNNPEPTH" + struct.pack("<IIQ", 1, 3, len(ids)))
        f.write(ids.tobytes())
    with open(vb, "wb") as f:
        f.write(struct.pack("<II", n, e))
        for name in ("x", "nx", "vde"):
            f.write(np.ascontiguousarray(v[name], np.float64).tobytes())
    got = binding.host_load_path_sidecar(pb, vb, v["label"], v["degree"])
    _check_against_reference(got, ids)
    # fail-loud on damaged inputs
    with open(pb, "r+b") as f:
        f.truncate(os.path.getsize(pb) - 4)
    with pytest.raises(binding.GnnpeError, match="bad header"):
        binding.host_load_path_sidecar(pb, vb, v["label"], v["degree"])
    with pytest.raises(binding.GnnpeError, match="cannot open"):
        binding.host_load_path_sidecar(str(tmp_path / "missing.bin"), vb, v["label"], v["degree"])
    # a header whose count wraps the size check (2^62 paths x 4 vertices x 4 bytes = 0 mod 2^64 on a 24-byte file) must be
    # rejected, not used to size the buffers
    crafted = str(tmp_path / "crafted.bin")
    with open(crafted, "wb") as f:
        f.write(b"GNNPEPTH" + struct.pack("<IIQ", 1, 4, 1 << 62))
    with pytest.raises(binding.GnnpeError, match="bad header"):
        binding.host_load_path_sidecar(crafted, vb, v["label"], v["degree"])


@pytest.mark.gpu
def test_cli_sidecars_round_trip(tmp_path):
    from gnnpe_amd import synth
    graph = os.path.join(TG, "data_graph.graph")
    v = np.load(os.path.join(TG, "vde_e2.npz"))
    d = str(tmp_path)
    synth.make_dataset_dir(d, 1)
    sn = np.argsort(v["degree"], kind="stable").astype(np.uint32)
    synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), sn, np.zeros(len(sn), np.uint32))
    r = subprocess.run([CLI, "-f", d + "/", "-d", graph, "-p", "1", "--sidecars", "--chunk", "70000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = binding.host_load_path_sidecar(os.path.join(d, "gnn-pe", "paths.bin"), os.path.join(d, "gnn-pe", "vde.bin"),
                                         v["label"], v["degree"])
    _check_against_reference(got, _golden_paths())
