"""CPU-only checks of the drop-in boundary: libgnnpe_hip.so loads, exports every symbol that
include/gnnpe_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from gnnpe_amd import binding


def _declared_functions(header="gnnpe_hip.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(gnnpe_[a-z0-9_]+)\s*\(", txt))


def test_header_and_binding_agree():
    decl = _declared_functions()
    assert decl == set(binding.SIGNATURES), (decl ^ set(binding.SIGNATURES))
    online = _declared_functions("gnnpe_online.h")
    assert online == set(binding.ONLINE_SIGNATURES) and not (online & decl), (online, online & decl)


def test_library_exports_every_declared_symbol():
    binding.build()
    lib = binding.load()
    for name in _declared_functions():
        assert hasattr(lib, name), name
    assert lib.gnnpe_abi_version() == binding.ABI_VERSION


def test_refinement_lives_in_a_library_of_its_own():
    """the refinement (out of SURVEY section 8's scope) is not in libgnnpe_hip.so; libgnnpe_online.so exports what
    include/gnnpe_online.h declares and finds the rest (contexts, loader, error text) in libgnnpe_hip.so."""
    binding.build()
    lib, online = binding.load(), binding.load_online()
    for name in _declared_functions("gnnpe_online.h"):
        assert hasattr(online, name), name
        assert not hasattr(lib, name), f"{name} is still exported by libgnnpe_hip.so"


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gnn-pe_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in src and "gnnpe_oracle" not in src, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_host_label_table_matches_reference_gen_vde_x():
    # R3 is host arithmetic inside the library (std::mt19937 + uniform_real_distribution)
    z = np.load(os.path.join(GOLDEN, "label_table.npz"))
    for e in (1, 2, 3, 8):
        assert np.array_equal(binding.host_label_table(256, e), z[f"e{e}"])


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(binding.GnnpeError, match="no HIP device|no CPU fallback"):
        binding.Engine(0)


def test_index_file_bytes_uses_each_builders_own_fan_out():
    """the pre-write 2 GiB guard must size index.dat with the fan-out of the builder that writes it.  Node capacity
    (4096 - 5) / (16 D + 4) (rtnode.cpp:27-28); the pair-major build fills min(capacity - 1, 64) entries per node, the tuple-array
    build (the multi-GPU path) min(capacity - 2, 64); one block per node + the header block (blk_file.cpp:38-52)."""
    lib = binding.load()

    def blocks(points, fan):
        level = [max(1, -(-points // fan))] if points else [1]
        while points and (len(level) == 1 or level[-1] > 1):
            level.append(-(-level[-1] // fan))
        return sum(level) + 1
    for D, fan0, fan1 in ((3, 64, 64), (6, 39, 38), (12, 19, 18), (24, 9, 8), (32, 6, 5)):  # e = 1, 2, 4, 8 at l = 2; l = 3 at e = 8
        for points in (0, 1, fan0, fan0 + 1, 6400, 415_545, 19_993_708, 33_600_000):
            assert lib.gnnpe_index_file_bytes(points, D, 0) == blocks(points, fan0) * 4096, (D, points)
            assert lib.gnnpe_index_file_bytes(points, D, 1) == blocks(points, fan1) * 4096, (D, points)
    # e = 1: 64 entries per node, not capacity - 1 = 77 -- 3.36e7 paths are past 2 GiB although 77 per node would fit
    assert lib.gnnpe_index_file_bytes(33_600_000, 3, 0) >= 1 << 31 > blocks(33_600_000, 77) * 4096
    # D = 6 on the multi-GPU path: 38 per node is 2.6 % more blocks than 39
    assert lib.gnnpe_index_file_bytes(19_993_708, 6, 1) > lib.gnnpe_index_file_bytes(19_993_708, 6, 0) > 1 << 31


def test_index_file_bytes_answers_zero_for_a_dimension_no_node_holds():
    """from D = 85 on a 4096-byte block holds fewer than three entries ((4096 - 5) / (16 D + 4), rtnode.cpp:27-28); the
    function used to loop for ever there (a fan-out of 1 never reaches a single root).  It needs no GPU and no context."""
    lib = binding.load()
    for D in (85, 96, 128, 254, 255, 1000, 0):
        for builder in (0, 1):
            assert lib.gnnpe_index_file_bytes(1_000_000, D, builder) == 0, (D, builder)
    # capacity 3 (D = 64 .. 84): both builders fill two entries per node
    for builder in (0, 1):
        assert lib.gnnpe_index_file_bytes(8, 84, builder) == (4 + 2 + 1 + 1) * 4096


def test_cli_refuses_index_for_a_width_no_node_holds(tmp_path):
    """gnnpe_main --index -e 32 (D = 96) dies with a message before touching the GPU or writing anything."""
    import subprocess
    from conftest import GOLDEN
    from gnnpe_amd import synth
    import numpy as np
    cli = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")
    tmp = str(tmp_path)
    synth.make_dataset_dir(tmp, 1)
    graph = os.path.join(GOLDEN, "test_graph", "data_graph.graph")
    r = subprocess.run([cli, "-f", tmp + "/", "-d", graph, "-m", "offline", "-p", "1", "-e", "32", "--index"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "node capacity below 3" in r.stderr, r.stderr
