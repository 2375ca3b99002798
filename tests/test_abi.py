"""CPU-only checks of the drop-in boundary: libgnnpe_hip.so loads, exports every symbol that
include/gnnpe_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from gnnpe_amd import binding


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "gnnpe_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(gnnpe_[a-z0-9_]+)\s*\(", txt))


def test_header_and_binding_agree():
    decl = _declared_functions()
    assert decl == set(binding.SIGNATURES), (decl ^ set(binding.SIGNATURES))


def test_library_exports_every_declared_symbol():
    binding.build()
    lib = binding.load()
    for name in _declared_functions():
        assert hasattr(lib, name), name
    assert lib.gnnpe_abi_version() == binding.ABI_VERSION


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
