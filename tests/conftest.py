import os
import sys

import pytest
import torch  # noqa: F401  (first: its bundled HIP runtime must be the one loaded in the process)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import gnnpe_amd  # noqa: E402,F401  (import shim for the gnn-pe_amd/ package directory)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def test_graph(oracle):
    """The reference's sample graph (Test/data_graph.graph) loaded through the oracle loader,
    with the degree-sorted processing order of gnnpe.py:71-72."""
    import numpy as np
    from gnnpe_amd import synth
    offs, nbrs, labels, meta = oracle.load_graph(os.path.join(GOLDEN, "test_graph", "data_graph.graph"))
    sn = synth.degree_order(offs)
    return dict(offsets=offs, nbrs=nbrs, labels=labels, meta=meta, sorted_nodes=sn,
                membership=np.zeros(meta["n"], np.uint32))


def small_cases():
    import numpy as np
    z = np.load(os.path.join(GOLDEN, "small_graphs.npz"))
    out = []
    for ci in range(int(z["n_cases"])):
        pre = f"c{ci}_"
        out.append({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
    return out
