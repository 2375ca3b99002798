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


# ---- runs of the UNMODIFIED reference that take a quarter of a minute of one host core each (it inserts 1.7e5 paths into its own R*-tree;
# it re-parses 1.4e7 text rows): started by the test that wrote their input files, collected by tests/test_zz_reference_consumers.py at
# the end of the session, so that the GPU tests in between do not wait for them
_BACKGROUND = {}


def start_reference_run(name, cmd, workdir):
    import subprocess
    import time
    out, err = os.path.join(workdir, name + ".stdout"), os.path.join(workdir, name + ".stderr")
    proc = subprocess.Popen(cmd, stdout=open(out, "w"), stderr=open(err, "w"))
    _BACKGROUND[name] = dict(proc=proc, out=out, err=err, started=time.time())


def finish_reference_run(name, timeout=1200):
    """(return code, stdout, stderr, seconds since the start) of a run started by start_reference_run; None if nobody started it."""
    import time
    job = _BACKGROUND.pop(name, None)
    if job is None:
        return None
    rc = job["proc"].wait(timeout=timeout)
    return rc, open(job["out"]).read(), open(job["err"]).read(), time.time() - job["started"]


def pytest_sessionfinish(session, exitstatus):
    for job in _BACKGROUND.values():  # (a session that stopped early: do not leave the reference running)
        if job["proc"].poll() is None:
            job["proc"].kill()
