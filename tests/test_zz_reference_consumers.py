"""The ends of two tests whose last step is a long single-core run of the UNMODIFIED reference (`ref_main -m online`) over files the
product wrote: the runs are started where the files are written (tests/test_gpu_cli.py, tests/test_gpu_index.py, through
conftest.start_reference_run) and collected here, at the end of the session, so that the GPU tests in between do not wait for them.
This module sorts behind every other test module on purpose."""
import json
import os
import re

import pytest

import conftest
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_reference_online_prints_the_known_answer_from_our_text_files():
    """tests/test_gpu_cli.py::test_test_graph_files_byte_exact_and_online_answer[1]: all_paths.txt / partition_paths.txt of Test/ written
    by gnnpe_main; the reference builds its own index.dat from them (RTree::insert) and must print the golden answer count."""
    got = conftest.finish_reference_run("test_graph_p1_online")
    if got is None:
        pytest.skip("the producing test did not run (or oracle/_ref/ref_main is not built)")
    rc, out, err, secs = got
    assert rc == 0, err[-500:]
    gold = json.load(open(os.path.join(GOLDEN, "test_graph", "golden.json")))["p1"]
    assert int(re.search(r"Answer Number: (\d+)", out).group(1)) == gold["answer_number"] == 45426


def test_reference_online_answers_from_every_partition_index_we_built():
    """tests/test_gpu_index.py::test_index_size_guard_and_every_partition_through_the_reference: G(70K, 700K), p = 4, all four index.dat
    built by `gnnpe_main --index`: the reference inserts nothing and prints the answer it printed from the trees it had inserted
    itself (tests/golden/large_index/reference_p4.json, a 33-minute run of the reference)."""
    got = conftest.finish_reference_run("g70_p4_online")
    if got is None:
        pytest.skip("the producing test did not run (or oracle/_ref/ref_main is not built)")
    rc, out, err, secs = got
    assert rc == 0, err[-300:]
    ref = json.load(open(os.path.join(GOLDEN, "large_index", "reference_p4.json")))
    assert "This R-Tree contains" not in out  # it inserted nothing: every partition's tree came from our files
    assert int(re.search(r"Answer Number: (\d+)", out).group(1)) == ref["answer_number"] == 2
