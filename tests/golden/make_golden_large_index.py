#!/usr/bin/env python3
"""Large index.dat through the untouched consumer (TEST INFRASTRUCTURE).

G(70K, 700K), l=2, e=2, p=1: 1.4e7 paths, ~3.8e5 node blocks (1.5 GB of index.dat) -- exercises what Test/ (15K blocks)
cannot: the reference online side's 100 000-entry heap (GNN-PE/include/heap/heap.h:3) and its block-id-indexed arrays
(custom.h:261,264,379).  It is also about the largest single partition the reference can read at all: its block file
seeks with 32-bit offsets (`fseek(fp, (bnum - act_block) * blocklength, SEEK_CUR)`, include/blockfile/blk_file.h:33),
so an index.dat of 2 GiB or more breaks the consumer whoever wrote it.  At BASELINE config 2 with p = 1 (2.0e7 paths)
the reference's OWN run dies that way: its insert-built tree reaches 2 147 819 520 bytes and `main -m online` ends
with SIGSEGV after 1 968 s (large_index/reference_config2_p1_crash.json, recorded by part A of this script at that size).

  part A (this container, CPU only, ~40 min):  the reference does everything itself -- `ref_main -m offline`, then
         `ref_main -m online`, whose first run builds index.dat by 2e7 R*-tree inserts -- and prints Answer Number.
  part B (GPU box, ~3 min):  `gnnpe_main -m offline --index` writes the text files and the bulk-loaded index.dat;
         the SAME reference online binary consumes them.  Same Answer Number, no heap overflow (exit code 0).

    python tests/golden/make_golden_large_index.py A      # writes large_index/reference.json + query.graph
    python tests/golden/make_golden_large_index.py B      # on the GPU box; writes large_index/ours.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import synth  # noqa: E402
from oracle import ref_main_path  # noqa: E402

OUT = os.path.join(HERE, "large_index")
QUERY = os.path.join(OUT, "query.graph")
N, M = 70_000, 700_000
P = int(os.environ.get("GNNPE_LARGE_INDEX_P", "1"))  # partitions; p = 4 is a size the reference's own heap survives


def make_query(g, seed=7):
    """A connected 6-vertex query cut out of the data graph (deterministic)."""
    rng = np.random.default_rng(seed)
    offs, nbrs = g["offsets"].astype(np.int64), g["nbrs"]
    while True:
        comp = [int(rng.integers(g["n"]))]
        for _ in range(400):
            if len(comp) == 6:
                break
            u = comp[int(rng.integers(len(comp)))]
            w = int(nbrs[int(rng.integers(offs[u], offs[u + 1]))]) if offs[u + 1] > offs[u] else u
            if w not in comp:
                comp.append(w)
        if len(comp) == 6:
            break
    idx = {v: i for i, v in enumerate(comp)}
    edges = sorted({(min(idx[v], idx[int(w)]), max(idx[v], idx[int(w)])) for v in comp for w in nbrs[offs[v]:offs[v + 1]] if int(w) in idx})
    deg = np.zeros(6, np.int64)
    for a, b in edges:
        deg[a] += 1
        deg[b] += 1
    with open(QUERY, "w") as f:
        f.write(f"t 6 {len(edges)}\n" + "".join(f"v {i} {int(g['labels'][v])} {int(deg[i])}\n" for i, v in enumerate(comp))
                + "".join(f"e {a} {b}\n" for a, b in edges))


def dataset(wd, g):
    gp = os.path.join(wd, "g.graph")
    synth.write_graph_file(gp, g)
    synth.make_dataset_dir(wd, P)
    synth.write_membership(os.path.join(wd, "gnn-pe", "membership.txt"), synth.degree_order(g["offsets"]), synth.block_membership(g["n"], P))
    return gp


def online(wd, gp):
    t0 = time.time()
    r = subprocess.run([ref_main_path(), "-f", wd + "/", "-d", gp, "-q", QUERY, "-m", "online", "-p", str(P)], capture_output=True, text=True)
    m = re.search(r"Answer Number: (\d+)", r.stdout)
    return dict(returncode=r.returncode, answer_number=int(m.group(1)) if m else None, seconds=round(time.time() - t0, 1),
                stdout_tail=r.stdout[-300:], stderr_tail=r.stderr[-300:])


def main():
    part = sys.argv[1] if len(sys.argv) > 1 else "A"
    os.makedirs(OUT, exist_ok=True)
    g = synth.gnm_graph(N, M)
    with tempfile.TemporaryDirectory() as wd:
        gp = dataset(wd, g)
        idx = os.path.join(wd, "gnn-pe", "partitions", "partition-0", "index.dat")
        if part == "A":
            make_query(g)
            t0 = time.time()
            subprocess.check_call([ref_main_path(), "-f", wd + "/", "-d", gp, "-m", "offline", "-p", str(P)], stdout=subprocess.DEVNULL)
            off_s = round(time.time() - t0, 1)
            res = online(wd, gp)
            hdr = np.fromfile(idx, np.int32, 6)  # partition 0
            res.update(offline_seconds=off_s, paths=int(open(os.path.join(wd, "gnn-pe", "all_paths.txt")).readline()),
                       index_bytes=os.path.getsize(idx), node_blocks=int(hdr[1]), what="reference offline + its own insert-built R*-tree")
            res["partitions"] = P
            json.dump(res, open(os.path.join(OUT, "reference.json" if P == 1 else f"reference_p{P}.json"), "w"), indent=1)
        else:
            cli = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")
            t0 = time.time()
            subprocess.check_call([cli, "-f", wd + "/", "-d", gp, "-m", "offline", "-p", str(P), "--index"], stdout=subprocess.DEVNULL)
            off_s = round(time.time() - t0, 2)
            res = online(wd, gp)
            hdr = np.fromfile(idx, np.int32, 6)
            res.update(offline_seconds=off_s, index_bytes=os.path.getsize(idx), node_blocks=int(hdr[1]),
                       what="gnnpe_main -m offline --index (bulk-loaded index.dat), consumed by the untouched reference online binary")
            ref = json.load(open(os.path.join(OUT, "reference.json" if P == 1 else f"reference_p{P}.json")))
            res["partitions"] = P
            res["reference_answer_number"] = ref["answer_number"]  # None: the reference's own tree overflows its heap
            # second opinion where the reference has none: this engine's own online side (filter + refinement on the GPU,
            # the reference's semantics; frozen round-1 code)
            r2 = subprocess.run([cli, "-f", wd + "/", "-d", gp, "-q", QUERY, "-m", "online", "-p", str(P)], capture_output=True, text=True)
            m2 = re.search(r"Answer Number: (\d+)", r2.stdout)
            res["gnnpe_main_online_answer_number"] = int(m2.group(1)) if m2 else None
            res["matches_reference"] = (res["returncode"] == 0 and res["answer_number"] is not None and
                                        res["answer_number"] == (ref["answer_number"] if ref["answer_number"] is not None
                                                                 else res["gnnpe_main_online_answer_number"]))
            json.dump(res, open(os.path.join(os.environ.get("GNNPE_LARGE_INDEX_OUT", OUT), "ours.json" if P == 1 else f"ours_p{P}.json"), "w"), indent=1)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
