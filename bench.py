#!/usr/bin/env python3
"""bench.py -- offline path-embedding build throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Launch rule: one process per GPU.  Under a launcher (WORLD_SIZE set) this file is one rank.  Started plainly with
--gpus N > 1 it starts N rank processes itself (`launch_ranks`: children of `python -m torch.distributed.run`, before this
process touches HIP), relays rank 0's JSON line and leaves with their exit code; N = 1 runs in this process.

One STEP = one full pass of the hot path over the synthetic 1M-vertex / 10M-edge labelled graph
(BASELINE.json configs[2]; l=2, e=2): vde [N>1: + all-gather of the vde rows] -> count (rank-sorted row blocks,
pair records, scan) -> fill (path ids + fp64 path embeddings written to HBM).  Inputs resident in HBM before the
timed region: this rank's CSR rows with their reverse positions, the halo rows (N>1: fetched once by all-to-all-v
when the graph is distributed -- graph structure, like the CSR itself), the processing order and the label table.
N>1 partitions the SAME graph across the ranks (configs[3]), so scaling is "strong".
value = paths of all ranks / max-over-ranks step time.

The JSON line also carries
  roofline     -- the dominant kernel (k_fill_ranked): algorithmic bytes (92 B/path at l=2,e=2) / its
                  launch duration, timed live with events on the launch stream; `step_frac` = the same bytes over
                  the whole step; `traffic` = HBM bytes per launch from this round's PMC passes (profiles/);
  cpu_baseline -- the UNMODIFIED reference `main -m offline` (oracle/_ref/ref_main), single thread, on a bounded
                  sample of the same generator (rank 0, N=1 only);
  index_build  -- second half of the metric: R-tree image on the device AND the index.dat files on disk (p=1, p=8);
  e2e          -- `gnnpe_main -m offline` wall-clock (load, emit, render, write) at configs 3 and 2;
  phases_ms    -- one-time work (graph distribution) separated from the per-step phases.
RCCL is mandatory for N>1: a failed init is an error (exit code != 0), never a silent downgrade.  Host-staged gloo
collectives are an explicit debugging mode (GNNPE_BENCH_SAME_DEVICE=1: all ranks on device 0).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import binding, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LEAF_TRAFFIC_BYTES = 27.7e9  # index leaf kernel (image alone) at config 3, per launch: 7.0 GB read + 20.7 GB written, profiles/r06_leaf_mem_pmc.txt
                            # (the leaf kernel is round 5's: no counter pass of it in round 6)
CLI = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_fill.json")


def bytes_per_path(L, e):
    """B_path = 4L (ids) + 8eL (pde) + 16 (two candidate reads: id + rank) + 8e (vde[c])."""
    return 4 * L + 8 * e * L + 16 + 8 * e


def cpu_baseline(sample_n, sample_m, seed, named=None):
    """Time the reference's own offline step on a bounded sample (or on a whole BASELINE config: `named`).  Only this leg
    (and tests / smoke) may touch oracle/."""
    from oracle import Oracle, ref_main_path
    g = synth.gnm_graph(sample_n, sample_m, seed=seed)
    sn = synth.degree_order(g["offsets"])
    P = synth.expected_paths_l2(g["offsets"])
    sample = f"G(n={sample_n}, m={sample_m}) same generator/seed family, l=2, p=1: {P} paths"
    if named:
        sample = f"{named} at FULL size: " + sample
    if os.path.exists(ref_main_path()):
        with tempfile.TemporaryDirectory() as wd:
            gp = os.path.join(wd, "g.graph")
            synth.write_graph_file(gp, g)
            synth.make_dataset_dir(wd, 1)
            synth.write_membership(os.path.join(wd, "gnn-pe", "membership.txt"), sn, np.zeros(sample_n, np.uint32))
            t0 = time.perf_counter()
            subprocess.check_call([ref_main_path(), "-f", wd + "/", "-d", gp, "-m", "offline", "-p", "1"],
                                  stdout=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            hdr = int(open(os.path.join(wd, "gnn-pe", "all_paths.txt")).readline())
            assert hdr == P, (hdr, P)
        kind = "reference"
        sample += "; whole `main -m offline` wall-clock (load + DFS/hash-set + both text files)"
    else:
        orc = Oracle()
        t0 = time.perf_counter()
        paths = orc.enumerate_dfs_hash(g["offsets"], g["nbrs"], sn, 3)
        orc.format_all_paths(paths)
        dt = time.perf_counter() - t0
        assert len(paths) == P
        kind = "port"
        sample += "; oracle hash-set DFS + text formatting"
    return dict(value=P / dt, unit="paths/s", cores=1, kind=kind, sample=sample, seconds=dt,
                note=("a BASELINE config timed on this box's host cores; " if named else "the small sample flatters the reference: its hash set still fits the caches here; ") +
                     "at the headline size (1M/10M, 2.0e8 paths) the same binary measured 1.14e5 paths/s (1 749 s, 29.9 GB RSS; BASELINE.md section 2.1)")


def cpu_baseline_all_cores(sample_n, sample_m, e, seed):
    """Second CPU leg: the oracle's all-core port of the SAME device-resident pass the GPU is
    timed on (vde + count + prefix + ids/pde fill into memory, closed form, OpenMP), outputs preallocated."""
    from oracle import Oracle
    orc = Oracle()
    g = synth.gnm_graph(sample_n, sample_m, seed=seed)
    sn = synth.degree_order(g["offsets"])
    P = synth.expected_paths_l2(g["offsets"])
    ids = np.zeros((P, 3), np.uint32)
    pde = np.zeros((P, 3 * e))
    ids[:] = 1  # touch the pages: the GPU side does not pay for allocation either
    pde[:] = 1.0
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        got = orc.offline_parallel(g["offsets"], g["nbrs"], g["labels"], sn, e, 0, ids, pde)[0]
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert got == P and int(ids[:, 1].astype(np.int64).sum()) > 0
    return dict(value=P / best, unit="paths/s", cores=orc.max_threads(), kind="port",
                sample=f"G(n={sample_n}, m={sample_m}), l=2, e={e}: {P} paths; oracle's OpenMP closed-form pass "
                       f"(vde + count + prefix + ids/pde fill into preallocated memory, no text)", seconds=best)


def online_filter_leg(eng, g, seed):
    """Cut a connected 8-vertex query out of the data graph, plan it on the host, filter on the device."""
    rng = np.random.default_rng(seed)
    offs, nbrs = g["offsets"].astype(np.int64), g["nbrs"]
    deg = np.diff(offs)
    chosen = None
    for _ in range(64):  # bounded: a start inside a component with fewer than 8 vertices is simply dropped
        start = int(rng.integers(g["n"]))
        comp, seen, tries = [start], {start}, 0
        while len(comp) < 8 and tries < 512:
            tries += 1
            u = comp[int(rng.integers(len(comp)))]
            if deg[u]:
                w = int(nbrs[int(rng.integers(offs[u], offs[u + 1]))])
                if w not in seen:
                    seen.add(w)
                    comp.append(w)
        if len(comp) == 8:
            chosen = comp
            break
    if chosen is None:
        return dict(skipped="no connected 8-vertex subgraph found in 64 attempts")
    idx = {v: i for i, v in enumerate(chosen)}
    edges = sorted({(min(idx[v], idx[int(w)]), max(idx[v], idx[int(w)])) for v in chosen for w in nbrs[offs[v]:offs[v + 1]]
                    if int(w) in idx})
    qdeg = np.zeros(8, np.int64)
    for a, b in edges:
        qdeg[a] += 1
        qdeg[b] += 1
    with tempfile.TemporaryDirectory() as wd:
        qp = os.path.join(wd, "q.graph")
        with open(qp, "w") as f:
            f.write(f"t 8 {len(edges)}\n" + "".join(f"v {i} {int(g['labels'][v])} {int(qdeg[i])}\n" for i, v in enumerate(chosen))
                    + "".join(f"e {a} {b}\n" for a, b in edges))
        plan = binding.host_query_plan(qp, 2)
    ms = []
    for _ in range(3):
        bm, t = eng.filter_candidates(plan)
        ms.append(t)
    cand = [int(np.unpackbits(r.view(np.uint8)).sum()) for r in bm]
    return dict(device_ms=min(ms), query_vertices=8, query_edges=len(edges), plan_paths=int(len(plan["vids"])),
                candidates_per_query_vertex=cand,
                what="leaf test of Partition::query (custom.h:404-431) on every path, fused with the enumeration (nothing emitted); no index, no files",
                reference="re-parses all_paths.txt (~95 s per 2e7 paths, custom.h:546-572) and inserts/loads the R-tree first")


def device_pass(torch, stream, local_rank, g, sn, labels, e, steps):
    """The timed step (vde + count + fill, inputs resident) on another graph of the same family: the config-2 line."""
    eng = binding.Engine(local_rank, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(labels, e))
    eng.vde(want=False)
    total = eng.count_paths(2)
    dev = torch.device("cuda", local_rank)
    ids = torch.empty((max(total, 1), 3), dtype=torch.int32, device=dev)
    pde = torch.empty((max(total, 1), 3 * e), dtype=torch.float64, device=dev)
    fill = []
    for it in range(steps + 2):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        eng.vde(want=False)
        t = eng.count_paths(2)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        eng.fill_paths_device(0, t, ids, pde, None)
        ev1.record()
        if it >= 2:
            fill.append((ev0, ev1))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    fms = float(np.mean([a.elapsed_time(b) for a, b in fill]))
    assert total == synth.expected_paths_l2(g["offsets"])
    eng.close()
    bpp = bytes_per_path(3, e)
    return dict(workload=f"config 2: G(n={g['n']}, m={g['m']}), l=2, e={e}, ids + pde", paths=total, ms_per_step=ms,
                value=total / (ms / 1e3), unit="paths/s", fill_ms=fms, fill_frac=total * bpp / (fms / 1e3) / 1e9 / HBM_PEAK_GBS)


def pge_leg(torch, stream, local_rank, g, labels, e, steps):
    """SURVEY 8(f)1, GNN-PGE offline (GNN-PGE/src/main.cpp:91-195): per-vertex path groups (segmented min / max over the
    neighbours' embeddings and label features) and the R-tree over the vertices' boxes, on the headline graph."""
    eng = binding.Engine(local_rank, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(np.arange(g["n"], dtype=np.uint32), np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(labels, e))
    eng.vde(want=False)
    pg, _ = eng.pge_groups_device()
    ev = []
    for it in range(steps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.pge_groups_device()
        b.record()
        if it >= 2:
            ev.append((a, b))
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    n, m2 = g["n"], len(g["nbrs"])
    # algorithmic bytes: per vertex its row bounds (8) and the two output rows (2 x 4e doubles); per adjacency entry the
    # neighbour id (4), its label (4) and its embedding (8e); the two memsets of the outputs are part of the call
    bytes_alg = n * (8 + 2 * 32 * e) + m2 * (8 + 8 * e)
    idx = []
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img, nbytes, hdr = eng.build_box_index_device(n, 2 * e, pg)  # every vertex' box, one partition (p = 1)
        torch.cuda.synchronize()
        idx.append((time.perf_counter() - t0) * 1e3)
    out = dict(workload=f"GNN-PGE offline on G(n={n}, m={m2 // 2}), e={e}: path_group + path_label_group of every vertex, R-tree over the {n} boxes",
               kernel="k_pge_groups", groups_ms=ms, algorithmic_bytes=bytes_alg, groups_frac=bytes_alg / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
               vertices_per_s=n / (ms / 1e3), index_first_call_ms=idx[0], index_build_ms=min(idx[1:]), index_bytes=int(nbytes),
               data_vertices_bin_bytes=4 + n * (12 + 8 + 8 * (3 * e + 8 * e)),
               note="bound by the 2m random gathers of vde[u] (one per adjacency entry), not by bytes; the reference computes the same "
                    "values with 2e gathers per entry on one thread (GNN-PGE/src/main.cpp:141-176)")
    eng.close()
    return out


def config5_leg(torch, stream, local_rank, labels, seed):
    """BASELINE config 5 on ONE GPU (the 8-GPU split of it is in tests/test_gpu_slabs_full.py): power-law 4M / 64M, l = 3
    (4-vertex paths: the reference's rule with the depth fixed, SURVEY D4 -- parity unpinned, the count is checked in the
    tests against the closed form sum_E (du-1)(dv-1) - 3T), e = 8.  The 4.2e13 paths fit nowhere, so the leg times the count
    (vde + per-row rank sort + k_deep3_count_hist / _coop + scans) and the emission (k_deep3_slices_fused, round 6: one wave per slice counts its kept rows, learns its first slot by look-back inside the unit and writes) on sampled ranges of 2^24 and 2^26 paths."""
    L, e = 4, 8
    t0 = time.perf_counter()
    g = synth.powerlaw_graph(4_000_000, 64_000_000, exponent=2.1, max_degree=3000, n_labels=labels, seed=seed)
    sn = synth.degree_order(g["offsets"])
    t_gen = time.perf_counter() - t0
    dev = torch.device("cuda", local_rank)
    eng = binding.Engine(local_rank, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(labels, e))
    eng.vde(want=False)
    total = eng.count_paths(3)  # sizes every internal buffer
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.vde(want=False)
    total2 = eng.count_paths(3)
    torch.cuda.synchronize()
    t_count = time.perf_counter() - t0
    assert total2 == total
    vms = []
    for _ in range(3):  # gen_vde alone (k_x_from_labels + k_vde + k_vde_hubs): rows up to 3 000 entries long
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        eng.vde(want=False)
        ev1.record()
        torch.cuda.synchronize()
        vms.append(ev0.elapsed_time(ev1))
    bpp = bytes_per_path(L, e)
    samples = []
    for log2_chunk in (24, 26):  # 2^24 paths = a few hundred work units (a starved chip); 2^26 = what a bulk emission queues
        chunk = 1 << log2_chunk
        ids = torch.empty((chunk, L), dtype=torch.int32, device=dev)
        pde = torch.empty((chunk, L * e), dtype=torch.float64, device=dev)
        for frac_at in (0.0, 0.37, 0.81):
            b = min(int(total * frac_at), total - chunk) if total > chunk else 0
            c = min(chunk, total - b)
            ms = []
            for _ in range(3):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                eng.fill_paths_device(b, b + c, ids, pde, None)
                ev1.record()
                torch.cuda.synchronize()
                ms.append(ev0.elapsed_time(ev1))
            m = min(ms[1:])
            samples.append(dict(first_path=b, paths=c, emit_ms=m, frac=c * bpp / (m / 1e3) / 1e9 / HBM_PEAK_GBS))
        del ids, pde
        torch.cuda.empty_cache()
    eng.close()
    deg = np.diff(g["offsets"].astype(np.int64))
    return dict(workload=f"config 5: power-law n=4000000 m=64000000 (max degree {int(deg.max())}), l=3, e=8, one GPU", paths=total,
                vde_count_s=t_count, vde_ms=min(vms), count_paths_per_s=total / t_count, kernel="k_deep3_slices_fused", bytes_per_path=bpp, emit_samples=samples,
                emit_frac=float(np.mean([x["frac"] for x in samples if x["paths"] >= 1 << 26] or [x["frac"] for x in samples])),
                emit_frac_note="mean over the 2^26-path ranges (18 GB of output each); the 2^24-path ranges are listed too",
                host_graph_generation_s=t_gen,
                parity="unpinned: the reference cannot run l=3 (SURVEY D4); the count equals the closed form sum_E (du-1)(dv-1) - 3T "
                       "computed by the oracle at full size (tests/test_gpu_slabs_full.py::test_config5_4m_64m_powerlaw_l3_e8)")


def index_l3_leg(torch, stream, local_rank, labels, l2_ms_per_gb):
    """R6 at l = 3: the triple-major index build (csrc/gnnpe_index_deep.hip.h) on two graphs whose partition image
    is the size of config 3's (22-26 GB): e = 2 (29 entries of 132 bytes per leaf) and config 5's e = 8 (six entries of 516 bytes), per
    byte of image beside the l = 2 pair-major build of the same run."""
    rows = []
    for n, m, e in ((70_000, 600_000, 2), (40_000, 220_000, 8)):
        g = synth.gnm_graph(n, m, n_labels=labels)
        eng = binding.Engine(local_rank, stream=stream.cuda_stream)
        eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
        eng.set_order(synth.degree_order(g["offsets"]), np.zeros(g["n"], np.uint32), 1)
        eng.set_label_table(binding.host_label_table(labels, e))
        eng.vde(want=False)
        full, cached = [], []
        for _ in range(3):
            total = eng.count_paths(3)
            for dst in (full, cached):  # the first build of a count sorts the units, the second reuses their order
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                _, nbytes, hdr = eng.build_index_partition_device(0)
                ev1.record()
                torch.cuda.synchronize()
                dst.append(ev0.elapsed_time(ev1))
        eng.close()
        gb = nbytes / 1e9
        rows.append(dict(workload=f"G({n}, {m}), l=3, e={e}, p=1", points=total, file_bytes=nbytes, leaves=hdr[4],
                         wallclock_ms=min(full[1:]), next_partition_ms=min(cached[1:]), ms_per_gb=min(full[1:]) / gb,
                         per_byte_vs_l2=(min(full[1:]) / gb) / l2_ms_per_gb if l2_ms_per_gb else None))
    return dict(builds=rows, builder="triple-major: units (s, b, c) x 64-entry pieces of c's row sorted by the 3-vertex key, one wave per leaf "
                                     "(k_tx_units, radix sort, k_tx_gather, k_tx_leaves, k_tx_inner)",
                note="per_byte_vs_l2 = (wallclock_ms / image GB) over the same ratio of index_build (the l = 2 pair-major build of this run); "
                     "until round 6 an l = 3 context took the tuple-array build (index_build.tuple_array_build_ms per 22 GB, after emitting the "
                     "enumeration twice to collect the partition's tuples)")


def leg_failed(out, key, ex):
    """An optional leg raised: the bench line keeps its headline and says what was lost (a leg is outside the timed steps)."""
    import traceback
    sys.stderr.write(f"[bench] leg {key} failed:\n{traceback.format_exc()}\n")
    if isinstance(out.get(key), dict):
        out[key]["error"] = f"{type(ex).__name__}: {ex}"[:400]
    else:
        out[key] = {"error": f"{type(ex).__name__}: {ex}"[:400]}


def guarded(out, key, fn, *a, **kw):
    try:
        out[key] = fn(*a, **kw)
    except Exception as ex:
        leg_failed(out, key, ex)


def e2e_leg(g, sn, p, index, label, allow_large=False):
    """Wall-clock of `gnnpe_main -m offline` on text inputs (load + emit + render + file writes [+ index.dat])."""
    free = shutil.disk_usage(tempfile.gettempdir()).free
    P = synth.expected_paths_l2(g["offsets"])
    need = P * 34 + (P * 120 if index else 0) + (400 << 20)
    if free < need:
        return dict(skipped=f"{label}: needs {need >> 30} GiB of scratch disk, {free >> 30} GiB free")
    with tempfile.TemporaryDirectory() as wd:
        gp = os.path.join(wd, "g.graph")
        synth.write_graph_file(gp, g)
        synth.make_dataset_dir(wd, p)
        synth.write_membership(os.path.join(wd, "gnn-pe", "membership.txt"), sn, synth.block_membership(g["n"], p))
        args = [CLI, "-f", wd + "/", "-d", gp, "-m", "offline", "-p", str(p), "--timing"] + (["--index"] if index else []) + (["--allow-large"] if allow_large else [])
        t0 = time.perf_counter()
        r = subprocess.run(args, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            return dict(error=r.stderr[-500:])
        t = json.loads(r.stderr.strip().splitlines()[-1])
        hdr = int(open(os.path.join(wd, "gnn-pe", "all_paths.txt")).readline())
        assert hdr == P == t["paths"], (hdr, P)
        out = dict(workload=label, partitions=p, seconds=dt, paths=P, paths_per_s=P / dt, in_process_s=t["end_to_end_s"],
                   load_s=t["load_s"], setup_s=t.get("setup_s"), vde_count_s=t["vde_count_s"],
                   emit_render_copy_s=t.get("emit_render_copy_s"), write_total_s=t.get("write_total_s"),
                   text_bytes=t["all_paths_bytes"] + t["partition_bytes"])
        if index:
            sizes = [os.path.getsize(os.path.join(wd, "gnn-pe", "partitions", f"partition-{i}", "index.dat")) for i in range(p)]
            out.update(index_build_s=t["index_build_s"], index_bytes=int(sum(sizes)))
        return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: one child process per GPU through torch.distributed.run
    (rendezvous on 127.0.0.1, a free port), the children's stdout / stderr inherited so rank 0's JSON line is this
    command's JSON line.  The parent never initialises HIP (no torch.cuda call, no engine); it only waits.  Returns the
    launcher's exit code (non-zero when any rank failed)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    global T_START
    T_START = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--vertices", dest="n", type=int, default=1_000_000)
    ap.add_argument("--edges", dest="m", type=int, default=10_000_000)
    ap.add_argument("--embedding", dest="e", type=int, default=2)
    ap.add_argument("--labels", type=int, default=64)
    ap.add_argument("--seed", type=int, default=synth.SEED)
    ap.add_argument("--powerlaw", action="store_true", help="power-law degrees (config-5 family) instead of G(n,m)")
    ap.add_argument("--max-degree", type=int, default=500)
    ap.add_argument("--ids-only", action="store_true", help="emit path ids only (28 B/path variant)")
    ap.add_argument("--fill-variant", type=int, default=4, choices=(1, 4))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-index", action="store_true", help="skip the index-build, online-filter and end-to-end legs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (file-writing) legs only")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 leg (power-law 4M/64M, l=3, e=8: about a minute of host graph generation)")
    ap.add_argument("--cpu-sample", type=str, default="30000,300000")
    ap.add_argument("--cpu-config2", choices=("auto", "yes", "no"), default="auto",
                    help="time the unmodified reference's `main -m offline` on BASELINE config 2 at full size (100K/1M, 2.0e7 paths: about "
                         "160 s on one host core) as `cpu_baseline`; auto = when the run is younger than --cpu-config2-before seconds by then")
    ap.add_argument("--cpu-config2-before", type=float, default=240.0)
    ap.add_argument("--placements", type=int, default=1,
                    help="candidate allocations the library's output pool draws (gnnpe_output_pool_create: the one the emit kernel "
                         "writes fastest is kept, the others freed); 1 (default) = one allocation that takes what comes -- the "
                         "pool then only measures which of the two emit shapes is faster into it")
    ap.add_argument("--compare-pool", type=int, default=3,
                    help="with --placements 1: also time the steps into the kept one of this many candidate allocations (rounds 2-3 "
                         "reported that as the headline); reported as roofline.drawn_pool, never as `value`; 0 = skip")
    ap.add_argument("--equal-paths", action="store_true",
                    help="N>1 slab planning: equal path counts instead of the fitted step-cost model (dist.STEP_COST_WEIGHTS)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N rank processes ourselves (children, never an exec) BEFORE anything
        # in this process touches HIP, relay their output, leave with their exit code
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    # debugging aid for boxes with one GPU: GNNPE_BENCH_SAME_DEVICE=1 puts every rank on device 0 and
    # carries the collectives over gloo (RCCL refuses two ranks on one device).  Never used by the driver.
    same_device = os.environ.get("GNNPE_BENCH_SAME_DEVICE") == "1"
    if same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    staged = same_device  # collectives over gloo with host staging instead of RCCL: explicit, for every rank alike
    # GNNPE_BENCH_FORCE_RCCL=1 (validation aid for single-GPU boxes): a 1-rank process group over RCCL and the N > 1 code of
    # this file -- slab rows, halo plan, vde all-gather, totals all-gather -- so that the multi-GPU path of the bench itself
    # has run before the first N > 1 launch (tests/test_gpu_rccl.py)
    force_rccl = os.environ.get("GNNPE_BENCH_FORCE_RCCL") == "1" and world == 1 and not same_device
    multi = world > 1 or force_rccl
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")  # (torch.distributed.run sets its own)
        if staged:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            # RCCL over xGMI, or no number at all: any failure here ends the run with a non-zero exit code on every
            # rank (the probe is a collective, so no rank can pass it alone)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            probe = torch.arange(world, dtype=torch.int32, device=device)
            got = torch.empty_like(probe)
            dist.all_to_all_single(got, probe, output_split_sizes=[1] * world, input_split_sizes=[1] * world)
            torch.cuda.synchronize()
            if got.tolist() != [rank] * world:
                raise SystemExit(f"rank {rank}: RCCL all-to-all-v probe returned {got.tolist()}")

    L, e = 3, args.e
    if args.powerlaw:
        g = synth.powerlaw_graph(args.n, args.m, exponent=2.1, max_degree=args.max_degree, n_labels=args.labels, seed=args.seed)
    else:
        g = synth.gnm_graph(args.n, args.m, n_labels=args.labels, seed=args.seed)
    sn = synth.degree_order(g["offsets"])
    mem = synth.block_membership(args.n, max(world, 1))

    from gnnpe_amd.dist import STEP_COST_WEIGHTS, SlabBuild, owned_rows, plan_slabs
    # a dedicated (non-null) stream shared by torch and the engine, so torch events bracket the
    # engine's kernels (handle 0 = "context's own stream" in the C-ABI)
    stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    eng = binding.Engine(local_rank, stream=stream.cuda_stream)
    bounds = plan_slabs(g["offsets"], sn, world, g["nbrs"], weights=(1.0, 0.0, 0.0) if args.equal_paths else STEP_COST_WEIGHTS)
    owned_entries = len(g["nbrs"])
    one_time = {}
    t_load = time.perf_counter()
    if not multi:
        eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    else:
        rows, roff, rnbr = owned_rows(g, sn, bounds, rank)
        owned_entries = int(roff[-1])
        eng.load_rows(args.n, g["labels"], rows, roff, rnbr, nbr_capacity=len(g["nbrs"]))
    eng.set_order(sn, mem, max(world, 1))
    eng.set_slab(int(bounds[rank]), int(bounds[rank + 1]))
    eng.set_label_table(binding.host_label_table(args.labels, e))
    eng.set_fill_variant(args.fill_variant)
    torch.cuda.synchronize()
    one_time["load_rows_revpos_ms"] = (time.perf_counter() - t_load) * 1e3
    sb = SlabBuild(eng, args.n, e, bounds, rank, world, device, nbr_capacity=len(g["nbrs"]), owned_entries=owned_entries,
                   force_collectives=force_rccl)
    if multi:  # distribute the graph: the halo rows arrive once and stay (like the CSR of a single-GPU run)
        t_h = time.perf_counter()
        sb.install_halo()
        torch.cuda.synchronize()
        one_time["halo_install_ms"] = (time.perf_counter() - t_h) * 1e3

    # first pass sizes the outputs (and every internal buffer); not timed
    total, base = sb.step()

    # Where the 12 GB of output land matters on this hardware (a resident store loop streams into a buffer at
    # ~5.0 or ~6.3 TB/s for the buffer's whole life, not predictable from its address).  Since round 4 the headline run takes
    # ONE allocation as it comes (`--placements 1`) and the library measures which EMIT SHAPE is faster into it
    # (gnnpe_emit_calibrate_device, run by gnnpe_output_pool_create on the buffer it keeps: start-vertex waves win in the fast
    # allocations, output tiles in the slow ones).  `--placements N` still draws N candidate allocations (kept for comparison;
    # the run then also times a plain allocation beside the kept one, roofline.plain_allocation).
    D_out = 0 if args.ids_only else e * L
    pool = binding.OutputPool(eng, max(total, 1), L, D_out, candidates=max(1, args.placements))
    pool_rep = pool.report()
    out_ids, out_pde = pool.ids, (pool.pde if D_out else None)
    cap_rows = pool.rows_cap
    ids_view = pool.ids_tensor(device)
    # both emit shapes timed into the kept buffer (the pool has done this already; again here for the numbers)
    shapes = eng.emit_calibrate_device(out_ids, out_pde, rows_cap=cap_rows) if (args.fill_variant == 4 and total > 0) else None

    fill_ms = []
    enqueue_only = args.fill_variant == 4 and e in (1, 2, 3, 4, 8)

    def one_step(timed, ids=None, pde=None, keep=None):
        """vde [+ all-gather] -> count -> fill, ENQUEUED: no read-back between the launches (the count leaves its total on the
        device, the fill clips against the buffers' capacity); N > 1 collects the ranks' totals after the fill is queued."""
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        o_ids, o_pde = (out_ids, out_pde) if ids is None else (ids, pde)
        if not enqueue_only:  # the generic pair-wave kernel (A/B baseline, widths without a specialised kernel): count with read-back
            if multi:
                sb.exchange_vde()
                t = sb.count_begin()
            else:
                eng.vde(want=False)
                t = sb._count_single()
            ev0.record()
            eng.fill_paths_device(0, t, o_ids, o_pde, None)
            ev1.record()
        else:
            if multi:
                sb.exchange_vde()
                sb.count_enqueue()  # count + async all-gather of the totals, straight from the engine's device word
            else:
                eng.vde(want=False)
                eng.count_paths_enqueue(2)
            ev0.record()
            eng.fill_paths_capped_device(cap_rows, o_ids, o_pde)
            ev1.record()
        if multi:
            sb.count_end()
        if timed:
            (fill_ms if keep is None else keep).append((ev0, ev1))

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    barrier()
    dt = time.perf_counter() - t0
    if not multi:
        assert (eng.count_total() if enqueue_only else sb.local_total) == total, "the timed steps counted a different number of paths than the sizing pass"
        sb.local_total = sb.global_total = total
        sb.base = 0
    else:
        assert sb.local_total == total

    # for comparison with rounds 2-3, whose headline was the best of 12 (5) candidate allocations: the same steps into the
    # kept one of `--compare-pool` candidates (transient memory: that many outputs; 0 = skip)
    drawn = None
    if not multi and args.placements == 1 and args.compare_pool > 1 and args.fill_variant == 4 and total >= (1 << 24):
        pool2 = binding.OutputPool(eng, max(total, 1), L, D_out, candidates=args.compare_pool)
        d_ev = []
        one_step(False, pool2.ids, pool2.pde if D_out else None)
        barrier()
        for _ in range(max(3, args.steps // 2)):
            one_step(True, pool2.ids, pool2.pde if D_out else None, d_ev)
        barrier()
        rep2 = pool2.report()
        d_ms = float(np.mean([a.elapsed_time(b) for a, b in d_ev]))
        drawn = dict(candidates=args.compare_pool, candidates_fill_ms=[round(x, 3) for x in rep2["candidates_ms"]], kept=rep2["kept"],
                     launch_ms=d_ms, kernel=eng.emit_kernel_name())
        pool2.close()
        eng.fill_paths_capped_device(cap_rows, out_ids, out_pde)  # the last fill decides what emit_kernel_name() reports below
        torch.cuda.synchronize()

    # the same steps into ONE plain allocation that takes what comes (what a caller without the pool gets)
    plain = None
    if not multi and args.placements > 1:
        p_ids = torch.empty((max(total, 1), L), dtype=torch.int32, device=device)
        p_pde = None if args.ids_only else torch.empty((max(total, 1), e * L), dtype=torch.float64, device=device)
        p_ev = []
        eng.emit_calibrate_device(p_ids, p_pde, rows_cap=max(total, 1))
        one_step(False, p_ids, p_pde)
        barrier()
        tp = time.perf_counter()
        for _ in range(max(3, args.steps // 2)):
            one_step(True, p_ids, p_pde, p_ev)
        barrier()
        plain = dict(ms_per_step=(time.perf_counter() - tp) * 1e3 / max(3, args.steps // 2),
                     fill_ms=float(np.mean([a.elapsed_time(b) for a, b in p_ev])))
        del p_ids, p_pde
        torch.cuda.empty_cache()

    # one more step, untimed, with a device sync after every phase: where the step time goes (reported, not `value`)
    def phase(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t) * 1e3
    per_step = {}
    if multi:
        _, per_step["vde_and_allgather_ms"] = phase(sb.exchange_vde)
        t_, per_step["count_ms"] = phase(sb.count)
    else:
        _, per_step["vde_ms"] = phase(lambda: eng.vde(want=False))
        t_, per_step["count_ms"] = phase(sb._count_single)
    _, per_step["fill_ms"] = phase(lambda: eng.fill_paths_device(0, t_, out_ids, out_pde, None))

    # sanity of what was just timed (outside the timed region): global path count = sum C(deg, 2) and the
    # middle-vertex checksum sum_paths(b) = sum_v v * C(deg v, 2), both closed forms of the input graph
    deg64 = np.diff(g["offsets"].astype(np.int64))
    want_paths = int((deg64 * (deg64 - 1) // 2).sum())
    want_mid = int((np.arange(args.n, dtype=np.int64) * (deg64 * (deg64 - 1) // 2)).sum())
    chk = torch.stack([ids_view[:total, 1].to(torch.int64).sum(), torch.tensor(total, device=device)]).to(torch.int64)
    if multi:
        chk_h = chk.cpu() if staged else chk
        dist.all_reduce(chk_h, op=dist.ReduceOp.SUM)
        chk = chk_h
    got_mid, got_paths = (int(x) for x in chk.tolist())
    if (got_paths, got_mid) != (want_paths, want_mid) or sb.global_total != want_paths:
        raise SystemExit(f"bench sanity check failed: paths {got_paths} (want {want_paths}), middle checksum {got_mid} "
                         f"(want {want_mid})")
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if staged else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    global_total = sb.global_total
    ms_per_step = dt * 1e3 / args.steps
    value = global_total / (ms_per_step / 1e3)

    fill_avg_ms = float(np.mean([a.elapsed_time(b) for a, b in fill_ms])) if fill_ms else float("nan")
    bpp = (4 * L + 16) if args.ids_only else bytes_per_path(L, e)
    achieved = total * bpp / (fill_avg_ms / 1e3) / 1e9
    # HBM traffic of the fill launch from the committed PMC passes (profiles/, same command and config): WRITE_SIZE +
    # the fabric read requests by size (TCC_EA0_RDREQ_{32,64,128}B); counters cannot be collected inside this run
    traffic, traffic_note = None, None
    kname = eng.emit_kernel_name() if args.fill_variant == 4 else "k_fill_edge_wave"
    if (os.path.exists(PMC_FILE) and world == 1 and args.fill_variant == 4 and not args.ids_only and not args.powerlaw
            and (args.n, args.m, e) == (1_000_000, 10_000_000, 2)):
        d = json.load(open(PMC_FILE)).get(kname, {}).get("derived", {})
        if "traffic_bytes" in d:
            traffic = d["traffic_bytes"] / 1e9
            traffic_note = (f"GB per launch from {os.path.relpath(PMC_FILE, ROOT)} (separate --pmc passes of this command): "
                            f"written {d['write_bytes'] / 1e9:.2f} + read {d['read_bytes'] / 1e9:.2f} (128-byte fabric requests)")
    peak_bytes = total * bpp / 1e9  # GB per launch
    cms = sorted(pool_rep["candidates_ms"])
    med_ms = float(np.median(cms)) if len(cms) > 1 else None
    roofline = dict(bound="hbm", kernel=kname, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source="committed profile (profiles/r06_pmc_fill.json: separate --pmc passes over the same kernel, graph and buffers' size), not this run",
                    traffic_note=traffic_note, bytes_per_path=bpp,
                    paths_per_launch=total, launch_ms=fill_avg_ms,
                    emit_shapes=None if shapes is None else dict(
                        starts_ms=round(shapes["starts_ms"], 3), starts_low_ms=round(shapes["starts_low_ms"], 3),
                        tiles_ms=round(shapes["tiles_ms"], 3), kept=shapes["kept"],
                        note="gnnpe_emit_calibrate_device: three emit launches timed into the output buffer (host clock around stream "
                             "synchronisations, best of two after a first touch), the fastest kept for it: starts = k_fill_ranked, one "
                             "wave per start vertex, one-shot in launch order (round 6; rounds 4-5: a resident grid of five workgroups per "
                             "CU with ticket counters); starts_low = the same kernel as a resident grid of three workgroups per CU, start "
                             "vertices in order from ticket counters; tiles = k_fill_tiles, one wave per 64-row output tile, launch order"),
                    output_pool=dict(candidates_fill_ms=[round(x, 3) for x in pool_rep["candidates_ms"]], kept=pool_rep["kept"],
                                     probe=pool_rep["probe"],
                                     frac_median_candidate=(peak_bytes / (med_ms / 1e3) / HBM_PEAK_GBS) if med_ms else None,
                                     frac_best_candidate=(peak_bytes / (cms[0] / 1e3) / HBM_PEAK_GBS) if med_ms else None,
                                     placements=args.placements,
                                     note="gnnpe_output_pool_create (product API, also behind gnnpe_main and offline.py); with --placements 1 (default "
                                          "since round 4) ONE allocation that takes what comes; with N > 1 independent candidate allocations, the emit "
                                          "kernel timed into each, the fastest kept and the others freed. `frac` is the kept buffer timed live over the steps"),
                    drawn_pool=None if drawn is None else dict(
                        drawn, frac=peak_bytes / (drawn["launch_ms"] / 1e3) / HBM_PEAK_GBS,
                        note="NOT the headline: the same steps into the fastest of `candidates` independent allocations (what rounds 2-3 "
                             "reported, with 5 and 12 candidates), for comparison with `frac`, which is one allocation as it came"),
                    plain_allocation=None if plain is None else dict(
                        launch_ms=plain["fill_ms"], frac=peak_bytes / (plain["fill_ms"] / 1e3) / HBM_PEAK_GBS, ms_per_step=plain["ms_per_step"],
                        value=global_total / (plain["ms_per_step"] / 1e3),
                        note="the same steps into one torch.empty allocation that takes what comes (`--placements 1` for a whole run like this)"),
                    step_frac=(global_total * bpp / (ms_per_step / 1e3) / 1e9) / (HBM_PEAK_GBS * world),
                    step_frac_note="the same algorithmic bytes over the whole step (vde + count + scan + fill), per GPU")

    # which kind of allocation the output buffer is: by what the default shape (start-vertex waves, one-shot since
    # round 6) reaches into it, as a fraction of 8 TB/s -- thresholds from profiles/r05_emit_ab.txt section 7 / r06_emit_oneshot.txt: fast
    # >= 0.80 (2.73-2.83 ms at config 3), between 0.74-0.80 (2.92-3.03), slow below; a slow buffer that takes the resident grid of three
    # workgroups per CU faster is `slow3`, the other kind `slow5`.  Top level so that BENCH records of different boxes compare like with like.
    emit_class, calibration_ms = None, None
    if shapes is not None and shapes["starts_ms"] > 0:  # (nothing is timed below 2^24 paths or on graphs with hub rows)
        f5 = peak_bytes / (shapes["starts_ms"] / 1e3) / HBM_PEAK_GBS
        emit_class = "fast" if f5 >= 0.80 else "between" if f5 >= 0.74 else ("slow3" if 0 < shapes["starts_low_ms"] < shapes["starts_ms"] else "slow5")  # (0: not timed -- at e > 2 shape 4 is shape 1)
        calibration_ms = dict(starts_one_shot=round(shapes["starts_ms"], 3), starts_resident_3_per_cu=round(shapes["starts_low_ms"], 3),
                              tiles=round(shapes["tiles_ms"], 3), kept=shapes["kept"])

    out = dict(metric="offline paths-embedded/sec + index-build wallclock, 1M-V/10M-E l=2",
               emit_class=emit_class, calibration_ms=calibration_ms,
               value_is="paths-embedded/sec of one device-resident pass (vde [+ all-gather] + count + scan + fill)",
               value=value, unit="paths/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=ms_per_step, higher_is_better=True, scaling="strong", vs_baseline=None,
               dtype="u32 ids + f64 embeddings", data="synthetic",
               config=dict(workload=f"{'power-law' if args.powerlaw else 'G'}(n={args.n}, m={args.m}) seed {args.seed}, {args.labels} labels, l=2, e={e}, "
                                    f"degree-sorted order; {'ids only' if args.ids_only else 'ids + pde'}",
                           paths=global_total, parallelism=f"slab{world}", fill_variant=args.fill_variant),
               roofline=roofline, sanity="path count and middle-vertex checksum match the closed forms",
               phases_ms=dict(one_time={k: round(v, 3) for k, v in one_time.items()},
                              per_step={k: round(v, 3) for k, v in per_step.items()},
                              note="one_time = distributing / loading the graph structure (rows, reverse positions, halo rows); "
                                   "per_step = what `value` times, here from one untimed step with a device synchronisation after every "
                                   "phase (the timed steps run without any)"))
    # what the first hardware SCALE record is to be judged against (per-rank kernels of config 4 emulated on one GPU,
    # scripts/emulate_rank.py, + the 16 MB vde all-gather over xGMI + launch gaps); strong scaling, so ms per step falls with N
    out["expected_ms_per_step"] = {"1": [3.55, 4.35], "2": [2.5, 2.9], "4": [1.5, 1.8], "8": [1.0, 1.3],
                                   "source": "PREFLIGHT.md / profiles/r06_emulate_rank.txt: every rank's per-step kernels emulated on one GPU (N = 2: 2.41-2.60 ms, "
                                             "N = 4: 1.40-1.55, N = 8: 0.88-0.94) + the 16 MB vde all-gather + launch gaps; N = 1 = this bench's own spread by "
                                             "allocation class"}
    if multi:
        out["halo"] = dict(sb.stats, owned_entries=owned_entries, slab=[int(bounds[rank]), int(bounds[rank + 1])],
                           local_paths=int(total), note="rank 0's share; halo rows are truncated to the slab's rank range")
        out["config"]["collectives"] = "gloo, staged through host memory (GNNPE_BENCH_SAME_DEVICE=1)" if staged else "rccl"

    legs = not multi and not args.no_index and not args.ids_only
    # index-build wallclock (second half of BASELINE.json's metric), measured outside the timed steps.
    # (a) device image: R*-tree file image of every path (p = 1), bulk-loaded on the device
    if legs:
        try:
            ib, ib_cached = [], []
            for _ in range(3):
                eng.count_paths(2)  # untimed: a new count invalidates the cached pair order, so the timed call below rebuilds it
                ev0 = torch.cuda.Event(enable_timing=True)
                ev1 = torch.cuda.Event(enable_timing=True)
                ev2 = torch.cuda.Event(enable_timing=True)
                ev0.record()
                img, nbytes, hdr = eng.build_index_partition_device(0)
                ev1.record()
                eng.build_index_partition_device(0)  # second partition-build of the same count: pair order reused
                ev2.record()
                torch.cuda.synchronize()
                ib.append(ev0.elapsed_time(ev1))
                ib_cached.append(ev1.elapsed_time(ev2))
            ib_tuple = []
            for _ in range(2):  # the tuple-array build (what a caller without the enumeration state uses: distributed ranks)
                ev0 = torch.cuda.Event(enable_timing=True)
                ev1 = torch.cuda.Event(enable_timing=True)
                ev0.record()
                eng.build_index_device(total, L, out_ids)
                ev1.record()
                torch.cuda.synchronize()
                ib_tuple.append(ev0.elapsed_time(ev1))
            # image AND the tree's auxiliary index (custom.h:268-364; the reference rebuilds it at every online start) in one
            # build: the leaf kernel computes the leaves' rows while it assembles them, the upper levels are a short pass
            fused_ms = []
            for _ in range(3):
                ev0 = torch.cuda.Event(enable_timing=True)
                ev1 = torch.cuda.Event(enable_timing=True)
                ev0.record()
                eng.build_index_partition_aux_device(0)  # pair order cached by the calls above: compare with next_partition_ms
                ev1.record()
                torch.cuda.synchronize()
                fused_ms.append(ev0.elapsed_time(ev1))
            # the generic pass over a finished image (what a foreign tree or an l = 3 image goes through), for comparison
            aux_ms = []
            img, nbytes, hdr = eng.build_index_partition_device(0)
            for _ in range(2):
                ev0 = torch.cuda.Event(enable_timing=True)
                ev1 = torch.cuda.Event(enable_timing=True)
                ev0.record()
                eng.aux_index_device_ptrs(img, nbytes, total, L, out_ids)
                ev1.record()
                torch.cuda.synchronize()
                aux_ms.append(ev0.elapsed_time(ev1))
            out["index_build"] = dict(wallclock_ms=min(ib[1:]), points=total, file_bytes=nbytes, node_blocks=hdr[1], leaves=hdr[4],
                                      image_and_aux_index_ms=min(fused_ms),
                                      aux_index_added_ms=min(fused_ms) - min(ib_cached),
                                      aux_index_note="Partition::build_auxiliary_index (custom.h:268-364): image_and_aux_index_ms builds the image "
                                                     "WITH the auxiliary index (leaf rows by the leaf kernel, inner nodes' rows and all keys by the "
                                                     "inner-node kernel) from the cached pair order -- compare next_partition_ms, the same build without it; "
                                                     "generic_aux_pass_ms is the stand-alone pass over a finished image (foreign trees, l = 3)",
                                      generic_aux_pass_ms=min(aux_ms),
                                      where="device image of index.dat (partition 0 of p = 1), pair-major build from the enumeration "
                                            "state: pair sort + leaves + upper levels; the files on disk are timed under e2e",
                                      next_partition_ms=min(ib_cached),
                                      hbm=dict(file_bytes_frac=nbytes / (min(ib[1:]) / 1e3) / 1e9 / HBM_PEAK_GBS,
                                               leaf_pass_traffic_bytes=LEAF_TRAFFIC_BYTES if (args.n, args.m, e) == (1_000_000, 10_000_000, 2) and not args.powerlaw else None,
                                               leaf_pass_traffic_frac=(LEAF_TRAFFIC_BYTES / (min(ib_cached) / 1e3) / 1e9 / HBM_PEAK_GBS)
                                               if (args.n, args.m, e) == (1_000_000, 10_000_000, 2) and not args.powerlaw else None,
                                               note="file_bytes_frac = the image's bytes over wallclock_ms (pair sort included); "
                                                    "leaf_pass_traffic_* = what the leaf kernel moves per launch at config 3 by the counters of the "
                                                    "committed profile (profiles/r06_leaf_mem_pmc.txt: 7.0 GB of 128-byte fabric reads + 20.7 GB of "
                                                    "64-byte writes), over next_partition_ms (leaf kernel + first-pair pass + inner nodes)"),
                                      next_partition_note="a further partition of the same count reuses the sorted pairs (p > 1: the order is built once)",
                                      tuple_array_build_ms=min(ib_tuple),
                                      first_call_ms=ib[0],
                                      first_call_note="the first call allocates the grow-only buffers (the 22 GB image, pair records) and loads "
                                                      "the kernels; later calls only enqueue kernels.  It follows the output pool's draw in this "
                                                      "process: hipMalloc of 24 GB right after 130 GB of candidates were freed waits for the driver "
                                                      "to reclaim them (38 ms when the pool drew 8, seconds after 12)",
                                      reference="~96 us per RTree::insert on the host (BASELINE.md): hours at this size")
            if (args.n, args.m, e) == (1_000_000, 10_000_000, 2) and not args.powerlaw and not args.no_config5:
                guarded(out["index_build"], "l3", index_l3_leg, torch, stream, local_rank, args.labels, min(ib[1:]) / (nbytes / 1e9))
        except Exception as ex:  # (an optional leg must not take the headline line with it)
            leg_failed(out, "index_build", ex)
    # next row: the online filter over the same paths -- query plan of an 8-vertex query cut out of the
    # data graph, leaf test of Partition::query on every enumerated path; outside the timed steps
    if legs and e == 2:
        guarded(out, "online_filter", online_filter_leg, eng, g, args.seed)
    del ids_view
    pool.close()
    eng.close()
    torch.cuda.empty_cache()
    g2 = None
    if legs and not args.powerlaw and (args.n, args.m) == (1_000_000, 10_000_000):
        g2 = synth.gnm_graph(100_000, 1_000_000, n_labels=args.labels, seed=args.seed)
        guarded(out, "config2", device_pass, torch, stream, local_rank, g2, synth.degree_order(g2["offsets"]), args.labels, e, args.steps)
    if legs and not args.powerlaw and e in (1, 2, 4, 8):
        guarded(out, "gnn_pge", pge_leg, torch, stream, local_rank, g, args.labels, e, args.steps)
    if legs and not args.no_config5 and not args.powerlaw and (args.n, args.m) == (1_000_000, 10_000_000):
        guarded(out, "config5", config5_leg, torch, stream, local_rank, args.labels, args.seed)
    if rank == 0:
        # (b) files on disk + end-to-end wall-clock of the drop-in CLI
        if legs and not args.no_e2e and not args.powerlaw and os.path.exists(CLI):
            big = (args.n, args.m) == (1_000_000, 10_000_000)
            name = "config 3: G(1M, 10M)" if big else f"G({args.n}, {args.m})"
            e2e = {}
            guarded(e2e, "text_p1", e2e_leg, g, sn, 1, False, name + ", p=1, text files only")
            # index files the UNTOUCHED reference online binary can read: every index.dat below 2 GiB (its block file seeks with
            # 32-bit arithmetic, blk_file.h:32-33; gnnpe_main --index refuses larger ones unless --allow-large)
            pc = 16 if big else max(1, int(np.ceil(synth.expected_paths_l2(g["offsets"]) * 108 / (1 << 31))))
            guarded(e2e, f"text_index_p{pc}", e2e_leg, g, sn, pc, True, name + f", p={pc}, text files + {pc} x index.dat, every file < 2 GiB (consumable)")
            if big:
                guarded(e2e, "text_index_p8_allow_large", e2e_leg, g, sn, 8, True, name + ", p=8, text files + 8 x index.dat of 2.7 GB (--allow-large: "
                        "not readable by the reference's online binary; kept for comparison with round 3)", allow_large=True)
            if big and g2 is not None:
                guarded(e2e, "config2_text_index_p2", e2e_leg, g2, synth.degree_order(g2["offsets"]), 2, True,
                        "config 2: G(100K, 1M), p=2, text files + 2 x index.dat (p=1 would be 2.1 GB)")
            e2e["reference"] = "BASELINE.md: config 3 offline 1 749 s, config 2 offline 158.8 s (1 thread); its index build is ~96 us per insert"
            out["e2e"] = e2e
            if "index_build" in out:
                out["index_build"]["files"] = {k: dict(seconds=v.get("index_build_s"), bytes=v.get("index_bytes"))
                                               for k, v in e2e.items() if isinstance(v, dict) and "index_build_s" in v}
        if world == 1 and not args.no_cpu_baseline:
            try:
                sn_, sm_ = (int(x) for x in args.cpu_sample.split(","))
                small = cpu_baseline(sn_, sm_, args.seed)
                out["cpu_baseline"] = small  # (replaced below by the config-2 run when there is time for it)
                guarded(out, "cpu_baseline_all_cores", cpu_baseline_all_cores, 300_000, 3_000_000, e, args.seed)
                from oracle import ref_main_path
                age = time.perf_counter() - T_START
                if os.path.exists(ref_main_path()) and (args.cpu_config2 == "yes" or (args.cpu_config2 == "auto" and age < args.cpu_config2_before)):
                    out["cpu_baseline"] = cpu_baseline(100_000, 1_000_000, args.seed, named="BASELINE config 2")
                    out["cpu_baseline_small_sample"] = small
                else:
                    out["cpu_baseline"]["config2_skipped"] = f"the run was {age:.0f} s old (limit {args.cpu_config2_before:.0f} s) or the reference binary is absent"
            except Exception as ex:  # (the line is still worth printing: the GPU numbers above were measured)
                leg_failed(out, "cpu_baseline", ex)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
