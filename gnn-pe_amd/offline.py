#!/usr/bin/env python3
"""Multi-GPU `-m offline`: one process per GPU, the data graph vertex-partitioned into slabs of the
processing order, halo exchange over RCCL (gnn-pe_amd/dist.py), and the SAME files the reference's
`main -m offline` writes (GNN-PE/src/main.cpp:98-119) -- identical for any number of ranks.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        gnn-pe_amd/offline.py -f <dataset dir>/ -d <graph> -p <partitions> [-e 2] [--index]

Output assembly: rank r's paths are one contiguous range of global path ids, so
  all_paths.txt               = "<P>\\n" + rank-order concatenation of the ranks' rendered rows
  partition-i/partition_paths.txt = "<count_i>\\n" + rank-order concatenation of the ranks' id lines
every rank renders its share on its GPU, the byte counts are all-gathered, and each rank writes its
bytes at its offset of the shared file (pwrite).  index.dat of partition i is built by rank i mod N
from the partition's path tuples, gathered in path-id order with one all-to-all-v per partition.

The single-GPU C++ tool `gnnpe_main` writes the same bytes; this driver exists for the partitioned
multi-GPU path named by BASELINE.json's north_star.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import binding  # noqa: E402
from gnnpe_amd.dist import SlabBuild, owned_rows, plan_slabs  # noqa: E402


def _pwrite_all(fd, data, offset):
    view = memoryview(data)
    done = 0
    while done < len(view):
        done += os.pwrite(fd, view[done:done + (256 << 20)], offset + done)


def _gather_sizes(value, world, staged):
    t = torch.zeros(world, dtype=torch.int64)
    mine = torch.tensor([int(value)], dtype=torch.int64)
    if not staged:
        t, mine = t.cuda(), mine.cuda()
    dist.all_gather_into_tensor(t, mine)
    return [int(x) for x in t.tolist()]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-f", "--file", dest="dataset", default="../Test/")
    ap.add_argument("-d", "--data", dest="graph", default="../Test/data_graph.graph")
    ap.add_argument("-p", "--partition", dest="p", type=int, default=5)
    ap.add_argument("-l", "--length", dest="l", type=int, default=2)
    ap.add_argument("-e", "--embedding", dest="e", type=int, default=2)
    ap.add_argument("--index", action="store_true", help="also write partition-i/index.dat")
    ap.add_argument("--chunk", type=int, default=16 << 20, help="paths rendered per device pass")
    ap.add_argument("--timing", action="store_true")
    ap.add_argument("--strict", action="store_true", help="refuse a graph file with a duplicate `e` line (simple graphs only)")
    ap.add_argument("-q", "--query", dest="query", default=None,
                    help="also answer this query graph (filter on every rank's slab, bitmaps OR-ed, refinement on rank 0)")
    args = ap.parse_args(argv)
    if args.l not in (2, 3):
        raise SystemExit("-l: only 2 and 3 are supported (SURVEY D4)")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # debugging aid for single-GPU boxes (see bench.py): every rank on device 0, collectives over gloo
    same_device = os.environ.get("GNNPE_BENCH_SAME_DEVICE") == "1"
    if same_device:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("offline.py needs a GPU: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if same_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    t0 = time.perf_counter()

    # R0 / R1 with the library's own loader; every rank reads the (small) text inputs, keeps only its rows
    # (a file with duplicate `e` lines loads as the reference loads it -- graph.cpp:211-218: the repeats count in `degree` and in
    # gen_vde's neighbour sums, the enumeration sees the de-duplicated rows; --strict refuses it, as rounds 1-5 did)
    g = binding.host_load_graph(args.graph, strict=args.strict)
    stored = None  # the rows with their repeats, when the file has any
    if g.get("simple_offsets") is not None:
        stored = dict(offsets=g["offsets"], nbrs=g["nbrs"])
        g = dict(g, offsets=g["simple_offsets"], nbrs=g["simple_nbrs"])  # what every structure below is built on
    n, L, e, p = g["n"], args.l + 1, args.e, args.p
    if rank == 0:
        print(f"|V|: {g['n']}, |E|: {g['m']}, |Σ|: {g['labels_count']}")
        print(f"Max Degree: {g['max_degree']}, Max Label Frequency: {g['max_label_frequency']}", flush=True)
    sn, mem = binding.host_read_membership(os.path.join(args.dataset, "gnn-pe", "membership.txt"), n, p)
    for i in range(p):
        d = os.path.join(args.dataset, "gnn-pe", "partitions", f"partition-{i}")
        if not os.path.isdir(d):
            raise SystemExit(f"missing directory {d}/ (the prep step creates it)")

    stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(stream)
    eng = binding.Engine(local_rank, stream=stream.cuda_stream)
    bounds = plan_slabs(g["offsets"], sn, world, g["nbrs"])
    if world == 1:
        eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
        owned_entries = len(g["nbrs"])
        if stored is not None:
            eng.set_multigraph_rows(stored["offsets"].astype(np.uint64), stored["nbrs"])
    else:
        rows, roff, rnbr = owned_rows(g, sn, bounds, rank)
        owned_entries = int(roff[-1])
        eng.load_rows(n, g["labels"], rows, roff, rnbr, nbr_capacity=len(g["nbrs"]) + owned_entries)
        if stored is not None:  # this rank's rows again, as stored: what gen_vde sums over
            _, moff, mnbr = owned_rows(stored, sn, bounds, rank)
            eng.set_multigraph_rows(moff, mnbr)
    eng.set_order(sn, mem, p)
    eng.set_slab(int(bounds[rank]), int(bounds[rank + 1]))
    eng.set_label_table(binding.host_label_table(max(g["labels_count"], 1), e))
    sb = SlabBuild(eng, n, e, bounds, rank, world, device, nbr_capacity=len(g["nbrs"]), owned_entries=owned_entries,
                   l=args.l)
    total, base = sb.step()  # halo exchange + vde + count; no fill yet
    P = sb.global_total
    if P > 0xFFFFFFFF:
        raise SystemExit(f"{P} paths exceed the reference's 32-bit path ids")
    t_count = time.perf_counter()

    # ---- optional: the online side over the partitioned graph (main.cpp:121-185 without the files) ----
    if args.query:
        if args.l != 2:
            raise SystemExit("--query needs -l 2")
        plan = binding.host_query_plan(args.query, e)
        if world > 1:
            eng.set_degrees(np.diff((stored or g)["offsets"].astype(np.int64)))  # slab-only engine: degrees of 2-hop vertices (stored rows' lengths)
        bm = sb.filter(plan)
        if rank == 0:
            print(len(plan["vids"]))
            print(f"Answer Number: {binding.host_refine(g, args.query, bm)}", flush=True)

    # ---- render this rank's share, chunk by chunk ----
    chunk = max(1, min(args.chunk, max(total, 1)))
    # the emitted rows go through the library's output pool (one allocation, filled once per chunk: no shape calibration)
    pool = binding.OutputPool(eng, chunk, L, 0, candidates=1, calibrate=False)
    ids = pool.ids_tensor(device)
    part = torch.empty(chunk, dtype=torch.int32, device=device)
    sel = torch.empty(chunk, dtype=torch.int64, device=device)
    text = torch.empty(chunk * (11 * L + 1) + 64, dtype=torch.uint8, device=device)
    all_txt, part_txt, part_cnt = [], [[] for _ in range(p)], [0] * p
    keep_ids = [[] for _ in range(p)] if args.index else None
    for b in range(0, total, chunk):
        c = min(total, b + chunk) - b
        eng.fill_paths_device(b, b + c, ids, None, None)
        eng.path_partitions_device(b, b + c, part)
        nb = eng.text_paths(c, L, ids, text, text.numel())
        eng.sync()
        all_txt.append(text[:nb].cpu().numpy().tobytes())
        for pid in range(p):
            k = eng.select_partition(c, part, pid, base + b, sel)
            part_cnt[pid] += k
            if k:
                nbp = eng.text_ids(k, sel, text, text.numel())
                eng.sync()
                part_txt[pid].append(text[:nbp].cpu().numpy().tobytes())
                if args.index:
                    keep_ids[pid].append(ids[(sel[:k] - (base + b)).long()].clone())
    staged = same_device or world == 1

    def assemble(path, header, pieces):
        """Every rank writes its pieces at its offset; rank 0 writes the header first."""
        mine = sum(len(x) for x in pieces)
        sizes = _gather_sizes(mine, world, staged) if world > 1 else [mine]
        hdr = header.encode()
        if rank == 0:
            with open(path, "wb") as f:
                f.write(hdr)
                f.truncate(len(hdr) + sum(sizes))
        if world > 1:
            dist.barrier()
        fd = os.open(path, os.O_WRONLY)
        off = len(hdr) + sum(sizes[:rank])
        for x in pieces:
            _pwrite_all(fd, x, off)
            off += len(x)
        os.close(fd)

    assemble(os.path.join(args.dataset, "gnn-pe", "all_paths.txt"), f"{P}\n", all_txt)  # main.cpp:110-119
    for pid in range(p):  # main.cpp:98-108
        cnt = sum(_gather_sizes(part_cnt[pid], world, staged)) if world > 1 else part_cnt[pid]
        assemble(os.path.join(args.dataset, "gnn-pe", "partitions", f"partition-{pid}", "partition_paths.txt"),
                 f"{cnt}\n", part_txt[pid])
    t_text = time.perf_counter()

    # ---- index.dat: partition pid is built by rank pid % world from the gathered tuples ----
    if args.index:
        for pid in range(p):
            owner = pid % world
            mine = torch.cat(keep_ids[pid]) if keep_ids[pid] else torch.empty((0, L), dtype=torch.int32, device=device)
            if world > 1:
                counts = _gather_sizes(mine.shape[0], world, staged)
                recv = torch.empty((sum(counts) if rank == owner else 0, L), dtype=torch.int32, device=device)
                in_splits = [0] * world
                in_splits[owner] = mine.shape[0]
                out_splits = counts if rank == owner else [0] * world
                sb._a2a(recv.view(-1), mine.reshape(-1), [x * L for x in out_splits], [x * L for x in in_splits])
                mine = recv
            if rank == owner:
                img, nbytes, hdr = eng.build_index_device(mine.shape[0], L, mine if mine.shape[0] else None)
                data = eng.copy_to_host(img, nbytes)
                with open(os.path.join(args.dataset, "gnn-pe", "partitions", f"partition-{pid}", "index.dat"), "wb") as f:
                    f.write(data.tobytes())
    if world > 1:
        dist.barrier()
    t_end = time.perf_counter()
    if args.timing and rank == 0:
        print(f'{{"paths": {P}, "gpus": {world}, "load_count_s": {t_count - t0:.3f}, "render_write_s": {t_text - t_count:.3f}, '
              f'"index_s": {t_end - t_text:.3f}, "end_to_end_s": {t_end - t0:.3f}}}', file=sys.stderr)
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
