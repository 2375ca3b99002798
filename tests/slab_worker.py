#!/usr/bin/env python3
"""One rank of a slab-partitioned run of the hot path, for the full-size multi-rank GPU tests (TEST INFRASTRUCTURE).

Launched by tests/test_gpu_slabs_full.py through torch.distributed.run (or directly for one rank).  Every rank loads
its slab's rows, installs the halo through dist.SlabBuild (the production exchange code), runs one step and writes
<out>/rank<r>.json: local / global path counts, the id base, an order-sensitive checksum of what it emitted
(gnnpe_rows_checksum_device, first_id = its global id base, so the ranks' checksums ADD to the single-rank one) and
the outcome of the size-independent properties.

Three ways to run R ranks:
  --threads R                      R ranks as threads of THIS process sharing device 0 (dist.ThreadRanks: device copies
                                   around thread barriers).  What the single-GPU box uses: a GPU box admits only a few
                                   processes per card, 8 rank processes are killed by its process guard.
  torchrun + GNNPE_BENCH_SAME_DEVICE=1   rank processes sharing device 0, collectives staged over gloo (<= 4 ranks).
  torchrun                         one GPU per rank over RCCL.
"""
import argparse
import json
import os
import sys
import threading
import traceback

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import binding, synth  # noqa: E402
from gnnpe_amd.dist import SlabBuild, ThreadRanks, owned_rows, plan_slabs  # noqa: E402

M64 = (1 << 64) - 1


def l2_properties(g, sn, dev, ids, pde, vde_ref, lo, hi):
    """Size-independent properties of one rank's emitted rows (all on the device): starts inside the slab, rows
    strictly increasing in (rank[s], b, c) (=> unique, reference order), (s,b) and (b,c) are edges, rank[c] > rank[s],
    pde rows are the vde gather.  Returns a dict of booleans."""
    n = g["n"]
    rank = np.empty(n, np.int64)
    rank[sn] = np.arange(n)
    rank_t = torch.from_numpy(rank).to(dev)
    s, b, c = ids[:, 0].long(), ids[:, 1].long(), ids[:, 2].long()
    out = {}
    out["starts_in_slab"] = bool(((rank_t[s] >= lo) & (rank_t[s] < hi)).all()) if len(s) else True
    out["rank_c_gt_rank_s"] = bool((rank_t[c] > rank_t[s]).all())
    key = (rank_t[s] * n + b) * n + c
    out["strictly_increasing"] = bool((key[1:] > key[:-1]).all())
    del key
    deg = torch.from_numpy(np.diff(g["offsets"].astype(np.int64))).to(dev)
    ekeys = torch.repeat_interleave(torch.arange(n, device=dev), deg) * n + torch.from_numpy(g["nbrs"].astype(np.int64)).to(dev)
    ok = True
    for u, v in ((s, b), (b, c)):
        q = u * n + v
        pos = torch.searchsorted(ekeys, q).clamp_(max=len(ekeys) - 1)
        ok = ok and bool((ekeys[pos] == q).all())
        del q, pos
    out["edges_exist"] = ok
    if pde is not None:
        vt = torch.from_numpy(vde_ref).to(dev)
        e = vt.shape[1]
        ok = True
        CH = 1 << 23
        for k, col in enumerate((s, b, c)):
            for a in range(0, len(col), CH):
                z = slice(a, min(a + CH, len(col)))
                ok = ok and bool((pde[z, e * k:e * (k + 1)] == vt[col[z]]).all())
        out["pde_is_vde_gather"] = ok
    return out


def oracle_l3_check(eng, g, sn, e, n_labels, total):
    """l = 3 at full size against a checker that is not the engine.  Parity stays unpinned -- no reference
    runs l = 3 -- but counts and rows come from the oracle's own code: orc_count_per_start_l3 (pinned to its plain DFS
    on small graphs, tests/test_oracle_deep.py) and orc_enumerate_starts (that DFS, for chosen start vertices)."""
    import time
    from oracle import Oracle
    orc = Oracle()
    n, L = g["n"], 4
    out = {}
    t0 = time.time()
    want = orc.count_per_start_l3(g["offsets"], g["nbrs"], sn)
    out["oracle_count_s"] = round(time.time() - t0, 1)
    got_total, got = eng.count_paths(3, per_start=True)
    out["per_start_equal"] = bool(np.array_equal(got, want)) and int(got_total) == total == int(want.sum(dtype=np.uint64))
    out["p4_closed_form"] = int(orc.count_p4(g["offsets"], g["nbrs"])[1])  # sum_E (du - 1)(dv - 1) - 3 T
    out["starts_with_paths"] = int((want > 0).sum())
    base = np.zeros(n + 1, np.uint64)
    np.cumsum(want, out=base[1:])
    # the embeddings the rows must carry: the oracle's gen_vde with e dimensions (custom.h:513-544)
    x, nx, vde = orc.gen_vde(g["offsets"], g["nbrs"], g["labels"], e)
    deg = np.diff(g["offsets"].astype(np.int64))
    rank = np.empty(n, np.int64)
    rank[sn] = np.arange(n)
    hub = int(np.argmax(deg))
    hub_nb = g["nbrs"][g["offsets"][hub]:g["offsets"][hub + 1]].astype(np.int64)
    pos = rank[hub_nb]
    k = int(np.argmin(np.abs(want[pos].astype(np.int64) - (1 << 21))))  # a start next to the hub with about 2^21 paths
    p_hub = int(pos[k])
    ranges = [("first", 0, min(total, 1 << 21)),
              ("through the hub", int(base[p_hub]), int(base[p_hub + 1])),
              ("last", max(0, total - (1 << 16)), total)]
    out["hub"] = dict(vertex=hub, degree=int(deg[hub]), start_position=p_hub, start_degree=int(deg[sn[p_hub]]))
    out["ranges"] = []
    for name, lo, hi in ranges:
        t0 = time.time()
        i0 = int(np.searchsorted(base, np.uint64(lo), side="right")) - 1
        i1 = int(np.searchsorted(base, np.uint64(hi), side="left"))  # starts i0 .. i1 - 1 cover [lo, hi)
        rows = orc.enumerate_starts(g["offsets"], g["nbrs"], sn, L, i0, want[i0:i1])
        a = lo - int(base[i0])
        rows = rows[a:a + (hi - lo)]
        pde = vde[rows].reshape(len(rows), L * e)  # gen_pde (custom.h:546-572): the vertices' vde rows side by side
        ids_g, pde_g, _ = eng.fill_paths(lo, hi, ids=True, pde=True, L=L)
        out["ranges"].append(dict(name=name, begin=lo, end=hi, starts=i1 - i0, seconds=round(time.time() - t0, 1),
                                  ids_equal=bool(np.array_equal(ids_g, rows)),
                                  pde_equal=bool(np.array_equal(pde_g.view(np.uint64), pde.view(np.uint64))),
                                  beyond_32_bits=bool(lo > (1 << 32))))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", required=True, help=".npz with offsets, nbrs, labels")
    ap.add_argument("--out", required=True)
    ap.add_argument("-l", type=int, default=2)
    ap.add_argument("-e", type=int, default=2)
    ap.add_argument("--labels", type=int, default=64)
    ap.add_argument("--props", type=int, default=1)
    ap.add_argument("--sample", type=int, default=0, help="l=3: rows per rank to emit and checksum (0 = all)")
    ap.add_argument("--ranges", default=None, help="world 1: JSON list of [begin, end) global id ranges to checksum")
    ap.add_argument("--weights", type=str, default="1,0,0", help="w_paths,w_owned,w_held of dist.plan_slabs")
    ap.add_argument("--threads", type=int, default=0, help="run this many ranks as threads of one process on device 0")
    ap.add_argument("--force-rccl", type=int, default=0,
                    help="world 1 only: a 1-rank process group over RCCL (backend nccl) and the N > 1 step -- halo plan, "
                         "all-to-all-v, vde all-gather, totals all-gather, enqueue-only count + capped fill")
    ap.add_argument("--oracle-l3", type=int, default=0,
                    help="world 1, l=3: every start vertex' path count against the oracle's own count (sorted-row merges, all cores), and "
                         "three global id ranges -- the first rows, the rows of a start vertex next to the highest-degree hub, the last rows "
                         "(behind the highest-ranked hub starts) -- against rows the oracle's DFS enumerates for the start vertices that "
                         "cover them: ids and all (l + 1) e doubles bit for bit")
    ap.add_argument("--oracle", type=int, default=0,
                    help="l=2: compare every emitted id and double with the oracle's all-core pass (needs ~15 GB of host memory per 2e8 paths)")
    args = ap.parse_args()

    z = np.load(args.graph)
    g = dict(n=len(z["labels"]), offsets=z["offsets"], nbrs=z["nbrs"], labels=z["labels"])

    args.oracle_rows = None
    if args.oracle and args.l == 2:
        from oracle import Oracle
        sn0 = synth.degree_order(g["offsets"])
        P, ovde, so, oids, opde = Oracle().offline_parallel(g["offsets"], g["nbrs"], g["labels"], sn0, args.e)
        args.oracle_rows = (P, oids, opde)

    if args.threads > 1:
        world = args.threads
        tr = ThreadRanks(world, timeout=1800.0)
        errors = []

        def body(rank):
            try:
                run_rank(args, g, rank, world, 0, "threads", tr.comm(rank))
            except BaseException:  # noqa: BLE001 -- any failure must release the peers' barriers
                errors.append((rank, traceback.format_exc()))
                tr.abort()

        ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errors:
            for r, tb in errors:
                sys.stderr.write(f"rank {r}:\n{tb}\n")
            sys.exit(1)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    same = os.environ.get("GNNPE_BENCH_SAME_DEVICE") == "1"
    if same:
        local_rank = 0
    backend = "none"
    grouped = world > 1 or args.force_rccl
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = "gloo" if same else "nccl"
        torch.cuda.set_device(local_rank)
        if same:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    run_rank(args, g, rank, world, local_rank, backend, None)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def run_rank(args, g, rank, world, local_rank, backend, comm):
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n, L, e = g["n"], args.l + 1, args.e
    sn = synth.degree_order(g["offsets"])
    mem = synth.block_membership(n, max(world, 1))
    bounds = plan_slabs(g["offsets"], sn, world, g["nbrs"], weights=tuple(float(x) for x in args.weights.split(",")))

    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = binding.Engine(local_rank, stream=stream.cuda_stream)
    owned_entries = len(g["nbrs"])
    cap = len(g["nbrs"]) * (2 if args.l == 3 else 1)
    forced = bool(getattr(args, "force_rccl", 0)) and world == 1
    if world == 1 and not forced:
        eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    else:
        rows, roff, rnbr = owned_rows(g, sn, bounds, rank)
        owned_entries = int(roff[-1])
        eng.load_rows(n, g["labels"], rows, roff, rnbr, nbr_capacity=cap + owned_entries)
    eng.set_order(sn, mem, max(world, 1))
    eng.set_slab(int(bounds[rank]), int(bounds[rank + 1]))
    eng.set_label_table(binding.host_label_table(args.labels, e))
    sb = SlabBuild(eng, n, e, bounds, rank, world, dev, nbr_capacity=cap, owned_entries=owned_entries, l=args.l,
                   comm=comm, force_collectives=forced)
    total, base = sb.step()
    if forced:
        # RCCL must really have carried the exchange: the vde table came back through the all-gather, the totals through
        # theirs; then the step again with nothing read back before the fill (enqueue-only count, capped fill)
        assert dist.get_backend() == "nccl" and sb.dist_on
        cap_rows = int(total) + 5
        ids2 = torch.zeros((cap_rows, L), dtype=torch.int32, device=dev)
        pde2 = torch.full((cap_rows, L * e), -1.0, dtype=torch.float64, device=dev)
        sb.step_enqueue(ids2, pde2, cap_rows)
        assert sb.count_end() == base and sb.local_total == total and sb.global_total == total
        ids1 = torch.empty((max(total, 1), L), dtype=torch.int32, device=dev)
        pde1 = torch.empty((max(total, 1), L * e), dtype=torch.float64, device=dev)
        eng.fill_paths_device(0, total, ids1, pde1, None)
        torch.cuda.synchronize()
        assert torch.equal(ids1[:total], ids2[:total]) and torch.equal(pde1[:total], pde2[:total])
        assert bool((ids2[total:] == 0).all()) and bool((pde2[total:] == -1.0).all())  # nothing beyond the total
        # a capacity below the total clips on the device
        small = max(int(total) // 3, 1)
        ids3 = torch.zeros((small + 7, L), dtype=torch.int32, device=dev)
        eng.count_paths_enqueue(args.l)
        eng.fill_paths_capped_device(small, ids3, None)
        torch.cuda.synchronize()
        assert torch.equal(ids3[:small], ids1[:small]) and bool((ids3[small:] == 0).all())
        assert eng.count_total() == total
        del ids1, ids2, ids3, pde1, pde2
    res = dict(rank=rank, world=world, backend=backend, total=int(total), base=int(base), global_total=int(sb.global_total),
               slab=[int(bounds[rank]), int(bounds[rank + 1])], halo=dict(sb.stats), owned_entries=int(owned_entries))

    if args.oracle_l3 and args.l == 3 and world == 1:
        res["oracle_l3"] = oracle_l3_check(eng, g, sn, e, args.labels, int(total))

    chunk = 1 << 24
    if args.ranges is not None:  # single rank: checksum the requested global ranges
        sums = []
        buf = torch.empty((chunk, L), dtype=torch.int32, device=dev)
        ranges_json = args.ranges
        if ranges_json.startswith("@"):  # a file somebody writes while this process is busy with the oracle's count: wait for it
            import time
            t_wait = time.time()
            while not os.path.exists(ranges_json[1:]):
                if time.time() - t_wait > 600:
                    raise SystemExit(f"{ranges_json[1:]} did not appear")
                time.sleep(0.2)
            ranges_json = open(ranges_json[1:]).read()
        for a, b in json.loads(ranges_json):
            acc = 0
            for c0 in range(a, b, chunk):
                c1 = min(c0 + chunk, b)
                eng.fill_paths_device(c0, c1, buf, None, None)
                acc = (acc + eng.rows_checksum_device(c1 - c0, L, buf, first_id=c0)) & M64
            sums.append(acc)
        res["range_checksums"] = sums
    else:
        emit = total if not args.sample else min(total, args.sample)
        res["emitted"] = int(emit)
        if args.l == 2 and args.props:
            ids = torch.empty((max(emit, 1), L), dtype=torch.int32, device=dev)
            pde = torch.empty((max(emit, 1), L * e), dtype=torch.float64, device=dev)
            eng.fill_paths_device(0, emit, ids, pde, None)
            torch.cuda.synchronize()
            res["checksum"] = eng.rows_checksum_device(emit, L, ids, first_id=base)
            ref = binding.Engine(local_rank)  # whole-graph context: the vde table every rank must agree with
            ref.load_csr(g["offsets"], g["nbrs"], g["labels"])
            ref.set_label_table(binding.host_label_table(args.labels, e))
            _, _, vde_ref = ref.vde()
            ref.close()
            res["props"] = l2_properties(g, sn, dev, ids[:emit], pde[:emit], vde_ref, int(bounds[rank]), int(bounds[rank + 1]))
            if args.oracle_rows is not None:  # this rank's rows = rows [base, base + total) of the single-rank output, bit for bit
                P, oids, opde = args.oracle_rows
                ok = int(sb.global_total) == P
                CH = 1 << 24

                # the oracle's rows of this rank on the device once (six comparisons follow: copying the rank's rows back each time cost
                # more than the fills)
                o_ids = torch.from_numpy(np.ascontiguousarray(oids[base:base + emit]).view(np.int32)).to(dev)
                o_pde = torch.from_numpy(np.ascontiguousarray(opde[base:base + emit]).view(np.int64)).to(dev)

                def rows_equal():
                    good = True
                    for a in range(0, emit, CH):
                        b = min(emit, a + CH)
                        good = good and torch.equal(ids[a:b], o_ids[a:b]) and torch.equal(pde[a:b].view(torch.int64), o_pde[a:b])
                    return bool(good)
                ok = ok and rows_equal()
                # every emit shape a caller can get, through the enqueue-only step (count without a read-back, capped fill), then
                # shape 0 after the library's calibration of THESE buffers
                kernels = {}
                for shape in (1, 4, 2, 3, 0):  # (3: the ticket waves of diagnostic builds; the shipped library answers with the tile kernel)
                    eng.set_emit_shape(shape)
                    kept = shape
                    if shape == 0:
                        kept = eng.emit_calibrate_device(ids, pde, rows_cap=max(emit, 1))["kept_shape"]
                    ids.zero_()
                    pde.zero_()
                    torch.cuda.current_stream(dev).synchronize()
                    eng.count_paths_enqueue(args.l)
                    eng.fill_paths_capped_device(max(emit, 1), ids, pde)
                    eng.sync()
                    kernels[str(shape)] = eng.emit_kernel_name()
                    ok = ok and eng.count_total() == total and kernels[str(shape)] == eng.EMIT_SHAPE_KERNELS[kept] and rows_equal()
                eng.set_emit_shape(0)
                del o_ids, o_pde
                res["emit_kernels"] = kernels
                res["oracle_exact"] = bool(ok)
            res["middle_sum"] = int(ids[:emit, 1].to(torch.int64).sum())
        else:
            buf = torch.empty((chunk, L), dtype=torch.int32, device=dev)
            acc = 0
            for c0 in range(0, emit, chunk):
                c1 = min(c0 + chunk, emit)
                eng.fill_paths_device(c0, c1, buf, None, None)
                acc = (acc + eng.rows_checksum_device(c1 - c0, L, buf, first_id=base + c0)) & M64
            res["checksum"] = acc
    os.makedirs(args.out, exist_ok=True)
    with open(os.path.join(args.out, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    if comm is not None:
        comm.barrier()
    eng.close()


if __name__ == "__main__":
    main()
