"""l=3 / wide-embedding stress run (BASELINE config 5 family) on one GPU: vde + count + chunked emit of
4-vertex paths (ids + pde) into a reusable device buffer, with the order-sensitive row checksum as the only
thing kept -- the outputs do not fit anywhere ("count + checksum only").  Not the headline bench.

    python scripts/bench_deep.py --vertices 100000 --edges 1000000 --embedding 2 [--powerlaw] [--max-paths N]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnpe_amd  # noqa: E402,F401
from gnnpe_amd import binding, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vertices", type=int, default=100_000)
    ap.add_argument("--edges", type=int, default=1_000_000)
    ap.add_argument("--embedding", type=int, default=2)
    ap.add_argument("--length", type=int, default=3)
    ap.add_argument("--labels", type=int, default=64)
    ap.add_argument("--powerlaw", action="store_true")
    ap.add_argument("--max-degree", type=int, default=2000)
    ap.add_argument("--chunk", type=int, default=1 << 26)
    ap.add_argument("--max-paths", type=int, default=0, help="emit only the first N paths (0 = all)")
    ap.add_argument("--ids-only", action="store_true")
    args = ap.parse_args()
    l, e, L = args.length, args.embedding, args.length + 1
    if args.powerlaw:
        g = synth.powerlaw_graph(args.vertices, args.edges, exponent=2.1, max_degree=args.max_degree, n_labels=args.labels)
    else:
        g = synth.gnm_graph(args.vertices, args.edges, n_labels=args.labels)
    sn = synth.degree_order(g["offsets"])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = binding.Engine(0, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(args.labels, e))

    def run():
        t0 = time.perf_counter()
        eng.vde(want=False)
        total = eng.count_paths(l)
        eng.sync()
        t1 = time.perf_counter()
        todo = min(total, args.max_paths) if args.max_paths else total
        chunk = max(1, min(args.chunk, todo))
        ids = torch.empty((chunk, L), dtype=torch.int32, device=dev)
        pde = None if args.ids_only else torch.empty((chunk, L * e), dtype=torch.float64, device=dev)
        chk = 0
        for b in range(0, todo, chunk):
            c = min(todo, b + chunk) - b
            eng.fill_paths_device(b, b + c, ids, pde, None)
            chk = (chk + eng.rows_checksum_device(c, L, ids, b)) & ((1 << 64) - 1)
        eng.sync()
        t2 = time.perf_counter()
        return total, todo, chk, t1 - t0, t2 - t1

    run()  # sizes every internal buffer
    total, todo, chk, t_count, t_fill = run()
    bpp = 4 * L + 16 + (0 if args.ids_only else 8 * e * L + 8 * e)  # 4L + 8eL + 16 + 8e
    print(json.dumps(dict(
        workload=f"{'power-law' if args.powerlaw else 'G(n,m)'} n={args.vertices} m={args.edges} max degree "
                 f"{int(np.diff(g['offsets'].astype(np.int64)).max())}, l={l}, e={e}, {'ids only' if args.ids_only else 'ids + pde'}",
        paths=total, emitted=todo, checksum=f"{chk:016x}", vde_count_ms=t_count * 1e3, emit_ms=t_fill * 1e3,
        paths_per_s=todo / t_fill if t_fill else None, bytes_per_path=bpp,
        algorithmic_GBps=todo * bpp / t_fill / 1e9 if t_fill else None)))
    eng.close()


if __name__ == "__main__":
    main()
