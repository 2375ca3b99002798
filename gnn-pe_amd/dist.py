"""One-process-per-GPU driver of the offline path.

The processing order (membership.txt line order, main.cpp:77-85) is cut into R contiguous SLABS,
one per rank.  Because every path is owned by its start vertex and start vertices are emitted in
processing order, rank r's paths are one contiguous range of global path ids: all_paths.txt is
the rank-order concatenation of the slabs' rows and partition_paths.txt the rank-order
concatenation of per-slab id lists, for any R (SURVEY 8(e) "Invariant").

Each rank holds the adjacency rows of its own start vertices.  Two kinds of work:

  one-time (graph structure; `install_halo`, the distributed analogue of loading the graph):
      all-to-all-v of the adjacency lists of the 1-hop middle vertices (requests, degrees, neighbour
      lists: RCCL over xGMI); l=3 repeats it for the rows two hops out.  The rows of the LAST hop are
      installed TRUNCATED: rank r never emits a path that ends on a vertex ranked before its slab
      (kept iff rank[c] > rank[s] >= bounds[r]), so those entries are dropped on arrival -- later slabs
      hold (and rank-sort, every step) only the upper part of every row.  Reverse positions and hub
      flags of the halo rows are built on arrival.  The halo stays resident across steps, exactly like
      the CSR of a single-GPU run (one rule for N = 1 and N > 1).
  per step:
      1. vde            -- local rows, then an all-gather of the vde rows (n x e doubles in total),
      2. count + scan   -- local; an all-gather of one uint64 per rank gives the global path-id base (started right
                           after the count, collected after the fill is enqueued: the fill needs the local total only),
      3. fill           -- local, into caller-provided device buffers.
There is no reduction across ranks anywhere on the offline side; the online filter (`filter`) takes the union
of the ranks' candidate bitmaps (all-gather + OR: RCCL has no bitwise reductions).

torch.distributed supplies the collectives (backend "nccl" = RCCL on GPUs, "gloo" in the CPU
tests); the engine behind `eng` is the C-ABI library (`binding.Engine`).  Tests substitute an
engine with the same methods to exercise this exchange logic on CPU.
"""
import threading

import numpy as np
import torch
import torch.distributed as dist


def _start_weights(offsets, sorted_nodes, nbrs):
    """Estimated paths per start vertex, in processing order (see plan_slabs)."""
    n = len(sorted_nodes)
    offs = offsets.astype(np.int64)
    deg = np.diff(offs)
    order = sorted_nodes.astype(np.int64)
    if nbrs is not None and len(nbrs):
        nd = (deg[nbrs.astype(np.int64)] - 1).astype(np.float64)
        csum = np.concatenate([[0.0], np.cumsum(nd)])
        two_hop = csum[offs[1:]] - csum[offs[:-1]]          # sum of (deg b - 1) over the neighbours of every vertex
    else:
        two_hop = deg.astype(np.float64) * max(float(deg.mean()) - 1.0 if n else 0.0, 1.0)
    # a neighbour-of-neighbour is an edge endpoint, i.e. drawn proportionally to degree: the chance that it
    # ranks after s is the share of edge endpoints held by later-ranked vertices, not the share of vertices
    dsorted = deg[order].astype(np.float64)
    later = (dsorted.sum() - np.cumsum(dsorted)) / max(dsorted.sum(), 1.0)
    return two_hop[order] * later + 1e-3, dsorted  # vertices without work still need a home


# Step cost of a slab in units of one emitted path, from per-rank kernel times of config 4 on one MI355X
# (scripts/emulate_rank.py, N = 8): a first fit gave step_ms ~ 0.11 + 0.0191 P + 0.0738 O + 0.0247 H with P = paths (M),
# O = adjacency entries of the slab's own rows (M: vde, pair records, per-pair work of the emit kernel), H = entries held
# and rank-sorted per step (M: own + truncated halo rows), i.e. weights 1 : 3.9 : 1.3; trying the neighbourhood on the
# box, 1 : 6 : 1.3 gave the flattest ranks (0.99 - 1.06 ms against 0.96 - 1.09).  Round 5 (the emit kernel is 10-15 % faster per path,
# the start records come from one pass): 1 : 6 : 1.3 now gives 0.80 - 0.957 ms, 1 : 3 : 0.7 the flattest ranks, 0.878 - 0.936
# (profiles/r05_emulate_rank8.txt).
STEP_COST_WEIGHTS = (1.0, 3.0, 0.7)


def plan_slabs(offsets, sorted_nodes, n_ranks, nbrs=None, weights=(1.0, 0.0, 0.0), entry_cost=None):
    """Cut the processing order into n_ranks contiguous slabs of roughly equal step time.

    Host-side prep (the analogue of the reference's partitioning script).  The paths of start s number
    count(s) = sum_{b in N(s)} |{c in N(b): rank[c] > rank[s]}|, estimated as
    (sum_{b in N(s)} (deg b - 1)) * (share of edge endpoints ranked after s): one pass over the adjacency (`nbrs`);
    without the adjacency the neighbour degrees are replaced by the mean degree.

    weights = (w_paths, w_owned, w_held): cost of a slab [lo, hi) =
        w_paths * paths + w_owned * (entries of its own rows) + w_held * (entries it holds),
    the held entries estimated as own + (all other rows, truncated: the share of edge endpoints ranked >= lo).
    (1, 0, 0) = equal path counts; STEP_COST_WEIGHTS = the fitted per-step kernel cost.  The last slabs own the
    high-degree vertices (many entries, few paths each), so they get fewer paths.  Returns uint32 bounds[n_ranks+1]."""
    n = len(sorted_nodes)
    if n == 0 or n_ranks <= 1:
        return np.array([0] + [n] * max(n_ranks, 1), np.uint32)
    if entry_cost is not None:  # round-2 draft interface: one weight on the held entries
        weights = (1.0, 0.0, float(entry_cost))
    w_p, w_o, w_h = (float(x) for x in weights)
    w, dsorted = _start_weights(offsets, sorted_nodes, nbrs)
    cw = np.concatenate([[0.0], np.cumsum(w)])          # paths before position i
    dc = np.concatenate([[0.0], np.cumsum(dsorted)])    # adjacency entries owned before position i
    tot_e = dc[-1]

    if w_o == 0.0 and w_h == 0.0:
        targets = cw[-1] * np.arange(1, n_ranks) / n_ranks
        cuts = np.searchsorted(cw[1:], targets).astype(np.int64)
        return np.maximum.accumulate(np.concatenate([[0], cuts, [n]]).astype(np.uint32))

    def cost(lo, hi):
        own = dc[hi] - dc[lo]
        share = (tot_e - dc[lo]) / max(tot_e, 1.0)       # edge endpoints ranked >= lo
        return w_p * (cw[hi] - cw[lo]) + w_o * own + w_h * (own + (tot_e - own) * share)

    def cuts_for(T):
        b = [0]
        for _ in range(n_ranks - 1):
            lo = b[-1]
            a, z = lo, n                                  # largest hi with cost(lo, hi) <= T
            while a < z:
                mid = (a + z + 1) >> 1
                if cost(lo, mid) <= T:
                    a = mid
                else:
                    z = mid - 1
            b.append(a)
        return b

    lo_t, hi_t = 0.0, cost(0, n)
    for _ in range(50):  # smallest budget T whose first n_ranks-1 slabs leave a last slab within T
        T = 0.5 * (lo_t + hi_t)
        b = cuts_for(T)
        if cost(b[-1], n) > T:
            lo_t = T
        else:
            hi_t = T
    return np.maximum.accumulate(np.array(cuts_for(hi_t) + [n], np.uint32))


def owned_rows(g, sorted_nodes, bounds, r):
    """Adjacency rows (in slab order) of rank r's start vertices: rows, row_offsets u64, row_nbrs."""
    rows = np.ascontiguousarray(sorted_nodes[int(bounds[r]):int(bounds[r + 1])], np.uint32)
    offs = g["offsets"].astype(np.int64)
    deg = offs[rows.astype(np.int64) + 1] - offs[rows.astype(np.int64)]
    roff = np.zeros(len(rows) + 1, np.uint64)
    np.cumsum(deg, out=roff[1:])
    tot = int(roff[-1])
    # gather the ragged rows without a Python loop
    idx = np.repeat(offs[rows.astype(np.int64)] - roff[:-1].astype(np.int64), deg) + np.arange(tot, dtype=np.int64)
    rnbr = np.ascontiguousarray(g["nbrs"][idx], np.uint32) if tot else np.zeros(0, np.uint32)
    return rows, roff, rnbr


class ThreadRanks:
    """In-process transport for R logical ranks that share ONE device: every rank is a thread of the same process with
    its own engine context and HIP stream, and a collective is device-to-device copies between the ranks' tensors
    around two thread barriers (the Python analogue of `gnnpe_main --gpus R --same-device --transport copy`).  It
    exists for single-GPU boxes, where RCCL refuses duplicate devices and a GPU box admits only a few processes per
    card: tests and debugging runs drive SlabBuild's exchange code with 8 ranks in one process.  Not a product path --
    with one GPU per rank the collectives are RCCL (`torch.distributed`, backend "nccl")."""

    def __init__(self, world, timeout=600.0):
        self.world, self.timeout = int(world), float(timeout)
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world

    def comm(self, rank):
        return _ThreadComm(self, int(rank))

    def abort(self):
        """Called by a rank that failed: the peers' barriers raise instead of waiting for it forever."""
        self._barrier.abort()


class _ThreadComm:
    backend = "threads"

    def __init__(self, ranks, rank):
        self.g, self.rank, self.world = ranks, rank, ranks.world

    def barrier(self):
        self.g._barrier.wait(self.g.timeout)

    @staticmethod
    def _sync(t):
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()

    def _publish(self, inp, splits):
        self._sync(inp)  # the engine wrote `inp` on this thread's stream; peers read it on theirs
        self.g._slots[self.rank] = (inp, np.concatenate([[0], np.cumsum(splits)]).astype(np.int64))
        self.barrier()

    def _retire(self, out):
        self._sync(out)
        self.barrier()  # nobody reuses its send buffer before every peer has read it
        self.g._slots[self.rank] = None

    def all_to_all_single(self, out, inp, out_splits, in_splits):
        self._publish(inp, in_splits)
        o = 0
        for p in range(self.world):
            src, offs = self.g._slots[p]
            k = int(offs[self.rank + 1] - offs[self.rank])
            assert k == int(out_splits[p]), (self.rank, p, k, out_splits[p])
            if k:
                out[o:o + k].copy_(src[int(offs[self.rank]):int(offs[self.rank]) + k])
            o += k
        self._retire(out)

    def all_gather_into_tensor(self, out, inp):
        self._publish(inp, [inp.numel()])
        k = inp.numel()
        for p in range(self.world):
            out[p * k:(p + 1) * k].copy_(self.g._slots[p][0].reshape(-1))
        self._retire(out)


class SlabBuild:
    """Per-rank state of the distributed offline build.  `eng` already holds this rank's rows
    (load_rows), the replicated order (set_order), slab (set_slab) and label table."""

    def __init__(self, eng, n, e, bounds, rank, world, device, nbr_capacity, owned_entries=None, group=None, l=2,
                 comm=None, force_collectives=False):
        """nbr_capacity: most neighbour entries this rank can RECEIVE (<= 2m); owned_entries: size
        of its own rows -- every peer may ask for all of them, so the send buffer holds
        (world-1) x owned_entries.  l: edges per path (2, or 3 = one more halo hop).  comm: a ThreadRanks.comm(rank)
        when the ranks are threads of one process sharing a device; None = torch.distributed.
        force_collectives: run the N > 1 step (halo plan, all-to-all-v, vde all-gather, all-gather of the totals) even
        with world == 1 -- a 1-rank process group over RCCL executes the multi-GPU code on a single-GPU box."""
        self.eng, self.n, self.e, self.l = eng, int(n), int(e), int(l)
        self.comm = comm
        self.dist_on = world > 1 or bool(force_collectives)
        self.bounds = np.ascontiguousarray(bounds, np.uint32)
        self.rank, self.world, self.device, self.group = rank, world, device, group
        i32 = dict(dtype=torch.int32, device=device)
        # rows I may need: < n.  rows peers may request from me: every peer can ask for all of mine.
        mine = int(self.bounds[rank + 1]) - int(self.bounds[rank])
        self.need = torch.zeros(max(self.n, 1), **i32)
        self.deg_in = torch.zeros(max(self.n, 1), **i32)
        self.req = torch.zeros(max(mine * max(world - 1, 1), 1), **i32)
        self.deg_out = torch.zeros(max(mine * max(world - 1, 1), 1), **i32)
        self.cap = int(max(nbr_capacity, 1))
        own = int(nbr_capacity if owned_entries is None else owned_entries)
        self.send_cap = int(max(own * max(world - 1, 1), 1))
        self.pack = torch.zeros(self.send_cap, **i32)
        self.nbr_in = torch.zeros(self.cap, **i32)
        lens = (self.bounds[1:].astype(np.int64) - self.bounds[:-1].astype(np.int64))
        self.maxlen = int(lens.max()) if len(lens) else 0
        self.vde_send = torch.zeros((max(self.maxlen, 1), self.e), dtype=torch.float64, device=device)
        self.vde_all = torch.zeros((world, max(self.maxlen, 1), self.e), dtype=torch.float64, device=device)
        self.tot_all = torch.zeros(world, dtype=torch.int64, device=device)
        self.stats = {}
        self._halo_plans = []  # per hop: what moved (ids, degrees, split sizes)
        self.halo_installed = False
        # the engine enqueues on its stream and the collectives on torch's current stream: they must be the same one
        if getattr(device, "type", "cpu") == "cuda" and hasattr(eng, "stream_handle"):
            cur = torch.cuda.current_stream(device).cuda_stream
            if eng.stream_handle != cur:
                raise RuntimeError("SlabBuild: the engine's stream must be torch's current stream on this device "
                                   "(Engine(..., stream=torch.cuda.current_stream().cuda_stream) inside a "
                                   "torch.cuda.stream(...) block), otherwise collectives race the engine's kernels")

    # Collectives.  With backend "nccl" (RCCL) device tensors go straight to the collective.  A gloo
    # group with device tensors (single-GPU debugging of the N>1 flow: several ranks sharing one
    # device, where RCCL refuses duplicate GPUs) stages through host memory.
    def _staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _a2a(self, out, inp, out_splits, in_splits):
        if self.comm is not None:
            return self.comm.all_to_all_single(out, inp, out_splits, in_splits)
        if self._staged(out):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                                   group=self.group)
            out.copy_(o)
            return
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)

    def _allgather(self, out_flat, inp_flat):
        if self.comm is not None:
            return self.comm.all_gather_into_tensor(out_flat, inp_flat)
        if self._staged(out_flat):
            o = torch.empty(out_flat.shape, dtype=out_flat.dtype)
            dist.all_gather_into_tensor(o, inp_flat.cpu(), group=self.group)
            out_flat.copy_(o)
            return
        dist.all_gather_into_tensor(out_flat, inp_flat, group=self.group)

    def install_halo(self, truncate=True):
        """Adjacency rows of the halo: graph structure, fetched ONCE and kept resident (drop + fetch again when
        called a second time, e.g. after changing the graph, the order or the slabs).  One hop for l=2, two for
        l=3.  Per hop: three small all-to-all-v (counts, ids, degrees), then pack at the owners -> one all-to-all-v
        of the neighbour lists -> append at the requester.  The last hop's rows are truncated to the entries ranked
        >= this slab's first position (see the module docstring) unless truncate=False."""
        self.eng.rows_drop_halo()
        self._halo_plans = []
        self.stats.update(halo_rows=0, halo_entries=0, served_rows=0, served_entries=0)
        hops = self.l - 1
        for hop in range(hops):  # l=2: rows of the middle vertices; l=3: also the rows they reference
            p = self._plan_hop()
            self._halo_plans.append(p)
            last = hop == hops - 1
            self._move_rows(p, int(self.bounds[self.rank]) if (truncate and last) else 0)
        if hasattr(self.eng, "rows_held"):
            self.stats["held_rows"], self.stats["held_entries"], self.stats["hub_rows"] = self.eng.rows_held()
        self.halo_installed = True

    # round-1 name, kept for callers that re-fetch explicitly
    exchange_halo = install_halo

    def _plan_hop(self):
        """One hop of the halo, metadata only: three small all-to-all-v (counts, ids, degrees)."""
        eng, R = self.eng, self.world
        need_counts = [int(x) for x in eng.halo_need(self.bounds, self.need, self.n)]
        n_need = sum(need_counts)
        # 1. how many rows does every peer want from me
        sc = torch.tensor(need_counts, dtype=torch.int64, device=self.device)
        rc = torch.zeros(R, dtype=torch.int64, device=self.device)
        self._a2a(rc, sc, [1] * R, [1] * R)
        req_counts = [int(x) for x in rc.tolist()]
        n_req = sum(req_counts)
        assert n_need <= self.need.numel() and n_req <= self.req.numel(), (n_need, n_req)
        # 2. the requested vertex ids
        req = self.req[:n_req]
        self._a2a(req, self.need[:n_need], req_counts, need_counts)
        # 3. their degrees, back to the requester
        deg_out = self.deg_out[:n_req]
        eng.rows_degree(n_req, req, deg_out)
        deg_in = self.deg_in[:n_need]
        self._a2a(deg_in, deg_out, need_counts, req_counts)
        send_sizes = self._segment_sums(deg_out, req_counts)
        recv_sizes = self._segment_sums(deg_in, need_counts)
        n_send, n_recv = sum(send_sizes), sum(recv_sizes)
        if n_send > self.send_cap or n_recv > self.cap:
            raise RuntimeError(f"halo buffers too small: send {n_send}/{self.send_cap}, recv {n_recv}/{self.cap}")
        return dict(need=self.need[:n_need].clone(), req=req.clone(), deg_in=deg_in.clone(), n_need=n_need, n_req=n_req,
                    send_sizes=send_sizes, recv_sizes=recv_sizes, n_send=n_send, n_recv=n_recv)

    def _move_rows(self, p, min_rank=0):
        """4. the adjacency lists themselves: pack at the owner, one all-to-all-v, append at the requester."""
        eng = self.eng
        eng.rows_pack(p["n_req"], p["req"], self.pack, self.send_cap)
        self._a2a(self.nbr_in[:p["n_recv"]], self.pack[:p["n_send"]], p["recv_sizes"], p["send_sizes"])
        eng.rows_append(p["n_need"], p["need"], p["deg_in"], self.nbr_in[:p["n_recv"]], p["n_recv"], min_rank)
        for k, v in (("halo_rows", p["n_need"]), ("halo_entries", p["n_recv"]), ("served_rows", p["n_req"]),
                     ("served_entries", p["n_send"])):
            self.stats[k] += v

    @staticmethod
    def _segment_sums(t, counts):
        if t.numel() == 0:
            return [0] * len(counts)
        cs = torch.cumsum(t.to(torch.int64), 0)
        ends = np.cumsum(counts)
        out, prev = [], 0
        vals = cs[torch.as_tensor(np.maximum(ends - 1, 0), device=t.device)].tolist()
        for k, c in enumerate(counts):
            cur = vals[k] if c > 0 else prev
            out.append(int(cur - prev))
            prev = cur
        return out

    def exchange_vde(self):
        """Per step: vde of the owned rows, then every rank's slab of the table (all-gather of n x e doubles)."""
        eng, R = self.eng, self.world
        b = self.bounds
        eng.vde(want=False)
        eng.vde_pack_slab(int(b[self.rank]), int(b[self.rank + 1]), self.vde_send)
        self._allgather(self.vde_all.view(-1), self.vde_send.view(-1))
        if hasattr(eng, "vde_unpack_all"):  # one launch for all the peers' slabs
            eng.vde_unpack_all(b, max(self.maxlen, 1), self.rank, self.vde_all)
            return
        for r in range(R):
            if r != self.rank and b[r + 1] > b[r]:
                eng.vde_unpack_slab(int(b[r]), int(b[r + 1]), self.vde_all[r])

    def count_enqueue(self):
        """The count and the all-gather of the ranks' totals ENQUEUED, nothing read back: the local total goes from the
        engine's device word straight into the collective's send buffer.  For steps whose outputs were sized by an
        earlier pass (fill_paths_capped_device clips on the device); count_end() collects the numbers afterwards."""
        if not hasattr(self, "_tot_dev"):
            self._tot_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.eng.count_paths_enqueue(self.l)
        self.eng.count_total_device(self._tot_dev)
        self._tot_mine = self._tot_dev
        self._tot_work = None
        if self.comm is None and not self._staged(self.tot_all):
            self._tot_work = dist.all_gather_into_tensor(self.tot_all, self._tot_mine, group=self.group, async_op=True)
        else:
            self._allgather(self.tot_all, self._tot_mine)

    def step_enqueue(self, out_ids, out_pde, cap_rows):
        """One rank-step with no host round trip before the fill: vde [+ all-gather] -> count -> totals all-gather
        (async) -> capped fill.  Returns nothing; count_end() afterwards yields base / totals (one synchronisation, after
        everything is enqueued)."""
        if self.dist_on:
            if not self.halo_installed:
                self.install_halo()
            self.exchange_vde()
        else:
            self.eng.vde(want=False)
        self.count_enqueue()
        self.eng.fill_paths_capped_device(cap_rows, out_ids, out_pde)

    def count_begin(self):
        """Local count, and the all-gather of the ranks' totals STARTED: the fill only needs the local total, so over
        RCCL the collective (and its host round trip) runs beside the fill kernel; count_end() collects it."""
        total = self.eng.count_paths(self.l)
        self.local_total = total
        self._tot_mine = torch.tensor([total], dtype=torch.int64, device=self.device)
        self._tot_work = None
        if self.comm is None and not self._staged(self.tot_all):
            self._tot_work = dist.all_gather_into_tensor(self.tot_all, self._tot_mine, group=self.group, async_op=True)
        else:
            self._allgather(self.tot_all, self._tot_mine)
        return total

    def count_end(self):
        """Global path-id base of this rank's rows and the global total, from the all-gather count_begin() started."""
        if self._tot_work is not None:
            self._tot_work.wait()
            self._tot_work = None
        tots = [int(x) for x in self.tot_all.tolist()]
        self.base = sum(tots[:self.rank])
        self.global_total = sum(tots)
        self.local_total = tots[self.rank]
        return self.base

    def count(self):
        total = self.count_begin()
        self.count_end()
        return total

    def step(self, out_ids=None, out_pde=None, out_pde_label=None):
        """One full pass of the hot path for this rank's slab; returns (local paths, global id base).  The halo rows
        are fetched by the first call (or an explicit install_halo) and stay resident."""
        if self.dist_on:
            if not self.halo_installed:
                self.install_halo()
            self.exchange_vde()
        else:
            self.eng.vde(want=False)
        total = self.count_begin() if self.dist_on else self._count_single()
        if out_ids is not None or out_pde is not None or out_pde_label is not None:
            self.eng.fill_paths_device(0, total, out_ids, out_pde, out_pde_label)
        if self.dist_on:
            self.count_end()
        return total, self.base

    def filter(self, plan, eps=1e-6):
        """Online filter over the whole partitioned graph: every rank tests its own slab's paths
        (call after step(): halo, vde and counts in place; slab-only engines need eng.set_degrees), then the
        candidate bitmaps are OR-ed across ranks -- the one reduction of the online side, the analogue of the union
        over partitions in main.cpp:165-171.  Returns the global bitmap [n_query_vertices x ceil(n/32)] uint32."""
        bm, _ = self.eng.filter_candidates(plan, eps)
        if self.dist_on:
            # union of the ranks' bitmaps: all-gather + OR (RCCL/NCCL reject BOR/BAND/BXOR, so no all-reduce)
            t = torch.from_numpy(np.ascontiguousarray(bm).view(np.int32)).reshape(-1)
            if self.comm is None and dist.get_backend(self.group) != "gloo":
                t = t.to(self.device)
            allb = torch.empty((self.world, t.numel()), dtype=t.dtype, device=t.device)
            if self.comm is not None:
                self.comm.all_gather_into_tensor(allb.view(-1), t)
            else:
                dist.all_gather_into_tensor(allb.view(-1), t, group=self.group)
            u = allb[0]
            for r in range(1, self.world):
                u = torch.bitwise_or(u, allb[r])
            bm = u.cpu().numpy().view(np.uint32).reshape(bm.shape)
        return bm

    def _count_single(self):
        self.local_total = self.global_total = self.eng.count_paths(self.l)
        self.base = 0
        return self.local_total
