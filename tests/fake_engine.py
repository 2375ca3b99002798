"""Oracle-backed stand-in for binding.Engine (TEST INFRASTRUCTURE): same method names, CPU tensors,
so the exchange logic in gnn-pe_amd/dist.py (SlabBuild) can run under gloo without a GPU.  The
local compute is the CPU oracle restricted to the rows the "rank" holds."""
import numpy as np
import torch


class FakeEngine:
    def __init__(self, oracle, n, labels, rows, row_offsets, row_nbrs, sorted_nodes, e):
        self.o, self.n, self.e = oracle, n, e
        self.labels = np.asarray(labels, np.uint32)
        self.sorted = np.asarray(sorted_nodes, np.uint32)
        self.rank = np.empty(n, np.int64)
        self.rank[self.sorted] = np.arange(n)
        self.owned = {int(v): np.asarray(row_nbrs[int(row_offsets[k]):int(row_offsets[k + 1])], np.uint32)
                      for k, v in enumerate(rows)}
        self.halo = {}
        self.slab = (0, n)
        self._vde = None

    def set_slab(self, b, e):
        self.slab = (b, e)

    def rows_drop_halo(self):
        self.halo = {}

    def _held(self):
        d = dict(self.owned)
        d.update(self.halo)
        return d

    def halo_need(self, bounds, buf, cap):
        held = self._held()
        need = set()
        for nb in held.values():  # every vertex referenced by a row on the "device" (second call = second hop)
            need.update(int(x) for x in nb)
        need = np.array(sorted(v for v in need if v not in held), np.int64)
        owner = np.searchsorted(np.asarray(bounds, np.int64), self.rank[need], side="right") - 1 if len(need) else need
        counts = np.zeros(len(bounds) - 1, np.uint64)
        pos = 0
        for r in range(len(bounds) - 1):
            ids = need[owner == r]
            counts[r] = len(ids)
            buf[pos:pos + len(ids)] = torch.from_numpy(ids.astype(np.int32))
            pos += len(ids)
        assert pos <= cap
        return counts

    def rows_degree(self, n_req, ids, out):
        held = self._held()
        for k in range(n_req):
            out[k] = len(held[int(ids[k])])

    def rows_pack(self, n_req, ids, out, cap):
        held = self._held()
        pos = 0
        for k in range(n_req):
            nb = held[int(ids[k])]
            out[pos:pos + len(nb)] = torch.from_numpy(nb.astype(np.int32))
            pos += len(nb)
        assert pos <= cap

    def rows_append(self, n_rows, ids, deg, nbrs, n_nbrs, min_rank=0):
        pos = 0
        for k in range(n_rows):
            d = int(deg[k])
            row = nbrs[pos:pos + d].numpy().astype(np.uint32)
            self.halo[int(ids[k])] = row[self.rank[row.astype(np.int64)] >= min_rank]  # truncated halo row
            pos += d
        assert pos == n_nbrs

    def _local_csr(self):
        held = self._held()
        deg = np.zeros(self.n, np.int64)
        for v, nb in held.items():
            deg[v] = len(nb)
        offs = np.zeros(self.n + 1, np.uint32)
        np.cumsum(deg, out=offs[1:])
        nbrs = np.zeros(int(offs[-1]), np.uint32)
        for v, nb in held.items():
            nbrs[offs[v]:offs[v] + len(nb)] = nb
        return offs, nbrs

    def vde(self, want=True):
        offs, nbrs = self._local_csr()
        self._x, _, self._vde = self.o.gen_vde(offs, nbrs, self.labels, self.e)

    def vde_pack_slab(self, b, e, buf):
        buf[:e - b] = torch.from_numpy(self._vde[self.sorted[b:e].astype(np.int64)])

    def vde_unpack_slab(self, b, e, buf):
        self._vde[self.sorted[b:e].astype(np.int64)] = buf[:e - b].numpy()

    def count_paths(self, l=2):
        offs, nbrs = self._local_csr()
        paths = self.o.enumerate_closed(offs, nbrs, self.sorted, l + 1)
        r = self.rank[paths[:, 0].astype(np.int64)]
        self._paths = paths[(r >= self.slab[0]) & (r < self.slab[1])]
        return len(self._paths)

    # enqueue-only count + capped fill (the real engine keeps the total on the device between the two)
    def count_paths_enqueue(self, l=2):
        self._total = self.count_paths(l)

    def count_total_device(self, dev_u64):
        dev_u64[0] = self._total

    def count_total(self):
        return self._total

    def fill_paths_capped_device(self, cap, out_ids, out_pde):
        self.fill_paths_device(0, min(cap, len(self._paths)), out_ids, out_pde, None)

    def fill_paths_device(self, b, e, out_ids, out_pde, out_pdl):
        p = self._paths[b:e]
        if out_ids is not None:
            out_ids[:len(p)] = torch.from_numpy(p.astype(np.int32))
        if out_pde is not None:
            out_pde[:len(p)] = torch.from_numpy(self._vde[p.astype(np.int64)].reshape(len(p), p.shape[1] * self.e))

    def filter_candidates(self, plan, eps=1e-6):
        offs, nbrs = self._local_csr()
        # degrees of ALL vertices (set_degrees on the real engine): the fake keeps the full degree table
        full_offs = np.zeros(self.n + 1, np.uint32)
        np.cumsum(self.full_degrees, out=full_offs[1:])
        sets = self.o.filter_candidates(self._paths, full_offs, self.labels, self._vde, plan["vids"], plan["labels"],
                                        plan["degrees"], plan["pde"], plan["n_vertices"], eps)
        bm = np.zeros((plan["n_vertices"], (self.n + 31) // 32), np.uint32)
        for u, ids in enumerate(sets):
            ids = ids.astype(np.int64)
            np.bitwise_or.at(bm[u], ids >> 5, np.uint32(1) << (ids & 31).astype(np.uint32))
        return bm, 0.0

    def set_degrees(self, degrees):
        self.full_degrees = np.asarray(degrees, np.int64)
