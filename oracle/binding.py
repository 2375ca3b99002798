"""ctypes binding of liboracle.so (gnnpe_oracle.c) -- TEST INFRASTRUCTURE ONLY.

numpy in, numpy out.  Every method names the reference lines its C function restates
(see gnnpe_oracle.h).  The product never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)


def build_oracle(force=False):
    """Compile liboracle.so (and, where /root/reference exists, oracle/_ref)."""
    src = os.path.join(_HERE, "gnnpe_oracle.c")
    stale = (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/GNN-PE/src/main.cpp") and not os.path.exists(ref_main_path()):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return _LIB


def ref_main_path():
    """The unmodified reference `main`, compiled by oracle/Makefile (may be absent)."""
    return os.path.join(_HERE, "_ref", "ref_main")


def ref_main_pge_path():
    """The unmodified GNN-PGE `main` (next row), compiled by oracle/Makefile (may be absent)."""
    return os.path.join(_HERE, "_ref", "ref_main_pge")


def ref_dump_path():
    return os.path.join(_HERE, "_ref", "ref_dump")


def ref_online_path():
    """oracle/ref_online.cpp harness: the reference's online filter (dump) / refinement (refine)."""
    return os.path.join(_HERE, "_ref", "ref_online")


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def bitmap_to_sets(bm, n):
    """rows of a candidate bitmap (uint32 words, bit v%32 of word v//32) -> sorted id arrays"""
    bits = np.unpackbits(np.ascontiguousarray(bm).view(np.uint8), axis=1, bitorder="little")[:, :n]
    return [np.flatnonzero(r).astype(np.uint32) for r in bits]


class Oracle:
    def __init__(self):
        build_oracle()
        L = C.CDLL(_LIB)
        L.orc_load_graph.restype = C.c_int
        L.orc_load_graph.argtypes = [C.c_char_p, _u32p, _u32p, C.POINTER(_u32p), C.POINTER(_u32p),
                                     C.POINTER(_u32p), _u32p, _u32p, _u32p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_read_membership.restype = C.c_int
        L.orc_read_membership.argtypes = [C.c_char_p, C.c_uint32, _u32p, _u32p]
        for name in ("orc_enumerate_dfs_hash", "orc_enumerate_closed"):
            fn = getattr(L, name)
            fn.restype = C.c_uint64
            fn.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, C.c_uint32, _u32p, C.c_uint64]
        L.orc_count_per_start.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, C.c_uint32, _u64p]
        L.orc_count_per_start_l3.restype = C.c_int
        L.orc_count_per_start_l3.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, _u64p]
        L.orc_enumerate_starts.restype = C.c_uint64
        L.orc_enumerate_starts.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, _u64p, _u32p]
        L.orc_gen_vde_x.argtypes = [C.c_uint32, C.c_uint32, _f64p]
        L.orc_gen_vde.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, C.c_uint32, _f64p, _f64p, _f64p]
        L.orc_gen_pde.argtypes = [C.c_uint64, C.c_uint32, _u32p, C.c_uint32, _u32p, _u32p, _f64p, _f64p,
                                  _f64p, _f64p, _u32p, _u32p]
        L.orc_write_all_paths.restype = C.c_int
        L.orc_write_all_paths.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, _u32p]
        L.orc_write_partition_paths.restype = C.c_int
        L.orc_write_partition_paths.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, _u32p, _u32p, C.c_uint32]
        L.orc_format_all_paths.restype = C.c_uint64
        L.orc_format_all_paths.argtypes = [C.c_uint64, C.c_uint32, _u32p, C.c_char_p]
        L.orc_index_validate.restype = C.c_int
        L.orc_index_validate.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                         _f64p, C.c_uint64, C.POINTER(C.c_int32)]
        L.orc_pge_groups.argtypes = [C.c_uint32, _u32p, _u32p, C.c_uint32, _f64p, _f64p, _f64p, _f64p]
        L.orc_pge_write_bin.restype = C.c_int
        L.orc_pge_write_bin.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, _u32p, _u32p, _f64p, _f64p, _f64p, _f64p,
                                        _f64p, C.c_double]
        L.orc_offline_parallel.restype = C.c_uint64
        L.orc_offline_parallel.argtypes = [C.c_uint32, _u32p, _u32p, _u32p, _u32p, C.c_uint32, C.c_int, _f64p, _u64p,
                                           _u32p, _f64p, C.c_uint64]
        L.orc_max_threads.restype = C.c_int
        L.orc_count_p4.restype = C.c_int
        L.orc_count_p4.argtypes = [C.c_uint32, _u32p, _u32p, _u64p, _u64p]
        L.orc_filter_candidates.argtypes = [C.c_uint64, C.c_uint32, _u32p, C.c_uint32, _u32p, _u32p, _f64p, C.c_uint32,
                                            C.c_uint32, _u32p, _u32p, _u32p, _f64p, C.c_double, _u32p]
        self.L = L

    # online filter (leaf test of Partition::query, custom.h:404-431)
    def filter_candidates(self, paths, offs, labels, vde, q_vids, q_labels, q_degrees, q_pde, n_qv, eps=1e-6):
        """Returns a list of sorted uint32 arrays, one per query vertex."""
        paths = np.ascontiguousarray(paths, np.uint32)
        P, Lp = paths.shape
        n = len(offs) - 1
        e = vde.shape[1]
        words = (n + 31) // 32
        bm = np.zeros((n_qv, words), np.uint32)
        qv = np.ascontiguousarray(q_vids, np.uint32)
        self.L.orc_filter_candidates(P, Lp, _p(paths, _u32p), n, _p(np.ascontiguousarray(offs, np.uint32), _u32p),
                                     _p(np.ascontiguousarray(labels, np.uint32), _u32p),
                                     _p(np.ascontiguousarray(vde, np.float64), _f64p), e, len(qv),
                                     _p(qv, _u32p), _p(np.ascontiguousarray(q_labels, np.uint32), _u32p),
                                     _p(np.ascontiguousarray(q_degrees, np.uint32), _u32p),
                                     _p(np.ascontiguousarray(q_pde, np.float64), _f64p), eps, _p(bm, _u32p))
        return bitmap_to_sets(bm, n)

    # all-core CPU port of the device-resident pass (bench.py's second CPU baseline)
    def count_p4(self, offs, nbrs):
        """(triangles, number of 4-vertex simple paths) in closed form -- the independent count for l = 3."""
        offs = np.ascontiguousarray(offs, np.uint32)
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        t, p4 = C.c_uint64(), C.c_uint64()
        if self.L.orc_count_p4(len(offs) - 1, _p(offs, _u32p), _p(nbrs, _u32p), C.byref(t), C.byref(p4)) != 0:
            raise MemoryError("orc_count_p4")
        return int(t.value), int(p4.value)

    def max_threads(self):
        return int(self.L.orc_max_threads())

    def offline_parallel(self, offs, nbrs, labels, sorted_nodes, e, threads=0, ids=None, pde=None, want=True):
        """Returns (P, vde, start_off, ids, pde); pass preallocated ids/pde to time the fill without allocation."""
        n = len(offs) - 1
        offs = np.ascontiguousarray(offs, np.uint32)
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        labels = np.ascontiguousarray(labels, np.uint32)
        sn = np.ascontiguousarray(sorted_nodes, np.uint32)
        vde = np.zeros((n, e))
        so = np.zeros(n + 1, np.uint64)
        cap = 0 if ids is None else len(ids)
        P = self.L.orc_offline_parallel(n, _p(offs, _u32p), _p(nbrs, _u32p), _p(labels, _u32p), _p(sn, _u32p), e, threads,
                                        _p(vde, _f64p), _p(so, _u64p), _p(ids, _u32p) if ids is not None else None,
                                        _p(pde, _f64p) if pde is not None else None, cap)
        if ids is None and want:
            ids = np.zeros((P, 3), np.uint32)
            pde = np.zeros((P, 3 * e))
            self.L.orc_offline_parallel(n, _p(offs, _u32p), _p(nbrs, _u32p), _p(labels, _u32p), _p(sn, _u32p), e, threads,
                                        _p(vde, _f64p), _p(so, _u64p), _p(ids, _u32p), _p(pde, _f64p), P)
        return P, vde, so, ids, pde

    # R0 graph.cpp:163-242
    def load_graph(self, path):
        n, m = C.c_uint32(), C.c_uint32()
        lc, md, mlf = C.c_uint32(), C.c_uint32(), C.c_uint32()
        po, pn, pl = _u32p(), _u32p(), _u32p()
        rc = self.L.orc_load_graph(path.encode(), C.byref(n), C.byref(m), C.byref(po), C.byref(pn),
                                   C.byref(pl), C.byref(lc), C.byref(md), C.byref(mlf))
        if rc != 0:
            raise OSError(f"orc_load_graph({path}) -> {rc}")
        offs = np.ctypeslib.as_array(po, shape=(n.value + 1,)).copy()
        nbrs = np.ctypeslib.as_array(pn, shape=(max(2 * m.value, 1),)).copy()[: 2 * m.value]
        labels = np.ctypeslib.as_array(pl, shape=(max(n.value, 1),)).copy()[: n.value]
        for p in (po, pn, pl):
            self.L.orc_free(p)
        meta = dict(n=n.value, m=m.value, labels_count=lc.value, max_degree=md.value, max_label_freq=mlf.value)
        return offs, nbrs, labels, meta

    # R1 main.cpp:77-85
    def read_membership(self, path, n):
        s = np.zeros(n, np.uint32)
        mem = np.zeros(n, np.uint32)
        rc = self.L.orc_read_membership(path.encode(), n, _p(s, _u32p), _p(mem, _u32p))
        if rc != 0:
            raise OSError(f"orc_read_membership({path}) -> {rc}")
        return s, mem

    def _enum(self, fn, offs, nbrs, sorted_nodes, L):
        n = len(offs) - 1
        offs = np.ascontiguousarray(offs, np.uint32)
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        sn = np.ascontiguousarray(sorted_nodes, np.uint32)
        P = fn(n, _p(offs, _u32p), _p(nbrs, _u32p), _p(sn, _u32p), L, None, 0)
        out = np.zeros((P, L), np.uint32)
        if P:
            P2 = fn(n, _p(offs, _u32p), _p(nbrs, _u32p), _p(sn, _u32p), L, _p(out, _u32p), P)
            assert P2 == P
        return out

    # R2 custom.h:52-92 (faithful hash-set DFS)
    def enumerate_dfs_hash(self, offs, nbrs, sorted_nodes, L=3):
        return self._enum(self.L.orc_enumerate_dfs_hash, offs, nbrs, sorted_nodes, L)

    # R2 closed form
    def enumerate_closed(self, offs, nbrs, sorted_nodes, L=3):
        return self._enum(self.L.orc_enumerate_closed, offs, nbrs, sorted_nodes, L)

    def count_per_start(self, offs, nbrs, sorted_nodes, L=3):
        n = len(offs) - 1
        offs = np.ascontiguousarray(offs, np.uint32)
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        sn = np.ascontiguousarray(sorted_nodes, np.uint32)
        c = np.zeros(n, np.uint64)
        self.L.orc_count_per_start(n, _p(offs, _u32p), _p(nbrs, _u32p), _p(sn, _u32p), L, _p(c, _u64p))
        return c

    def count_per_start_l3(self, offs, nbrs, sorted_nodes):
        """count_per_start(L = 4) at sizes the DFS cannot walk (sorted-row merges, all cores); counts by processing position."""
        n = len(offs) - 1
        offs, nbrs = np.ascontiguousarray(offs, np.uint32), np.ascontiguousarray(nbrs, np.uint32)
        sn = np.ascontiguousarray(sorted_nodes, np.uint32)
        c = np.zeros(n, np.uint64)
        if self.L.orc_count_per_start_l3(n, _p(offs, _u32p), _p(nbrs, _u32p), _p(sn, _u32p), _p(c, _u64p)) != 0:
            raise MemoryError("orc_count_per_start_l3")
        return c

    def enumerate_starts(self, offs, nbrs, sorted_nodes, L, first, counts):
        """The closed-form DFS for the start vertices at processing positions first .. first + len(counts) - 1, whose row counts
        are `counts`: rows in emission order (P x L uint32)."""
        n = len(offs) - 1
        offs, nbrs = np.ascontiguousarray(offs, np.uint32), np.ascontiguousarray(nbrs, np.uint32)
        sn = np.ascontiguousarray(sorted_nodes, np.uint32)
        so = np.zeros(len(counts) + 1, np.uint64)
        np.cumsum(np.asarray(counts, np.uint64), out=so[1:])
        out = np.zeros((int(so[-1]), L), np.uint32)
        got = self.L.orc_enumerate_starts(n, _p(offs, _u32p), _p(nbrs, _u32p), _p(sn, _u32p), L, int(first), len(counts), _p(so, _u64p),
                                          _p(out, _u32p))
        if got != int(so[-1]):
            raise ValueError("a start vertex has a different number of paths than its count")
        return out

    # R3 custom.h:492-511
    def gen_vde_x(self, label, e):
        x = np.zeros(e, np.float64)
        self.L.orc_gen_vde_x(int(label), e, _p(x, _f64p))
        return x

    def label_table(self, n_labels, e):
        return np.stack([self.gen_vde_x(l, e) for l in range(n_labels)]) if n_labels else np.zeros((0, e))

    # R4 custom.h:513-544
    def gen_vde(self, offs, nbrs, labels, e):
        n = len(offs) - 1
        offs = np.ascontiguousarray(offs, np.uint32)
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        labels = np.ascontiguousarray(labels, np.uint32)
        x = np.zeros((n, e))
        nx = np.zeros((n, e))
        vde = np.zeros((n, e))
        self.L.orc_gen_vde(n, _p(offs, _u32p), _p(nbrs, _u32p), _p(labels, _u32p), e,
                           _p(x, _f64p), _p(nx, _f64p), _p(vde, _f64p))
        return x, nx, vde

    # R5 custom.h:546-572
    def gen_pde(self, paths, e, offs, labels, x, vde):
        P, L = paths.shape
        paths = np.ascontiguousarray(paths, np.uint32)
        offs = np.ascontiguousarray(offs, np.uint32)
        labels = np.ascontiguousarray(labels, np.uint32)
        x = np.ascontiguousarray(x, np.float64)
        vde = np.ascontiguousarray(vde, np.float64)
        pde = np.zeros((P, e * L))
        pdl = np.zeros((P, e * L))
        pl = np.zeros((P, L), np.uint32)
        pd = np.zeros((P, L), np.uint32)
        self.L.orc_gen_pde(P, L, _p(paths, _u32p), e, _p(offs, _u32p), _p(labels, _u32p), _p(x, _f64p),
                           _p(vde, _f64p), _p(pde, _f64p), _p(pdl, _f64p), _p(pl, _u32p), _p(pd, _u32p))
        return pde, pdl, pl, pd

    # R7 main.cpp:98-119
    def write_all_paths(self, path, paths):
        paths = np.ascontiguousarray(paths, np.uint32)
        rc = self.L.orc_write_all_paths(path.encode(), paths.shape[0], paths.shape[1], _p(paths, _u32p))
        if rc:
            raise OSError(path)

    def write_partition_paths(self, path, paths, membership, pid):
        paths = np.ascontiguousarray(paths, np.uint32)
        mem = np.ascontiguousarray(membership, np.uint32)
        rc = self.L.orc_write_partition_paths(path.encode(), paths.shape[0], paths.shape[1],
                                              _p(paths, _u32p), _p(mem, _u32p), pid)
        if rc:
            raise OSError(path)

    def format_all_paths(self, paths):
        paths = np.ascontiguousarray(paths, np.uint32)
        P, L = paths.shape
        nbytes = self.L.orc_format_all_paths(P, L, _p(paths, _u32p), None)
        buf = C.create_string_buffer(nbytes)
        self.L.orc_format_all_paths(P, L, _p(paths, _u32p), buf)
        return buf.raw

    # R6 decoder/validator
    def index_validate(self, img):
        """img: bytes of an index.dat.  Returns dict(hdr..., leaf_son, leaf_pt, height) or raises."""
        hdr = (C.c_int32 * 8)()
        rc = self.L.orc_index_validate(img, len(img), hdr, None, None, 0, None)
        if rc != 0:
            raise ValueError(f"index.dat invalid: code {rc}, header {list(hdr)}")
        nd, dim = hdr[3], hdr[2]
        son = np.zeros(nd, np.int32)
        pt = np.zeros((nd, dim))
        h = C.c_int32()
        rc = self.L.orc_index_validate(img, len(img), hdr, _p(son, C.POINTER(C.c_int32)), _p(pt, _f64p), nd,
                                       C.byref(h))
        assert rc == 0
        keys = ["blocklength", "n_blocks", "dim", "num_data", "dnodes", "inodes", "root_is_data", "root"]
        out = dict(zip(keys, list(hdr)))
        out.update(leaf_son=son, leaf_pt=pt, height=h.value)
        return out

    # Partition::build_auxiliary_index, custom.h:268-364 (numpy restatement; struct Auxiliary_Index
    # custom.h:152-165).  Pinned by tests/golden/aux_index/ (dumps of the compiled reference's own constructor).
    def aux_index(self, img, L, path_degrees, path_pde_label):
        """img: bytes of an index.dat; path_degrees [n x L], path_pde_label [n x D]: the partition's paths in partition
        order (what a leaf entry's `son` indexes, custom.h:243,274).  The same recursive walk from the root as the
        reference: leaves reduce their paths (custom.h:270-313), inner nodes recurse into every child first, set the
        child's key from the entry's upper bounds (custom.h:319-328) and reduce the children (custom.h:330-360).
        Returns key [N], degrees [N x L] uint32, label_mbr [N x 2D] (lo0, hi0, lo1, hi1, ...), by node block id."""
        img = bytes(img)
        nblk = int(np.frombuffer(img, np.int32, 1, 4)[0])
        dim = int(np.frombuffer(img, np.int32, 1, 8)[0])
        root = int(np.frombuffer(img, np.int32, 1, 25)[0])
        esz = 16 * dim + 4
        key = np.zeros(nblk)                       # Auxiliary_Index(): key 0, degrees 0, label_mbr 0 (custom.h:159-164)
        deg = np.zeros((nblk, L), np.uint32)
        mbr = np.zeros((nblk, 2 * dim))
        path_degrees = np.asarray(path_degrees)
        path_pde_label = np.asarray(path_pde_label)

        def walk(b):
            base = (b + 1) * 4096                  # blk_file.cpp:110
            level = int(np.frombuffer(img, np.int8, 1, base)[0])
            ne = int(np.frombuffer(img, np.int32, 1, base + 1)[0])
            if ne == 0:
                return
            ent = np.frombuffer(img, np.uint8, ne * esz, base + 5).reshape(ne, esz)
            son = ent[:, 16 * dim:].copy().view(np.int32).reshape(-1)
            if level == 0:
                d, lo, hi = path_degrees[son], path_pde_label[son], path_pde_label[son]
            else:
                bounces = ent[:, :16 * dim].copy().view(np.float64).reshape(ne, 2 * dim)
                for i, c in enumerate(son):
                    walk(int(c))
                    k = 0.0
                    for j in range(dim):            # custom.h:324-328: subtracted one by one, in this order
                        k -= float(bounces[i, 2 * j + 1])
                    key[c] = k
                d, lo, hi = deg[son], mbr[son, 0::2], mbr[son, 1::2]
            deg[b] = d.max(axis=0)
            mbr[b, 0::2] = lo.min(axis=0)
            mbr[b, 1::2] = hi.max(axis=0)

        walk(root)
        return key, deg, mbr

    # GNN-PGE offline (GNN-PGE/src/main.cpp:91-195)
    def pge_groups(self, offs, nbrs, e, x, vde):
        n = len(offs) - 1
        offs = np.ascontiguousarray(offs, np.uint32)
        nbrs = np.ascontiguousarray(nbrs, np.uint32)
        x = np.ascontiguousarray(x, np.float64)
        vde = np.ascontiguousarray(vde, np.float64)
        pg = np.zeros((n, 4 * e))
        plg = np.zeros((n, 4 * e))
        self.L.orc_pge_groups(n, _p(offs, _u32p), _p(nbrs, _u32p), e, _p(x, _f64p), _p(vde, _f64p), _p(pg, _f64p),
                              _p(plg, _f64p))
        return pg, plg

    def pge_write_bin(self, path, e, offs, labels, x, nx, vde, pg, plg, key_fill=0.0):
        n = len(offs) - 1
        a = [np.ascontiguousarray(t, np.float64) for t in (x, nx, vde, pg, plg)]
        offs = np.ascontiguousarray(offs, np.uint32)
        labels = np.ascontiguousarray(labels, np.uint32)
        rc = self.L.orc_pge_write_bin(path.encode(), n, e, _p(offs, _u32p), _p(labels, _u32p), *[_p(t, _f64p) for t in a],
                                      key_fill)
        if rc:
            raise OSError(path)
