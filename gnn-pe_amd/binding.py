"""ctypes view of the C-ABI in include/gnnpe_hip.h (libgnnpe_hip.so).

This is plumbing for bench.py and the tests: numpy arrays / raw device pointers go in, nothing
torch-typed crosses the boundary.  `load()` raises if the HIP library is missing or cannot be
loaded -- there is deliberately no CPU fallback (the oracle lives in oracle/ and is never
imported from here).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GNNPE_LIB_PATH: the A/B scripts under scripts/ point it at the diagnostic build, `make -C gnn-pe_amd DIAG=1 diag`)
LIB_PATH = os.environ.get("GNNPE_LIB_PATH") or os.path.join(_HERE, "libgnnpe_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gnnpe_hip.h")

_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes): one row per declaration in include/gnnpe_hip.h
SIGNATURES = {
    "gnnpe_abi_version": (C.c_int, []),
    "gnnpe_last_error": (C.c_char_p, []),
    "gnnpe_create": (_vp, [C.c_int]),
    "gnnpe_destroy": (None, [_vp]),
    "gnnpe_set_stream": (C.c_int, [_vp, _vp]),
    "gnnpe_sync": (C.c_int, [_vp]),
    "gnnpe_dev_alloc": (C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    "gnnpe_dev_free": (C.c_int, [_vp, _vp]),
    "gnnpe_copy_to_host": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "gnnpe_device_count": (C.c_int, []),
    "gnnpe_get_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "gnnpe_copy_device": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "gnnpe_gather_rows_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, C.c_uint64, _vp, _vp]),
    "gnnpe_write_device_file": (C.c_int, [_vp, _vp, C.c_uint64, C.c_char_p]),
    "gnnpe_load_csr": (C.c_int, [_vp, C.c_uint32, _u32p, _u32p, _u32p]),
    "gnnpe_load_rows": (C.c_int, [_vp, C.c_uint32, _u32p, C.c_uint32, _u32p, _u64p, _u32p, C.c_uint64]),
    "gnnpe_set_order": (C.c_int, [_vp, _u32p, _u32p, C.c_uint32]),
    "gnnpe_set_slab": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "gnnpe_set_label_table": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _f64p]),
    "gnnpe_host_label_table": (C.c_int, [C.c_uint32, C.c_uint32, _f64p]),
    "gnnpe_host_load_graph": (C.c_int, [C.c_char_p, _u32p, _u32p, C.POINTER(_u32p), C.POINTER(_u32p), C.POINTER(_u32p),
                                        _u32p]),
    "gnnpe_host_load_multigraph": (C.c_int, [C.c_char_p, _u32p, _u32p, C.POINTER(_u32p), C.POINTER(_u32p), C.POINTER(_u32p),
                                             _u32p, C.POINTER(_u32p), C.POINTER(_u32p)]),
    "gnnpe_set_multigraph_rows": (C.c_int, [_vp, C.c_uint32, _u64p, _u32p]),
    "gnnpe_host_read_membership": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, _u32p, _u32p]),
    "gnnpe_host_free": (None, [_vp]),
    "gnnpe_halo_need": (C.c_int, [_vp, C.c_uint32, _u32p, _vp, C.c_uint64, _u64p]),
    "gnnpe_rows_degree": (C.c_int, [_vp, C.c_uint64, _vp, _vp]),
    "gnnpe_rows_pack": (C.c_int, [_vp, C.c_uint64, _vp, _vp, C.c_uint64]),
    "gnnpe_rows_append": (C.c_int, [_vp, C.c_uint64, _vp, _vp, _vp, C.c_uint64, C.c_uint32]),
    "gnnpe_rows_drop_halo": (C.c_int, [_vp]),
    "gnnpe_rows_held": (C.c_int, [_vp, _u64p, _u64p, _u32p]),
    "gnnpe_vde": (C.c_int, [_vp, _f64p, _f64p, _f64p]),
    "gnnpe_vde_device_ptr": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "gnnpe_vde_pack_slab": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp]),
    "gnnpe_vde_unpack_slab": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp]),
    "gnnpe_vde_unpack_all": (C.c_int, [_vp, C.c_uint32, _u32p, C.c_uint32, C.c_uint32, _vp]),
    "gnnpe_count_paths": (C.c_int, [_vp, C.c_uint32, _u64p, _u64p]),
    "gnnpe_fill_paths": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _u32p, _f64p, _f64p]),
    "gnnpe_fill_paths_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp, _vp, _vp]),
    "gnnpe_build_index_partition_aux_device": (C.c_int, [_vp, C.c_uint32, C.POINTER(_vp), _u64p, C.POINTER(C.c_int32), C.POINTER(_vp),
                                                         C.POINTER(_vp), C.POINTER(_vp), _u32p]),
    "gnnpe_output_pool_create": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "gnnpe_output_pool_acquire": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _u64p]),
    "gnnpe_output_pool_report": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_float), _u32p, _u32p, C.POINTER(C.c_int)]),
    "gnnpe_output_pool_destroy": (None, [_vp]),
    "gnnpe_count_paths_enqueue": (C.c_int, [_vp, C.c_uint32]),
    "gnnpe_count_total": (C.c_int, [_vp, _u64p]),
    "gnnpe_count_total_device": (C.c_int, [_vp, _vp]),
    "gnnpe_fill_paths_capped_device": (C.c_int, [_vp, C.c_uint64, _vp, _vp]),
    "gnnpe_path_partitions_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp]),
    "gnnpe_text_paths": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, _vp, C.c_uint64, _u64p]),
    "gnnpe_text_ids": (C.c_int, [_vp, C.c_uint64, _vp, _vp, C.c_uint64, _u64p]),
    "gnnpe_select_partition": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint32, C.c_uint64, _vp, _u64p]),
    "gnnpe_rows_checksum_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, C.c_uint64, _u64p]),
    "gnnpe_host_query_plan": (C.c_int, [C.c_char_p, C.c_uint32, _u32p, _u32p, C.POINTER(_u32p), C.POINTER(_u32p),
                                        C.POINTER(_u32p), C.POINTER(_f64p)]),
    "gnnpe_host_load_path_sidecar": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, _u32p, _u32p, _u64p, _u32p, _u32p,
                                               C.POINTER(_u32p), C.POINTER(_u32p), C.POINTER(_u32p), C.POINTER(_f64p),
                                               C.POINTER(_f64p)]),
    "gnnpe_host_write_paths_header": (C.c_int, [_vp, C.c_uint32, C.c_uint64]),
    "gnnpe_host_load_partition_sidecar": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, _u32p, _u32p, _u64p, _u32p,
                                                    _u32p, C.POINTER(_u32p), C.POINTER(_u32p), C.POINTER(_u32p),
                                                    C.POINTER(_u32p), C.POINTER(_f64p), C.POINTER(_f64p)]),
    "gnnpe_aux_index_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint32, _vp, C.POINTER(_vp), C.POINTER(_vp),
                                         C.POINTER(_vp), _u32p, _u32p]),
    "gnnpe_build_aux_index": (C.c_int, [_vp, C.c_uint32, C.c_char_p]),
    "gnnpe_host_load_aux_index": (C.c_int, [C.c_char_p, _u32p, _u32p, _u32p, C.POINTER(_f64p), C.POINTER(_u32p),
                                            C.POINTER(_f64p)]),
    "gnnpe_pinned_alloc": (C.c_int, [C.c_uint64, C.POINTER(_vp)]),
    "gnnpe_pinned_free": (None, [_vp]),
    "gnnpe_set_degrees": (C.c_int, [_vp, _u32p]),
    "gnnpe_filter_candidates": (C.c_int, [_vp, C.c_uint32, _u32p, _u32p, _u32p, _f64p, C.c_uint32, C.c_double, _u32p,
                                          _f64p]),
    "gnnpe_build_index_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, C.POINTER(_vp), _u64p,
                                           C.POINTER(C.c_int32)]),
    "gnnpe_build_index": (C.c_int, [_vp, C.c_uint32, C.c_char_p]),
    "gnnpe_build_index_files": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
    "gnnpe_build_index_partition_device": (C.c_int, [_vp, C.c_uint32, C.POINTER(_vp), _u64p, C.POINTER(C.c_int32)]),
    "gnnpe_build_box_index_device": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _vp, C.POINTER(_vp), _u64p,
                                               C.POINTER(C.c_int32)]),
    "gnnpe_pge_groups": (C.c_int, [_vp, _f64p, _f64p]),
    "gnnpe_pge_device_ptr": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "gnnpe_pge_build_index": (C.c_int, [_vp, C.c_uint64, _u32p, C.c_char_p]),
    "gnnpe_fill_kernel_name": (C.c_char_p, []),
    "gnnpe_set_fill_variant": (C.c_int, [_vp, C.c_int]),
    "gnnpe_set_emit_shape": (C.c_int, [_vp, C.c_int]),
    "gnnpe_index_file_bytes": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_int]),
    "gnnpe_emit_kernel_name": (C.c_char_p, [_vp]),
    "gnnpe_emit_calibrate_device": (C.c_int, [_vp, C.c_uint64, _vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
}

ABI_VERSION = 6  # GNNPE_ABI_VERSION of include/gnnpe_hip.h
_lib = None


class GnnpeError(RuntimeError):
    pass


# include/gnnpe_online.h (libgnnpe_online.so): the refinement -- out of SURVEY section 8's scope, a library of its own since round 6
ONLINE_LIB_PATH = os.path.join(_HERE, "libgnnpe_online.so")
ONLINE_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "gnnpe_online.h")
ONLINE_SIGNATURES = {
    "gnnpe_host_refine": (C.c_int, [C.c_uint32, _u32p, _u32p, _u32p, C.c_char_p, _u32p, C.c_uint64, _u64p]),
    "gnnpe_refine": (C.c_int, [_vp, C.c_char_p, _u32p, C.c_uint64, _u64p, _f64p]),
}
_online = None


def load_online():
    """dlopen libgnnpe_online.so (it links against libgnnpe_hip.so, which load() brings in first).  Raises if absent."""
    global _online
    if _online is not None:
        return _online
    load()
    if not os.path.exists(ONLINE_LIB_PATH):
        raise GnnpeError(f"{ONLINE_LIB_PATH} is missing: build it with `make -C gnn-pe_amd`")
    lib = C.CDLL(ONLINE_LIB_PATH)
    for name, (res, args) in ONLINE_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _online = lib
    return lib


def build(force=False):
    """Compile libgnnpe_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))] + [HEADER_PATH]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "libgnnpe_hip.so", "libgnnpe_online.so"])
    return LIB_PATH


def load():
    """dlopen libgnnpe_hip.so and bind every symbol the header declares.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GnnpeError(f"{LIB_PATH} is missing: build it with `make -C gnn-pe_amd` "
                         "(there is no CPU fallback for the HIP engine)")
    try:
        # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's): whichever is loaded first
        # serves the whole process, and torch cannot initialise on the other one.  Load torch's first.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    if lib.gnnpe_abi_version() != ABI_VERSION:
        raise GnnpeError(f"ABI version {lib.gnnpe_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def _np(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype)


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _dev(t):
    """Raw device pointer of a torch tensor / int / None."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


def host_label_table(n_labels, e):
    """R3: gen_vde_x (custom.h:492-511) for labels 0..n_labels-1 -- host arithmetic inside the library."""
    out = np.zeros((n_labels, e))
    rc = load().gnnpe_host_label_table(n_labels, e, _ptr(out, _f64p))
    if rc:
        raise GnnpeError(load().gnnpe_last_error().decode())
    return out


def simple_rows(offsets, nbrs):
    """The rows the reference's DFS + hash set amount to (custom.h:68-77): every ascending list without its repeats.
    offsets [n + 1], nbrs: a CSR as graph.cpp:211-233 leaves a file with duplicate `e` lines.  Returns (offsets, nbrs) uint32."""
    offsets, nbrs = _np(offsets, np.uint32), _np(nbrs, np.uint32)
    n = len(offsets) - 1
    row = np.repeat(np.arange(n, dtype=np.uint32), np.diff(offsets.astype(np.int64)))
    keep = np.ones(len(nbrs), bool)
    if len(nbrs):
        keep[1:] = (nbrs[1:] != nbrs[:-1]) | (row[1:] != row[:-1])
    so = np.zeros(n + 1, np.uint32)
    so[1:] = np.cumsum(np.bincount(row[keep], minlength=n))
    return so, np.ascontiguousarray(nbrs[keep])


def host_load_graph(path, strict=True):
    """R0 through the library's own loader (host/graph_loader.cpp).  Returns a dict like synth graphs.  strict=False: a file with
    duplicate `e` lines is loaded as the reference loads it (offsets / nbrs keep the repeats) and the dict also holds
    simple_offsets / simple_nbrs, the rows the enumeration runs on (None for a simple graph).  Self-loops are refused either way
    (the reference's loader leaves a slot uninitialised for them: graph.cpp:211-218)."""
    lib = load()
    n, m = C.c_uint32(), C.c_uint32()
    po, pn, pl = _u32p(), _u32p(), _u32p()
    meta = (C.c_uint32 * 3)()
    if not strict:
        pso, psn = _u32p(), _u32p()
        rc = lib.gnnpe_host_load_multigraph(path.encode(), C.byref(n), C.byref(m), C.byref(po), C.byref(pn), C.byref(pl), meta,
                                            C.byref(pso), C.byref(psn))
    else:
        rc = lib.gnnpe_host_load_graph(path.encode(), C.byref(n), C.byref(m), C.byref(po), C.byref(pn), C.byref(pl), meta)
    if rc == -1:
        raise FileNotFoundError(lib.gnnpe_last_error().decode())
    if rc:
        raise GnnpeError(lib.gnnpe_last_error().decode())
    offs = np.ctypeslib.as_array(po, shape=(n.value + 1,)).copy()
    nbrs = np.ctypeslib.as_array(pn, shape=(max(2 * m.value, 1),)).copy()[: 2 * m.value]
    labels = np.ctypeslib.as_array(pl, shape=(max(n.value, 1),)).copy()[: n.value]
    extra = {}
    if not strict:
        if pso:
            so = np.ctypeslib.as_array(pso, shape=(n.value + 1,)).copy()
            extra = dict(simple_offsets=so, simple_nbrs=np.ctypeslib.as_array(psn, shape=(max(int(so[-1]), 1),)).copy()[: int(so[-1])])
            lib.gnnpe_host_free(pso)
            lib.gnnpe_host_free(psn)
        else:
            extra = dict(simple_offsets=None, simple_nbrs=None)
    for p in (po, pn, pl):
        lib.gnnpe_host_free(p)
    return dict(n=n.value, m=m.value, offsets=offs, nbrs=nbrs, labels=labels, labels_count=meta[0], max_degree=meta[1],
                max_label_frequency=meta[2], **extra)


def host_query_plan(path, e):
    """Query side of the online filter (main.cpp:136-151) through the library's host planner.
    Returns dict(n_vertices, vids, labels, degrees (n_paths x 3), pde (n_paths x 3e))."""
    lib = load()
    nv, npth = C.c_uint32(), C.c_uint32()
    pv, pl, pd = _u32p(), _u32p(), _u32p()
    pp = _f64p()
    rc = lib.gnnpe_host_query_plan(path.encode(), int(e), C.byref(nv), C.byref(npth), C.byref(pv), C.byref(pl), C.byref(pd),
                                   C.byref(pp))
    if rc == -1:
        raise FileNotFoundError(lib.gnnpe_last_error().decode())
    if rc:
        raise GnnpeError(lib.gnnpe_last_error().decode())
    k = npth.value
    out = dict(n_vertices=nv.value)
    for name, ptr in (("vids", pv), ("labels", pl), ("degrees", pd)):
        out[name] = np.ctypeslib.as_array(ptr, shape=(max(k * 3, 1),)).copy()[: k * 3].reshape(k, 3)
        lib.gnnpe_host_free(ptr)
    out["pde"] = np.ctypeslib.as_array(pp, shape=(max(k * 3 * e, 1),)).copy()[: k * 3 * e].reshape(k, 3 * e)
    lib.gnnpe_host_free(pp)
    return out


def host_refine(g, query_path, bitmap, limit=0xFFFFFFFF):
    """Refinement half of the online step on the host (custom.h:634-932): answer count from the filter's bitmap."""
    lib = load()
    out = C.c_uint64()
    o, nb, lb = _np(g["offsets"], np.uint32), _np(g["nbrs"], np.uint32), _np(g["labels"], np.uint32)
    bm = _np(bitmap, np.uint32)
    rc = load_online().gnnpe_host_refine(len(o) - 1, _ptr(o, _u32p), _ptr(nb, _u32p), _ptr(lb, _u32p), query_path.encode(),
                               _ptr(bm, _u32p), int(limit), C.byref(out))
    if rc:
        raise GnnpeError(lib.gnnpe_last_error().decode())
    return out.value


def host_load_path_sidecar(paths_bin, vde_bin, labels, degrees):
    """gen_pde's per-path arrays (custom.h:546-572) from the binary sidecars of `gnnpe_main --sidecars`."""
    lib = load()
    lab, deg = _np(labels, np.uint32), _np(degrees, np.uint32)
    P, L, e = C.c_uint64(), C.c_uint32(), C.c_uint32()
    pv, pl, pd = _u32p(), _u32p(), _u32p()
    pp, px = _f64p(), _f64p()
    rc = lib.gnnpe_host_load_path_sidecar(paths_bin.encode(), vde_bin.encode(), len(lab), _ptr(lab, _u32p), _ptr(deg, _u32p),
                                          C.byref(P), C.byref(L), C.byref(e), C.byref(pv), C.byref(pl), C.byref(pd), C.byref(pp),
                                          C.byref(px))
    if rc:
        raise GnnpeError(lib.gnnpe_last_error().decode())
    n, Lv, D = P.value, L.value, L.value * e.value
    out = {}
    for name, ptr, w in (("vids", pv, Lv), ("labels", pl, Lv), ("degrees", pd, Lv), ("pde", pp, D), ("pde_label", px, D)):
        out[name] = np.ctypeslib.as_array(ptr, shape=(max(n * w, 1),)).copy()[: n * w].reshape(n, w)
        lib.gnnpe_host_free(ptr)
    return out


def host_load_partition_sidecar(paths_bin, vde_bin, partition_paths_txt, labels, degrees):
    """the partition's copy of the paths (Partition::Partition, custom.h:205-216) from the sidecars and the
    partition's partition_paths.txt: gen_pde's arrays for its rows only, plus the global path ids."""
    lib = load()
    lab, deg = _np(labels, np.uint32), _np(degrees, np.uint32)
    P, L, e = C.c_uint64(), C.c_uint32(), C.c_uint32()
    pi, pv, pl, pd = _u32p(), _u32p(), _u32p(), _u32p()
    pp, px = _f64p(), _f64p()
    rc = lib.gnnpe_host_load_partition_sidecar(paths_bin.encode(), vde_bin.encode(), partition_paths_txt.encode(), len(lab),
                                               _ptr(lab, _u32p), _ptr(deg, _u32p), C.byref(P), C.byref(L), C.byref(e),
                                               C.byref(pi), C.byref(pv), C.byref(pl), C.byref(pd), C.byref(pp), C.byref(px))
    if rc:
        raise GnnpeError(lib.gnnpe_last_error().decode())
    n, Lv, D = P.value, L.value, L.value * e.value
    out = {}
    for name, ptr, w in (("path_ids", pi, 1), ("vids", pv, Lv), ("labels", pl, Lv), ("degrees", pd, Lv), ("pde", pp, D),
                         ("pde_label", px, D)):
        out[name] = np.ctypeslib.as_array(ptr, shape=(max(n * w, 1),)).copy()[: n * w].reshape(n, w)
        lib.gnnpe_host_free(ptr)
    out["path_ids"] = out["path_ids"].reshape(-1)
    return out


def host_load_aux_index(path):
    """aux_index.bin -> what Partition::build_auxiliary_index (custom.h:268-364) computes, by node block id."""
    lib = load()
    N, L, D = C.c_uint32(), C.c_uint32(), C.c_uint32()
    pk, pd, pm = _f64p(), _u32p(), _f64p()
    rc = lib.gnnpe_host_load_aux_index(path.encode(), C.byref(N), C.byref(L), C.byref(D), C.byref(pk), C.byref(pd), C.byref(pm))
    if rc:
        raise GnnpeError(lib.gnnpe_last_error().decode())
    n = N.value
    out = dict(L=L.value, D=D.value)
    for name, ptr, w in (("key", pk, 1), ("degrees", pd, L.value), ("label_mbr", pm, 2 * D.value)):
        out[name] = np.ctypeslib.as_array(ptr, shape=(max(n * w, 1),)).copy()[: n * w].reshape(n, w)
        lib.gnnpe_host_free(ptr)
    out["key"] = out["key"].reshape(-1)
    return out


def host_read_membership(path, n, p):
    """R1 (main.cpp:77-85) through the library's reader."""
    lib = load()
    sn = np.zeros(n, np.uint32)
    mem = np.zeros(n, np.uint32)
    rc = lib.gnnpe_host_read_membership(path.encode(), n, p, _ptr(sn, _u32p), _ptr(mem, _u32p))
    if rc:
        raise GnnpeError(lib.gnnpe_last_error().decode())
    return sn, mem


class _DevArray:
    """A typed view of raw device memory for torch (`torch.as_tensor(view, device=...)` shares it, no copy)."""

    def __init__(self, ptr, shape, typestr, owner=None):
        self.owner = owner  # keeps the pool (and through it the engine) alive for as long as a tensor shares the memory
        self.__cuda_array_interface__ = dict(shape=tuple(int(x) for x in shape), typestr=typestr, data=(int(ptr), False),
                                             version=2, strides=None)


class OutputPool:
    """gnnpe_output_pool_*: the emit kernel's output buffers (ids: rows_cap x L uint32, pde: rows_cap x D doubles).  One
    allocation by default; with `candidates` > 1 the fastest of that many independent allocations, each timed with the emit
    kernel (or, without a count on the engine, a streaming write), the others freed before the constructor returns.  With an
    l = 2 count on the engine the pool also times both emit shapes into the buffer it keeps (gnnpe_emit_calibrate_device).
    `ids` / `pde` are raw device pointers (ints); ids_tensor() / pde_tensor() are torch views of the same memory: close() refuses
    while one of them is alive (Engine.close() does not ask -- drop the views before closing the engine).  Create the pool
    AFTER eng.count_paths() so that the probe is the emit kernel itself."""

    NO_CALIBRATION = 0x80000000  # GNNPE_POOL_NO_CALIBRATION

    def __init__(self, eng, rows_cap, L, D, candidates=1, calibrate=True):
        self.eng, self.lib = eng, eng.lib
        self.rows_cap, self.L, self.D = int(max(rows_cap, 1)), int(L), int(D)
        h = C.c_void_p()
        eng._ck(self.lib.gnnpe_output_pool_create(eng.ctx, self.rows_cap, self.L, self.D, int(candidates) | (0 if calibrate else self.NO_CALIBRATION), C.byref(h)))
        self.h = h
        self._views = []  # weak references to the arrays handed to torch
        eng._pools.append(self)
        self._refresh()

    def _refresh(self):
        a, b, cap = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self.eng._ck(self.lib.gnnpe_output_pool_acquire(self.h, C.byref(a), C.byref(b), C.byref(cap)))
        self.ids, self.pde = int(a.value or 0), int(b.value or 0)

    def report(self):
        ms = (C.c_float * 64)()
        n, kept, kern = C.c_uint32(), C.c_uint32(), C.c_int()
        self.eng._ck(self.lib.gnnpe_output_pool_report(self.h, 64, ms, C.byref(n), C.byref(kept), C.byref(kern)))
        return dict(candidates_ms=[round(float(ms[i]), 4) for i in range(n.value)], kept=int(kept.value),
                    probe="emit kernel" if kern.value else "streaming write")

    def ids_tensor(self, device):
        import torch
        return torch.as_tensor(self._view(self.ids, (self.rows_cap, self.L), "<i4"), device=device)

    def pde_tensor(self, device):
        import torch
        return torch.as_tensor(self._view(self.pde, (self.rows_cap, self.D), "<f8"), device=device) if self.D else None

    def _view(self, ptr, shape, typestr):
        import weakref
        a = _DevArray(ptr, shape, typestr, owner=self)
        self._views.append(weakref.ref(a))
        return a

    def close(self, force=False):
        """Frees the pool's memory.  A pool never outlives its engine: Engine.close() closes its pools first (force)."""
        if not force and any(r() is not None for r in self._views):
            raise GnnpeError("OutputPool.close(): a tensor from ids_tensor() / pde_tensor() still shares the pool's memory; drop it first")
        if self.h and self.eng.ctx:
            self.lib.gnnpe_output_pool_destroy(self.h)
            if self in self.eng._pools:
                self.eng._pools.remove(self)
        self.h = None

    def __del__(self):
        try:
            self.close(force=True)
        except Exception:
            pass


class Engine:
    """One context = one GPU.  Method names follow the reference functions they replace."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        self.ctx = self.lib.gnnpe_create(int(device))
        if not self.ctx:
            raise GnnpeError("gnnpe_create: " + self.lib.gnnpe_last_error().decode())
        self.n = 0
        self.e = 0
        self.slab = (0, 0)
        self.total = None
        self._pools = []
        self.stream_handle = None  # None = the context's own stream
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if self.ctx:
            for p in list(getattr(self, "_pools", [])):
                p.close(force=True)
            self.lib.gnnpe_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise GnnpeError(f"[{rc}] " + self.lib.gnnpe_last_error().decode())

    def set_stream(self, stream):
        """stream: raw hipStream_t as int (torch.cuda.current_stream().cuda_stream), or None."""
        self._ck(self.lib.gnnpe_set_stream(self.ctx, C.c_void_p(stream) if stream else None))
        self.stream_handle = int(stream) if stream else None

    def sync(self):
        self._ck(self.lib.gnnpe_sync(self.ctx))

    # R0: Static_Graph::loadGraphFromFile output (graph.cpp:163-242)
    def load_csr(self, offsets, nbrs, labels):
        offsets, nbrs, labels = _np(offsets, np.uint32), _np(nbrs, np.uint32), _np(labels, np.uint32)
        n = len(offsets) - 1
        assert len(labels) == n and len(nbrs) == offsets[-1]
        self._ck(self.lib.gnnpe_load_csr(self.ctx, n, _ptr(offsets, _u32p), _ptr(nbrs, _u32p), _ptr(labels, _u32p)))
        self.n = n
        self.slab = (0, n)

    def load_multigraph(self, offsets, nbrs, labels):
        """R0 for a CSR with repeated entries, as the reference's loader leaves a file with duplicate `e` lines: the simple rows
        for the enumeration (custom.h:68-77), the stored rows for gen_vde and the degree columns (include/gnnpe_hip.h)."""
        so, sn = simple_rows(offsets, nbrs)
        self.load_csr(so, sn, labels)
        if len(sn) != len(nbrs):
            self.set_multigraph_rows(_np(offsets, np.uint32).astype(np.uint64), nbrs)

    def set_multigraph_rows(self, row_offsets, row_nbrs):
        row_offsets, row_nbrs = _np(row_offsets, np.uint64), _np(row_nbrs, np.uint32)
        self._ck(self.lib.gnnpe_set_multigraph_rows(self.ctx, len(row_offsets) - 1, _ptr(row_offsets, _u64p), _ptr(row_nbrs, _u32p)))

    def load_rows(self, n, labels, rows, row_offsets, row_nbrs, nbr_capacity=0):
        labels, rows = _np(labels, np.uint32), _np(rows, np.uint32)
        row_offsets, row_nbrs = _np(row_offsets, np.uint64), _np(row_nbrs, np.uint32)
        self._ck(self.lib.gnnpe_load_rows(self.ctx, n, _ptr(labels, _u32p), len(rows), _ptr(rows, _u32p),
                                          _ptr(row_offsets, _u64p), _ptr(row_nbrs, _u32p), int(nbr_capacity)))
        self.n = n
        self.slab = (0, n)

    # R1: main.cpp:77-85
    def set_order(self, sorted_nodes, membership, p):
        sn, mem = _np(sorted_nodes, np.uint32), _np(membership, np.uint32)
        assert len(sn) == self.n and len(mem) == self.n
        self._ck(self.lib.gnnpe_set_order(self.ctx, _ptr(sn, _u32p), _ptr(mem, _u32p), int(p)))

    def set_slab(self, begin, end):
        self._ck(self.lib.gnnpe_set_slab(self.ctx, int(begin), int(end)))
        self.slab = (int(begin), int(end))

    # R3: gen_vde_x table (custom.h:492-511)
    def set_label_table(self, table):
        t = _np(table, np.float64)
        assert t.ndim == 2
        self._ck(self.lib.gnnpe_set_label_table(self.ctx, t.shape[0], t.shape[1], _ptr(t, _f64p)))
        self.e = t.shape[1]

    # R4: gen_vde (custom.h:513-544)
    def vde(self, want=True):
        if not want:
            self._ck(self.lib.gnnpe_vde(self.ctx, None, None, None))
            return None
        x = np.zeros((self.n, self.e))
        nx = np.zeros((self.n, self.e))
        v = np.zeros((self.n, self.e))
        self._ck(self.lib.gnnpe_vde(self.ctx, _ptr(x, _f64p), _ptr(nx, _f64p), _ptr(v, _f64p)))
        return x, nx, v

    def vde_device_ptr(self):
        a, b = _vp(), _vp()
        self._ck(self.lib.gnnpe_vde_device_ptr(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def vde_pack_slab(self, begin, end, dev_buf):
        self._ck(self.lib.gnnpe_vde_pack_slab(self.ctx, begin, end, _dev(dev_buf)))

    def vde_unpack_slab(self, begin, end, dev_buf):
        self._ck(self.lib.gnnpe_vde_unpack_slab(self.ctx, begin, end, _dev(dev_buf)))

    # R2: dfs + VectorHash (custom.h:52-92), count half
    def vde_unpack_all(self, bounds, stride, skip_rank, dev_buf):
        b = _np(bounds, np.uint32)
        self._ck(self.lib.gnnpe_vde_unpack_all(self.ctx, len(b) - 1, _ptr(b, _u32p), int(stride), int(skip_rank), _dev(dev_buf)))

    def count_paths(self, l=2, per_start=False):
        tot = C.c_uint64()
        ps = np.zeros(self.slab[1] - self.slab[0], np.uint64) if per_start else None
        self._ck(self.lib.gnnpe_count_paths(self.ctx, l, _ptr(ps, _u64p), C.byref(tot)))
        self.total = tot.value
        self.l = l
        return (tot.value, ps) if per_start else tot.value

    def count_paths_enqueue(self, l=2):
        """The count enqueued only (no read-back); count_total() fetches the number, count_total_device() copies it
        to a device word a collective can send."""
        self._ck(self.lib.gnnpe_count_paths_enqueue(self.ctx, l))
        self._total = None  # resolved by the first reader of .total (one read-back)
        self.l = l

    @property
    def total(self):
        """Paths of the last count; after an enqueue-only count the first read fetches it (count_total)."""
        if getattr(self, "_total", None) is None and getattr(self, "l", None) is not None:
            self.count_total()
        return getattr(self, "_total", None)

    @total.setter
    def total(self, value):
        self._total = value

    def count_total(self):
        tot = C.c_uint64()
        self._ck(self.lib.gnnpe_count_total(self.ctx, C.byref(tot)))
        self.total = tot.value
        return tot.value

    def count_total_device(self, dev_u64):
        self._ck(self.lib.gnnpe_count_total_device(self.ctx, _dev(dev_u64)))

    def fill_paths_capped_device(self, cap_rows, dev_vids=None, dev_pde=None):
        """Rows [0, min(total, cap_rows)) into buffers of cap_rows rows; the host never learns the total."""
        self._ck(self.lib.gnnpe_fill_paths_capped_device(self.ctx, int(cap_rows), _dev(dev_vids), _dev(dev_pde)))

    # R2 + R5: emit half (gen_pde, custom.h:546-572)
    def fill_paths(self, begin=0, end=None, ids=True, pde=True, pde_label=False, L=None):
        end = self.total if end is None else end
        cnt = end - begin
        L = getattr(self, "l", 2) + 1 if L is None else L
        D = self.e * L
        v = np.zeros((cnt, L), np.uint32) if ids else None
        p = np.zeros((cnt, D)) if pde else None
        q = np.zeros((cnt, D)) if pde_label else None
        self._ck(self.lib.gnnpe_fill_paths(self.ctx, begin, end, _ptr(v, _u32p), _ptr(p, _f64p), _ptr(q, _f64p)))
        return v, p, q

    def fill_paths_device(self, begin, end, dev_vids=None, dev_pde=None, dev_pde_label=None):
        self._ck(self.lib.gnnpe_fill_paths_device(self.ctx, begin, end, _dev(dev_vids), _dev(dev_pde),
                                                  _dev(dev_pde_label)))

    def rows_checksum_device(self, n_rows, L, dev_ids, first_id=0):
        out = C.c_uint64()
        self._ck(self.lib.gnnpe_rows_checksum_device(self.ctx, int(n_rows), int(L), _dev(dev_ids), int(first_id),
                                                     C.byref(out)))
        return out.value

    def set_degrees(self, degrees):
        d = _np(degrees, np.uint32)
        assert len(d) == self.n
        self._ck(self.lib.gnnpe_set_degrees(self.ctx, _ptr(d, _u32p)))

    # online filter (Partition::query, custom.h:366-489)
    def filter_candidates(self, plan, eps=1e-6):
        """plan: dict from host_query_plan.  Returns (bitmap [n_query_vertices x ceil(n/32)] uint32, device ms)."""
        nv = int(plan["n_vertices"])
        bm = np.zeros((nv, (self.n + 31) // 32), np.uint32)
        ms = C.c_double()
        v, l, d = (_np(plan[k], np.uint32) for k in ("vids", "labels", "degrees"))
        p = _np(plan["pde"], np.float64)
        self._ck(self.lib.gnnpe_filter_candidates(self.ctx, len(v), _ptr(v, _u32p), _ptr(l, _u32p), _ptr(d, _u32p),
                                                  _ptr(p, _f64p), nv, float(eps), _ptr(bm, _u32p), C.byref(ms)))
        return bm, ms.value

    def refine(self, query_path, bitmap, limit=0xFFFFFFFF):
        """Refinement on the device (custom.h:634-932): (answers, device ms) from the filter's candidate bitmap."""
        out, ms = C.c_uint64(), C.c_double()
        bm = _np(bitmap, np.uint32)
        self._ck(load_online().gnnpe_refine(self.ctx, query_path.encode(), _ptr(bm, _u32p), int(limit), C.byref(out), C.byref(ms)))
        return out.value, ms.value

    def path_partitions_device(self, begin, end, dev_part):
        self._ck(self.lib.gnnpe_path_partitions_device(self.ctx, begin, end, _dev(dev_part)))

    def set_fill_variant(self, v):
        self._ck(self.lib.gnnpe_set_fill_variant(self.ctx, int(v)))

    def set_emit_shape(self, shape):
        """0 = whatever gnnpe_emit_calibrate_device measured faster into the buffer (start-vertex waves where nothing was measured),
        1 = one wave per start vertex, one-shot in launch order (k_fill_ranked), 2 = one wave per output tile (k_fill_tiles), 3 = persistent waves taking
        tiles from ticket counters (k_fill_tickets: measured, never chosen -- it lives in diagnostic builds, `make DIAG=1`; the
        shipped library answers with the tile kernel), 4 = shape 1's kernel as a resident grid of three workgroups per CU with ticket counters.
        Graphs with hub rows always take the start-vertex kernel."""
        self._ck(self.lib.gnnpe_set_emit_shape(self.ctx, int(shape)))

    EMIT_SHAPES = (1, 2, 4)
    EMIT_SHAPE_NAMES = {1: "starts", 2: "tiles", 3: "tickets", 4: "starts_low"}
    EMIT_SHAPE_KERNELS = {1: "k_fill_ranked", 2: "k_fill_tiles", 3: "k_fill_tiles", 4: "k_fill_ranked"}

    def has_emit_shape(self, shape):
        return int(shape) in self.EMIT_SHAPES

    def emit_kernel_name(self):
        return self.lib.gnnpe_emit_kernel_name(self.ctx).decode()

    def emit_calibrate_device(self, dev_vids=None, dev_pde=None, rows_cap=None):
        """Times the emit shapes 1 (start-vertex waves, one-shot), 4 (the same kernel as a resident grid of three workgroups per CU) and 2 (output tiles) into these
        buffers and keeps the fastest for them: dict(starts_ms, starts_low_ms, tiles_ms, kept, kept_shape).  rows_cap = the
        buffers' capacity in rows (default: the tensors' first dimension)."""
        if rows_cap is None:
            t = dev_pde if dev_pde is not None else dev_vids
            rows_cap = int(t.shape[0]) if hasattr(t, "shape") else (1 << 62)
        ms, k = (C.c_float * 5)(), C.c_int()
        self._ck(self.lib.gnnpe_emit_calibrate_device(self.ctx, int(rows_cap), _dev(dev_vids), _dev(dev_pde), ms, C.byref(k)))
        return dict(starts_ms=ms[1], starts_low_ms=ms[4], tiles_ms=ms[2], kept=self.EMIT_SHAPE_NAMES[k.value], kept_shape=k.value)

    # halo exchange helpers
    def halo_need(self, slab_bounds, dev_ids, cap):
        b = _np(slab_bounds, np.uint32)
        counts = np.zeros(len(b) - 1, np.uint64)
        self._ck(self.lib.gnnpe_halo_need(self.ctx, len(b) - 1, _ptr(b, _u32p), _dev(dev_ids), int(cap),
                                          _ptr(counts, _u64p)))
        return counts

    def rows_degree(self, n_req, dev_ids, dev_deg):
        self._ck(self.lib.gnnpe_rows_degree(self.ctx, int(n_req), _dev(dev_ids), _dev(dev_deg)))

    def rows_pack(self, n_req, dev_ids, dev_out, cap):
        self._ck(self.lib.gnnpe_rows_pack(self.ctx, int(n_req), _dev(dev_ids), _dev(dev_out), int(cap)))

    def rows_append(self, n_rows, dev_ids, dev_deg, dev_nbrs, n_nbrs, min_rank=0):
        """min_rank > 0: entries ranked before that processing position are dropped (last-hop halo rows)."""
        self._ck(self.lib.gnnpe_rows_append(self.ctx, int(n_rows), _dev(dev_ids), _dev(dev_deg), _dev(dev_nbrs),
                                            int(n_nbrs), int(min_rank)))

    def rows_held(self):
        """(rows, neighbour entries, hub rows) on the device: own rows plus the installed (truncated) halo."""
        r, e, h = C.c_uint64(), C.c_uint64(), C.c_uint32()
        self._ck(self.lib.gnnpe_rows_held(self.ctx, C.byref(r), C.byref(e), C.byref(h)))
        return r.value, e.value, h.value

    def rows_drop_halo(self):
        self._ck(self.lib.gnnpe_rows_drop_halo(self.ctx))

    # R7: text rendering (main.cpp:98-119)
    def text_paths(self, n_rows, L, dev_vids, dev_text=None, cap=0):
        nb = C.c_uint64()
        self._ck(self.lib.gnnpe_text_paths(self.ctx, int(n_rows), int(L), _dev(dev_vids), _dev(dev_text), int(cap),
                                           C.byref(nb)))
        return nb.value

    def text_ids(self, n, dev_ids, dev_text=None, cap=0):
        nb = C.c_uint64()
        self._ck(self.lib.gnnpe_text_ids(self.ctx, int(n), _dev(dev_ids), _dev(dev_text), int(cap), C.byref(nb)))
        return nb.value

    def select_partition(self, n, dev_part, pid, id_base, dev_ids):
        cnt = C.c_uint64()
        self._ck(self.lib.gnnpe_select_partition(self.ctx, int(n), _dev(dev_part), int(pid), int(id_base),
                                                 _dev(dev_ids), C.byref(cnt)))
        return cnt.value

    # R6: index.dat (custom.h:235-257)
    def build_index_device(self, cnt, L, dev_vids):
        img, nb = _vp(), C.c_uint64()
        hdr = (C.c_int32 * 8)()
        self._ck(self.lib.gnnpe_build_index_device(self.ctx, int(cnt), int(L), _dev(dev_vids), C.byref(img),
                                                   C.byref(nb), hdr))
        return img.value, nb.value, list(hdr)

    def build_index_partition_device(self, pid):
        """index.dat image of partition pid straight from the enumeration state (pair-major build when eligible)."""
        img, nb = _vp(), C.c_uint64()
        hdr = (C.c_int32 * 8)()
        self._ck(self.lib.gnnpe_build_index_partition_device(self.ctx, int(pid), C.byref(img), C.byref(nb), hdr))
        return img.value, nb.value, list(hdr)

    def build_box_index_device(self, cnt, dim, dev_boxes):
        """cnt x 2*dim doubles (lo0, hi0, lo1, hi1, ...) -> same image format (GNN-PGE/include/custom.h:171-177)."""
        img, nb = _vp(), C.c_uint64()
        hdr = (C.c_int32 * 8)()
        self._ck(self.lib.gnnpe_build_box_index_device(self.ctx, int(cnt), int(dim), _dev(dev_boxes), C.byref(img),
                                                       C.byref(nb), hdr))
        return img.value, nb.value, list(hdr)

    def build_index(self, pid, path):
        self._ck(self.lib.gnnpe_build_index(self.ctx, int(pid), path.encode()))

    def build_index_files(self, paths, aux_paths=None):
        """index.dat (and optionally aux_index.bin) of partitions 0..len(paths)-1, the index files written side by side."""
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        aux = None if aux_paths is None else (C.c_char_p * len(aux_paths))(*[p.encode() for p in aux_paths])
        self._ck(self.lib.gnnpe_build_index_files(self.ctx, len(paths), arr, aux))

    def build_index_partition_aux_device(self, pid, fetch=False):
        """index.dat image of partition pid AND its auxiliary index in one build (leaf rows by the leaf kernel).  Returns
        (image ptr, nbytes, hdr, key ptr, degrees ptr, label_mbr ptr, n_nodes); fetch=True returns host arrays instead of the
        three pointers."""
        img, nb = _vp(), C.c_uint64()
        hdr = (C.c_int32 * 8)()
        k, d, m = _vp(), _vp(), _vp()
        N = C.c_uint32()
        self._ck(self.lib.gnnpe_build_index_partition_aux_device(self.ctx, int(pid), C.byref(img), C.byref(nb), hdr, C.byref(k),
                                                                 C.byref(d), C.byref(m), C.byref(N)))
        if not fetch:
            return img.value, nb.value, list(hdr), k.value, d.value, m.value, N.value
        L, D, n = self.l + 1, (self.l + 1) * self.e, N.value
        key, deg, mbr = np.zeros(n), np.zeros((n, L), np.uint32), np.zeros((n, 2 * D))
        for host, dev in ((key, k), (deg, d), (mbr, m)):
            if host.nbytes:
                self._ck(self.lib.gnnpe_copy_to_host(self.ctx, host.ctypes.data_as(_vp), dev, host.nbytes))
        return img.value, nb.value, list(hdr), key, deg, mbr, n

    def aux_index_device_ptrs(self, dev_image, nbytes, cnt, L, dev_tuples):
        """The same pass, results left on the device: (key ptr, degrees ptr, label_mbr ptr, n_nodes, D) -- context-owned
        arrays, valid until the next call."""
        k, d, m = _vp(), _vp(), _vp()
        N, D = C.c_uint32(), C.c_uint32()
        self._ck(self.lib.gnnpe_aux_index_device(self.ctx, _dev(dev_image), int(nbytes), int(cnt), int(L), _dev(dev_tuples),
                                                 C.byref(k), C.byref(d), C.byref(m), C.byref(N), C.byref(D)))
        return k.value, d.value, m.value, N.value, D.value

    def aux_index_device(self, dev_image, nbytes, cnt, L, dev_tuples):
        """Partition::build_auxiliary_index (custom.h:268-364) over an index.dat image in device memory; dev_tuples =
        the partition's paths [cnt x L] in partition order.  Returns host copies: key[N], degrees[N x L], label_mbr[N x 2D]."""
        k, d, m = _vp(), _vp(), _vp()
        N, D = C.c_uint32(), C.c_uint32()
        self._ck(self.lib.gnnpe_aux_index_device(self.ctx, _dev(dev_image), int(nbytes), int(cnt), int(L), _dev(dev_tuples),
                                                 C.byref(k), C.byref(d), C.byref(m), C.byref(N), C.byref(D)))
        n, dim = N.value, D.value
        return dict(key=self.copy_to_host(k.value, n * 8).view(np.float64),
                    degrees=self.copy_to_host(d.value, n * L * 4).view(np.uint32).reshape(n, L),
                    label_mbr=self.copy_to_host(m.value, n * 2 * dim * 8).view(np.float64).reshape(n, 2 * dim), L=int(L), D=dim)

    def build_aux_index(self, pid, path):
        self._ck(self.lib.gnnpe_build_aux_index(self.ctx, int(pid), path.encode()))

    def copy_to_host(self, dev_ptr, nbytes):
        out = np.zeros(nbytes, np.uint8)
        self._ck(self.lib.gnnpe_copy_to_host(self.ctx, out.ctypes.data_as(_vp), C.c_void_p(dev_ptr), int(nbytes)))
        return out

    # GNN-PGE offline (GNN-PGE/src/main.cpp:91-195)
    def pge_groups(self):
        pg = np.zeros((self.n, 4 * self.e))
        plg = np.zeros((self.n, 4 * self.e))
        self._ck(self.lib.gnnpe_pge_groups(self.ctx, _ptr(pg, _f64p), _ptr(plg, _f64p)))
        return pg, plg

    def pge_groups_device(self):
        """The same computation left on the device: (path_group, path_label_group) device addresses."""
        self._ck(self.lib.gnnpe_pge_groups(self.ctx, None, None))
        a, b = _vp(), _vp()
        self._ck(self.lib.gnnpe_pge_device_ptr(self.ctx, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def pge_build_index(self, vertices, path):
        v = _np(vertices, np.uint32)
        self._ck(self.lib.gnnpe_pge_build_index(self.ctx, len(v), _ptr(v, _u32p), path.encode()))
