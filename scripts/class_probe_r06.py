#!/usr/bin/env python3
"""Round 6, one more bounded probe of the allocation classes: the emit kernel timed into the SAME PHYSICAL PAGES
  (A1) mapped at one virtual address in process A,
  (A2) unmapped and mapped again at a second virtual address in process A,
  (B)  imported through a POSIX file descriptor and mapped in a SECOND process B,
next to a plain allocation of each process.  hipMemCreate / hipMemExportToShareableHandle / hipMemAddressReserve / hipMemMap through
ctypes on the HIP runtime torch has loaded; the fill is the product's (gnnpe_fill_paths_device at config 3, shapes 1 / 4 / 2 as the
calibration times them).  What round 3 left open (profiles/r03_buffer_classes.txt): the same chunks at another address changed class,
two chunk sets at one address did not -- but a 12 GB handle kept one rate through twelve windows of one reservation, "fast in one
process and slow in the next".  This probe asks the per-process half directly.

usage: python scripts/class_probe_r06.py            launcher (never touches the GPU): starts A and B joined by a socket pair
Output -> profiles/r06_class_probe.txt."""
import ctypes as C
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def launcher():
    a, b = socket.socketpair(socket.AF_UNIX, socket.SOCK_STREAM)
    for s in (a, b):
        os.set_inheritable(s.fileno(), True)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for rnd in range(rounds):  # fresh processes per round: the class is said to be drawn per process
        print(f"## round {rnd}", flush=True)
        pa = subprocess.Popen([sys.executable, os.path.abspath(__file__), "A", str(a.fileno())], pass_fds=[a.fileno()])
        pb = subprocess.Popen([sys.executable, os.path.abspath(__file__), "B", str(b.fileno())], pass_fds=[b.fileno()])
        ra, rb = pa.wait(), pb.wait()
        if ra or rb:
            raise SystemExit(f"role A exit {ra}, role B exit {rb}")


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class Flags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", C.c_void_p),
                ("allocFlags", Flags)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


def role(which, sock_fd):
    import numpy as np
    import torch
    import gnnpe_amd  # noqa: F401
    from gnnpe_amd import binding, synth
    sock = socket.socket(fileno=sock_fd)
    # the HIP runtime torch brought into the process (loaded privately by its extension modules: found by path)
    paths = {ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln}
    hip = C.CDLL(sorted(paths)[0])
    for name in ("hipMemCreate", "hipMemAddressReserve", "hipMemMap", "hipMemSetAccess", "hipMemUnmap", "hipMemExportToShareableHandle",
                 "hipMemImportFromShareableHandle", "hipMemGetAllocationGranularity", "hipMemRelease", "hipMemAddressFree"):
        getattr(hip, name).restype = C.c_int

    def ck(rc, what):
        if rc != 0:
            raise SystemExit(f"{which}: {what} failed with hipError {rc}")

    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    g = synth.gnm_graph(1_000_000, 10_000_000)
    sn = synth.degree_order(g["offsets"])
    eng = binding.Engine(0, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"])
    eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(64, 2))
    eng.vde(want=False)
    total = eng.count_paths(2)
    MiB2 = 2 << 20
    pde_bytes = (total * 48 + MiB2 - 1) // MiB2 * MiB2
    ids_bytes = (total * 12 + MiB2 - 1) // MiB2 * MiB2
    prop = Prop(1, 1, Loc(1, 0), None, Flags(0, 0, 0))  # pinned device memory of device 0, exportable as a POSIX fd
    gran = C.c_size_t()
    ck(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1), "hipMemGetAllocationGranularity")
    size = (pde_bytes + ids_bytes + gran.value - 1) // gran.value * gran.value

    def time_fill(base, label):
        out = []
        for shape, name in ((1, "starts x5"), (4, "starts x3"), (2, "tiles")):
            eng.set_emit_shape(shape)
            ts = []
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.fill_paths_device(0, total, base + pde_bytes, base, None)
                eng.sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            out.append(f"{name} {min(ts[1:]):.3f}")
        print(f"{which} {label:46s} at {base:#x}: " + "  ".join(out) + " ms", flush=True)

    def map_at(handle, reserve_extra=0):
        ptr = C.c_void_p()
        ck(hip.hipMemAddressReserve(C.byref(ptr), C.c_size_t(size + reserve_extra), C.c_size_t(0), None, C.c_ulonglong(0)), "hipMemAddressReserve")
        ck(hip.hipMemMap(ptr, C.c_size_t(size), C.c_size_t(0), handle, C.c_ulonglong(0)), "hipMemMap")
        acc = Access(Loc(1, 0), 3)
        ck(hip.hipMemSetAccess(ptr, C.c_size_t(size), C.byref(acc), C.c_size_t(1)), "hipMemSetAccess")
        return ptr.value

    handle = C.c_void_p()
    keep = []
    if which == "A":
        ck(hip.hipMemCreate(C.byref(handle), C.c_size_t(size), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
        fd = C.c_int(-1)
        ck(hip.hipMemExportToShareableHandle(C.byref(fd), handle, 1, C.c_ulonglong(0)), "hipMemExportToShareableHandle")
        va1 = map_at(handle)
        time_fill(va1, "shared pages, first mapping")
        ck(hip.hipMemUnmap(C.c_void_p(va1), C.c_size_t(size)), "hipMemUnmap")
        va2 = map_at(handle, reserve_extra=1 << 30)  # (va1 stays reserved, so this is another address)
        time_fill(va2, "shared pages, second mapping (same process)")
        plain = torch.empty(size, dtype=torch.uint8, device=dev)
        time_fill(plain.data_ptr(), "plain allocation of this process")
        for k in range(2):  # more pages of the same kind: another handle, created and mapped here
            h2 = C.c_void_p()
            ck(hip.hipMemCreate(C.byref(h2), C.c_size_t(size), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
            time_fill(map_at(h2), f"own pages {k} (hipMemCreate + hipMemMap)")
        for k in range(2):
            more = torch.empty(size, dtype=torch.uint8, device=dev)
            time_fill(more.data_ptr(), f"another plain allocation {k}")
            keep.append(more)
        torch.cuda.synchronize()
        socket.send_fds(sock, [str(size).encode()], [fd.value])  # B measures while A is idle
        sock.recv(16)
        time_fill(va2, "shared pages, second mapping, after B")
    else:
        msg, fds, _, _ = socket.recv_fds(sock, 64, 1)
        assert int(msg.decode()) == size, (msg, size)
        # (the header does not say whether osHandle is the descriptor or points at it: the pointer form first -- an implementation that
        # wants the value answers it with an error instead of dereferencing a small integer)
        fdv = C.c_int(fds[0])
        rc = hip.hipMemImportFromShareableHandle(C.byref(handle), C.byref(fdv), 1)
        if rc != 0:
            rc = hip.hipMemImportFromShareableHandle(C.byref(handle), C.c_void_p(fds[0]), 1)
        ck(rc, "hipMemImportFromShareableHandle")
        va = map_at(handle)
        time_fill(va, "shared pages, imported into a second process")
        plain = torch.empty(size, dtype=torch.uint8, device=dev)
        time_fill(plain.data_ptr(), "plain allocation of this process")
        h2 = C.c_void_p()
        ck(hip.hipMemCreate(C.byref(h2), C.c_size_t(size), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
        time_fill(map_at(h2), "own pages (hipMemCreate + hipMemMap)")
        torch.cuda.synchronize()
        sock.send(b"done")
    eng.close()
    sock.detach()


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] in ("A", "B"):
        role(sys.argv[1], int(sys.argv[2]))
    else:
        launcher()
