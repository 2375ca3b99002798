"""gnn-pe_amd -- MI355X-native offline path-embedding engine for GNN-PE.

Holds only what the hot path needs:
  csrc/     hand-written HIP kernels for gfx950 + the C-ABI (include/gnnpe_hip.h)
  host/     C++17 host side: `main -m offline` CLI mirror, loader, writers
  binding   ctypes view of the C-ABI for bench.py / tests (no torch types cross the boundary)
  synth     deterministic synthetic graph generator + prep-step file layout
  dist      one-process-per-GPU driver (slab partition, halo exchange over torch.distributed)

Importing the package does not load the HIP library; `binding.load()` does, and raises if the
library is missing -- there is no CPU fallback.
"""
from . import synth  # noqa: F401

__all__ = ["synth", "binding", "dist"]
