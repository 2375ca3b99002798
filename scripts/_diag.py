"""The A/B scripts in this directory switch kernels and knobs by environment variable between two launches of one process.  The shipped
library reads its (eight) variables once per context and does not contain the A/B knobs at all (round 6); the diagnostic build does both:
`make -C gnn-pe_amd DIAG=1 diag` -> gnn-pe_amd/libgnnpe_hip_diag.so, which gnnpe_amd.binding loads when GNNPE_LIB_PATH names it.
`import _diag` FIRST in a script that needs it (before gnnpe_amd.binding is imported)."""
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(_ROOT, "gnn-pe_amd", "libgnnpe_hip_diag.so")
if "GNNPE_LIB_PATH" not in os.environ:
    if not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "gnn-pe_amd"), "DIAG=1", "-j", "8", "diag"])
    os.environ["GNNPE_LIB_PATH"] = _LIB
