"""The count kernel's pieces on their own (GNNPE_ROWS_PROBE=1|2|3, diagnostic kernels k_rows_rank_probe): run under
rocprofv3 by scripts/rows_probe.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _diag  # noqa: F401  (the diagnostic build: this script's knobs live there)
import gnnpe_amd
from gnnpe_amd import binding, synth
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
eng = binding.Engine(0)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
want = synth.expected_paths_l2(g["offsets"])
assert eng.count_paths(2) == want
for mode in ("1", "2", "3"):
    os.environ["GNNPE_ROWS_PROBE"] = mode
    assert eng.count_paths(2) == want  # the real kernel still runs; the probe launches follow it
os.environ.pop("GNNPE_ROWS_PROBE")
eng.close()
