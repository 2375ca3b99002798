"""Leaf kernel of the pair-major index build at config 3: workgroup -> leaf group mapping by XCD (GNNPE_LEAF_XCD_CHUNK, read at
every build).  0 = launch order (neighbouring workgroups, which sit on different XCDs, take neighbouring leaves); C > 0 = the
workgroup on XCD x takes the x-th run of C consecutive leaf groups of every 8 C.  Same process, same buffers, builds from the
cached pair order, cases interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, hashlib
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _diag  # noqa: F401  (the diagnostic build: this script's knobs live there)
import gnnpe_amd
from gnnpe_amd import binding, synth
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 10_000_000)
g = synth.gnm_graph(n, m)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
eng.count_paths(2)
chunks = [0, 1, 4, 16, 64, 1024, 16384, 80000]
def timed(build):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); build(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for case, build in (("image", lambda: eng.build_index_partition_device(0)), ("image + aux", lambda: eng.build_index_partition_aux_device(0))):
    build(); build()
    res = {c: [] for c in chunks}
    for rnd in range(4):
        for c in chunks:
            os.environ["GNNPE_LEAF_XCD_CHUNK"] = str(c)
            timed(build)
            res[c].append(min(timed(build) for _ in range(3)))
    for c in chunks:
        print(f"[{case:12s}] chunk {c:6d}: " + " ".join(f"{t:.3f}" for t in res[c]) + f"  min {min(res[c]):.3f} ms", flush=True)
os.environ["GNNPE_LEAF_XCD_CHUNK"] = "0"
eng.close()
