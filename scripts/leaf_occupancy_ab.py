"""Index leaf kernel (k_pack_leaves_pairs, one leaf per wave, launch order) at BASELINE config 3 from the cached pair order: resident
workgroups per CU capped by dynamic LDS nobody touches (GNNPE_LEAF_LDS_PAD), same process, same image buffer.  usage: leaf_occupancy_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import _diag  # noqa: F401  (the diagnostic build: this script's knobs live there)
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
g = synth.gnm_graph(1_000_000, 10_000_000); sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
eng.count_paths(2)
eng.build_index_partition_device(0)
ref = None
for rnd in range(2):
    for pad in ("0", "4600", "10500", "16500", "24000", "37000"):
        os.environ["GNNPE_LEAF_LDS_PAD"] = pad
        ts = []
        for _ in range(4):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); img, nbytes, hdr = eng.build_index_partition_device(0); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        print(f"round {rnd} pad {pad:>6s}: image from the cached pair order min {min(ts):.3f} ms ({nbytes} bytes)", flush=True)
eng.close()
