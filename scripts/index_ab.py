"""Same-process timing of the pair-major index build at config 3 (same image allocation for every call).  The cases
are environment settings read per call; the knobs this was written for (strip width, occupancy, grid, knock-outs of the
leaf kernel's reads / stores / MBR) were temporary and are gone from the library -- DESIGN.md section 3 has the results."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import binding, synth
cases = sys.argv[1:] or ["", "GNNPE_LEAF_STRIP_ALL=1"]
g = synth.gnm_graph(1_000_000, 10_000_000)
sn = synth.degree_order(g["offsets"])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = binding.Engine(0, stream=stream.cuda_stream)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
eng.count_paths(2)
KEYS = ["GNNPE_LEAF_STRIP_ALL", "GNNPE_LEAF_DYNLDS", "GNNPE_LEAF_GRID", "GNNPE_LEAF_KNOCK", "GNNPE_LEAF_EXP"]
sums = {}
for rnd in range(3):
    for case in cases:
        for k in KEYS: os.environ.pop(k, None)
        for kv in filter(None, case.split(",")):
            k, v = kv.split("="); os.environ[k] = v
        img, nb, hdr = eng.build_index_partition_device(0)
        ts = []
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); img, nb, hdr = eng.build_index_partition_device(0); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        if rnd == 0:
            head = eng.copy_to_host(img, 64 << 20)
            sums[case] = hashlib.md5(head.tobytes()).hexdigest()
        print(f"round {rnd} [{case}]: further-partition build min {min(ts):.3f} median {sorted(ts)[2]:.3f} ms  ({nb/1e9:.2f} GB)", flush=True)
print("first 64 MiB identical across cases:", len(set(sums.values())) == 1)
eng.close()
