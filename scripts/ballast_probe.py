"""Which part of the device memory is the slow class?  Fresh process per case: a ballast of K GB is allocated first and KEPT, then the 2.4 + 9.6 GB
output buffers; the emit kernel (config 3) is timed into them.   python scripts/ballast_probe.py [rounds] [K ...]"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 2 and sys.argv[1] == "case":
    K = float(sys.argv[2])
    sys.path.insert(0, os.path.dirname(HERE))
    import numpy as np, torch
    import gnnpe_amd
    from gnnpe_amd import binding, synth
    dev = torch.device("cuda:0")
    ballast = None
    if K > 0 and len(sys.argv) > 3 and sys.argv[3] == "early":
        ballast = torch.empty(int(K * (1 << 30)), dtype=torch.uint8, device=dev); ballast[::4096] = 1
    g = synth.gnm_graph(1_000_000, 10_000_000)
    sn = synth.degree_order(g["offsets"])
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    eng = binding.Engine(0, stream=stream.cuda_stream)
    eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(g["n"], np.uint32), 1)
    eng.set_label_table(binding.host_label_table(64, 2)); eng.vde(want=False)
    total = eng.count_paths(2)
    if K > 0 and ballast is None:
        ballast = torch.empty(int(K * (1 << 30)), dtype=torch.uint8, device=dev); ballast[::4096] = 1
    ids = torch.empty((total, 3), dtype=torch.int32, device=dev); pde = torch.empty((total, 6), dtype=torch.float64, device=dev)
    eng.fill_paths_device(0, total, ids, pde, None); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); eng.fill_paths_device(0, total, ids, pde, None); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"ballast {K:5.1f} GB {'(before the engine)' if len(sys.argv) > 3 else '(before the outputs)':21s} emit {min(ts):.3f} ms = {total * 92 / min(ts) / 1e-3 / 8e12:.3f}   ids @ {ids.data_ptr():#x} pde @ {pde.data_ptr():#x}", flush=True)
    eng.close()
else:
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ks = sys.argv[2:] or ["0", "1", "4", "8", "16", "32"]
    for rnd in range(rounds):
        for k in ks:
            subprocess.call([sys.executable, os.path.abspath(__file__), "case"] + k.split(":"))
