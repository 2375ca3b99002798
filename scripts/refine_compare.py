"""Device vs host refinement (gnnpe_refine vs gnnpe_host_refine) on queries cut out of a synthetic graph:
    python scripts/refine_compare.py <vertices> <edges> <labels> <query vertices>"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import gnnpe_amd
from gnnpe_amd import synth, binding
from make_golden_online import cut_query
n, m, nl, qs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
g = synth.gnm_graph(n, m, n_labels=nl)
sn = synth.degree_order(g["offsets"])
eng = binding.Engine(0)
eng.load_csr(g["offsets"], g["nbrs"], g["labels"]); eng.set_order(sn, np.zeros(n, np.uint32), 1)
eng.set_label_table(binding.host_label_table(nl, 2)); eng.vde(want=False); eng.count_paths(2)
rng = np.random.default_rng(3)
wd = tempfile.mkdtemp()
for k in range(3):
    qp = os.path.join(wd, f"q{k}.graph")
    open(qp, "w").write(cut_query(g["offsets"].astype(np.int64), g["nbrs"], g["labels"], qs, rng))
    plan = binding.host_query_plan(qp, 2)
    bm, fms = eng.filter_candidates(plan)
    t0 = time.time(); a_dev, dms = eng.refine(qp, bm); t_dev = time.time() - t0
    t0 = time.time(); a_host = binding.host_refine(g, qp, bm); t_host = time.time() - t0
    print(f"labels {nl} query {qs}v: filter {fms:.3f} ms; answers dev {a_dev} host {a_host}; refine device {dms:.2f} ms (call {t_dev*1e3:.2f}), host {t_host*1e3:.2f} ms")
