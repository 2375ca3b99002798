"""End-to-end `gnnpe_main -m online` at a BASELINE config on the GPU box: generate the .graph + membership.txt and a
connected query cut out of the data graph, answer it (GPU filter + host refinement), report the timing JSON."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import gnnpe_amd
from gnnpe_amd import synth
from make_golden_online import cut_query
n, m, qsize = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
extra = sys.argv[4:]
root = tempfile.mkdtemp(prefix="gnnpe_onl_", dir=os.environ.get("TMPDIR", "/tmp"))
g = synth.gnm_graph(n, m)
gp = os.path.join(root, "g.graph")
synth.write_graph_file(gp, g)
synth.make_dataset_dir(root, 1)
synth.write_membership(os.path.join(root, "gnn-pe", "membership.txt"), synth.degree_order(g["offsets"]), np.zeros(n, np.uint32))
rng = np.random.default_rng(5)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnn-pe_amd", "gnnpe_main")
for k in range(3):
    qp = os.path.join(root, f"q{k}.graph")
    open(qp, "w").write(cut_query(g["offsets"].astype(np.int64), g["nbrs"], g["labels"], qsize, rng))
    t0 = time.time()
    r = subprocess.run([cli, "-f", root + "/", "-d", gp, "-q", qp, "-m", "online", "-p", "1", "--timing"] + extra,
                       capture_output=True, text=True)
    print("rc", r.returncode, "wall_s", round(time.time() - t0, 2), r.stdout.strip().splitlines()[-1], "|", r.stderr.strip()[-300:])
subprocess.run(["rm", "-rf", root])
