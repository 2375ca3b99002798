"""End-to-end `gnnpe_main -m offline` at a BASELINE config on the GPU box: generate the .graph +
membership.txt, run the CLI with --timing (and optionally --index), report sizes and md5 of outputs."""
import hashlib, json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gnnpe_amd
from gnnpe_amd import synth
n, m, p = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
extra = sys.argv[4:]
root = tempfile.mkdtemp(prefix="gnnpe_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
t0 = time.time()
g = synth.gnm_graph(n, m)
sn = synth.degree_order(g["offsets"])
gp = os.path.join(root, "g.graph")
synth.write_graph_file(gp, g)
synth.make_dataset_dir(root, p)
synth.write_membership(os.path.join(root, "gnn-pe", "membership.txt"), sn, synth.block_membership(n, p))
t_gen = time.time() - t0
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnn-pe_amd", os.environ.get("GNNPE_CLI", "gnnpe_main"))
t0 = time.time()
r = subprocess.run([cli, "-f", root + "/", "-d", gp, "-m", "offline", "-p", str(p), "--timing"] + extra, capture_output=True, text=True)
wall = time.time() - t0
print("rc", r.returncode, "wall_s", round(wall, 2), "gen_s", round(t_gen, 1))
print(r.stdout.strip())
print(r.stderr.strip()[-1500:])
ap = os.path.join(root, "gnn-pe", "all_paths.txt")
if os.path.exists(ap):
    print("all_paths bytes", os.path.getsize(ap), "header", open(ap).readline().strip(), "expected", synth.expected_paths_l2(g["offsets"]))
    for i in range(p):
        f = os.path.join(root, "gnn-pe", "partitions", f"partition-{i}", "index.dat")
        if os.path.exists(f):
            print("index", i, os.path.getsize(f))
subprocess.run(["rm", "-rf", root])
