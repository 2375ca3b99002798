"""Online query time of the UNTOUCHED reference binary with (a) the index.dat it builds itself and (b) the
bulk-loaded index.dat written by gnnpe_main --index, on the reference's sample graph (runs on the GPU box)."""
import os, re, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gnnpe_amd
from gnnpe_amd import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_main")
CLI = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")
graph = os.path.join(ROOT, "tests", "golden", "test_graph", "data_graph.graph")
query = os.path.join(ROOT, "tests", "golden", "test_graph", "query_graph.graph")
deg = np.array([int(l.split()[3]) for l in open(graph) if l.startswith("v")])
sn = np.argsort(deg, kind="stable").astype(np.uint32)
for p in (1, 2):
    res = {}
    for who in ("reference-built", "bulk-loaded"):
        d = tempfile.mkdtemp()
        synth.make_dataset_dir(d, p)
        synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), sn, (np.arange(len(deg)) % p).astype(np.uint32))
        if who == "reference-built":
            subprocess.check_call([REF, "-f", d + "/", "-d", graph, "-m", "offline", "-p", str(p)], stdout=subprocess.DEVNULL)
            t0 = time.time()
            subprocess.check_output([REF, "-f", d + "/", "-d", graph, "-q", query, "-m", "online", "-p", str(p)])
            build = time.time() - t0
        else:
            t0 = time.time()
            subprocess.check_call([CLI, "-f", d + "/", "-d", graph, "-p", str(p), "--index"], stdout=subprocess.DEVNULL)
            build = time.time() - t0
        times = []
        for _ in range(3):
            out = subprocess.check_output([REF, "-f", d + "/", "-d", graph, "-q", query, "-m", "online", "-p", str(p)], text=True)
            m = re.search(r"Answer Number: (\d+) Query Time \(ms\): ([0-9.e+-]+)", out)
            times.append(float(m.group(2)))
        size = sum(os.path.getsize(os.path.join(d, "gnn-pe", "partitions", f"partition-{i}", "index.dat")) for i in range(p))
        res[who] = (int(m.group(1)), min(times), size, build)
        subprocess.run(["rm", "-rf", d])
    print(f"p={p}:", {k: dict(answer=v[0], query_ms=round(v[1], 2), index_bytes=v[2], first_run_or_build_s=round(v[3], 2)) for k, v in res.items()})
