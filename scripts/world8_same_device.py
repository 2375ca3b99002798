"""First-SCALE dry run of BOTH multi-GPU drivers at config 4 on ONE device.

  (1) `gnnpe_main -m offline -p 8`                       one context, the reference's files (md5 of all_paths.txt + the 8 partition_paths.txt)
  (2) `gnnpe_main --gpus 8 --same-device`                eight rank threads, eight contexts on device 0, exchanges as device copies
                                                         -> the same md5s; per-rank halo statistics from --timing
  (3) `python bench.py --gpus R` (its own launcher, GNNPE_BENCH_SAME_DEVICE=1: R rank PROCESSES on device 0, collectives staged over
      gloo) for R in 2, 4 -- a GPU box admits at most 6 processes on its card, so the 8-rank bench cannot run here; the 8-rank slab
      code itself is (2) and tests/test_gpu_slabs_full.py (threads).
Prints everything next to the expectation of DESIGN.md section 6.   usage: world8_same_device.py [n m]"""
import hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gnnpe_amd
from gnnpe_amd import synth

n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 10_000_000)
p = 8
cli = os.path.join(ROOT, "gnn-pe_amd", "gnnpe_main")
root = tempfile.mkdtemp(prefix="gnnpe_w8_", dir=os.environ.get("TMPDIR", "/tmp"))
g = synth.gnm_graph(n, m)
sn = synth.degree_order(g["offsets"])
gp = os.path.join(root, "g.graph")
synth.write_graph_file(gp, g)
mem = synth.block_membership(n, p)
print(f"# config 4: G(n={n}, m={m}), l=2, e=2, p={p}; {synth.expected_paths_l2(g['offsets'])} paths", flush=True)


def md5s(d):
    out = {}
    for rel in ["gnn-pe/all_paths.txt"] + [f"gnn-pe/partitions/partition-{i}/partition_paths.txt" for i in range(p)]:
        h = hashlib.md5()
        with open(os.path.join(d, rel), "rb") as f:
            for blk in iter(lambda: f.read(1 << 24), b""):
                h.update(blk)
        out[rel] = h.hexdigest()
    return out


res = {}
for name, extra in (("one_context", []), ("gpus8_same_device", ["--gpus", "8", "--same-device"])):
    d = os.path.join(root, name)
    os.makedirs(d)
    synth.make_dataset_dir(d, p)
    synth.write_membership(os.path.join(d, "gnn-pe", "membership.txt"), sn, mem)
    t0 = time.time()
    r = subprocess.run([cli, "-f", d + "/", "-d", gp, "-m", "offline", "-p", str(p), "--timing"] + extra, capture_output=True, text=True)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    t = json.loads(r.stderr.strip().splitlines()[-1])
    res[name] = (md5s(d), t, wall)
    print(f"{name}: wall {wall:.2f} s, end_to_end_s {t['end_to_end_s']}, paths {t['paths']}", flush=True)
    subprocess.run(["rm", "-rf", d])
same = res["one_context"][0] == res["gpus8_same_device"][0]
print("md5 of all_paths.txt and the 8 partition_paths.txt: --gpus 8 --same-device ==", "one context" if same else "DIFFERENT", flush=True)
for k, v in res["one_context"][0].items():
    print("   ", v, k)
assert same
t8 = res["gpus8_same_device"][1]
print("per-rank statistics of --gpus 8 (DESIGN.md section 6 table: rank 0 holds 15.0 M entries, rank 3 13.5 M, rank 7 5.4 M):")
for rk in t8.get("ranks", []):
    print("   ", json.dumps(rk))

print("\n# bench.py --gpus R started plainly (its own launcher), R rank processes on device 0, collectives staged over gloo")
expect = {2: "2.2-2.5", 4: "1.4-1.6", 8: "1.1-1.3"}
for R in (2, 4):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GNNPE_BENCH_SAME_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(R), "--vertices", str(n), "--edges", str(m), "--steps", "5",
                        "--warmup", "2", "--no-index", "--no-cpu-baseline", "--no-config5"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    dline = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(f"R={R}: n_gpus {dline['n_gpus']}, paths {dline['config']['paths']}, sanity: {dline['sanity']}; ms_per_step {dline['ms_per_step']:.3f} "
          f"(ranks SHARE one device and stage through the host: not a rate; expected on {R} devices over RCCL: {expect[R]} ms), "
          f"expected_ms_per_step field: {dline.get('expected_ms_per_step')}")
    print("    rank 0:", json.dumps({k: dline['halo'][k] for k in ('owned_entries', 'halo_rows', 'halo_entries', 'held_entries', 'local_paths', 'slab') if k in dline['halo']}),
          "per-step phases:", json.dumps(dline["phases_ms"]["per_step"]))
subprocess.run(["rm", "-rf", root])
