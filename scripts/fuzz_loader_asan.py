"""Malformed graph / membership files against a sanitizer build of the host side (no GPU needed: the loader and the input checks run before
the first device call).  Build:  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
-o /tmp/gnnpe_main_asan host/main.cpp host/slab_offline.cpp host/graph_loader.cpp -L. -lgnnpe_hip -ldl -Wl,-rpath,$PWD   (in gnn-pe_amd/).
    python scripts/fuzz_loader_asan.py /tmp/gnnpe_main_asan [cases]
Passes when no run dies of a signal and none prints a sanitizer report; exit codes of rejected inputs are the loader's own."""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cli = sys.argv[1]
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
src = open(os.path.join(ROOT, "tests", "golden", "test_graph", "data_graph.graph")).read().split("\n")
rng = random.Random(7)
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
bad = 0
def mutate(lines):
    lines = list(lines)
    k = rng.randrange(10)
    i = rng.randrange(len(lines))
    if k == 0: lines = lines[:i]                                   # truncated file
    elif k == 1: lines[i] = lines[i][: rng.randrange(len(lines[i]) + 1)]  # truncated line
    elif k == 2: lines[i] = lines[i].replace(" ", "  \t", 1)       # odd whitespace
    elif k == 3: lines[i] = "e 99999999999999999999 1"             # overflowing id
    elif k == 4: lines[i] = "e -5 3"                               # negative id
    elif k == 5: lines[i] = "v x y z"                              # not numbers
    elif k == 6: lines.insert(i, lines[i])                         # duplicate line
    elif k == 7: lines[i] = "e 7 7"                                # self loop
    elif k == 8: lines[0] = "t 5 99999999"                         # header lies
    else: lines[i] = "".join(chr(rng.randrange(1, 255)) for _ in range(rng.randrange(1, 40)))  # bytes
    return lines
with tempfile.TemporaryDirectory() as d:
    os.makedirs(os.path.join(d, "gnn-pe"))
    for c in range(cases):
        lines = src
        for _ in range(rng.randrange(1, 4)): lines = mutate(lines)
        gp = os.path.join(d, "g.graph")
        open(gp, "w", errors="surrogateescape").write("\n".join(lines))
        mem = os.path.join(d, "gnn-pe", "membership.txt")
        if rng.random() < 0.5:
            open(mem, "w").write("\n".join(str(rng.randrange(-3, 5000)) for _ in range(rng.randrange(0, 4000))))
        elif os.path.exists(mem):
            os.remove(mem)
        r = subprocess.run([cli, "-d", gp, "-f", d + "/", "-m", "offline", "-p", str(rng.choice([1, 2, 8]))], capture_output=True, env=env, timeout=120)
        err = r.stderr.decode(errors="replace")
        if r.returncode < 0 or "Sanitizer" in err or "runtime error" in err:
            bad += 1
            print(f"case {c}: rc {r.returncode}\n{err[-1500:]}")
print(f"{cases} malformed inputs, {bad} sanitizer findings")
sys.exit(1 if bad else 0)
