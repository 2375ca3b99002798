cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_deep
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_deep -- python3 scripts/deep_emit_profile.py 26 > gpurun_out/r06_deep_emit.log 2>&1
tail -8 gpurun_out/r06_deep_emit.log
python3 - <<'PY'
import csv, glob, re
rows = []
for f in glob.glob("gpurun_out/prof_deep/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void gnnpe::", "")))
rows.sort()
# the last 12 fill calls: each = k_deep_unit_range ... k_deep3_slices<8,true>
calls, cur = [], []
for s, e, n in rows:
    if n.startswith("k_deep_unit_range"):
        if cur: calls.append(cur)
        cur = []
    cur.append((s, e, n))
if cur: calls.append(cur)
out = open("gpurun_out/r06_deep_emit_calls.txt", "w")
for c in calls[-12:]:
    t0 = c[0][0]
    line = "call: " + "  ".join(f"{n[:28]} {(e - s) / 1e6:.3f}@{(s - t0) / 1e6:.3f}" for s, e, n in c if "k_deep" in n or "Scan" in n or "scan" in n) + f"  | span {(c[-1][1] - t0) / 1e6:.3f} ms"
    print(line); out.write(line + "\n")
PY
