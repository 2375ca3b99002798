#!/bin/bash
# Round 6, the count phase: kernel times and fabric requests of the step's count side at config 3, for the shipped
# library and for the diagnostic build with 8-byte pair records (GNNPE_ROWS_PAIR8=1).  Output -> profiles/r06_rows_probe.txt.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {  # tag, env...
  tag=$1; shift
  echo "## $tag"
  env "$@" python3 scripts/count_phase_r06.py 30 || return 1
  rm -rf gpurun_out/cp_trace_$tag
  ( export "$@"; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cp_trace_$tag -- python3 scripts/count_phase_r06.py 10 > gpurun_out/cp_trace_$tag.log 2>&1 )
  python3 - "$tag" <<'PY'
import csv, glob, re, sys
per = {}
for f in glob.glob(f"gpurun_out/cp_trace_{sys.argv[1]}/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void gnnpe::", "")
        per.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 10:
        print(f"  {k[:70]:70s} launches {len(v):3d}  median {sorted(v)[len(v) // 2]:.4f} ms")
PY
  i=0
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i+1))
    rm -rf gpurun_out/cp_pmc_${tag}_$i
    ( export "$@"; timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/cp_pmc_${tag}_$i -- python3 scripts/count_phase_r06.py 2 > gpurun_out/cp_pmc_${tag}_$i.log 2>&1 )
    python3 - "$tag" "$i" <<'PY'
import csv, glob, re, sys
per = {}
for f in glob.glob(f"gpurun_out/cp_pmc_{sys.argv[1]}_{sys.argv[2]}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_rows_rank_multi" in n or "k_start_scan" in n or "k_vde<" in n:
            k = re.search(r"k_[a-z_]+", n).group(0)
            per.setdefault((k, r["Counter_Name"]), {}).setdefault(r["Dispatch_Id"], 0.0)
            per[(k, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
for (k, c), v in sorted(per.items()):
    vals = list(v.values())
    print(f"  {k:20s} {c:28s} mean {sum(vals) / len(vals):.4g} per launch ({len(vals)} launches)")
PY
  done
}
run shipped GNNPE_X=0
run diag GNNPE_LIB_PATH=$GRAFT_REPO_ROOT/gnn-pe_amd/libgnnpe_hip_diag.so
run diag_pair8 GNNPE_LIB_PATH=$GRAFT_REPO_ROOT/gnn-pe_amd/libgnnpe_hip_diag.so GNNPE_ROWS_PAIR8=1
