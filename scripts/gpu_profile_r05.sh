#!/bin/bash
# Round-5 profile set (one gpurun call): kernel trace of the bench command (-> profiles/r05_kernel_stats.csv through
# scripts/summarize_trace_r05.py) and the fabric counters of the emit kernels (-> profiles/r05_pmc_fill.json through scripts/emit_pmc_json.py).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r05}
rm -rf gpurun_out/${R}_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -- python3 bench.py --steps 10 --warmup 2 --compare-pool 0 --no-cpu-baseline --no-index --no-config5 > gpurun_out/${R}_trace.log 2>&1
echo "trace rc=$?"
tail -c 400 gpurun_out/${R}_trace.log
EMIT_AB3_ARGS="--shapes starts,tiles,tickets" bash scripts/emit_pmc3.sh 1 2 > gpurun_out/${R}_emit_pmc.txt 2>&1
echo "pmc rc=$?"
