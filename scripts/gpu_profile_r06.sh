#!/bin/bash
# Round-6 profile set (one gpurun call): kernel trace of the bench command (-> profiles/r06_kernel_stats.csv through
# scripts/summarize_trace_r06.py) and the fabric counters of the emit kernels (-> profiles/r06_pmc_fill.json through scripts/emit_pmc_json.py;
# the A/B script switches shapes by environment, so it runs on the diagnostic build: scripts/_diag.py).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r06}
rm -rf gpurun_out/${R}_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -- python3 bench.py --steps 10 --warmup 2 --compare-pool 0 --no-cpu-baseline --no-index --no-config5 > gpurun_out/${R}_trace.log 2>&1
echo "trace rc=$?"
tail -c 300 gpurun_out/${R}_trace.log
python3 scripts/summarize_trace_r06.py $R 10 2 > gpurun_out/${R}_kernel_stats_print.txt 2>&1; head -30 gpurun_out/${R}_kernel_stats_print.txt
EMIT_AB3_ARGS="--shapes starts,tiles" bash scripts/emit_pmc3.sh 1 2 > gpurun_out/${R}_emit_pmc.txt 2>&1
echo "pmc rc=$?"; tail -20 gpurun_out/${R}_emit_pmc.txt
