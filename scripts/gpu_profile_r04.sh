#!/bin/bash
# Round-4 profile set (one gpurun call): kernel trace of the bench command, counters of both emit kernels, the count kernel's
# counters, the index legs' trace.  Summaries: scripts/summarize_trace_r04.py, scripts/emit_pmc_json.py, scripts/trace_table.py.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r04}
rm -rf gpurun_out/${R}_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -- python3 bench.py --steps 10 --warmup 2 --compare-pool 0 --no-cpu-baseline --no-index --no-config5 > gpurun_out/${R}_trace.log 2>&1
echo "trace rc=$?"
tail -c 600 gpurun_out/${R}_trace.log
