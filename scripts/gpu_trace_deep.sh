#!/bin/bash
# Kernel trace of the l = 3 path on the config-5 graph (power-law 4M / 64M, e = 8): count kernels and k_deep3 over the first
# 2^26 paths in four chunks (scripts/bench_deep.py).  Summary -> profiles/<round>_deep_kernel_stats.csv
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r03}
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_deep_trace -- python3 scripts/bench_deep.py --vertices 4000000 --edges 64000000 --powerlaw --max-degree 3000 --embedding 8 --max-paths 67108864 --chunk 16777216 > gpurun_out/${R}_deep_trace.log 2>&1
echo "deep trace rc=$?"
tail -1 gpurun_out/${R}_deep_trace.log | cut -c1-600
