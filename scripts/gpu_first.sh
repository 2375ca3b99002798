#!/bin/bash
# first GPU session: parity tests, smoke, bench, kernel-trace profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
timeout 600 python bench.py --steps 10 --warmup 2 --fill-variant 1 --no-cpu-baseline > gpurun_out/bench_v1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
tail -5 gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench.log gpurun_out/bench_v1.log gpurun_out/bench_prof.log
