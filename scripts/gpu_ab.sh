#!/bin/bash
# A/B session: parity tests, then bench variants (with and without pde), profile of the default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
for v in ${VARIANTS:-4}; do
  timeout 600 python bench.py --steps 10 --warmup 2 --fill-variant $v --no-cpu-baseline > gpurun_out/bench_v$v.log 2>&1
  timeout 600 python bench.py --steps 10 --warmup 2 --fill-variant $v --no-cpu-baseline --ids-only > gpurun_out/bench_v${v}_ids.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --fill-variant ${PROF_VARIANT:-4} > gpurun_out/bench_prof.log 2>&1
tail -n 3 gpurun_out/pytest_gpu.log
for f in gpurun_out/bench_v*.log; do echo "== $f"; tail -n 1 $f | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('ms_per_step', round(d['ms_per_step'], 3), 'value', '%.3e' % d['value'], 'fill_ms', round(d['roofline']['launch_ms'], 3), 'frac', round(d['roofline']['frac'], 3))
except Exception as e:
    print('ERR', e)
"; done
