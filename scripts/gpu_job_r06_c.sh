cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_deep.py tests/test_gpu_pool.py -x -q -m gpu > gpurun_out/r06_t6.log 2>&1 || { tail -40 gpurun_out/r06_t6.log; exit 1; }
tail -3 gpurun_out/r06_t6.log
timeout -k 10 600 python3 scripts/deep_emit_profile.py 26 > gpurun_out/r06_deep_emit2.log 2>&1; tail -4 gpurun_out/r06_deep_emit2.log
timeout -k 10 600 python3 scripts/deep_emit_profile.py 24 > gpurun_out/r06_deep_emit3.log 2>&1; tail -4 gpurun_out/r06_deep_emit3.log
