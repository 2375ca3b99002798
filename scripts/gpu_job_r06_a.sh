cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_emit_tiles.py tests/test_gpu_fuzz.py tests/test_gpu_offline_driver.py tests/test_gpu_multigraph.py tests/test_gpu_rccl.py -x -q -m gpu > gpurun_out/r06_t5.log 2>&1 || { tail -30 gpurun_out/r06_t5.log; exit 1; }
tail -3 gpurun_out/r06_t5.log
bash scripts/count_phase_r06.sh > gpurun_out/r06_rows_probe.txt 2>&1; tail -60 gpurun_out/r06_rows_probe.txt
python bench.py --steps 20 --warmup 3 --no-index --no-config5 --no-cpu-baseline --compare-pool 0 > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err; python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_a.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['emit_class'], d['calibration_ms'], d['phases_ms']['per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
