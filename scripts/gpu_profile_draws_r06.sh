cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for R in r06d r06e r06f r06g; do
  rm -rf gpurun_out/${R}_trace
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_trace -- python3 bench.py --steps 10 --warmup 2 --compare-pool 0 --no-cpu-baseline --no-index --no-config5 > gpurun_out/${R}_trace.log 2>&1 || exit 1
  python3 scripts/summarize_trace_r06.py $R 10 2 > gpurun_out/${R}_kernel_stats_print.txt 2>&1
  sed -n 3,4p gpurun_out/${R}_kernel_stats_print.txt
done
