cd "$GRAFT_REPO_ROOT"
for e in 8 4 3; do
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-index --embedding $e 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('e=$e ms_per_step', round(d['ms_per_step'], 3), 'value', '%.3e' % d['value'], 'fill_ms', round(d['roofline']['launch_ms'], 3), 'frac', round(d['roofline']['frac'], 3), 'bpp', d['roofline']['bytes_per_path'])"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
