export TMPDIR=/tmp
timeout 900 python bench.py --repeats 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('runs', ['%.3e'%x for x in c['runs']], 'median', d['value'])"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "adaptation or fused or statistical" 2>&1 | tail -2
