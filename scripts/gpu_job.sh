export TMPDIR=/tmp
for kb in 64 160; do
  AHMC_NUTS_LDS_WG_KB=$kb AHMC_DEBUG=1 timeout 300 python bench.py --config cfg5 --steps 2 --warmup 0 --repeats 1 --no-cpu-baseline 2> /tmp/err_$kb.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('cfg5 LDS/WG $kb KB', 'e2e %.3e  warm %.3e  draw %.3e' % (d['value'], c['warmup_phase']['value'], c['post_adaptation']['value']))"
  grep "k_nuts<" /tmp/err_$kb.txt | sort | uniq -c | head -3
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "multiwave or geometries" 2>&1 | tail -2
