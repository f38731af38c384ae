O=gpurun_out/r3o; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "two_pipelines or cfg4" 2>&1 | tail -5 ) > $O/pytest.log; tail -2 $O/pytest.log
for np in 2 3 4; do
  AHMC_DENSE_PIPES=$np timeout 600 python bench.py --config cfg4 --steps 6 --warmup 0 --repeats 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('pipes $np', 'e2e %.3e  warm %.3e  draw %.3e  TF %.1f' % (d['value'], c['warmup_phase']['value'], c['post_adaptation']['value'], d['roofline']['achieved']))"
done
for env in "X=0" "AHMC_NUTS_LDS_SLOTS=2" "AHMC_NUTS_LDS_SLOTS=3" "AHMC_NUTS_WAVES_PER_CU=4" "AHMC_NUTS_WAVES_PER_CU=12"; do
  env $env timeout 300 python bench.py --config cfg5 --steps 2 --warmup 0 --repeats 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('cfg5 $env', 'e2e %.3e  warm %.3e  draw %.3e' % (d['value'], c['warmup_phase']['value'], c['post_adaptation']['value']))"
done
