# scratch: whatever the last gpurun call of the session ran (see scripts/README.md)
export TMPDIR=/tmp
mkdir -p gpurun_out/r3ay
( AHMC_NUTS_ORDER_REFRESH=1 AHMC_NUTS_BATCH=62 timeout 14 python bench.py --config cfg3 --steps 20 --warmup 0 --repeats 1 --ess 0 --no-cpu-baseline 2> gpurun_out/r3ay/cfg3.err | tail -1 ) > gpurun_out/r3ay/cfg3.json
python -c "
import json
d=json.loads(open('gpurun_out/r3ay/cfg3.json').read().strip().splitlines()[-1]); k=d['config']
print('cfg3 refresh, 62 per launch: e2e %.4e warm %.4e draw %.4e' % (d['value'], k['warmup_phase']['value'], k['post_adaptation']['value']))"
