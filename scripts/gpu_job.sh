# scratch: whatever the last gpurun call of the session ran (see scripts/README.md)
export TMPDIR=/tmp
O=gpurun_out/r3ax; mkdir -p $O
for n in 16 250; do
( AHMC_NUTS_FIRST_BATCH=$n timeout 30 python bench.py --config cfg3 --steps 20 --warmup 0 --repeats 1 --ess 0 --no-cpu-baseline 2> $O/cfg3_$n.err | tail -1 ) > $O/cfg3_$n.json
python - $n <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r3ax/cfg3_{sys.argv[1]}.json').read().strip().splitlines()[-1]); k=d['config']
print('cfg3 first batch', sys.argv[1], ': e2e %.4e warm %.4e draw %.4e' % (d['value'], k['warmup_phase']['value'], k['post_adaptation']['value']))
PY
done
