# scratch: whatever the last gpurun call of the session ran (see scripts/README.md)
export TMPDIR=/tmp
O=gpurun_out/r3as; mkdir -p $O
for w in 16 12 8; do
  ( AHMC_NUTS_WAVES_PER_CU=$w timeout 300 python bench.py --config cfg3 --steps 20 --warmup 0 --repeats 1 --ess 0 --no-cpu-baseline 2> $O/bench_w$w.err | tail -1 ) > $O/bench_w$w.json
  python - $O/bench_w$w.json $w <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print('cfg3 waves/CU', sys.argv[2], 'e2e %.4e  warm %.4e  draw %.4e' % (d['value'], c['warmup_phase']['value'], c['post_adaptation']['value']))
PY
done
