export TMPDIR=/tmp
O=gpurun_out/r3am; mkdir -p $O
for cfg in cfg3 cfg4 cfg5; do
  st=20; [ $cfg = cfg5 ] && st=2
  ( timeout 900 python bench.py --config $cfg --steps $st --warmup 1 --no-cpu-baseline 2> $O/bench_$cfg.err | tail -1 ) > $O/bench_$cfg.json
  python - $O/bench_$cfg.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[1], 'e2e %.4e  warm %.4e  draw %.4e runs %s' % (d['value'], c['warmup_phase']['value'], c['post_adaptation']['value'], c['runs']))
PY
  PROFILE_STEPS=2 PROFILE_PASSES=none bash scripts/profile_head.sh $cfg > /dev/null 2>&1
done
