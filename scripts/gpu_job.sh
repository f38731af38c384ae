export TMPDIR=/tmp
O=gpurun_out/r3ah; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "prefetch or bulk_sample or fused_warmup or host_draws" 2>&1 | tail -5
for pf in 0 1 0 1; do
  ( AHMC_NORMALS_PREFETCH=$pf timeout 600 python bench.py --no-cpu-baseline 2> $O/bench_$pf.err | tail -1 ) > $O/bench_$pf.json
  python - $O/bench_$pf.json $pf <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print('prefetch', sys.argv[2], 'e2e %.4e  warm %.4e  draw %.4e runs %s' % (d['value'], c['warmup_phase']['value'], c['post_adaptation']['value'], c['runs']), d['roofline']['frac'])
PY
done
