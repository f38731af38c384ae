O=gpurun_out/r3t; mkdir -p $O; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_user_targets.py -m gpu -q -x -k "kernel_target" 2>&1 | tail -5 ) > $O/pytest.log; tail -2 $O/pytest.log
( timeout 600 python bench.py --config cfg3 --warmup 1 --no-cpu-baseline 2> $O/cfg3.err | tail -1 ) > $O/bench_cfg3.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3t/bench_cfg3.json").read().strip().splitlines()[-1]); c=d["config"]
print('cfg3', 'e2e %.4e  warm %.4e  draw %.4e runs %s ess %s' % (d['value'], c['warmup_phase']['value'], c['post_adaptation']['value'], c['runs'], (c.get('ess') or {}).get('ess_per_sec')))
PY
