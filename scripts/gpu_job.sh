export TMPDIR=/tmp
O=gpurun_out/r3ai; mkdir -p $O
for v in base dtw5 dtw6 dtw8 base dtw6; do
  lib=advancedhmc.jl_amd/csrc/variants/libahmc_hip_$v.so; [ $v = base ] && lib=advancedhmc.jl_amd/csrc/libahmc_hip.so
  ( AHMC_HIP_LIB=$lib timeout 600 python bench.py --config cfg4 --steps 4 --warmup 0 --repeats 1 --ess 0 --no-cpu-baseline 2> $O/bench_$v.err | tail -1 ) > $O/bench_$v.json
  python - $O/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[2], 'e2e %.4e  warm %.4e  draw %.4e' % (d['value'], c['warmup_phase']['value'], c['post_adaptation']['value']))
PY
done
