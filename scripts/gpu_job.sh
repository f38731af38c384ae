export TMPDIR=/tmp
O=gpurun_out/r3ak; mkdir -p $O
AHMC_HIP_LIB=advancedhmc.jl_amd/csrc/variants/libahmc_hip_zpf.so timeout 600 python -m pytest tests -m gpu -x -q -k "bulk_sample or fused_warmup or cfg2_pipeline or nuts_transitions or full_size_slice" 2>&1 | tail -3
for v in base zpf base zpf; do
  lib=advancedhmc.jl_amd/csrc/variants/libahmc_hip_$v.so; [ $v = base ] && lib=advancedhmc.jl_amd/csrc/libahmc_hip.so
  ( AHMC_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline 2> $O/bench_$v.err | tail -1 ) > $O/bench_$v.json
  python - $O/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[2], 'e2e %.4e  warm %.4e  draw %.4e runs %s' % (d['value'], c['warmup_phase']['value'], c['post_adaptation']['value'], [round(x/1e9,4) for x in c['runs']]))
PY
done
