O=$PWD/gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_external_target_gpu.py -q -m gpu -x -k "dense or zz or external or user_density or hip_" > $O/dense_ext.log 2>&1; echo "exit $?" >> $O/dense_ext.log; tail -4 $O/dense_ext.log | cut -c1-200
python scripts/ext_trace.py 16384; python scripts/ext_trace.py 65536
timeout 300 python scripts/ext_bench.py --chains 16384 --transitions 4 2>/dev/null | tail -1
timeout 400 python bench.py --config cfg4 --no-cpu-baseline --steps 6 --warmup 0 --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('cfg4 e2e %.3e draw %.3e'%(d['value'], c['post_adaptation']['value']))"
