mkdir -p gpurun_out/r2c; O=gpurun_out/r2c
HIP_LAUNCH_BLOCKING=1 timeout 300 python scripts/dbg_multiwave.py 1000 1 > $O/dbg_1000.log 2>&1; echo "exit $?" >> $O/dbg_1000.log; tail -15 $O/dbg_1000.log
HIP_LAUNCH_BLOCKING=1 timeout 300 python scripts/dbg_multiwave.py 1000 0 > $O/dbg_1000_norealign.log 2>&1; echo "exit $?" >> $O/dbg_1000_norealign.log; tail -6 $O/dbg_1000_norealign.log
timeout 900 python -m pytest tests -q -m gpu -rf --timeout 600 --deselect "tests/test_gpu_parity.py::test_multiwave_chains" > $O/gpu_suite.log 2>&1; echo "exit $?" >> $O/gpu_suite.log; tail -60 $O/gpu_suite.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --ess 0 --repeats 1 > $O/bench_base.json 2> $O/bench_base.err; python -c "
import json,sys
d=json.loads(open('$O/bench_base.json').read().strip().splitlines()[-1]); c=d['config']; print('BASE', d['value'], c['warmup_phase']['value'], c['post_adaptation']['value'], d['roofline']['dominant']['leapfrogs_per_s_in_kernel'], d['roofline']['other']['leapfrogs_per_s_in_kernel'])"
AHMC_HIP_LIB=$PWD/advancedhmc.jl_amd/csrc/variants/libahmc_hip_mfma.so timeout 300 python bench.py --no-cpu-baseline --ess 0 --repeats 1 > $O/bench_mfma.json 2> $O/bench_mfma.err; python -c "
import json,sys
d=json.loads(open('$O/bench_mfma.json').read().strip().splitlines()[-1]); c=d['config']; print('MFMA', d['value'], c['warmup_phase']['value'], c['post_adaptation']['value'], d['roofline']['dominant']['leapfrogs_per_s_in_kernel'], d['roofline']['other']['leapfrogs_per_s_in_kernel'], c['max_abs_mean'], c['max_abs_var_minus_1'])"; tail -3 $O/bench_mfma.err
AHMC_HIP_LIB=$PWD/advancedhmc.jl_amd/csrc/variants/libahmc_hip_mfma.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "nuts_transitions or nuts_geometries or cfg2_pipeline or full_size_slice or phasepoint_and_leapfrog" > $O/mfma_parity.log 2>&1; tail -5 $O/mfma_parity.log
