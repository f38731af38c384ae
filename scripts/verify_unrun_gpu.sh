#!/bin/bash
# First GPU call of the next round: run what round 1 wrote after its GPU minutes were spent.
#   gpurun --timeout 1500 -- 'bash scripts/verify_unrun_gpu.sh'
# 1. the regular GPU suite (the xfail-non-strict tests of tests/test_zz_external_target_gpu.py report XPASS / XFAIL),
# 2. the same file with --runxfail so that a failure shows its traceback,
# 3. a short bench line (the kernels are those of the last measured build: scripts/isa_digest.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/verify
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu -x -rxX > $O/gpu_suite.log 2>&1
echo "suite exit $?" >> $O/gpu_suite.log
timeout 900 python -m pytest tests/test_zz_external_target_gpu.py -q -m gpu --runxfail --timeout 300 > $O/unrun.log 2>&1
echo "unrun exit $?" >> $O/unrun.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
# 4. what an ask / tell request costs (device closure, then host closure)
timeout 300 python scripts/ext_bench.py --chains 16384 --transitions 4 > $O/ext_bench_device.json 2> $O/ext_bench_device.err
timeout 300 python scripts/ext_bench.py --chains 16384 --transitions 4 --host > $O/ext_bench_host.json 2> $O/ext_bench_host.err
tail -5 $O/gpu_suite.log; tail -40 $O/unrun.log; cat $O/bench.json $O/ext_bench_device.json $O/ext_bench_host.json
