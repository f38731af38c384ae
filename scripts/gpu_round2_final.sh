# the round's closing GPU call: the whole GPU suite, smoke(), the driver's bench line, and the profiles at HEAD
O=$PWD/gpurun_out/r2final; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -rf --timeout 600 > $O/gpu_suite.log 2>&1; echo "exit $?" >> $O/gpu_suite.log; grep -v "^  File\|^Thread\|Extension modules" $O/gpu_suite.log | tail -6 | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err; tail -c 1200 $O/bench_driver_line.json
bash scripts/profile_head.sh cfg2 cfg3 > $O/profile.log 2>&1
PROFILE_STEPS=2 bash scripts/profile_head.sh cfg5 > $O/profile5.log 2>&1
PROFILE_STEPS=4 bash scripts/profile_head.sh cfg4 > $O/profile4.log 2>&1; tail -c 300 $O/profile4.log
