mkdir -p gpurun_out/r2b; O=gpurun_out/r2b
timeout 900 python -m pytest tests -q -m gpu -rf --timeout 600 > $O/gpu_suite.log 2>&1; echo "exit $?" >> $O/gpu_suite.log; tail -40 $O/gpu_suite.log
timeout 300 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -c 3000 $O/bench_cfg2.json; tail -5 $O/bench_cfg2.err
AHMC_GEOMETRY=32,4 timeout 300 python bench.py --no-cpu-baseline --ess 0 --repeats 1 > $O/bench_cfg2_g32e4.json 2> $O/bench_cfg2_g32e4.err; tail -c 1500 $O/bench_cfg2_g32e4.json
timeout 300 python bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; tail -c 1500 $O/bench_cfg3.json; tail -3 $O/bench_cfg3.err
timeout 400 python bench.py --config cfg4 --no-cpu-baseline --steps 2 --warmup 0 --repeats 1 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; tail -c 1500 $O/bench_cfg4.json; tail -3 $O/bench_cfg4.err
timeout 400 python bench.py --config cfg5 --no-cpu-baseline --steps 2 --warmup 0 --repeats 1 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; tail -c 1500 $O/bench_cfg5.json; tail -3 $O/bench_cfg5.err
bash scripts/profile_head.sh cfg2 > $O/profile.log 2>&1; tail -c 3000 $O/profile.log
