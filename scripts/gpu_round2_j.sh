O=$PWD/gpurun_out/r2j; mkdir -p $O
bash scripts/profile_head.sh cfg2 cfg3 > $O/profile.log 2>&1; tail -c 600 $O/profile.log
PROFILE_STEPS=2 bash scripts/profile_head.sh cfg5 > $O/profile5.log 2>&1; tail -c 400 $O/profile5.log
timeout 300 python scripts/ext_bench.py --chains 16384 --transitions 4 > $O/ext_bench_device.json 2> $O/ext_bench_device.err; cat $O/ext_bench_device.json; tail -3 $O/ext_bench_device.err
timeout 300 python scripts/ext_bench.py --chains 16384 --transitions 4 --host > $O/ext_bench_host.json 2> $O/ext_bench_host.err; cat $O/ext_bench_host.json; tail -3 $O/ext_bench_host.err
