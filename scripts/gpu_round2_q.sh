O=$PWD/gpurun_out/r2q; mkdir -p $O
PYTHONFAULTHANDLER=1 AHMC_BENCH_FORCE_DIST=1 NCCL_DEBUG=WARN timeout 300 python -X faulthandler bench.py --steps 2 --repeats 1 --no-cpu-baseline --ess 0 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "exit $?"; tail -c 600 $O/bench_forcedist.json; grep -v "^  File \"/usr" $O/bench_forcedist.err | tail -25 | cut -c1-250
