O=$PWD/gpurun_out/r2s; mkdir -p $O
AHMC_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 2 --repeats 1 --no-cpu-baseline --ess 0 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "exit $?"; grep '^{' $O/bench_forcedist.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forcedist', d['value'], d['n_gpus'], d['config']['gather'], d['config']['gathered_draws'], d['config']['max_abs_mean'])"
