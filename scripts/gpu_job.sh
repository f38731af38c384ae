# scratch: whatever the last gpurun call of the session ran (see scripts/README.md)
export TMPDIR=/tmp
O=gpurun_out/r3ar; mkdir -p $O
( AHMC_DEBUG=1 timeout 300 python bench.py --config cfg3 --steps 20 --warmup 0 --repeats 1 --ess 0 --no-cpu-baseline 2> $O/bench_dbg.err | tail -1 ) > $O/bench_dbg.json
grep "ahmc\]" $O/bench_dbg.err | sort | uniq -c
