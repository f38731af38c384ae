#!/bin/bash
O=gpurun_out/r4e; mkdir -p $O
( timeout 900 python -m pytest tests/test_pipeline_parity.py -m gpu -q --tb=short 2>&1 | tail -60 ) > $O/pytest_pipeline.log
export AHMC_DEBUG=1
AB_REPEATS=3 AB_ARGS="--config cfg3" bash scripts/ab_bench.sh $O/cfg3 base 2>&1 | tee $O/cfg3_ab.txt
AB_REPEATS=3 AB_ARGS="--config cfg2" bash scripts/ab_bench.sh $O/cfg2 base 2>&1 | tee $O/cfg2_ab.txt
grep -h "sched" $O/cfg3/base.err | head -40; grep -h "sched" $O/cfg2/base.err | head -40
tail -5 $O/pytest_pipeline.log
