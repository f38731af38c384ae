#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
( timeout 900 python -m pytest tests/test_pipeline_parity.py -m gpu -q --tb=short 2>&1 | tail -60 ) > $O/pytest_pipeline.log
export AHMC_DEBUG=1
AB_REPEATS=2 AB_ARGS="--config cfg3" bash scripts/ab_bench.sh $O/cfg3 base \
  "base@r4:AHMC_NUTS_DRAW_BATCH=4" "base@r3:AHMC_NUTS_DRAW_BATCH=3" "base@r2:AHMC_NUTS_DRAW_BATCH=2" \
  "base@w32:AHMC_NUTS_DRAW_BATCH=4,AHMC_NUTS_WARM_BATCH=32,AHMC_NUTS_WARM_REFRESH=1" \
  "base@w8:AHMC_NUTS_DRAW_BATCH=4,AHMC_NUTS_WARM_BATCH=8,AHMC_NUTS_WARM_REFRESH=1" 2>&1 | tee $O/cfg3_ab.txt
AB_REPEATS=2 AB_ARGS="--config cfg2" bash scripts/ab_bench.sh $O/cfg2 base "base@s0:AHMC_NUTS_SCHED=0" \
  "base@w64:AHMC_NUTS_WARM_BATCH=64,AHMC_NUTS_WARM_REFRESH=1" 2>&1 | tee $O/cfg2_ab.txt
AB_REPEATS=1 AB_ARGS="--config cfg5 --steps 2" bash scripts/ab_bench.sh $O/cfg5 base "base@s0:AHMC_NUTS_SCHED=0,AHMC_NUTS_ORDER_REFRESH=0" 2>&1 | tee $O/cfg5_ab.txt
grep -h "sched" $O/cfg3/base.err | head -20; grep -h "sched" $O/cfg2/base.err | head -12; grep -h sched $O/cfg5/base.err | head
tail -5 $O/pytest_pipeline.log
