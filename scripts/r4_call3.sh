#!/bin/bash
O=gpurun_out/r4c; mkdir -p $O
( timeout 600 python -m pytest tests/test_pipeline_parity.py -m gpu -q --tb=short 2>&1 | tail -150 ) > $O/pytest_pipeline.log
( AHMC_NUTS_LOGW=1 timeout 600 python -m pytest tests/test_pipeline_parity.py -m gpu -q --tb=short -k fused_warmup_equals 2>&1 | tail -60 ) > $O/pytest_pipeline_logw.log
AB_REPEATS=2 AB_ARGS="--config cfg3" bash scripts/ab_bench.sh $O/cfg3 \
  "base@r4:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=4" \
  "base@r2:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=2" \
  "base@b16:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_BATCH=16" 2>&1 | tee $O/cfg3_ab.txt
AB_REPEATS=1 AB_ARGS="--config cfg2" bash scripts/ab_bench.sh $O/cfg2 base \
  "base@r250:AHMC_NUTS_ORDER_REFRESH=1" \
  "base@r64:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=64" \
  "base@r16:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=16" \
  "base@r8:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=8" 2>&1 | tee $O/cfg2_ab.txt
tail -8 $O/pytest_pipeline.log; tail -3 $O/pytest_pipeline_logw.log
