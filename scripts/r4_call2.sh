#!/bin/bash
O=gpurun_out/r4b; mkdir -p $O
( timeout 600 python -m pytest tests/test_pipeline_parity.py -m gpu -q 2>&1 | tail -60 ) > $O/pytest_pipeline.log
( AHMC_NUTS_LOGW=1 timeout 600 python -m pytest tests/test_pipeline_parity.py -m gpu -q -k fused_warmup_equals 2>&1 | tail -30 ) > $O/pytest_pipeline_logw.log
( timeout 300 python scripts/dbg_fused_multiwave.py 2048 hier 64 2>&1 | tail -120 ) > $O/dbg_2048.log
( timeout 300 python scripts/dbg_fused_multiwave.py 600 hier 96 2>&1 | tail -120 ) > $O/dbg_600.log
( AHMC_NUTS_LOGW=1 timeout 300 python scripts/dbg_fused_multiwave.py 2048 hier 64 2>&1 | tail -120 ) > $O/dbg_2048_logw.log
AB_REPEATS=2 AB_ARGS="--config cfg3" bash scripts/ab_bench.sh $O/cfg3 \
  "base@r16:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=16" \
  "base@r8:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=8" \
  "base@r31f:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=31,AHMC_NUTS_FIRST_BATCH=8" 2>&1 | tee $O/cfg3_ab.txt
tail -5 $O/pytest_pipeline.log; tail -3 $O/pytest_pipeline_logw.log
