#!/bin/bash
# round 4, first GPU call: the suite (with the new pipeline-parity tests) → smoke → default bench, then the cfg3 schedule A/B
bash scripts/gpu_check.sh r4a
O=gpurun_out/r4a
AB_REPEATS=2 AB_ARGS="--config cfg3" bash scripts/ab_bench.sh $O/cfg3 base \
  "base@r250:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=250" \
  "base@r125:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=125" \
  "base@r62:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=62" \
  "base@r31:AHMC_NUTS_ORDER_REFRESH=1,AHMC_NUTS_DRAW_BATCH=31" \
  "base@n62:AHMC_NUTS_DRAW_BATCH=62" 2>&1 | tee $O/cfg3_ab.txt
