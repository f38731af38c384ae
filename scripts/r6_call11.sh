#!/bin/bash
mkdir -p gpurun_out/r6k
for B in 1000 500 250 125 60; do
  AHMC_NUTS_BATCH=$B timeout 300 python bench.py --config cfg3 --no-cpu-baseline --ess 0 --repeats 1 --warmup 0 --detail $PWD/gpurun_out/r6k/b$B.json > gpurun_out/r6k/b$B.line 2> gpurun_out/r6k/b$B.err
  python - gpurun_out/r6k/b$B.json $B <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]; r = d["roofline"]
w = r["dominant"] if "warm" in r["dominant"]["phase"] else r["other"]
print("batch %5s  e2e %.3e  warm-up wall %.3e (%d launches, in-kernel %.3e)  draws wall %.3e  draw-batch %s" % (sys.argv[2], d["value"], c["warmup_phase"]["value"], w["launches"], w["leapfrogs_per_s_in_kernel"], c["post_adaptation"]["value"], c.get("draw_launch_length_found_by_the_engine")))
PY
done 2>&1 | tee gpurun_out/r6k/cfg3_batch.txt
