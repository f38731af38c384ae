#!/bin/bash
mkdir -p gpurun_out/r6f
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --config cfg4 --steps 6 --warmup 1 --no-cpu-baseline --ess 0 --repeats 1 $BARGS --detail $PWD/gpurun_out/r6f/$name.json > gpurun_out/r6f/$name.line 2> gpurun_out/r6f/$name.err
  python - gpurun_out/r6f/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]; r = d["roofline"]
    print("%-16s e2e %.3e lf/s = %.1f TFLOP/s (frac %.3f)  warm %.3e draw %.3e" % (sys.argv[2], d["value"], r["achieved"], r["frac"], c["warmup_phase"]["value"], c["post_adaptation"]["value"]))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
}
{
BARGS="--dtype f32"; run f32_d512_n2w2 AHMC_DENSE_EPOCH_NCT=2 AHMC_DENSE_EPOCH_WPE=2; run f32_d512_n1w4 AHMC_DENSE_EPOCH_NCT=1
BARGS="--dtype f32 --dim 256"; run f32_d256_n2w2 AHMC_DENSE_EPOCH_NCT=2 AHMC_DENSE_EPOCH_WPE=2
BARGS="--dtype f32 --dim 384"; run f32_d384_n2w2 AHMC_DENSE_EPOCH_NCT=2 AHMC_DENSE_EPOCH_WPE=2
} > gpurun_out/r6f/dense_ab.txt 2>&1
cat gpurun_out/r6f/dense_ab.txt
