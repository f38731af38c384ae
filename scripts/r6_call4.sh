#!/bin/bash
mkdir -p gpurun_out/r6d
T="tests/test_gpu_parity.py::test_dense_epoch_kernel_equals_step_synchronous_kernels tests/test_gpu_parity.py::test_cfg4_shape_against_oracle"
timeout 1200 python -m pytest $T -q -x -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/r6d/tests.log
tail -n 4 gpurun_out/r6d/tests.log
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --config cfg4 --steps 6 --warmup 1 --no-cpu-baseline --ess 0 --repeats 1 $BARGS --detail $PWD/gpurun_out/r6d/$name.json > gpurun_out/r6d/$name.line 2> gpurun_out/r6d/$name.err
  python - gpurun_out/r6d/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]; r = d["roofline"]
    print("%-16s e2e %.3e lf/s = %.1f TFLOP/s (frac %.3f)  warm %.3e draw %.3e  launches %s" % (sys.argv[2], d["value"], r["achieved"], r["frac"], c["warmup_phase"]["value"], c["post_adaptation"]["value"], r.get("launches_since_create")))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
}
{
BARGS="--dtype f32"; run f32_epoch_nct2; run f32_epoch_nct1 AHMC_DENSE_EPOCH_NCT=1; run f32_step AHMC_DENSE_EPOCH=0
BARGS="--dim 384"; run f64_d384_epoch; run f64_d384_step AHMC_DENSE_EPOCH=0
BARGS="--dim 256"; run f64_d256_epoch; run f64_d256_v1 AHMC_DENSE_EPOCH_V=1
BARGS="--dim 768 --dtype f32"; run f32_d768_epoch; run f32_d768_step AHMC_DENSE_EPOCH=0
BARGS=""; run f64_d512_epoch2; run f64_d512_v1 AHMC_DENSE_EPOCH_V=1
} > gpurun_out/r6d/dense_ab.txt 2>&1
cat gpurun_out/r6d/dense_ab.txt
