#!/bin/bash
mkdir -p gpurun_out/r6r
timeout 1300 python -m pytest tests/test_gpu_parity.py::test_dense_epoch_kernel_equals_step_synchronous_kernels tests/test_gpu_parity.py::test_cfg4_shape_against_oracle -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r6r/tests.log
tail -n 3 gpurun_out/r6r/tests.log
run() { name=$1; shift
  env "$@" timeout 400 python bench.py --config cfg4 --steps 6 --warmup 1 --no-cpu-baseline --ess 0 --repeats 1 $BARGS --detail $PWD/gpurun_out/r6r/$name.json > gpurun_out/r6r/$name.line 2> gpurun_out/r6r/$name.err
  python - gpurun_out/r6r/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]; r = d["roofline"]
    print("%-16s e2e %.3e lf/s = %.1f TFLOP/s (frac %.3f)  warm %.3e draw %.3e  launches %s" % (sys.argv[2], d["value"], r["achieved"], r["frac"], c["warmup_phase"]["value"], c["post_adaptation"]["value"], r.get("launches_since_create")))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
}
{
BARGS="--dim 1024 --chains 4096"; run f64_d1024_epoch; run f64_d1024_step AHMC_DENSE_EPOCH=0
BARGS="--dim 1024 --chains 4096 --dtype f32"; run f32_d1024_epoch; run f32_d1024_step AHMC_DENSE_EPOCH=0
BARGS="--dim 768"; run f64_d768_step
BARGS=""; run f64_d512_default
BARGS="--dtype f32"; run f32_d512_default
} > gpurun_out/r6r/dense_ab.txt 2>&1
cat gpurun_out/r6r/dense_ab.txt
