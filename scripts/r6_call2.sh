#!/bin/bash
# round 6, GPU call 2: k_dense_epoch2 — parity of both shapes, then cfg4 A/B against round 4's kernel
mkdir -p gpurun_out/r6b
T="tests/test_gpu_parity.py::test_dense_epoch_kernel_equals_step_synchronous_kernels tests/test_gpu_parity.py::test_cfg4_shape_against_oracle"
timeout 900 python -m pytest $T -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r6b/tests_default.log
AHMC_DENSE_EPOCH_NCT=2 timeout 900 python -m pytest $T -q -x -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r6b/tests_nct2.log
tail -3 gpurun_out/r6b/tests_default.log gpurun_out/r6b/tests_nct2.log
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --config cfg4 --steps 6 --warmup 1 --no-cpu-baseline --ess 0 --repeats 1 --detail $PWD/gpurun_out/r6b/$name.json > gpurun_out/r6b/$name.line 2> gpurun_out/r6b/$name.err
  python - gpurun_out/r6b/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]; r = d["roofline"]
    print("%-10s e2e %.3e lf/s = %.1f TFLOP/s  warm %.3e draw %.3e  launches %s" % (sys.argv[2], d["value"], r["achieved"], c["warmup_phase"]["value"], c["post_adaptation"]["value"], r.get("launches_since_create")))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
}
{
run v1 AHMC_DENSE_EPOCH_V=1
run v2nct2 AHMC_DENSE_EPOCH_NCT=2
run v2nct1 AHMC_DENSE_EPOCH_NCT=1
run v1b AHMC_DENSE_EPOCH_V=1
run v2nct1b AHMC_DENSE_EPOCH_NCT=1
} > gpurun_out/r6b/cfg4_ab.txt 2>&1
cat gpurun_out/r6b/cfg4_ab.txt
