#!/bin/bash
mkdir -p gpurun_out/r6n
timeout 900 python -m pytest tests/test_pipeline_parity.py tests/test_instantiations.py "tests/test_gpu_parity.py::test_bulk_sample_equals_stepwise" "tests/test_gpu_parity.py::test_host_draws_double_buffered" tests/test_v3_state_gather.py -q -x -p no:cacheprovider -m gpu 2>&1 | tail -6 > gpurun_out/r6n/tests.log
tail -n 3 gpurun_out/r6n/tests.log
run() { name=$1; cfg=$2; shift; shift
  env "$@" timeout 400 python bench.py --config $cfg --no-cpu-baseline --ess 0 --repeats 2 $BARGS --detail $PWD/gpurun_out/r6n/$name.json > gpurun_out/r6n/$name.line 2> gpurun_out/r6n/$name.err
  python - gpurun_out/r6n/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("%-12s e2e %.4e  warm %.4e  draw %.4e  runs %s" % (sys.argv[2], d["value"], c["warmup_phase"]["value"], c["post_adaptation"]["value"], ["%.4e" % x for x in c.get("runs", [])]))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
{
BARGS=""
run cfg2_pre cfg2
run cfg2_off cfg2 AHMC_NORMALS_PREFETCH=0
run cfg2_pre2 cfg2
run cfg2_off2 cfg2 AHMC_NORMALS_PREFETCH=0
run cfg3_pre cfg3
run cfg3_off cfg3 AHMC_NORMALS_PREFETCH=0
BARGS="--steps 2 --warmup 0"
run cfg5_pre cfg5
run cfg5_off cfg5 AHMC_NORMALS_PREFETCH=0
} > gpurun_out/r6n/prefetch.txt 2>&1
cat gpurun_out/r6n/prefetch.txt
