#!/bin/bash
mkdir -p gpurun_out/r6g
V=$PWD/advancedhmc.jl_amd/csrc/variants/libahmc_hip_e16.so
AHMC_HIP_LIB=$V AHMC_GEOMETRY=128,16 timeout 600 python -m pytest "tests/test_gpu_parity.py::test_multiwave_chains" -q -x -p no:cacheprovider -k "2048 or 1500 or 1000" 2>&1 | tail -8 > gpurun_out/r6g/tests_e16.log
tail -n 3 gpurun_out/r6g/tests_e16.log
run() { name=$1; shift
  env "$@" timeout 400 python bench.py --config cfg5 --steps 2 --warmup 0 --no-cpu-baseline --ess 0 --repeats 1 --detail $PWD/gpurun_out/r6g/$name.json > gpurun_out/r6g/$name.line 2> gpurun_out/r6g/$name.err
  python - gpurun_out/r6g/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("%-10s e2e %.3e  warm %.3e  draw %.3e  |mean| %.2e |var-1| %s" % (sys.argv[2], d["value"], c["warmup_phase"]["value"], c["post_adaptation"]["value"], c.get("max_abs_mean"), c.get("max_abs_var_minus_1")))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
{
run base
run e16 AHMC_HIP_LIB=$V AHMC_GEOMETRY=128,16
run v256 AHMC_HIP_LIB=$V
} > gpurun_out/r6g/cfg5.txt 2>&1
cat gpurun_out/r6g/cfg5.txt
