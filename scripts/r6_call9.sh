#!/bin/bash
mkdir -p gpurun_out/r6i
timeout 1500 python -m pytest tests/test_gpu_parity.py::test_multiwave_chains tests/test_gpu_parity.py::test_cfg5_full_size_slices_against_oracle "tests/test_pipeline_parity.py" tests/test_instantiations.py -q -p no:cacheprovider -m gpu -k "hier or cfg5 or multiwave or 2048 or 600 or instantiation" 2>&1 | tail -12 > gpurun_out/r6i/tests.log
tail -n 4 gpurun_out/r6i/tests.log
V=$PWD/advancedhmc.jl_amd/csrc/variants/libahmc_hip_nopre.so
run() { name=$1; shift
  env "$@" timeout 400 python bench.py --config cfg5 --steps 2 --warmup 0 --no-cpu-baseline --ess 0 --repeats 1 --detail $PWD/gpurun_out/r6i/$name.json > gpurun_out/r6i/$name.line 2> gpurun_out/r6i/$name.err
  python - gpurun_out/r6i/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("%-10s e2e %.4e  warm %.4e  draw %.4e  |mean| %.3e" % (sys.argv[2], d["value"], c["warmup_phase"]["value"], c["post_adaptation"]["value"], c.get("max_abs_mean")))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
{
run pre
run nopre AHMC_HIP_LIB=$V
run pre2
run nopre2 AHMC_HIP_LIB=$V
} > gpurun_out/r6i/cfg5.txt 2>&1
cat gpurun_out/r6i/cfg5.txt
