#!/bin/bash
mkdir -p gpurun_out/r6t
run() { name=$1; cfg=$2; shift; shift
  env "$@" timeout 400 python bench.py --config $cfg --no-cpu-baseline --ess 0 --repeats 1 $BARGS --detail $PWD/gpurun_out/r6t/$name.json > gpurun_out/r6t/$name.line 2> gpurun_out/r6t/$name.err
  python - gpurun_out/r6t/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("%-14s e2e %.4e  warm %.4e  draw %.4e" % (sys.argv[2], d["value"], c["warmup_phase"]["value"], c["post_adaptation"]["value"]))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
}
{
BARGS="--steps 6 --warmup 1"
run cfg4_base cfg4
run cfg4_chunk128 cfg4 AHMC_DENSE_CHUNK=128
run cfg4_chunk32 cfg4 AHMC_DENSE_CHUNK=32
run cfg4_pipes1 cfg4 AHMC_DENSE_SPLIT=0
run cfg4_pipes3 cfg4 AHMC_DENSE_PIPES=3
run cfg4_pipes4 cfg4 AHMC_DENSE_PIPES=4
run cfg4_base2 cfg4
BARGS=""
run cfg2_base cfg2
run cfg2_g32e4 cfg2 AHMC_GEOMETRY=32,4
} > gpurun_out/r6t/sweeps.txt 2>&1
cat gpurun_out/r6t/sweeps.txt
