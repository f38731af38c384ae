#!/bin/bash
# round 6, GPU call 1: the suite under margin-aware parity (no -x: the whole record), then cfg3 thread geometries under the round-4 schedule
mkdir -p gpurun_out/r6a
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r6a/suite.log
cp gpurun_out/parity_margins.json gpurun_out/r6a/ 2>/dev/null
V=$PWD/advancedhmc.jl_amd/csrc/variants/libahmc_hip_geo.so
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --config cfg3 --no-cpu-baseline --ess 0 --repeats 2 --detail $PWD/gpurun_out/r6a/$name.json > gpurun_out/r6a/$name.line 2> gpurun_out/r6a/$name.err
  python - gpurun_out/r6a/$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]
    print("%-10s e2e %.3e  warm %.3e  draw %.3e  runs %s  draw-batch %s" % (sys.argv[2], d["value"], c["warmup_phase"]["value"], c["post_adaptation"]["value"], c.get("runs"), c.get("draw_launch_length_found_by_the_engine")))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-600:])
PY
}
{
run base
run geo162 AHMC_HIP_LIB=$V
run geo84 AHMC_HIP_LIB=$V AHMC_GEOMETRY=8,4
run geo48 AHMC_HIP_LIB=$V AHMC_GEOMETRY=4,8
run base2
run geo84b AHMC_HIP_LIB=$V AHMC_GEOMETRY=8,4
} > gpurun_out/r6a/cfg3_geometries.txt 2>&1
cat gpurun_out/r6a/cfg3_geometries.txt
tail -5 gpurun_out/r6a/suite.log
