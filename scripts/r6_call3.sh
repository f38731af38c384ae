#!/bin/bash
mkdir -p gpurun_out/r6c
V=$PWD/advancedhmc.jl_amd/csrc/variants/libahmc_hip_prof.so
for spec in "v1 AHMC_DENSE_EPOCH_V=1" "nct2 AHMC_DENSE_EPOCH_NCT=2" "nct1 AHMC_DENSE_EPOCH_NCT=1"; do
  set -- $spec; name=$1; shift
  env AHMC_HIP_LIB=$V AHMC_DEBUG=1 "$@" timeout 300 python bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline --ess 0 --repeats 1 --detail $PWD/gpurun_out/r6c/$name.json > gpurun_out/r6c/$name.line 2> gpurun_out/r6c/$name.err
  echo "== $name"; grep -E "k_dense_epoch2<|cycles per workgroup-step" gpurun_out/r6c/$name.err | sort | uniq -c | sort -rn | head -3; grep "cycles per workgroup-step" gpurun_out/r6c/$name.err | tail -1
  python -c "
import json; d=json.load(open('gpurun_out/r6c/$name.json')); print('$name', d['value'], d['roofline']['achieved'])"
done
