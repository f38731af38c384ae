#!/bin/bash
mkdir -p gpurun_out/r6h
timeout 1300 python -m pytest tests/test_zz_external_target_gpu.py tests/test_gpu_parity.py::test_dense_transitions tests/test_gpu_parity.py::test_dense_epoch_kernel_equals_step_synchronous_kernels tests/test_gpu_parity.py::test_cfg4_shape_against_oracle -q -p no:cacheprovider -m gpu 2>&1 | tail -25 > gpurun_out/r6h/tests.log
tail -n 4 gpurun_out/r6h/tests.log
{
for spec in "gen" "classic TC=classic" "strict TC=strict" "temper TEMPER=1.02" "classic_f32 TC=classic DTYPE=f32" "strict_f32 TC=strict DTYPE=f32" "temper_f32 TEMPER=1.02 DTYPE=f32"; do
  set -- $spec; name=$1; shift
  for mode in epoch step old; do
    extra=""; [ $mode = step ] && extra="AHMC_DENSE_EPOCH=0"; [ $mode = old ] && extra="AHMC_DENSE_POOL=0 AHMC_DENSE_EPOCH=0"
    [ $mode = old ] && [ $name = gen ] && continue
    echo "== $name $mode: $(env ADAPT=100 STEPS=100 $extra "$@" timeout 300 python scripts/dense_bench.py 2>&1 | grep -E "MFMA|epoch launches" | tr '\n' ' ')"
  done
done
} > gpurun_out/r6h/variants.txt 2>&1
cat gpurun_out/r6h/variants.txt
