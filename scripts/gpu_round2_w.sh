O=$PWD/gpurun_out/r2w; mkdir -p $O
bash scripts/ab_bench.sh $O base typed 2>&1 | tee $O/ab.log
AHMC_HIP_LIB=$PWD/advancedhmc.jl_amd/csrc/variants/libahmc_hip_typed.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "nuts_transitions or cfg2_pipeline or full_size_slice or fused or bulk" > $O/typed_parity.log 2>&1; tail -3 $O/typed_parity.log | cut -c1-200
