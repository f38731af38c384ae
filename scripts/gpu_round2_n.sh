O=$PWD/gpurun_out/r2n; mkdir -p $O
bash scripts/ab_bench.sh $O base inl inl@i128:AHMC_NUTS_INLINE_NORMALS=1 inl@i500:AHMC_NUTS_INLINE_NORMALS=1,AHMC_NUTS_BATCH=500 inl@i1000:AHMC_NUTS_INLINE_NORMALS=1,AHMC_NUTS_BATCH=1000 2>&1 | tee $O/ab.log
AHMC_HIP_LIB=$PWD/advancedhmc.jl_amd/csrc/variants/libahmc_hip_inl.so AHMC_NUTS_INLINE_NORMALS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "nuts or bulk or fused or cfg2 or full_size" > $O/inl_parity.log 2>&1; tail -4 $O/inl_parity.log | cut -c1-200
