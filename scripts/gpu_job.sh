export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O
V=$PWD/advancedhmc.jl_amd/csrc/variants
AHMC_HIP_LIB=$V/libahmc_hip_pfs.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "nuts or full_size or cfg2 or fused" 2>&1 | tail -2
AB_ARGS="--warmup 1" bash scripts/ab_bench.sh $O/cfg2 base pfs base@2 pfs@2
bash scripts/profile_head.sh cfg2 2>&1 | tail -1
