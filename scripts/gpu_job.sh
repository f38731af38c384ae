export TMPDIR=/tmp
bash scripts/gpu_check.sh r3ag
bash scripts/profile_head.sh cfg2 > gpurun_out/r3ag/profile_cfg2.log 2>&1; tail -3 gpurun_out/r3ag/profile_cfg2.log
