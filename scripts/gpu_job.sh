bash scripts/gpu_check.sh r3x
export TMPDIR=/tmp
bash scripts/profile_head.sh cfg2 2>&1 | tail -1
