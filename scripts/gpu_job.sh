bash scripts/gpu_check.sh r3q
