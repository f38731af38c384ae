# scratch: whatever the last gpurun call of the session ran (see scripts/README.md)
export TMPDIR=/tmp
bash scripts/gpu_check.sh r3at
