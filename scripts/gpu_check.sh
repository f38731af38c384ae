#!/bin/bash
# The driver's round-end sequence on a GPU box, in one gpurun call:  GPU parity suite → smoke() → the default bench line.
#   gpurun --timeout 900 -- 'bash scripts/gpu_check.sh r3a'        (outputs under gpurun_out/<tag>/)
# Optional second argument: extra pytest args (e.g. '-k dense').
tag=${1:-check}; shift || true
out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
du -sm --exclude=.git --exclude=gpurun_out . > "$out/tree_mb.txt" 2>&1
( timeout 1500 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -40 ) > "$out/pytest_gpu.log"; echo "pytest rc=${PIPESTATUS[0]}" >> "$out/pytest_gpu.log"
( timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -5 ) > "$out/smoke.log"
( timeout 900 python bench.py 2> "$out/bench.err" | tail -1 ) > "$out/bench_line.json"
tail -3 "$out/pytest_gpu.log"; cat "$out/smoke.log"; cut -c1-600 "$out/bench_line.json"
