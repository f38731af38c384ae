#!/bin/bash
# cfg3 warm-up: bimodal from process to process? six processes, the warm-up phase's wall time against the device time of its launches
mkdir -p gpurun_out/r6j
for i in 1 2 3 4 5 6 7 8; do
  AHMC_DEBUG= timeout 300 python bench.py --config cfg3 --no-cpu-baseline --ess 0 --repeats 1 --warmup 0 --detail $PWD/gpurun_out/r6j/p$i.json > gpurun_out/r6j/p$i.line 2> gpurun_out/r6j/p$i.err
  python - gpurun_out/r6j/p$i.json $i <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d["config"]; r = d["roofline"]
w = r["dominant"] if "warm" in r["dominant"]["phase"] else r["other"]
dr = r["other"] if "warm" in r["dominant"]["phase"] else r["dominant"]
print("proc %s  e2e %.3e  warm-up wall %.3e lf/s (%.1f ms)  warm-up kernels: %d launches, %.1f ms each, in-kernel %.3e lf/s   draws wall %.3e in-kernel %.3e" % (
    sys.argv[2], d["value"], c["warmup_phase"]["value"], c["warmup_phase"]["ms_per_transition"] * c["n_adapts"], w["launches"], w["avg_launch_ms"], w["leapfrogs_per_s_in_kernel"],
    c["post_adaptation"]["value"], dr["leapfrogs_per_s_in_kernel"]))
PY
done 2>&1 | tee gpurun_out/r6j/cfg3_procs.txt
rocm-smi --showclocks 2>/dev/null | head -20 >> gpurun_out/r6j/cfg3_procs.txt
