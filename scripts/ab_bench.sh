#!/bin/bash
# A/B of kernel variants on cfg2: ab_bench.sh OUTDIR name[:ENV=VAL,...] ...   (name "base" = the shipped library; other names = csrc/variants/libahmc_hip_<name>.so)
O=$1; shift; mkdir -p $O
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
  lib=""; base=${name%%@*}
  [ "$base" != "base" ] && lib="AHMC_HIP_LIB=$PWD/advancedhmc.jl_amd/csrc/variants/libahmc_hip_$base.so"
  env $lib $envs timeout 300 python bench.py --no-cpu-baseline --no-secondary --ess 0 --repeats ${AB_REPEATS:-1} --detail $PWD/$O/$name.json ${AB_ARGS:-} > $O/$name.line 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); c = d["config"]; r = d["roofline"]   # (the FULL record: bench.py --detail)
    print("%-14s e2e %.3e  warm %.3e (%.2f ms/tr, %.1f lf/tr)  draw %.3e (%.2f ms/tr, %.1f lf/tr)  in-kernel draw %.3e warm %.3e  |mean| %.1e |var-1| %.1e" % (
        sys.argv[2], d["value"], c["warmup_phase"]["value"], c["warmup_phase"]["ms_per_transition"], c["warmup_phase"]["mean_leapfrogs_per_transition"],
        c["post_adaptation"]["value"], c["post_adaptation"]["ms_per_transition"], c["post_adaptation"]["mean_leapfrogs_per_transition"],
        r["dominant"]["leapfrogs_per_s_in_kernel"] if r["dominant"]["phase"] == "draws" else r["other"]["leapfrogs_per_s_in_kernel"],
        r["other"]["leapfrogs_per_s_in_kernel"] if r["dominant"]["phase"] == "draws" else r["dominant"]["leapfrogs_per_s_in_kernel"],
        c["max_abs_mean"], c["max_abs_var_minus_1"] or 0))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
