export TMPDIR=/tmp
PROFILE_STEPS=2 PROFILE_EXTRA="--transitions-per-step 3" PROFILE_PASSES="fetch write sq1" PROFILE_PASS_TIMEOUT=170 bash scripts/profile_head.sh cfg4 > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/prof_cfg4/summary.json"))
print(d.get("missing_passes"), d.get("bench_plain",{}).get("value"))
for r in d.get("per_kernel_counters",[])[:5]: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})
PY
ls -la gpurun_out/prof_cfg4/ | head -30
