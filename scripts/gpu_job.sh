O=gpurun_out/r3i; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_external_target_gpu.py tests/test_v3_state_gather.py -m gpu -q -x -k "dense or cfg4 or Dense or state" 2>&1 | tail -40 ) > $O/pytest_dense.log
tail -5 $O/pytest_dense.log
for steps in 6 20; do
  timeout 900 python bench.py --config cfg4 --steps $steps --warmup 0 --repeats 1 --no-cpu-baseline > $O/cfg4_s$steps.json 2> $O/cfg4_s$steps.err
  python - $O/cfg4_s$steps.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(sys.argv[1], "e2e %.3e  warm %.3e  draw %.3e  TF %.1f  lf/tr %.0f/%.0f" % (d["value"], c["warmup_phase"]["value"], c["post_adaptation"]["value"], d["roofline"]["achieved"], c["warmup_phase"]["mean_leapfrogs_per_transition"], c["post_adaptation"]["mean_leapfrogs_per_transition"]))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
tail -3 $O/cfg4_s6.err
