O=$PWD/gpurun_out/r2k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "dense" > $O/dense_tests.log 2>&1; echo "exit $?" >> $O/dense_tests.log; tail -5 $O/dense_tests.log | cut -c1-300
for v in "split1_b256:" "split0_b256:AHMC_DENSE_SPLIT=0" "split1_b64:AHMC_NUTS_BATCH=64" "split0_b64:AHMC_DENSE_SPLIT=0,AHMC_NUTS_BATCH=64"; do
  n=${v%%:*}; e=$(echo ${v#*:} | tr ',' ' ')
  env $e timeout 600 python bench.py --config cfg4 --no-cpu-baseline --steps 6 --warmup 0 --repeats 1 > $O/cfg4_$n.json 2> $O/cfg4_$n.err
  python - "$O/cfg4_$n.json" "$n" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print("%-12s e2e %.3e (%.1f TF)  warm %.3e  draw %.3e (%.1f TF)  lf/tr %.0f"%(sys.argv[2], d["value"], d["roofline"]["achieved"], c["warmup_phase"]["value"], c["post_adaptation"]["value"], c["post_adaptation"]["value"]*4*512*512/1e12, c["post_adaptation"]["mean_leapfrogs_per_transition"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
done
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ext_kt -o ext -- python scripts/ext_bench.py --chains 16384 --transitions 2 > $O/ext_kt.json 2> $O/ext_kt.err
python - <<'PY'
import sqlite3,glob
f=glob.glob('gpurun_out/r2k/ext_kt/**/*_results.db', recursive=True)
if f:
    c=sqlite3.connect(f[0]).cursor()
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 12"): print("%9.1f ms %6d calls %9.1f us avg %5.1f%% %s"%(r[2]/1e6,r[1],r[3]/1e3,r[4],r[0][:80]))
PY
