O=gpurun_out/r3m; mkdir -p $O; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ADAPT=10 DRAWS=10 timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python scripts/user_target_bench.py kernel > $O/kt.log 2>&1
python - <<'PY'
import glob, sqlite3
f=glob.glob("gpurun_out/r3m/kt/**/*_results.db", recursive=True)
cur=sqlite3.connect(f[0]).cursor()
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 12"):
    print("%10.1f ms %7d calls %9.1f us %5.1f%%  %s" % (r[2]/1e6, r[1], r[3]/1e3, r[4], r[0][:100]))
PY
tail -3 $O/kt.log
find $O -name "*.db" -size +8M -delete
