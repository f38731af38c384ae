O=$PWD/gpurun_out/r2p; mkdir -p $O
AHMC_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 2 --repeats 1 --no-cpu-baseline --ess 50 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; python -c "
import json; d=json.loads(open('$O/bench_forcedist.json').read().strip().splitlines()[-1]); print('forcedist', d['value'], d['n_gpus'], d['config']['gather'], d['config']['gathered_draws'], d['config']['max_abs_mean'])"; tail -3 $O/bench_forcedist.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --ess 0 > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; python -c "
import json; d=json.loads(open('$O/bench_torchrun1.json').read().strip().splitlines()[-1]); print('torchrun1', d['value'], d['n_gpus'], d['config']['gather'])"; tail -3 $O/bench_torchrun1.err
AHMC_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace -d $O/roctx -o roctx -- python bench.py --steps 1 --warmup 0 --repeats 1 --no-cpu-baseline --ess 0 --chains 4096 > $O/roctx.json 2> $O/roctx.err
python - <<'PY'
import sqlite3,glob
f=glob.glob('gpurun_out/r2p/roctx/**/*_results.db', recursive=True)
if f:
    c=sqlite3.connect(f[0]).cursor()
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    print([t for t in tabs if 'marker' in t.lower() or 'region' in t.lower()][:10])
    for t in tabs:
        if 'marker' in t.lower() and 'view' not in t.lower():
            try: print(t, c.execute(f"select count(*) from {t}").fetchone())
            except Exception as e: print(t, e)
    try:
        for r in c.execute("select name, count(*) from regions group by name order by count(*) desc limit 12"): print(r)
    except Exception as e: print("regions:", e)
PY
