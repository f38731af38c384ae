"""Distil the rocprofv3 (rocpd SQLite) outputs of scripts/profile_round.sh into one JSON: per-kernel
time stats and, for the dominant kernel's timed launches, per-launch counter sums."""
import glob, json, os, sqlite3, sys
from collections import defaultdict

O = sys.argv[1]
N_TIMED = int(os.environ.get("N_TIMED_LAUNCHES", "4"))  # bench.py default: 128 timed transitions = 4 launches of 32
out = {}


def db(name):
    f = glob.glob(os.path.join(O, name, "**", "*_results.db"), recursive=True)
    return sqlite3.connect(f[0]).cursor() if f else None


cur = db("kt")
dom = None
if cur:
    out["kernel_stats"] = [dict(zip(("name", "calls", "total_us", "average_us", "percent"), r))
                           for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 12")]
    dom = out["kernel_stats"][0]["name"]
    # the sampling-phase kernel is k_nuts MODE 0 (the warm-up runs MODE 3, a different instantiation that can
    # well be the larger share of a short bench): that is the one the bench's roofline is about
    import re
    for k in out["kernel_stats"]:
        if re.search(r"k_nuts<\w+, \d+, \d+, 0, \d+>", k["name"]):
            dom = k["name"]
            break
    d = cur.execute("select start, duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size "
                    "from kernels where name = ? order by start", (dom,)).fetchall()
    out["dominant_kernel"] = dom
    out["dominant_launches_total"] = len(d)
    out["dominant_timed_launch_ms"] = [x[1] / 1e6 for x in d[-N_TIMED:]]
    out["dominant_launch_shape"] = dict(zip(("grid", "workgroup", "lds_bytes", "vgpr", "agpr", "sgpr", "scratch"), d[-1][2:]))

counters = {}
for name in ("fetch", "write", "sq1", "sq2", "sq3", "grbm"):
    cur = db(name)
    if not cur:
        continue
    per = defaultdict(lambda: defaultdict(float))  # dispatch -> counter -> value summed over instances
    for did, cn, v in cur.execute("select dispatch_id, counter_name, value from counters_collection where kernel_name = ?", (dom,)):
        per[did][cn] += v
    for did in sorted(per)[-N_TIMED:]:
        for c, v in per[did].items():
            counters.setdefault(c, []).append(v)
out["counters_per_timed_launch"] = {c: sum(v) / len(v) for c, v in counters.items()}
for n in ("bench_plain", "bench_under_rocprof"):
    try:
        out[n] = json.loads(open(os.path.join(O, n + ".json")).read().strip().splitlines()[-1])
    except Exception as e:
        out[n] = repr(e)
print(json.dumps(out, indent=1))
