#!/usr/bin/env python
"""Do the dense engine's two pipelines overlap?  Reads a rocprofv3 --kernel-trace database (rocpd SQLite) of a cfg4
run and reports, for the GEMM and the tree kernel: launch shape (registers, LDS, scratch, grid), mean duration, the
queues they ran on, and how much of their device time was concurrent with a kernel of the OTHER queue
(sum of durations vs length of the union of the intervals).  Prints JSON.

    rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python bench.py --config cfg4 --steps 1 --warmup 0 --repeats 1 --ess 0 --no-cpu-baseline
    python scripts/dense_timeline.py gpurun_out/tl > gpurun_out/dense_timeline.json
"""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

f = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)
cur = sqlite3.connect(f[0]).cursor()
rows = cur.execute("select name, queue_id, start, end, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
                   "from kernels order by start").fetchall()
short = lambda n: n.split("(")[0][:60]
by = defaultdict(list)
for r in rows:
    by[short(r[0])].append(r)
out = {"dispatches": len(rows), "kernels": {}}
for n, rs in sorted(by.items(), key=lambda kv: -sum(r[3] - r[2] for r in kv[1]))[:8]:
    d = [r[3] - r[2] for r in rs]
    out["kernels"][n] = {"calls": len(rs), "total_ms": sum(d) / 1e6, "mean_us": sum(d) / len(d) / 1e3, "queues": sorted({r[1] for r in rs}),
                         "grid_x_last": rs[-1][4], "workgroup": rs[-1][5], "lds": rs[-1][6], "scratch": rs[-1][7], "vgpr": rs[-1][8], "agpr": rs[-1][9],
                         "sgpr": rs[-1][10]}
# union of all kernel intervals vs their sum: 1.0 = fully serial
iv = sorted((r[2], r[3]) for r in rows)
union, s0, e0 = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > e0:
        union += e0 - s0
        s0, e0 = s, e
    else:
        e0 = max(e0, e)
union += e0 - s0
total = sum(e - s for s, e in iv)
out["sum_of_durations_ms"] = total / 1e6
out["union_ms"] = union / 1e6
out["span_ms"] = (iv[-1][1] - iv[0][0]) / 1e6
out["concurrency"] = total / union          # 1.0 = never two kernels at once
out["idle_fraction_of_span"] = 1 - union / (iv[-1][1] - iv[0][0])
# a window of the bulk phase: 24 consecutive dispatches starting where (late in the run) the tree kernel still runs at its
# full grid, and the statistics of the full-grid launches only (the tail of a batch is launch-latency bound)
tmax = max((r[4] for r in rows if "k_d_tree" in r[0]), default=0)
gmax = max((r[4] for r in rows if "k_dgemm<" in r[0]), default=0)
full_t = [r for r in rows if "k_d_tree" in r[0] and r[4] >= 0.9 * tmax]
full_g = [r for r in rows if "k_dgemm<" in r[0] and r[4] >= 0.9 * gmax]
out["full_grid"] = {"tree_grid": tmax, "gemm_grid": gmax, "tree_calls": len(full_t), "gemm_calls": len(full_g),
                    "tree_mean_us": sum(r[3] - r[2] for r in full_t) / max(1, len(full_t)) / 1e3,
                    "gemm_mean_us": sum(r[3] - r[2] for r in full_g) / max(1, len(full_g)) / 1e3}
for q in sorted({r[1] for r in full_g}):
    st = [r[2] for r in full_g if r[1] == q]
    gaps = sorted(b - a for a, b in zip(st, st[1:]))
    if gaps:
        out["full_grid"]["gemm_period_us_median_queue%d" % q] = gaps[len(gaps) // 2] / 1e3
start = next((i for i in range(int(len(rows) * 0.7), len(rows)) if "k_d_tree" in rows[i][0] and rows[i][4] >= 0.9 * tmax), int(len(rows) * 0.8))
t0 = rows[start][2]
out["window"] = [{"t_us": (r[2] - t0) / 1e3, "dur_us": (r[3] - r[2]) / 1e3, "queue": r[1], "grid": r[4], "kernel": short(r[0])[:28]} for r in rows[start:start + 24]]
# per pair (GEMM on one queue, tree kernel on the other): time both were running
g = [r for r in rows if "k_dgemm" in r[0]]
t = [r for r in rows if "k_d_tree" in r[0]]
j, both = 0, 0
for r in g:
    while j < len(t) and t[j][3] <= r[2]:
        j += 1
    k = j
    while k < len(t) and t[k][2] < r[3]:
        both += max(0, min(r[3], t[k][3]) - max(r[2], t[k][2]))
        k += 1
out["gemm_and_tree_concurrent_ms"] = both / 1e6
out["tree_total_ms"] = sum(r[3] - r[2] for r in t) / 1e6
out["gemm_total_ms"] = sum(r[3] - r[2] for r in g) / 1e6
print(json.dumps(out, indent=1))
