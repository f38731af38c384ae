#!/usr/bin/env python
"""Sum rocprofv3 PMC counters per kernel over one or more pass directories (rocpd SQLite) and join them with the kernel
trace of the same command: calls, device time, HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, KiB units; the gfx950 correction of
MI355X_MICROARCH.md), bytes per call, TB/s over the kernel's own time, VALU / MFMA instructions per call.

    python scripts/pmc_by_kernel.py <kt_dir> <pmc_dir> [<pmc_dir> ...] > summary.json

Written for the dense engine (k_dgemm, k_d_tree, ...), whose bench-sized runs have too many dispatches for
scripts/profile_head.sh's five passes; scripts/profile_dense_counters.sh runs it on a short job."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def db(d):
    f = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    return sqlite3.connect(f[0]).cursor() if f else None


kt = db(sys.argv[1])
ks = {r[0]: {"calls": r[1], "total_ms": r[2] / 1e3, "mean_us": r[3], "percent": r[4]}   # (the view reports microseconds)
      for r in kt.execute("select name,total_calls,total_duration,average,percentage from top_kernels")}
cnt = defaultdict(lambda: defaultdict(float))
ncall = defaultdict(dict)
for d in sys.argv[2:]:
    cur = db(d)
    if not cur:
        continue
    for kn, cn, v, n in cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
        cnt[kn][cn] += v
        ncall[kn][cn] = n
rows = []
for kn, c in cnt.items():
    k = ks.get(kn)
    if not k:
        continue
    r = {"kernel": kn[:90], **k}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        nb = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        calls = ncall[kn].get("FETCH_SIZE") or k["calls"]
        r["hbm_gbytes"] = nb / 1e9
        r["hbm_bytes_per_call"] = nb / calls
        r["hbm_tb_per_s_over_kernel_time"] = nb / 1e12 / (k["total_ms"] / 1e3) if k["total_ms"] else None
        r["read_fraction"] = 2 * c["FETCH_SIZE"] / (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) if nb else None
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"):
        if name in c:
            r[name + "_per_call"] = c[name] / (ncall[kn].get(name) or k["calls"])
    rows.append(r)
rows.sort(key=lambda r: -r["total_ms"])
print(json.dumps({"kernels": rows[:12]}, indent=1))
