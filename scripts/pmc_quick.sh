#!/bin/bash
# one PMC pass (instruction counts) over the bench command -> VALU/SALU instructions per leapfrog of the sampling-phase kernel
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace -d $O/sq1 -o sq1 -- python bench.py --no-cpu-baseline > $O/bench.json 2> $O/err.txt
python - <<PY
import sqlite3, glob, json, re
db = sqlite3.connect(glob.glob("$O/sq1/**/*_results.db", recursive=True)[0]); cur = db.cursor()
b = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
per = {}
for did, name, cn, v in cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
    if re.search(r"k_nuts<\w+, \d+, \d+, 0, \d+>", name): per.setdefault(did, {}).setdefault(cn, 0.0); per[did][cn] += v
last = sorted(per)[-2:]
lf = b["roofline"]["leapfrogs_per_launch"]
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
    print(c, sum(per[d][c] for d in last) / len(last) / lf, "per leapfrog")
print("value", b["value"])
PY
rm -rf $O/sq1
