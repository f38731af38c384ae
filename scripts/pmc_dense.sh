#!/bin/bash
# PMC pass over the cfg4 script: MFMA utilisation of k_dgemm (dense engine)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmcd; rm -rf $O; mkdir -p $O; cd $R
ADAPT=10 STEPS=16 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $O/p -o p -- python scripts/dense_bench.py > $O/log.txt 2>&1
python - <<PY
import sqlite3, glob, json
db = sqlite3.connect(glob.glob("$O/p/**/*_results.db", recursive=True)[0]); cur = db.cursor()
agg = {}
for name, grid, cn, v, dur in cur.execute("select kernel_name, grid_size, counter_name, value, duration from counters_collection"):
    k = "k_dgemm (64x64 tiles)" if "k_dgemm<" in name else ("k_dgemm_small" if "k_dgemm_small" in name else ("k_d_tree" if "k_d_tree<" in name else None))
    if not k: continue
    a = agg.setdefault(k, {}); a[cn] = a.get(cn, 0.0) + v
out = {}
for k, a in agg.items():
    gui = a.get("GRBM_GUI_ACTIVE", 0) / 8  # summed over the 8 XCDs
    out[k] = {"counters": a, "mfma_busy_fraction": (a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 256 * 4)) if gui else None,
              "mfma_f64_instructions": a.get("SQ_INSTS_VALU_MFMA_F64", 0)}
print(json.dumps(out, indent=1))
PY
tail -2 $O/log.txt
rm -rf $O/p
