#!/usr/bin/env python
"""Aggregate the per-wave timelines of MANY launches of one phase (scripts/wave_timeline.py analyses one): fill and lockstep of
every launch, and of the phase as a whole — Σ wave durations ÷ (wave slots × Σ launch durations).
    python scripts/wave_timeline_agg.py <prefix> <mode> [chains_per_wave] [wave_slots] [skip_first_n_launches]"""
import glob
import json
import re
import sys

import numpy as np

prefix, mode = sys.argv[1], sys.argv[2]
cpw = int(sys.argv[3]) if len(sys.argv) > 3 else 1
slots = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
skip = int(sys.argv[5]) if len(sys.argv) > 5 else 0
files = sorted(glob.glob(f"{prefix}.*.mode{mode}.bin"), key=lambda f: int(re.search(r"\.(\d+)\.mode", f).group(1)))[skip:]
rows = []
for f in files:
    a = np.fromfile(f, dtype=np.uint64).reshape(-1, 8)
    a = a[a[:, 1] > 0]
    if not len(a):
        continue
    t0, t1 = a[:, 0].astype(np.float64) * 1e-8, a[:, 1].astype(np.float64) * 1e-8
    L = t1.max() - t0.min()
    rows.append({"launch_s": L, "busy": float((t1 - t0).sum()), "leaf_steps": float(a[:, 2].sum()), "alive": float(a[:, 3].sum()),
                 "re": float(a[:, 4].sum()), "trans": float(a[:, 5].max()), "waves": int(len(a)), "longest_wave_share": float((t1 - t0).max() / L)})
tot = {k: sum(r[k] for r in rows) for k in ("launch_s", "busy", "leaf_steps", "alive", "re")}
out = {"launches": len(rows), "transitions_per_launch": sorted({int(r["trans"]) for r in rows}), "waves_per_launch": rows[0]["waves"] if rows else 0,
       "phase_s": tot["launch_s"], "fill": tot["busy"] / (slots * tot["launch_s"]), "lockstep": tot["alive"] / (cpw * tot["leaf_steps"]),
       "reintegration_steps_per_leaf_step": tot["re"] / tot["leaf_steps"], "chain_leapfrogs": tot["alive"],
       "fill_by_launch_quartiles": [float(np.percentile([r["busy"] / (slots * r["launch_s"]) for r in rows], q)) for q in (0, 25, 50, 75, 100)],
       "longest_wave_share_of_its_launch_median": float(np.median([r["longest_wave_share"] for r in rows]))}
print(json.dumps(out, indent=1))
