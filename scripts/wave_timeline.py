#!/usr/bin/env python
"""Per-wave timeline of one k_nuts launch, from a MEASUREMENT build of the engine (scripts/build_variant.py tl
-DAHMC_WAVE_TIMELINE=1; AHMC_WAVE_TIMELINE_OUT=<prefix> makes every MODE 0 / MODE 3 launch write <prefix>.<n>.mode<m>.bin):
eight 64-bit words per wave — start, end (100 MHz wall clock), leaf steps, Σ over leaf steps of the chains still building,
re-integration steps, transitions, HW_ID, XCC_ID.

    python scripts/wave_timeline.py <file.bin> [chains_per_wave] [wave_slots]

Prints what separates "the launch waits for a few waves" from "the waves are slow" from "lockstep":
  fill        Σ wave durations ÷ (wave slots × launch duration): how full the chip's wave slots were
  lockstep    Σ alive ÷ (chains per wave × Σ leaf steps): the share of lane groups doing a leaf in a leaf step
  step time   a wave's duration ÷ its (leaf + re-integration) steps, by start time (crowded phase vs the thin end)
"""
import json
import sys

import numpy as np


def analyse(path, cpw=1, slots=4096):
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
    a = a[a[:, 1] > 0]
    t0, t1 = a[:, 0].astype(np.float64) * 1e-8, a[:, 1].astype(np.float64) * 1e-8  # 100 MHz → seconds
    steps, alive, re, trans = (a[:, k].astype(np.float64) for k in (2, 3, 4, 5))
    T0, T1 = t0.min(), t1.max()
    L = T1 - T0
    dur = t1 - t0
    out = {"waves": int(len(a)), "launch_s": L, "fill": float(dur.sum() / (slots * L)),
           "lockstep": float(alive.sum() / (cpw * steps.sum())), "leaf_steps": float(steps.sum()), "chain_leapfrogs": float(alive.sum()),
           "reintegration_steps_per_leaf_step": float(re.sum() / steps.sum()),
           "longest_wave_s": float(dur.max()), "longest_wave_share_of_launch": float(dur.max() / L),
           "longest_wave_leaf_steps": float(steps[np.argmax(dur)])}
    # resident waves over time (20 bins)
    edges = np.linspace(T0, T1, 21)
    res = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None)
        res.append(round(float(ov.sum() / (hi - lo)), 1))
    out["resident_waves_by_twentieth_of_the_launch"] = res
    # time per step (µs) by start time
    st = dur / np.maximum(steps + re, 1) * 1e6
    order = np.argsort(t0)
    q = np.array_split(order, 8)
    out["us_per_step_by_start_octile"] = [round(float(np.median(st[i])), 3) for i in q]
    out["leaf_steps_per_wave_by_start_octile"] = [round(float(np.median(steps[i])), 0) for i in q]
    out["start_s_by_octile"] = [round(float(np.median(t0[i]) - T0), 3) for i in q]
    # the last 5 % of the launch: how many waves are still running and how fast
    late = t1 > T0 + 0.95 * L
    out["waves_ending_in_last_5pct"] = int(late.sum())
    out["us_per_step_of_those"] = round(float(np.median(st[late])), 3) if late.any() else None
    # time-area by what a wave was doing cannot be split further here; XCC spread of the longest 64 waves
    top = np.argsort(dur)[-64:]
    out["xcc_of_longest_64"] = np.bincount((a[top, 7] & 0xF).astype(np.int64), minlength=8).tolist()
    return out


if __name__ == "__main__":
    cpw = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    slots = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    print(json.dumps(analyse(sys.argv[1], cpw, slots), indent=1))
