#!/usr/bin/env python
"""What a k_nuts launch of cfg3 costs under different dispatch orders, from the work traces of cfg3_trace.py (CPU only).
A wave = 4 chains in lockstep (a transition costs its longest tree, + 0.5 re-integration steps per leaf step), 4 waves per
wave slot as on the GPU (16 384 waves on 4 096 slots), slots take the next wave of the dispatch order when they fall free.
"by eps (today)" reproduces what the per-wave timeline MEASURED on the MI355X: lockstep 0.37, fill 0.50 (profiles/
r3_cfg3_wave_timeline_draws.json: 0.366 / 0.53)."""
import heapq
import os

import numpy as np

d = np.load(os.environ.get('TRACE', '/tmp/cfg3_trace.npz')); Wa, Wd, eps = d['Wa'].astype(np.int64), d['Wd'].astype(np.int64), d['eps']
n_dr, N = Wd.shape
tot = Wd.sum(0)
q = [0, .01, .5, .9, .99, .999, 1]
print("per-chain sum over the draws: quantiles", dict(zip(q, np.quantile(tot, q).astype(int))), " mean", tot.mean())
print("eps quantiles", dict(zip(q, np.quantile(eps, q).round(4))))
print("corr(log work, log eps) %.3f   corr(log draws work, log warm-up work) %.3f  corr(log first 64, log rest) %.3f  corr(log last-100 warm-up, log draws) %.3f" % (
    np.corrcoef(np.log(tot), np.log(eps))[0,1], np.corrcoef(np.log(tot), np.log(Wa.sum(0)))[0,1],
    np.corrcoef(np.log(Wd[:64].sum(0)), np.log(Wd[64:].sum(0)))[0,1], np.corrcoef(np.log(Wa[-100:].sum(0)), np.log(tot))[0,1]))

def wave_work(order, W, cpw=4, reint=0.5):
    g = order.reshape(-1, cpw)
    m = W[:, g].max(axis=2)                # (transitions, waves): leaf steps of a transition = its longest tree
    return (m * (1 + reint)).sum(0)        # + re-integration (max over the groups again, ≈ 0.5 per leaf step measured)

def makespan(work_in_dispatch_order, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for w in work_in_dispatch_order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + w)
    return max(h)

def report(name, order, W, cpw=4, slots=None, resort=True):
    slots = slots or N // cpw // 4         # the GPU's ratio: 16 384 waves on 4 096 slots
    ww = wave_work(order, W, cpw)
    ms = makespan(ww, slots)
    ideal = W.sum() * 1.5 / cpw / slots    # if every group were always busy and the slots perfectly packed
    print(f"{name:58s} lockstep {W.sum()*1.5/(cpw*ww.sum()):.3f}  fill {ww.sum()/(slots*ms):.3f}  makespan/ideal {ms/ideal:.2f}  heaviest wave/makespan {ww.max()/ms:.2f}")
    return ms

order_eps = np.argsort(eps)                               # ascending step size (today's first-launch order)
order_true = np.argsort(-tot)                             # oracle: by the work the chains are about to do
order_warm = np.argsort(-Wa.sum(0))
order_warm100 = np.argsort(-Wa[-100:].sum(0))
print("\none launch of %d transitions, %d chains, 4 chains per wave, %d slots" % (n_dr, N, N // 16))
report("by eps (today)", order_eps, Wd)
report("by the warm-up's total work", order_warm, Wd)
report("by the warm-up's last 100 transitions", order_warm100, Wd)
report("by the draws' own totals (perfect predictor of the chain totals)", order_true, Wd)
# perfect LPT on the true WAVE works (upper bound for any order with these groups)
ww = wave_work(order_true, Wd); ms = makespan(np.sort(ww)[::-1], N // 16)
print(f"{'  … waves then dispatched heaviest first':58s} fill {ww.sum()/(N//16*ms):.3f}")
# first batch then the rest ordered by the first batch's work
for fb in (16, 64, 250):
    o1 = order_eps
    w1 = wave_work(o1, Wd[:fb]); m1 = makespan(w1, N // 16)
    o2 = np.argsort(-Wd[:fb].sum(0))
    w2 = wave_work(o2, Wd[fb:]); m2 = makespan(w2, N // 16)
    ideal = Wd.sum() * 1.5 / 4 / (N // 16)
    print(f"first launch {fb:4d} by eps, the rest by its measured work: (makespan1 + makespan2)/ideal = {(m1+m2)/ideal:.2f}")
# equal launches, each ordered by the previous launch's work
for k in (8, 4, 2):
    L = n_dr // k; tot_ms = 0; o = order_eps
    for b in range(k):
        W = Wd[b*L:(b+1)*L]; tot_ms += makespan(wave_work(o, W), N // 16); o = np.argsort(-W.sum(0))
    print(f"{k} equal launches, each ordered by the one before: Σ makespan / ideal = {tot_ms/(Wd.sum()*1.5/4/(N//16)):.2f}")
# one chain per wave (no lockstep), 4x the waves on the same slots, half the lanes idle is NOT priced here
report("one chain per wave, by eps", order_eps, Wd, cpw=1, slots=N // 16)
report("one chain per wave, by true totals", order_true, Wd, cpw=1, slots=N // 16)

# hybrid: the heaviest x % of the chains alone in their waves (a heavy wave is then one chain's serial time, not four chains'
# lockstep), the rest four per wave; dispatch: single-chain waves first.  With the true totals (bound) and with eps (realistic).
print()
for name, key in (("true totals", -tot), ("eps", eps)):
    o = np.argsort(key)
    for x in (0.02, 0.05, 0.1, 0.25):
        nh = int(N * x) // 4 * 4
        w1 = wave_work(o[:nh], Wd, cpw=1)
        w4 = wave_work(o[nh:], Wd, cpw=4)
        ww = np.concatenate([w1, w4])
        ms = makespan(ww, N // 16)
        print(f"heaviest {x:4.0%} by {name:11s} alone in their waves: makespan/ideal {ms / (Wd.sum() * 1.5 / 4 / (N // 16)):.2f}  fill {ww.sum() / (N // 16 * ms):.3f}")

# several launches: which work orders launch b?  the first launch's (what the engine does today), everything so far, the launch before
print()
ideal = Wd.sum() * 1.5 / 4 / (N // 16)
for k in (1, 2, 4, 8, 16, 32):
    L = n_dr // k
    res = {}
    for mode in ("first launch's work (today)", "cumulative work", "the launch before"):
        o = np.argsort(eps)
        tot_ms = 0
        cum = np.zeros(N)
        for b in range(k):
            W = Wd[b * L:(b + 1) * L]
            tot_ms += makespan(wave_work(o, W), N // 16)
            cum += W.sum(0)
            if mode.startswith("first"):
                if b == 0:
                    o = np.argsort(-W.sum(0))
            elif mode.startswith("cumulative"):
                o = np.argsort(-cum)
            else:
                o = np.argsort(-W.sum(0))
        res[mode] = tot_ms / ideal
    print(f"{k:3d} launches of {L:4d}: " + "  ".join(f"{m}: {v:.2f}" for m, v in res.items()))
