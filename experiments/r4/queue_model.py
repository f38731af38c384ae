#!/usr/bin/env python
"""Event-driven model of PERSISTENT waves fed from a priority queue (CPU only; traces from cfg3_trace.py): an item is the next
block of B transitions of a group of 4 chains (or of a single chain, four of which a wave pops together), its priority the
estimated remaining work (blocks left × the work of its previous block); a wave that finishes an item pushes the successor
and pops the heaviest item waiting.  No launch barrier, fresh predictions — against one launch ordered by step size."""
import heapq
import os

import numpy as np

d = np.load(os.environ.get('TRACE', '/tmp/cfg3_trace.npz')); Wa, Wd, eps = d['Wa'].astype(np.int64), d['Wd'].astype(np.int64), d['eps']
n_dr, N = Wd.shape
SLOTS = N // 16
REINT = 0.5
ideal = Wd.sum() * (1 + REINT) / 4 / SLOTS

def sim(B, regroup, prio="last", cpw=4, turn_cost=0.0):
    """items: fixed groups of 4 chains (regroup=False) or single chains popped four at a time from the heaviest bucket (True).
    prio: 'last' = work of the item's previous block; 'none' = FIFO."""
    nb = n_dr // B
    blk = Wd[:nb*B].reshape(nb, B, N)                     # (block, transition, chain)
    t_now = 0.0
    slots = [(0.0, s) for s in range(SLOTS)]              # (time free, id)
    heapq.heapify(slots)
    busy_area = 0.0
    useful = 0.0
    if not regroup:
        order = np.argsort(eps).reshape(-1, cpw)          # groups formed once, by step size
        ng = order.shape[0]
        nextb = np.zeros(ng, int)
        last = np.zeros(ng)                               # priority
        ready = [(-0.0, g) for g in range(ng)]            # max-heap on priority
        heapq.heapify(ready)
        events = []                                       # (time, group) completions
        running = 0
        end = 0.0
        # initial fill: groups in eps order (heavy first = small eps)
        ready = [(-1e30 + i, g) for i, g in enumerate(range(ng))]  # every group once first, in eps order
        heapq.heapify(ready)
        free = SLOTS
        t = 0.0
        while ready or events:
            while free and ready:
                _, g = heapq.heappop(ready)
                b = nextb[g]
                w = blk[b][:, order[g]].max(axis=1).sum() * (1 + REINT) + turn_cost * B
                heapq.heappush(events, (t + w, g, w))
                busy_area += w
                free -= 1
            t, g, w = heapq.heappop(events)
            free += 1
            end = max(end, t)
            nextb[g] += 1
            if nextb[g] < nb:
                pr = -w * (nb - nextb[g]) if prio == "last" else t   # estimated remaining work
                heapq.heappush(ready, (pr, g))
        return end, busy_area
    else:
        nextb = np.zeros(N, int)
        lastw = np.zeros(N)
        # ready chains keyed by priority (heaviest recent work first); a wave takes the 4 heaviest ready chains
        ready = [(-1e30 + i, c) for i, c in enumerate(np.argsort(eps))]
        heapq.heapify(ready)
        events = []
        free = SLOTS
        t = 0.0
        end = 0.0
        while ready or events:
            while free and len(ready) >= cpw or (free and ready and not events):
                take = [heapq.heappop(ready)[1] for _ in range(min(cpw, len(ready)))]
                # the chains of a wave need not be at the same block: each runs ITS next block of B transitions, in lockstep
                mats = np.stack([blk[nextb[c]][:, c] for c in take], axis=1)
                w = mats.max(axis=1).sum() * (1 + REINT) + turn_cost * B
                heapq.heappush(events, (t + w, tuple(take), w, tuple(mats.sum(0))))
                busy_area += w
                free -= 1
            if not events:
                break
            t, take, w, own = heapq.heappop(events)
            free += 1
            end = max(end, t)
            for c, ow in zip(take, own):
                nextb[c] += 1
                if nextb[c] < nb:
                    heapq.heappush(ready, (-ow * (nb - nextb[c]) if prio == "last" else t, c))
        return end, busy_area

print("today (one launch, by eps): makespan/ideal 5.41, lockstep 0.367, fill 0.50  [pack.py]")
for B in (8, 16, 32, 64, 125):
    for regroup in (False, True):
        end, area = sim(B, regroup)
        nbt = (n_dr // B) * B
        W = Wd[:nbt].sum() * (1 + REINT)
        print(f"B={B:4d} {'chains re-grouped per block' if regroup else 'fixed groups            '}: makespan/ideal {end/(W/4/SLOTS):.2f}  lockstep {W/(4*area):.3f}  fill {area/(SLOTS*end):.3f}")
