#!/usr/bin/env python
"""Model of the ASYNCHRONOUS lockstep for chains that share a wave — prepared for round 4.

Two things, both in plain Python on top of the second oracle (oracle/ahmc_ref.py: same arithmetic, same Philox draws):

1. `GroupMachine` — the NUTS transition (MultinomialTS + GeneralisedNoUTurn, src/trajectory.jl:626-742) as a RESUMABLE state
   machine whose `step()` is exactly ONE leapfrog: a tree leaf with its merges / park / end of subtree / end of doubling, or one
   step of the candidate's re-integration, with the end of the transition and the prologue of the next one folded into the
   step that completes it.  It is `k_nuts`' iterative formulation (ahmc_nuts.hpp: leaf loop, merge_level, park, top of the
   doubling, re-integration by leaf index) with every loop counter made a member — i.e. what a lane group of the asynchronous
   kernel carries.  `check_machine()` runs it against `ahmc_ref.nuts_transition` chain by chain: same candidates, same
   statistics, bit for bit.
2. `simulate_wave()` — CPW machines advanced together, one `step()` each per wave step, against today's schedule (all groups at
   the same doubling / leaf, a transition ends when its longest tree and its longest re-integration have ended), priced with
   the VALU counts of DESIGN §7 item 1.  Reports wave-instructions per chain-leapfrog for: today's lockstep, the asynchronous
   schedule with per-group merge counts (a wave step runs max over groups of them), and the same with new transitions started
   on multiples of `align` wave steps.

    python experiments/r4/async_lockstep_model.py            # check + the estimate on funnel chains
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ahmc_ref as R  # noqa: E402

TREE, REINT = 0, 1


def ctz(n):
    return (n & -n).bit_length() - 1


class GroupMachine:
    """one chain; `step()` = one leapfrog.  Events of the last step are left in `self.ev` for the cost model:
    merges (count), top (end of a doubling), turn (end of a transition + prologue of the next), kind (TREE / REINT)."""

    def __init__(self, seed, chain, h, eps, theta0, n_transitions, max_depth=10, delta_max=1000.0, iteration0=0):
        self.seed, self.chain, self.h, self.eps, self.max_depth, self.delta_max = seed, chain, h, eps, max_depth, delta_max
        self.n_transitions, self.kt, self.iteration0 = n_transitions, 0, iteration0
        self.z = R.phasepoint(h, list(theta0), [0.0] * len(theta0))
        self.draws, self.stats = [], []
        self.finished = False
        self._prologue()

    # ---- transition prologue (src/sampler.jl:54-57) + the set-up of the first doubling ----
    def _prologue(self):
        self.rng = R.Rng(self.seed, self.chain, self.iteration0 + self.kt)
        z0 = R.refresh(self.rng, self.h, self.z)
        self.z0, self.H0 = z0, R.energy(z0)
        self.cur = self.oth = z0
        self.pos_cur = self.pos_oth = 0
        self.cur_is_left = False
        self.w_tree, self.sa_tree, self.na_tree, self.dh_tree, self.ck_tree = 0.0, 0.0, 0, 0.0, 0
        self.A_tree = list(z0.r)
        self.depth, self.numerical = 0, False
        self.jw = 0
        self.LA, self.LRF, self.S = {}, {}, {}
        self._begin_doubling()
        self.phase = TREE

    def _begin_doubling(self):
        vleft = self.rng.rand_bool()  # direction (:693)
        self.v = -1 if vleft else 1
        if vleft != self.cur_is_left:
            if self.jw > 0:
                self.cur, self.oth = self.oth, self.cur
            self.pos_cur, self.pos_oth = self.pos_oth, self.pos_cur
            self.cur_is_left = vleft
        self.leaf, self.nleaf = 0, 1 << self.jw
        self.sub_term = False

    def _dots(self, A, ra, rb):
        h = self.h
        return R.dot(A, h.dHdr(ra)), R.dot(A, h.dHdr(rb))

    def step(self):
        assert not self.finished
        self.ev = dict(kind=self.phase, merges=0, top=0, turn=0)
        if self.phase == REINT:
            self.zc = R.step(self.eps, self.h, self.zc, 1 if self.ck_tree > 0 else -1)
            self.re_left -= 1
            if self.re_left == 0:
                self._epilogue()
            return
        # ---- one leaf (:638-647) ----
        h, v = self.h, self.v
        self.leaf += 1
        leaf, nm = self.leaf, ctz(self.leaf)
        self.cur = R.step(self.eps, h, self.cur, v)
        self.pos_cur += v
        ne = R.neg_energy(self.cur)
        dH = -ne - self.H0
        sa_c, na_c, dh_c, ck_c, w_c = math.exp(R.jl_min(0.0, -dH)), 1, dH, self.pos_cur, self.H0 + ne
        sub_term = not (-self.H0 < self.delta_max + ne)  # Termination(::MultinomialTS, …) (:503-507)
        self.numerical = self.numerical or sub_term
        A_c, RF_c = self.cur.r, self.cur.r
        merged = 0
        for lvl in range(nm):  # merges after this leaf: one per trailing zero bit of `leaf` (:649-673)
            if sub_term:
                break
            self.ev["merges"] += 1
            A_p, RF_p = self.LA[lvl], self.LRF[lvl]
            w_p, sa_p, na_p, dh_p, ck_p = self.S[lvl]
            w_new = R.logaddexp(w_p, w_c)
            if w_new < w_p + self.rng.randexp():  # keep the first-built half's candidate (:191-195)
                ck_c = ck_p
            w_c = w_new
            sa_c, na_c = sa_p + sa_c, na_p + na_c
            dh_c = R.maxabs(dh_p, dh_c) if v > 0 else R.maxabs(dh_c, dh_p)
            A_c = [a + b for a, b in zip(A_p, A_c)]
            d0, d1 = self._dots(A_c, RF_p, self.cur.r)
            sub_term = d0 <= 0 or d1 <= 0
            RF_c = RF_p
            merged = lvl + 1
        end_subtree = False
        if sub_term:
            # enclosing unfinished subtrees still absorb the statistics of their first halves (:666)
            pend = ((leaf - 1) >> merged) << merged
            q = merged
            while (pend >> q) != 0:
                if (pend >> q) & 1:
                    _, sa_p, na_p, dh_p, _ = self.S[q]
                    sa_c, na_c = sa_p + sa_c, na_p + na_c
                    dh_c = R.maxabs(dh_p, dh_c) if v > 0 else R.maxabs(dh_c, dh_p)
                q += 1
            end_subtree = True
        elif leaf < self.nleaf:
            self.LA[nm], self.LRF[nm] = A_c, RF_c  # park the finished level-nm subtree until its sibling is built
            self.S[nm] = (w_c, sa_c, na_c, dh_c, ck_c)
        else:
            end_subtree = True
        if not end_subtree:
            return
        # ---- top level of the doubling loop (:708-722) ----
        self.ev["top"] = 1
        if not sub_term:
            self.depth += 1
            if self.w_tree < w_c + self.rng.randexp():  # mh_accept: biased progressive sampling (:202-206)
                self.ck_tree = ck_c
        self.sa_tree += sa_c
        self.na_tree += na_c
        self.dh_tree = R.maxabs(dh_c, self.dh_tree) if v < 0 else R.maxabs(self.dh_tree, dh_c)
        self.w_tree = R.logaddexp(self.w_tree, w_c)
        self.A_tree = [a + b for a, b in zip(self.A_tree, A_c)]
        d0, d1 = self._dots(self.A_tree, self.cur.r, self.oth.r)
        turn = d0 <= 0 or d1 <= 0
        self.jw += 1
        if sub_term or turn or self.jw >= self.max_depth:
            # Transition(zcand, stats) (:725-741): the candidate is a leaf INDEX; re-integrate from z0
            self.re_left = abs(self.ck_tree)
            self.zc = self.z0
            if self.re_left == 0:
                self._epilogue()
            else:
                self.phase = REINT
        else:
            self._begin_doubling()

    def _epilogue(self):
        self.ev["turn"] = 1
        zc = self.zc
        H = R.energy(zc)
        self.stats.append(dict(n_steps=self.na_tree, acceptance_rate=self.sa_tree / self.na_tree, log_density=zc.lp, hamiltonian_energy=H,
                               hamiltonian_energy_error=H - self.H0, max_hamiltonian_energy_error=self.dh_tree, tree_depth=self.depth,
                               numerical_error=self.numerical))
        self.draws.append((list(zc.theta), list(zc.r)))
        self.z = zc
        self.kt += 1
        if self.kt >= self.n_transitions:
            self.finished = True
        else:
            self._prologue()


def check_machine(n_chains=6, n_transitions=12, D=8, eps=0.35, target="funnel", seed=0x5EED0003):
    """GroupMachine == ahmc_ref.nuts_transition (candidate and every statistic, bit for bit)"""
    import random

    rnd = random.Random(1)
    fn = R.funnel if target == "funnel" else R.iso_gaussian
    h = R.Hamiltonian([1.0] * D, fn, D)
    nt = R.NUTS(R.MultinomialTS, R.GENERALISED, eps)
    total = 0
    for c in range(n_chains):
        th0 = [rnd.random() for _ in range(D)]
        draws, stats = R.sample_chain(seed, c, h, nt, th0, n_transitions)
        m = GroupMachine(seed, c, h, eps, th0, n_transitions)
        while not m.finished:
            m.step()
        assert len(m.stats) == n_transitions
        for k, (a, b) in enumerate(zip(stats, m.stats)):
            for key in b:
                assert a[key] == b[key], (c, k, key, a[key], b[key])
            assert draws[k][0] == m.draws[k][0], (c, k, "theta")
        total += sum(s["n_steps"] for s in stats)
    return total


# VALU wave-instructions (`k_nuts<double,64,2,0,0>`; the (16,2) instantiation is ≈ 10 % leaner per part):
COST = dict(leaf=89, merge=80, park=8, top=95, turn=585, reint=40)


def simulate_wave(machines_factory, cpw=4, align=0):
    """→ dict(today=…, async_=…) wave-instructions per chain-leapfrog.  `machines_factory()` returns `cpw` fresh machines."""
    # --- today's schedule: groups share (transition, doubling, leaf); replay each chain alone and combine per transition ---
    ms = machines_factory()
    per = []  # per chain: list over transitions of (list over doublings of (leaves, merges per leaf list), reint steps)
    for m in ms:
        tr, cur_d, leaves, merges, re = [], [], 0, [], 0
        while not m.finished:
            m.step()
            e = m.ev
            if e["kind"] == TREE:
                leaves += 1
                merges.append(e["merges"])
                if e["top"]:
                    cur_d.append((leaves, merges))
                    leaves, merges = 0, []
            else:
                re += 1
            if e["turn"]:
                tr.append((cur_d, re))
                cur_d, re = [], 0
        per.append(tr)
    n_tr = min(len(t) for t in per)
    lf = sum(sum(lv for lv, _ in per[g][k][0]) for g in range(cpw) for k in range(n_tr))
    today = 0
    for k in range(n_tr):
        nd = max(len(per[g][k][0]) for g in range(cpw))
        for d in range(nd):
            nl = max((per[g][k][0][d][0] if d < len(per[g][k][0]) else 0) for g in range(cpw))
            for i in range(1, nl + 1):
                # a leaf step runs while any group is still building; its merges: one per trailing zero (any group merging)
                mg = max((per[g][k][0][d][1][i - 1] if d < len(per[g][k][0]) and i <= per[g][k][0][d][0] else 0) for g in range(cpw))
                today += COST["leaf"] + COST["merge"] * mg + COST["park"]
            today += COST["top"]
        today += COST["reint"] * max(per[g][k][1] for g in range(cpw)) + COST["turn"]
    # --- asynchronous: one step() of every unfinished group per wave step ---
    ms = machines_factory()
    asyn, wave_steps, wait = 0, 0, [0] * cpw
    done_tr = [0] * cpw
    while any((not m.finished) and done_tr[g] < n_tr for g, m in enumerate(ms)):
        kinds, mg, top, turn = [], 0, 0, 0
        for g, m in enumerate(ms):
            if m.finished or done_tr[g] >= n_tr:
                continue
            if wait[g] > 0:      # (alignment: a group that has just started a transition idles until the wave step count
                wait[g] -= 1     #  is a multiple of `align`)
                continue
            m.step()
            e = m.ev
            kinds.append(e["kind"])
            mg = max(mg, e["merges"])
            top |= e["top"]
            if e["turn"]:
                turn = 1
                done_tr[g] += 1
                if align:
                    wait[g] = (-(wave_steps + 1)) % align
        wave_steps += 1
        if kinds:
            # one leapfrog for everybody (leaf-priced when any group is on a leaf), merges = max over groups, tops / turns by the whole wave
            asyn += (COST["leaf"] if TREE in kinds else COST["reint"]) + COST["merge"] * mg + COST["park"] + COST["top"] * top + COST["turn"] * turn
        else:
            asyn += 4
    return dict(chain_leapfrogs=lf, today=today / lf, async_=asyn / lf, wave_steps=wave_steps, transitions=n_tr)


def main():
    n = check_machine()
    print(f"GroupMachine == ahmc_ref.nuts_transition on funnel chains ({n} leapfrogs compared)")
    n = check_machine(target="iso", eps=0.6, D=16, n_chains=4)
    print(f"GroupMachine == ahmc_ref.nuts_transition on iso Gaussian chains ({n} leapfrogs compared)")
    import random

    D, eps, n_tr = 32, 0.09, int(os.environ.get("TRANSITIONS", 40))
    h = R.Hamiltonian([1.0] * D, R.funnel, D)
    for cpw in (4,):
        for wave in range(int(os.environ.get("WAVES", 3))):
            rnd = random.Random(100 + wave)
            th0 = [[rnd.random() for _ in range(D)] for _ in range(cpw)]

            def factory():
                return [GroupMachine(0x5EED0003, 1000 * wave + g, h, eps, th0[g], n_tr, iteration0=7) for g in range(cpw)]
            for align in (0, 8, 32):
                r = simulate_wave(factory, cpw, align)
                print(f"wave {wave} (D={D} funnel, eps={eps}, {cpw} chains, {r['transitions']} transitions, {r['chain_leapfrogs']} leapfrogs): "
                      f"today {r['today']:.1f}  asynchronous(align={align}) {r['async_']:.1f} wave-instructions per chain-leapfrog")


if __name__ == "__main__":
    main()
