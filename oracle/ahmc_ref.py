"""ahmc_ref.py — a SECOND, independent restatement of the reference's dynamic and static transitions.

TEST INFRASTRUCTURE ONLY (like everything under oracle/).  Transcribed from the Julia source of
AdvancedHMC.jl v0.8.6 — struct for struct, function for function, recursion and all, in plain Python with
float64 scalars and lists — not from oracle/ahmc_oracle.cpp.  Its job is to
pin the C++ oracle's CONTROL logic (tree building, sampler combination, termination, statistics): the two
restatements must agree bit for bit on the same Philox streams (tests/test_oracle_cross.py).  What the two
share by construction is only the specification that is ours, not the reference's: the Philox4x32-10 stream
layout of include/ahmc_hip.h (`ahmc_seed`) and the order in which a D-vector is summed (index order).

Citations are path:line in the AdvancedHMC.jl checkout.
"""
import math

INF = float("inf")
RNG_MOMENTUM, RNG_TRANSITION, RNG_JITTER = 0, 1, 2
M32 = 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al. 2011) and the stream layout: counter = (chain, iteration, purpose, slot),
# key = (seed_lo, seed_hi)
# ------------------------------------------------------------------------------------------------
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


class Rng:
    def __init__(self, seed, chain, iteration):
        self.k0, self.k1, self.chain, self.iter = seed & M32, (seed >> 32) & M32, chain & M32, iteration & M32
        self.draw = 0  # sequential scalar draws of one transition

    def raw(self, purpose, slot):
        return philox4x32_10(self.chain, self.iter, purpose, slot & M32, self.k0, self.k1)

    @staticmethod
    def u53(hi, lo):
        return (float(((hi >> 5) << 26) | (lo >> 6)) + 0.5) * (1.0 / 9007199254740992.0)

    def uniform(self, purpose, slot):
        p = self.raw(purpose, slot)
        return self.u53(p[0], p[1])

    def normal(self, purpose, d):  # Box–Muller on pair d // 2
        p = self.raw(purpose, d >> 1)
        u1, u2 = self.u53(p[0], p[1]), self.u53(p[2], p[3])
        rad = math.sqrt(-2.0 * math.log(u1))
        ang = 6.283185307179586476925286766559 * u2
        return rad * math.sin(ang) if (d & 1) else rad * math.cos(ang)

    # the `rng` argument of the dynamic transition: one 32-bit word per draw, in call order
    def _word(self):
        k = self.draw
        self.draw += 1
        return self.raw(RNG_TRANSITION, k >> 2)[k & 3]

    def rand(self):
        return (float(self._word()) + 0.5) * 2.3283064365386962890625e-10

    def rand_bool(self):
        return (self._word() >> 31) != 0

    def randexp(self):
        return -math.log(self.rand())


# ------------------------------------------------------------------------------------------------
# LogExpFunctions.logaddexp
# ------------------------------------------------------------------------------------------------
def logaddexp(x, y):
    if x == y:
        delta = 0.0
    else:
        delta = abs(x - y)
    mx = x if x > y else y
    if math.isnan(x) or math.isnan(y):
        mx = float("nan")
    return mx + math.log1p(math.exp(-delta))


def jl_min(a, b):  # Base.min propagates NaN
    if math.isnan(a) or math.isnan(b):
        return float("nan")
    return a if a < b else b


def dot(a, b):
    s = 0.0
    for x, y in zip(a, b):
        s += x * y
    return s


# ------------------------------------------------------------------------------------------------
# Hamiltonian: metric (src/metric.jl, src/hamiltonian.jl:50-59,155-177) + target
# ------------------------------------------------------------------------------------------------
def cholesky_upper(M):
    """U upper triangular with UᵀU = M (cholesky(Symmetric(M⁻¹)).U, src/metric.jl:104-109), column by column"""
    D = len(M)
    U = [[0.0] * D for _ in range(D)]
    for j in range(D):
        for i in range(j + 1):
            s = M[i][j]
            for k in range(i):
                s -= U[k][i] * U[k][j]
            U[i][j] = math.sqrt(s) if i == j else s / U[i][i]
    return U


class Hamiltonian:
    def __init__(self, minv, logdensity_and_gradient, D=None):
        # None = UnitEuclideanMetric; list of floats = DiagEuclideanMetric's M⁻¹; list of rows = DenseEuclideanMetric's M⁻¹
        self.minv = minv
        self.fn = logdensity_and_gradient
        self.D = len(minv) if minv is not None else D
        self.dense = minv is not None and isinstance(minv[0], list)
        self.chol = cholesky_upper(minv) if self.dense else None

    def dHdr(self, r):  # ∂H∂r (:50-68)
        if self.minv is None:
            return list(r)
        if self.dense:  # M⁻¹ * r
            out = []
            for row in self.minv:
                s = 0.0
                for m, x in zip(row, r):
                    s += m * x
                out.append(s)
            return out
        return [m * x for m, x in zip(self.minv, r)]

    def neg_energy_r(self, r):  # neg_energy(h, r, θ) (:155-184)
        s = 0.0
        if self.minv is None:
            for x in r:
                s += x * x
        elif self.dense:  # mul!(_temp, M⁻¹, r); -dot(r, _temp) / 2
            s = dot(r, self.dHdr(r))
        else:
            for m, x in zip(self.minv, r):
                s += (x * x) * m
        return -s / 2

    def dHdtheta(self, theta):  # ∂H∂θ (:45-48): DualValue(ℓπ, -∇ℓπ)
        v, g = self.fn(theta)
        return v, [-x for x in g]

    def rand_momentum(self, rng):  # src/metric.jl:290-309
        z = [rng.normal(RNG_MOMENTUM, d) for d in range(self.D)]
        if self.minv is None:
            return z
        if self.dense:  # ldiv!(cholM⁻¹, r) (:311-320): back substitution with the upper factor
            for i in range(self.D - 1, -1, -1):
                s = z[i]
                for j in range(i + 1, self.D):
                    s -= self.chol[i][j] * z[j]
                z[i] = s / self.chol[i][i]
            return z
        return [x / math.sqrt(m) for x, m in zip(z, self.minv)]


def iso_gaussian(theta):  # test/common.jl:40-44, 52-56 with m = 0, s = 1
    log2pi = 1.8378770664093454835606594728112
    v = 0.0
    for x in theta:
        v += -(log2pi + x * x) / 2
    return v, [-x for x in theta]


def funnel(theta):  # research/notebooks/geweke_test.ipynb cell 4 (the arithmetic of include/ahmc_hip.h's family)
    log2pi = 1.8378770664093454835606594728112
    D = len(theta)
    y = theta[0]
    ss = 0.0
    for x in theta[1:]:
        ss += x * x
    try:
        ey = math.exp(-y)
    except OverflowError:
        ey = INF
    nm1 = float(D - 1)
    v = -(log2pi + 2 * math.log(3.0) + y * y / 9) / 2 - nm1 * (log2pi + y) / 2 - ss * ey / 2
    g = [-y / 9 - nm1 / 2 + ss * ey / 2] + [-x * ey for x in theta[1:]]
    return v, g


class PhasePoint:  # src/hamiltonian.jl:88-107
    def __init__(self, theta, r, lp, g, lk):
        self.theta, self.r, self.g = theta, r, g
        self.lp = lp if math.isfinite(lp) else -INF  # non-finite values → -Inf (:95-104)
        self.lk = lk if math.isfinite(lk) else -INF

    def isfinite(self, h):  # :141-142: values and gradients of ℓπ and ℓκ
        return (math.isfinite(self.lp) and math.isfinite(self.lk) and all(math.isfinite(x) for x in self.g)
                and all(math.isfinite(x) for x in h.dHdr(self.r)))


def phasepoint(h, theta, r, lp=None, g=None):  # :115-119
    if lp is None:
        lp, g = h.dHdtheta(theta)
    return PhasePoint(theta, r, lp, g, h.neg_energy_r(r))


def energy(z):  # :149,194
    return -(z.lp + z.lk)


def neg_energy(z):
    return z.lp + z.lk


# ------------------------------------------------------------------------------------------------
# step (src/integrator.jl:216-265), Leapfrog only
# ------------------------------------------------------------------------------------------------
def temper(alpha, r, i, is_half, n_steps):  # TemperedLeapfrog (src/integrator.jl:198-209); alpha None = the other integrators
    if alpha is None:
        return r
    i_temper = 2 * (i - 1) + 1 + (0 if is_half else 1)
    s = math.sqrt(alpha)
    return [x * s for x in r] if i_temper <= n_steps else [x / s for x in r]


def step(eps, h, z, n_steps=1, full_trajectory=False, alpha=None):
    fwd = n_steps > 0
    n_steps = abs(n_steps)
    e = eps if fwd else -eps
    res = []
    theta, r, value, gradient = z.theta, z.r, z.lp, z.g
    for i in range(1, n_steps + 1):
        r = temper(alpha, r, i, True, n_steps)
        r = [a - e / 2 * b for a, b in zip(r, gradient)]
        dr = h.dHdr(r)
        theta = [a + e * b for a, b in zip(theta, dr)]
        value, gradient = h.dHdtheta(theta)
        r = [a - e / 2 * b for a, b in zip(r, gradient)]
        r = temper(alpha, r, i, False, n_steps)
        z = phasepoint(h, theta, r, value, gradient)
        res.append(z)
        if not z.isfinite(h):
            break
    return res if full_trajectory else z


# ------------------------------------------------------------------------------------------------
# trajectory samplers (src/trajectory.jl:90-206)
# ------------------------------------------------------------------------------------------------
class SliceTS:
    def __init__(self, zcand, lu, n):
        self.zcand, self.lu, self.n = zcand, lu, n

    @classmethod
    def initial(cls, rng, z0):  # :143-144
        return cls(z0, neg_energy(z0) - rng.randexp(), 1)

    @classmethod
    def leaf(cls, s, H0, zcand):  # :163-165
        return cls(zcand, s.lu, int(s.lu <= neg_energy(zcand)))

    @staticmethod
    def combine_rng(rng, s1, s2):  # :178-183
        n = s1.n + s2.n
        zcand = s1.zcand if n * rng.rand() < s1.n else s2.zcand
        return SliceTS(zcand, s1.lu, n)

    @staticmethod
    def combine_z(zcand, s1, s2):  # :185-189
        return SliceTS(zcand, s1.lu, s1.n + s2.n)

    @staticmethod
    def mh_accept(rng, s, sp):  # :202
        return s.n * rng.rand() < sp.n

    def termination(self, delta_max, H0, Hp):  # :500-502
        return Termination(False, not (self.lu < delta_max + -Hp))


class MultinomialTS:
    def __init__(self, zcand, lw):
        self.zcand, self.lw = zcand, lw

    @classmethod
    def initial(cls, rng, z0):  # :155
        return cls(z0, 0.0)

    @classmethod
    def leaf(cls, s, H0, zcand):  # :174-176
        return cls(zcand, H0 + neg_energy(zcand))

    @staticmethod
    def combine_rng(rng, s1, s2):  # :191-195
        lw = logaddexp(s1.lw, s2.lw)
        zcand = s1.zcand if lw < s1.lw + rng.randexp() else s2.zcand
        return MultinomialTS(zcand, lw)

    @staticmethod
    def combine_z(zcand, s1, s2):  # :197-200
        return MultinomialTS(zcand, logaddexp(s1.lw, s2.lw))

    @staticmethod
    def mh_accept(rng, s, sp):  # :204-206
        return s.lw < sp.lw + rng.randexp()

    def termination(self, delta_max, H0, Hp):  # :503-507
        return Termination(False, not (-H0 < delta_max + -Hp))


# ------------------------------------------------------------------------------------------------
# Termination, BinaryTree, U-turn criteria (src/trajectory.jl:454-623)
# ------------------------------------------------------------------------------------------------
class Termination:
    def __init__(self, dynamic, numerical):
        self.dynamic, self.numerical = dynamic, numerical

    def __mul__(self, o):  # :493-495
        return Termination(self.dynamic or o.dynamic, self.numerical or o.numerical)

    def isterminated(self):
        return self.dynamic or self.numerical


class BinaryTree:  # :512-520
    def __init__(self, zleft, zright, rho, sum_alpha, n_alpha, dH_max):
        self.zleft, self.zright, self.rho, self.sum_alpha, self.n_alpha, self.dH_max = zleft, zright, rho, sum_alpha, n_alpha, dH_max


def maxabs(a, b):  # :526
    return a if abs(a) > abs(b) else b


def combine_trees(tl, tr):  # :533-542, TurnStatistic combine :466-467
    rho = None if tl.rho is None else [a + b for a, b in zip(tl.rho, tr.rho)]
    return BinaryTree(tl.zleft, tr.zright, rho, tl.sum_alpha + tr.sum_alpha, tl.n_alpha + tr.n_alpha, maxabs(tl.dH_max, tr.dH_max))


def generalised_uturn_criterion(rho, p_sharp_minus, p_sharp_plus):  # :619-621
    return (dot(rho, p_sharp_minus) <= 0) or (dot(rho, p_sharp_plus) <= 0)


CLASSIC, GENERALISED, STRICT = 0, 1, 2


def isterminated_tree(tc, h, t, tleft, tright):
    if tc == CLASSIC:  # :551-557
        z0, z1 = t.zleft, t.zright
        dth = [b - a for a, b in zip(z0.theta, z1.theta)]
        s = (dot(dth, h.dHdr([-x for x in z0.r])) >= 0) or (dot([-x for x in dth], h.dHdr(z1.r)) >= 0)
        return Termination(s, False)
    s1 = Termination(generalised_uturn_criterion(t.rho, h.dHdr(t.zleft.r), h.dHdr(t.zright.r)), False)  # :566-570
    if tc == GENERALISED:
        return s1
    # StrictGeneralisedNoUTurn (:579-617)
    rho2 = [a + b for a, b in zip(tleft.rho, tright.zleft.r)]
    s2 = Termination(generalised_uturn_criterion(rho2, h.dHdr(t.zleft.r), h.dHdr(tright.zleft.r)), False)
    rho3 = [a + b for a, b in zip(tleft.zright.r, tright.rho)]
    s3 = Termination(generalised_uturn_criterion(rho3, h.dHdr(tleft.zright.r), h.dHdr(t.zright.r)), False)
    return s1 * s2 * s3


# ------------------------------------------------------------------------------------------------
# build_tree (:626-675) and the dynamic transition (:677-742)
# ------------------------------------------------------------------------------------------------
class NUTS:
    def __init__(self, TS, tc, eps, max_depth=10, delta_max=1000.0, temper_alpha=None):
        self.TS, self.tc, self.eps, self.max_depth, self.delta_max, self.temper_alpha = TS, tc, eps, max_depth, delta_max, temper_alpha


def build_tree(rng, nt, h, z, sampler, v, j, H0):
    if j == 0:
        zp = step(nt.eps, h, z, v, alpha=nt.temper_alpha)
        Hp = energy(zp)
        dH = Hp - H0
        alpha = math.exp(jl_min(0.0, -dH))
        sp = nt.TS.leaf(sampler, H0, zp)
        rho = None if nt.tc == CLASSIC else zp.r
        return BinaryTree(zp, zp, rho, alpha, 1, dH), sp, sp.termination(nt.delta_max, H0, Hp)
    tree1, sampler1, term1 = build_tree(rng, nt, h, z, sampler, v, j - 1, H0)
    if not term1.isterminated():
        if v == -1:
            tree2, sampler2, term2 = build_tree(rng, nt, h, tree1.zleft, sampler, v, j - 1, H0)
            tl, tr = tree2, tree1
        else:
            tree2, sampler2, term2 = build_tree(rng, nt, h, tree1.zright, sampler, v, j - 1, H0)
            tl, tr = tree1, tree2
        tree1 = combine_trees(tl, tr)
        sampler1 = nt.TS.combine_rng(rng, sampler1, sampler2)
        term1 = term1 * term2 * isterminated_tree(nt.tc, h, tree1, tl, tr)
    return tree1, sampler1, term1


def nuts_transition(rng, h, nt, z0):
    H0 = energy(z0)
    tree = BinaryTree(z0, z0, None if nt.tc == CLASSIC else z0.r, 0.0, 0, 0.0)
    sampler = nt.TS.initial(rng, z0)
    termination = Termination(False, False)
    zcand = z0
    j = 0
    while not termination.isterminated() and j < nt.max_depth:
        vleft = rng.rand_bool()
        if vleft:
            treep, samplerp, termp = build_tree(rng, nt, h, tree.zleft, sampler, -1, j, H0)
            tl, tr = treep, tree
        else:
            treep, samplerp, termp = build_tree(rng, nt, h, tree.zright, sampler, 1, j, H0)
            tl, tr = tree, treep
        if not termp.isterminated():
            j = j + 1
            if nt.TS.mh_accept(rng, sampler, samplerp):
                zcand = samplerp.zcand
        tree = combine_trees(tl, tr)
        sampler = nt.TS.combine_z(zcand, sampler, samplerp)
        termination = termination * termp * isterminated_tree(nt.tc, h, tree, tl, tr)
    H = energy(zcand)
    stat = dict(n_steps=tree.n_alpha, is_accept=True, acceptance_rate=tree.sum_alpha / tree.n_alpha, log_density=zcand.lp,
                hamiltonian_energy=H, hamiltonian_energy_error=H - H0, max_hamiltonian_energy_error=tree.dH_max, tree_depth=j,
                numerical_error=termination.numerical)
    return zcand, stat


# ------------------------------------------------------------------------------------------------
# static transition, EndPointTS (src/trajectory.jl:271-340, mh_accept_ratio :855-880)
# ------------------------------------------------------------------------------------------------
def hmc_transition(rng, h, eps, L, z, temper_alpha=None):
    H0 = energy(z)
    zp = step(eps, h, z, L, alpha=temper_alpha)
    Hp = energy(zp)
    is_accept = Hp < H0 + (-math.log(rng.uniform(RNG_TRANSITION, 0)))  # :858: one 53-bit draw (static transitions keep 53 bits)
    alpha = jl_min(1.0, math.exp(H0 - Hp))
    zn = zp if is_accept else z
    zn = PhasePoint(zn.theta, [-x for x in zn.r], zn.lp, zn.g, zn.lk)  # :283
    H = energy(zn)
    stat = dict(n_steps=L, is_accept=is_accept, acceptance_rate=alpha, log_density=zn.lp, hamiltonian_energy=H,
                hamiltonian_energy_error=H - H0, numerical_error=not math.isfinite(Hp))
    return zn, stat


# ------------------------------------------------------------------------------------------------
# transition(rng, h, κ, z) (src/sampler.jl:48-58): refresh, then the trajectory's transition
# ------------------------------------------------------------------------------------------------
def refresh(rng, h, z, alpha=0.0):
    """FullMomentumRefreshment (src/hamiltonian.jl:213-220); alpha != 0: PartialMomentumRefreshment(α) (:243-254)"""
    xi = h.rand_momentum(rng)
    if alpha == 0.0:
        return phasepoint(h, z.theta, xi)
    s = math.sqrt(1 - alpha * alpha)
    return phasepoint(h, z.theta, [alpha * a + s * b for a, b in zip(z.r, xi)])


def jitter(rng, eps0, jit):  # JitteredLeapfrog (src/integrator.jl:140-143): one uniform per transition, on its own stream
    return eps0 * (1 + jit * (2 * rng.uniform(RNG_JITTER, 0) - 1))


def sample_chain(seed, chain, h, kernel, theta0, n_transitions, iteration0=0):
    """n_transitions of one chain from θ0; kernel = NUTS(...) or ("hmc", eps, L).  Returns (draws, stats)."""
    z = phasepoint(h, list(theta0), [0.0] * len(theta0))
    draws, stats = [], []
    for it in range(n_transitions):
        rng = Rng(seed, chain, iteration0 + it)
        z = refresh(rng, h, z)
        if isinstance(kernel, NUTS):
            z, st = nuts_transition(rng, h, kernel, z)
        else:
            _, eps, L = kernel
            z, st = hmc_transition(rng, h, eps, L, z)
        draws.append((list(z.theta), list(z.r)))
        stats.append(st)
    return draws, stats


# ------------------------------------------------------------------------------------------------
# static transition, MultinomialTS (src/trajectory.jl:369-390; randcat src/utilities.jl:51-59)
# ------------------------------------------------------------------------------------------------
COUPLED_CHAIN = 0xFFFFFFFF


def hmc_multinomial_transition(rng, h, eps, L, z):
    H0 = energy(z)
    # n_steps_fwd = rand_coupled(rng, 0:n_steps) (:373): ONE draw shared by all chains — the stream of chain 0xFFFFFFFF
    shared = Rng((rng.k1 << 32) | rng.k0, COUPLED_CHAIN, rng.iter)
    n_fwd = int(math.floor(shared.uniform(RNG_TRANSITION, 0) * float(L + 1)))
    n_fwd = min(n_fwd, L)
    zs_fwd = step(eps, h, z, n_fwd, full_trajectory=True)
    zs_bwd = step(eps, h, z, -(L - n_fwd), full_trajectory=True) if L - n_fwd > 0 else []
    zs = list(reversed(zs_bwd)) + [z] + zs_fwd  # vcat(reverse(zs_bwd)..., z, zs_fwd...) (:377)
    lw = [-energy(zz) for zz in zs]  # unnormalised log weights (:379)
    mx = -INF
    for v in lw:
        mx = v if v > mx else mx
    se = 0.0
    for v in lw:
        se += math.exp(v - mx)
    lse = mx + math.log(se)
    u = rng.uniform(RNG_TRANSITION, 0)
    cum, idx = 0.0, 0
    while cum < u and idx < len(zs):  # randcat: the first index whose cumulated probability reaches u
        cum += math.exp(lw[idx] - lse)
        idx += 1
    zp = zs[max(idx, 1) - 1]
    sa = 0.0
    for v in lw:
        sa += math.exp(jl_min(0.0, -((-v) - H0)))
    alpha = sa / float(len(lw))  # mean MH acceptance over the trajectory (:385-388)
    zn = PhasePoint(zp.theta, [-x for x in zp.r], zp.lp, zp.g, zp.lk)
    H = energy(zn)
    stat = dict(n_steps=L, is_accept=True, acceptance_rate=alpha, log_density=zn.lp, hamiltonian_energy=H,
                hamiltonian_energy_error=H - H0, numerical_error=not math.isfinite(energy(zp)))
    return zn, stat


# ------------------------------------------------------------------------------------------------
# adaptation (src/adaptation/stepsize.jl, massmatrix.jl, stan_adaptor.jl; glue src/sampler.jl:3-22,72-90), one chain
# ------------------------------------------------------------------------------------------------
class DualAveraging:  # NesterovDualAveraging(δ, ϵ) (stepsize.jl:168-172) with DAState (:25-31)
    def __init__(self, delta, eps, gamma=0.05, t0=10.0, kappa=0.75):
        self.delta, self.gamma, self.t0, self.kappa = delta, gamma, t0, kappa
        self.m, self.eps, self.mu, self.x_bar, self.H_bar = 0, eps, math.log(10 * eps), 0.0, 0.0

    def adapt(self, alpha):  # adapt_stepsize! (:178-210)
        m = self.m + 1
        eta_H = 1.0 / (m + self.t0)
        H_bar = (1.0 - eta_H) * self.H_bar + eta_H * (self.delta - jl_min(1.0, alpha))
        x = self.mu - H_bar * (math.sqrt(m) / self.gamma)
        eta_x = m ** (-self.kappa)
        x_bar = (1.0 - eta_x) * self.x_bar + eta_x * x
        eps = math.exp(x)
        if not math.isfinite(eps):  # (:198-202)
            m, eps, x_bar, H_bar = self.m, self.eps, self.x_bar, self.H_bar
        self.m, self.eps, self.x_bar, self.H_bar = m, eps, x_bar, H_bar

    def reset(self):  # (:40-46)
        self.m, self.mu, self.x_bar, self.H_bar = 0, math.log(10 * self.eps), 0.0, 0.0

    def finalize(self):  # (:55-58)
        self.eps = math.exp(self.x_bar)


class WelfordVar:  # massmatrix.jl:86-157
    def __init__(self, D, n_min=10):
        self.n, self.n_min, self.mu, self.M, self.var = 0, n_min, [0.0] * D, [0.0] * D, [1.0] * D

    def push(self, s):  # (:141-149)
        self.n += 1
        n = float(self.n)
        delta = [a - b for a, b in zip(s, self.mu)]
        self.mu = [a + d / n for a, d in zip(self.mu, delta)]
        self.M = [a + d * d * ((n - 1) / n) for a, d in zip(self.M, delta)]

    def get_estimation(self):  # (:152-157)
        n = float(self.n)
        return [n / ((n + 5) * (n - 1)) * m + 1e-3 * (5 / (n + 5)) for m in self.M]

    def update(self):  # update!(ve) (:60-62)
        if self.n >= self.n_min:
            self.var = self.get_estimation()

    def reset(self):  # (:134-139)
        self.n = 0
        self.mu = [0.0] * len(self.mu)
        self.M = [0.0] * len(self.M)


class NutpieVar:  # massmatrix.jl:172-250: Welford estimators of the positions and of the gradients
    def __init__(self, D, n_min=10):
        self.pos, self.grad, self.n, self.n_min, self.var = WelfordVar(D, n_min), WelfordVar(D, n_min), 0, n_min, [1.0] * D

    def push_point(self, theta, gradient):  # push!(nv, z::PhasePoint) (:238-243); gradient = z.ℓπ.gradient
        self.n += 1
        self.pos.push(theta)
        self.grad.push(gradient)

    def update(self):
        if self.n >= self.n_min:  # get_estimation (:246-250)
            self.var = [math.sqrt(a / b) for a, b in zip(self.pos.get_estimation(), self.grad.get_estimation())]

    def reset(self):  # (:228-232)
        self.n = 0
        self.pos.reset()
        self.grad.reset()


class WelfordCov:  # massmatrix.jl:283-340
    def __init__(self, D, n_min=10):
        self.n, self.n_min, self.mu = 0, n_min, [0.0] * D
        self.M = [[0.0] * D for _ in range(D)]
        self.var = [[1.0 if i == j else 0.0 for j in range(D)] for i in range(D)]  # (`cov`; named var like the Diag estimators)

    def push(self, s):  # (:323-331)
        self.n += 1
        n = float(self.n)
        delta = [a - b for a, b in zip(s, self.mu)]
        self.mu = [a + d / n for a, d in zip(self.mu, delta)]
        D = len(s)
        for i in range(D):
            for j in range(D):
                self.M[i][j] = self.M[i][j] + (s[i] - self.mu[i]) * delta[j]

    def update(self):  # update!(ce) + get_estimation (:272-281, :334-340)
        if self.n >= self.n_min:
            n = float(self.n)
            D = len(self.mu)
            self.var = [[n / ((n + 5) * (n - 1)) * self.M[i][j] + (1e-3 * (5 / (n + 5)) if i == j else 0.0) for j in range(D)] for i in range(D)]

    def reset(self):
        D = len(self.mu)
        self.n, self.mu, self.M = 0, [0.0] * D, [[0.0] * D for _ in range(D)]


class NaiveHMCAdaptor:  # Adaptation.jl:41-64: both adaptors at every iteration, no windows
    def __init__(self, pc, ssa):
        self.pc, self.ssa = pc, ssa

    def initialize(self, n_adapts):
        pass

    def adapt(self, z, alpha):
        self.ssa.adapt(alpha)
        push_position_or_point(self.pc, z)
        self.pc.update()


def push_position_or_point(pc, z):
    if isinstance(pc, NutpieVar):
        pc.push_point(z.theta, z.g)
    else:
        pc.push(z.theta)


def stan_windows(init_buffer, term_buffer, window_size, n_adapts):  # initialize! (stan_adaptor.jl:13-50)
    window_start, window_end = init_buffer + 1, n_adapts - term_buffer
    splits = []
    next_window = init_buffer + window_size
    while next_window <= window_end:
        if next_window + 2 * window_size > window_end:
            next_window = window_end
        splits.append(next_window)
        window_size *= 2
        next_window += window_size
    if splits and splits[-1] == n_adapts:
        splits.pop()
    return window_start, window_end, splits


class StanHMCAdaptor:  # stan_adaptor.jl:94-159
    def __init__(self, pc, ssa, init_buffer=75, term_buffer=50, window_size=25):
        self.pc, self.ssa, self.ib, self.tb, self.ws = pc, ssa, init_buffer, term_buffer, window_size
        self.i, self.window_start, self.window_end, self.splits = 0, 0, 0, []

    def initialize(self, n_adapts):
        self.window_start, self.window_end, self.splits = stan_windows(self.ib, self.tb, self.ws, n_adapts)

    def adapt(self, z, alpha):  # (:137-159)
        self.i += 1
        self.ssa.adapt(alpha)
        if self.window_start <= self.i <= self.window_end:
            push_position_or_point(self.pc, z)
            if self.i in self.splits:
                self.pc.update()
        if self.i in self.splits:
            self.ssa.reset()
            self.pc.reset()


def sample_chain_adapted(seed, chain, fn, minv0, eps0, kernel_of, theta0, n_samples, n_adapts, delta=0.8, windows=(75, 50, 25),
                         estimator=WelfordVar, naive=False):
    """sample(rng, h, κ, θ, n_samples, StanHMCAdaptor(WelfordVar, NesterovDualAveraging(δ, ϵ)), n_adapts)
    (src/sampler.jl:159-248) for one chain with a DiagEuclideanMetric; kernel_of(eps) -> NUTS(...) or ("hmc", eps, L) /
    ("hmc_mn", eps, L).  Returns (draws, stats, final eps, final M⁻¹)."""
    D = len(theta0)
    h = Hamiltonian([list(row) for row in minv0] if isinstance(minv0[0], list) else list(minv0), fn, D)
    eps = eps0
    adaptor = (NaiveHMCAdaptor(estimator(D), DualAveraging(delta, eps0)) if naive
               else StanHMCAdaptor(estimator(D), DualAveraging(delta, eps0), *windows))
    z = phasepoint(h, list(theta0), [0.0] * D)
    draws, stats = [], []
    for i in range(1, n_samples + 1):
        rng = Rng(seed, chain, i - 1)
        z = refresh(rng, h, z)
        kernel = kernel_of(eps)
        if not isinstance(kernel, NUTS) and kernel[0] == "hmcda":  # FixedIntegrationTime(λ): nsteps (src/trajectory.jl:241-243)
            kernel = ("hmc", eps, max(1, int(math.floor(kernel[2] / eps))))
        if isinstance(kernel, NUTS):
            z, st = nuts_transition(rng, h, kernel, z)
        elif kernel[0] == "hmc":
            z, st = hmc_transition(rng, h, kernel[1], kernel[2], z)
        else:
            z, st = hmc_multinomial_transition(rng, h, kernel[1], kernel[2], z)
        st["step_size"] = eps
        if i <= n_adapts:  # adapt!(h, κ, adaptor, i, n_adapts, z, α) (src/sampler.jl:72-90)
            if i == 1:
                adaptor.initialize(n_adapts)
            adaptor.adapt(z, st["acceptance_rate"])
            if i == n_adapts:
                adaptor.ssa.finalize()
            v = adaptor.pc.var
            h = Hamiltonian([list(row) for row in v] if isinstance(v[0], list) else list(v), fn, D)  # update(h, adaptor): renew(metric, getM⁻¹)
            eps = adaptor.ssa.eps                          # update(κ, adaptor): nominal step size ← getϵ
        draws.append((list(z.theta), list(z.r)))
        stats.append(st)
    return draws, stats, eps, h.minv


# ------------------------------------------------------------------------------------------------
# find_good_stepsize (src/trajectory.jl:753-837)
# ------------------------------------------------------------------------------------------------
RNG_FINDEPS = 3


def find_good_stepsize(seed, chain, iteration, h, theta, initial_step_size=0.1, max_n_iters=100):
    eps = epsp = float(initial_step_size)
    loghalf = math.log(0.5)
    log_a_min, log_a_cross, log_a_max = 2 * loghalf, loghalf, math.log(0.75)
    d, invd = 2.0, 0.5
    rng = Rng(seed, chain, iteration)
    z0 = [rng.normal(RNG_FINDEPS, k) for k in range(h.D)]  # rand_momentum on the search's own stream
    if h.dense:
        raise NotImplementedError
    r = z0 if h.minv is None else [x / math.sqrt(m) for x, m in zip(z0, h.minv)]
    z = phasepoint(h, list(theta), r)
    H = energy(z)

    def A(e):  # (:753-757)
        return energy(step(e, h, z))

    Hp = A(eps)
    dH = H - Hp
    ratio_too_high = dH > log_a_cross
    for _ in range(max_n_iters):
        epsp = d * eps if ratio_too_high else invd * eps
        Hp = A(eps)  # (evaluated at ϵ, not ϵ′: :799-800)
        dH = H - Hp
        if ratio_too_high != (dH > log_a_cross):
            break
        eps = epsp
    eps, epsp = (eps, epsp) if eps <= epsp else (epsp, eps)  # minmax
    for _ in range(max_n_iters):
        mid = eps / 2 + epsp / 2  # Statistics.middle
        Hp = A(mid)
        dH = H - Hp
        if dH > log_a_max:
            eps = mid
        elif dH < log_a_min:
            epsp = mid
        else:
            eps = mid
            break
    return eps
