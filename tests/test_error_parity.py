"""Error behaviour of the C ABI: the HIP engine and the CPU oracle return the SAME status for the same misuse.

The reference rejects bad input with exceptions (`@argcheck`, `ArgumentError`, MethodError — cited per probe); the boundary turns them into status
codes (include/ahmc_hip.h: AHMC_ERR_ARGUMENT, _UNSUPPORTED, _STATE …).  Each probe below is one misuse — or one edge of the valid range — applied to
a fresh context of either library through the raw entry points; the statuses must agree probe by probe, and a context that refused a call must
still run a transition afterwards.
"""
import ctypes as C

import numpy as np
import pytest

import ahmc_amd as A
from ahmc_amd import _capi as capi

D, N = 6, 40


def _cfg(**kw):
    k = capi.KernelCfg()
    k.nuts, k.sampler, k.criterion, k.max_depth, k.delta_max, k.L, k.lambda_, k.refresh_alpha = 1, capi.TS_MULTINOMIAL, capi.TC_GENERALISED, 5, 1000.0, 0, 0.0, 0.0
    for key, v in kw.items():
        setattr(k, key, v)
    return k


def _arr(shape, fill=0.1):
    return np.full(shape, fill, dtype=np.float64, order="F")


# (name, fn(engine) -> status).  `e._ctx` is the context, `e.lib.dll` the library; every call returns the int32 status of include/ahmc_hip.h
def _p(e):
    return capi.as_ptr


PROBES = [
    ("leapfrog of 0 steps", lambda e: e.lib.dll.ahmc_leapfrog(e._ctx, 0)),
    ("step size: zero length", lambda e: e.lib.dll.ahmc_set_stepsize(e._ctx, capi.as_ptr(_arr(N)), 0)),
    ("step size: wrong length", lambda e: e.lib.dll.ahmc_set_stepsize(e._ctx, capi.as_ptr(_arr(N + 3)), N + 3)),
    ("step size: null pointer", lambda e: e.lib.dll.ahmc_set_stepsize(e._ctx, None, N)),
    ("integrator: unknown kind", lambda e: e.lib.dll.ahmc_set_integrator(e._ctx, 7, 0.0)),
    ("integrator: tempered α = 0", lambda e: e.lib.dll.ahmc_set_integrator(e._ctx, capi.INTEGRATOR_TEMPERED if hasattr(capi, "INTEGRATOR_TEMPERED") else 2, 0.0)),
    ("integrator: jitter < 0", lambda e: e.lib.dll.ahmc_set_integrator(e._ctx, capi.INTEGRATOR_JITTERED if hasattr(capi, "INTEGRATOR_JITTERED") else 1, -0.5)),
    ("metric: unknown kind", lambda e: e.lib.dll.ahmc_set_metric(e._ctx, 9, capi.as_ptr(_arr(D)), D)),
    ("metric: diag of wrong length", lambda e: e.lib.dll.ahmc_set_metric(e._ctx, capi.METRIC_DIAG, capi.as_ptr(_arr(D + 2)), D + 2)),
    ("metric: dense of wrong size", lambda e: e.lib.dll.ahmc_set_metric(e._ctx, capi.METRIC_DENSE, capi.as_ptr(_arr(D * D + 1)), D * D + 1)),
    ("metric: null pointer", lambda e: e.lib.dll.ahmc_set_metric(e._ctx, capi.METRIC_DIAG, None, D)),
    ("nuts: max_depth 0", lambda e: e.lib.dll.ahmc_nuts_transition(e._ctx, 0, 1000.0, capi.TC_GENERALISED, capi.TS_MULTINOMIAL)),
    ("nuts: max_depth -1", lambda e: e.lib.dll.ahmc_nuts_transition(e._ctx, -1, 1000.0, capi.TC_GENERALISED, capi.TS_MULTINOMIAL)),
    ("nuts: max_depth 40", lambda e: e.lib.dll.ahmc_nuts_transition(e._ctx, 40, 1000.0, capi.TC_GENERALISED, capi.TS_MULTINOMIAL)),
    ("nuts: unknown criterion", lambda e: e.lib.dll.ahmc_nuts_transition(e._ctx, 5, 1000.0, 11, capi.TS_MULTINOMIAL)),
    ("nuts: EndPointTS", lambda e: e.lib.dll.ahmc_nuts_transition(e._ctx, 5, 1000.0, capi.TC_GENERALISED, capi.TS_ENDPOINT)),
    ("nuts: unknown sampler", lambda e: e.lib.dll.ahmc_nuts_transition(e._ctx, 5, 1000.0, capi.TC_GENERALISED, 9)),
    ("nuts: Δ_max = 0", lambda e: e.lib.dll.ahmc_nuts_transition(e._ctx, 5, 0.0, capi.TC_GENERALISED, capi.TS_MULTINOMIAL)),
    ("nuts: Δ_max NaN", lambda e: e.lib.dll.ahmc_nuts_transition(e._ctx, 5, float("nan"), capi.TC_GENERALISED, capi.TS_MULTINOMIAL)),
    ("hmc: L = 0, λ = 0", lambda e: e.lib.dll.ahmc_hmc_transition(e._ctx, 0, 0.0, capi.TS_ENDPOINT)),
    ("hmc: L < 0", lambda e: e.lib.dll.ahmc_hmc_transition(e._ctx, -3, 0.0, capi.TS_ENDPOINT)),
    ("hmc: SliceTS", lambda e: e.lib.dll.ahmc_hmc_transition(e._ctx, 4, 0.0, capi.TS_SLICE)),
    ("hmc: λ < 0", lambda e: e.lib.dll.ahmc_hmc_transition(e._ctx, 0, -1.0, capi.TS_ENDPOINT)),
    ("hmc: λ with per-chain ϵ", lambda e: e.lib.dll.ahmc_hmc_transition(e._ctx, 0, 1.0, capi.TS_ENDPOINT)),
    ("sample: n_samples = 0", lambda e: e.lib.dll.ahmc_sample(e._ctx, C.byref(_cfg()), 0, 0, 0, None)),
    ("sample: n_samples < 0", lambda e: e.lib.dll.ahmc_sample(e._ctx, C.byref(_cfg()), -4, 0, 0, None)),
    ("sample: n_adapts > n_samples", lambda e: e.lib.dll.ahmc_sample(e._ctx, C.byref(_cfg()), 3, 9, 0, None)),
    ("sample: n_adapts < 0", lambda e: e.lib.dll.ahmc_sample(e._ctx, C.byref(_cfg()), 3, -1, 0, None)),
    ("sample: null kernel", lambda e: e.lib.dll.ahmc_sample(e._ctx, None, 3, 0, 0, None)),
    ("sample: refresh α = 1", lambda e: e.lib.dll.ahmc_sample(e._ctx, C.byref(_cfg(refresh_alpha=1.0)), 2, 0, 0, None)),
    ("sample: refresh α < 0", lambda e: e.lib.dll.ahmc_sample(e._ctx, C.byref(_cfg(refresh_alpha=-0.2)), 2, 0, 0, None)),
    ("sample: refresh α > 1", lambda e: e.lib.dll.ahmc_sample(e._ctx, C.byref(_cfg(refresh_alpha=1.5)), 2, 0, 0, None)),
    ("sample_from: i_first = 0", lambda e: e.lib.dll.ahmc_sample_from(e._ctx, C.byref(_cfg()), 0, 3, 0, 0, None)),
    ("sample_from: i_first > n_samples", lambda e: e.lib.dll.ahmc_sample_from(e._ctx, C.byref(_cfg()), 9, 3, 0, 0, None)),
    ("sample: static L = 0", lambda e: e.lib.dll.ahmc_sample(e._ctx, C.byref(_cfg(nuts=0, sampler=capi.TS_ENDPOINT, L=0)), 2, 0, 0, None)),
    ("adaptor_init: unknown kind", lambda e: e.lib.dll.ahmc_adaptor_init(e._ctx, 17, 0.8, 75, 50, 25)),
    ("adaptor_init: δ = 0", lambda e: e.lib.dll.ahmc_adaptor_init(e._ctx, capi.ADAPT_STEPSIZE, 0.0, 75, 50, 25)),
    ("adaptor_init: δ = 1", lambda e: e.lib.dll.ahmc_adaptor_init(e._ctx, capi.ADAPT_STEPSIZE, 1.0, 75, 50, 25)),
    ("adaptor_init: negative buffer", lambda e: e.lib.dll.ahmc_adaptor_init(e._ctx, capi.ADAPT_STAN, 0.8, -1, 50, 25)),
    ("adaptor_init: window 0", lambda e: e.lib.dll.ahmc_adaptor_init(e._ctx, capi.ADAPT_STAN, 0.8, 75, 50, 0)),
    ("var estimator: unknown", lambda e: e.lib.dll.ahmc_set_var_estimator(e._ctx, 9)),
    ("adapt without an adaptor", lambda e: e.lib.dll.ahmc_adapt(e._ctx, 1, 10, None, None)),
    ("adapt: i = 0", lambda e: (e.lib.dll.ahmc_adaptor_init(e._ctx, capi.ADAPT_STEPSIZE, 0.8, 75, 50, 25), e.lib.dll.ahmc_adapt(e._ctx, 0, 10, None, None))[1]),
    ("find_good_stepsize: ϵ0 = 0", lambda e: e.lib.dll.ahmc_find_good_stepsize(e._ctx, 0.0, 100)),
    ("find_good_stepsize: ϵ0 < 0", lambda e: e.lib.dll.ahmc_find_good_stepsize(e._ctx, -0.1, 100)),
    ("find_good_stepsize: no iterations", lambda e: e.lib.dll.ahmc_find_good_stepsize(e._ctx, 0.1, 0)),
    ("get_stat: unknown field", lambda e: e.lib.dll.ahmc_get_stat(e._ctx, 99, capi.as_ptr(_arr(N)))),
    ("get_stat: null pointer", lambda e: e.lib.dll.ahmc_get_stat(e._ctx, 0, None)),
    ("get_info: unknown key", lambda e: e.lib.dll.ahmc_get_info(e._ctx, 99, C.byref(C.c_int64()))),
    ("refresh: α = 1", lambda e: e.lib.dll.ahmc_refresh_momentum(e._ctx, 1.0)),
    ("refresh: α < 0", lambda e: e.lib.dll.ahmc_refresh_momentum(e._ctx, -0.5)),
    ("ext_advance without a run", lambda e: e.lib.dll.ahmc_ext_advance(e._ctx, None, None)),
    ("ext_pending without a run", lambda e: e.lib.dll.ahmc_ext_pending(e._ctx, C.byref(C.c_int64()), None, None)),
    ("set_phasepoint: null θ", lambda e: e.lib.dll.ahmc_set_phasepoint(e._ctx, None, None, None, None)),
    ("gather_state: null", lambda e: e.lib.dll.ahmc_gather_state(e._ctx, None)),
    ("ref_compat toggles", lambda e: e.lib.dll.ahmc_set_ref_compat(e._ctx, 1) or e.lib.dll.ahmc_set_ref_compat(e._ctx, 0)),
]


def _ext_cfg_probe(**kw):
    return lambda e: e.lib.dll.ahmc_ext_begin(e._ctx, C.byref(_cfg(**kw)), 1)


# the ask / tell protocol of a user density evaluated by the caller (ahmc_ext_*): an engine whose target is AHMC_TARGET_EXTERNAL
EXT_PROBES = [
    ("ext_begin: valid", _ext_cfg_probe()),
    ("ext_begin: n_trans = 0", lambda e: e.lib.dll.ahmc_ext_begin(e._ctx, C.byref(_cfg()), 0)),
    ("ext_begin: null kernel", lambda e: e.lib.dll.ahmc_ext_begin(e._ctx, None, 1)),
    ("ext_begin: max_depth 0", _ext_cfg_probe(max_depth=0)),
    ("ext_begin: max_depth 40", _ext_cfg_probe(max_depth=40)),
    ("ext_begin: refresh α = 1", _ext_cfg_probe(refresh_alpha=1.0)),
    ("ext_begin: refresh α < 0", _ext_cfg_probe(refresh_alpha=-0.3)),
    ("ext_begin: unknown sampler", _ext_cfg_probe(sampler=9)),
    ("ext_begin: unknown criterion", _ext_cfg_probe(criterion=9)),
    ("ext_begin: static SliceTS", _ext_cfg_probe(nuts=0, sampler=capi.TS_SLICE, L=3)),
    ("ext_begin: static L = 0", _ext_cfg_probe(nuts=0, sampler=capi.TS_ENDPOINT, L=0)),
    ("ext_begin: static λ with per-chain ϵ", _ext_cfg_probe(nuts=0, sampler=capi.TS_ENDPOINT, L=0, lambda_=1.0)),
    ("ext_begin twice", lambda e: (e.lib.dll.ahmc_ext_begin(e._ctx, C.byref(_cfg()), 1), e.lib.dll.ahmc_ext_begin(e._ctx, C.byref(_cfg()), 1))[1]),
    ("ext_advance: null results inside a run", lambda e: (e.lib.dll.ahmc_ext_begin(e._ctx, C.byref(_cfg()), 1), e.lib.dll.ahmc_ext_advance(e._ctx, None, None))[1]),
    ("fused transition on an external target", lambda e: e.lib.dll.ahmc_nuts_transition(e._ctx, 5, 1000.0, capi.TC_GENERALISED, capi.TS_MULTINOMIAL)),
    ("leapfrog on an external target", lambda e: e.lib.dll.ahmc_leapfrog(e._ctx, 3)),
    ("ext_cancel without a run", lambda e: e.lib.dll.ahmc_ext_cancel(e._ctx)),
    ("ext_find_good_stepsize_begin: ϵ0 = 0", lambda e: e.lib.dll.ahmc_ext_find_good_stepsize_begin(e._ctx, 0.0, 10)),
]

# where the HIP engine may say AHMC_ERR_UNSUPPORTED (2) to what the oracle — the reference's behaviour — serves: engine limits named in include/ahmc_hip.h
ENGINE_LIMITS = {"nuts: max_depth 40", "ext_begin: max_depth 40"}


def _ext_engine(lib):
    def fn(theta):
        return -0.5 * (theta ** 2).sum(axis=0), -theta
    h = A.Hamiltonian(A.DiagEuclideanMetric((D, N)), A.ExternalTarget(D, fn))
    e = A.Engine(h, N, rng=3, lib=lib)
    e.set_integrator(A.Leapfrog(np.full(N, 0.2)))
    e.set_position(np.random.default_rng(1).normal(size=(D, N)))
    return e


def _run_ext_probes(lib):
    out = {}
    for name, fn in EXT_PROBES:
        e = _ext_engine(lib)
        try:
            out[name] = int(fn(e))
            e.lib.dll.ahmc_ext_cancel(e._ctx)
            e.transition(A.HMCKernel(A.Trajectory(A.MultinomialTS, A.Leapfrog(np.full(N, 0.2)), A.GeneralisedNoUTurn(max_depth=3))))   # the ask / tell loop still works
            assert np.isfinite(e.theta()).all(), name
        finally:
            e.close()
    return out


def _engine(lib, with_point=True):
    h = A.Hamiltonian(A.DiagEuclideanMetric((D, N)), A.IsoGaussian(D))
    e = A.Engine(h, N, rng=3, lib=lib)
    e.set_integrator(A.Leapfrog(np.full(N, 0.2)))
    if with_point:
        e.set_position(np.random.default_rng(1).normal(size=(D, N)))
    return e


def _run_probes(lib, with_point):
    out = {}
    for name, fn in PROBES:
        e = _engine(lib, with_point)
        try:
            out[name] = int(fn(e))
            if with_point:   # a context that refused (or served) the call is still good for a transition
                e.transition(A.HMCKernel(A.Trajectory(A.MultinomialTS, A.Leapfrog(np.full(N, 0.2)), A.GeneralisedNoUTurn(max_depth=4))))
                assert np.isfinite(e.theta()).all(), name
        finally:
            e.close()
    return out


def test_probe_table_on_the_oracle(oracle):
    """(CPU) the table itself: every probe returns a status of the header's enumeration, most of them refusals"""
    st = _run_probes(oracle, True)
    assert all(0 <= v <= 6 for v in st.values()), st
    assert sum(v != 0 for v in st.values()) >= 22, st
    st0 = _run_probes(oracle, False)   # before any phase point exists: the calls that need one are refused as a STATE error or earlier as arguments
    assert all(0 <= v <= 6 for v in st0.values()), st0
    ste = _run_ext_probes(oracle)
    assert all(0 <= v <= 6 for v in ste.values()) and ste["ext_begin: valid"] == 0 and sum(v != 0 for v in ste.values()) >= 10, ste


@pytest.mark.gpu
@pytest.mark.parametrize("with_point", [True, False])
def test_error_statuses_agree_between_the_hip_engine_and_the_oracle(hip, oracle, with_point):
    sg, so = _run_probes(hip, with_point), _run_probes(oracle, with_point)
    if with_point:
        sg.update(_run_ext_probes(hip))
        so.update(_run_ext_probes(oracle))
    diff = {k: (sg[k], so[k]) for k in sg if sg[k] != so[k] and not (k in ENGINE_LIMITS and (sg[k], so[k]) == (2, 0))}
    assert not diff, f"probe: (HIP engine, oracle) statuses differ: {diff}"
    if with_point and hip.backend.startswith("hip"):     # (not in a dry run of this file on the oracle)
        assert all(sg[k] == 2 for k in ENGINE_LIMITS), {k: sg[k] for k in ENGINE_LIMITS}
