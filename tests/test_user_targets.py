"""User log-densities ON THE DEVICE — the reference's plugin surface (`h.∂ℓπ∂θ(θ)`, /root/reference/src/hamiltonian.jl:45-48;
LogDensityProblems behind src/AdvancedHMC.jl:163-186) without a host round trip:

  * target PLUGIN (ahmc_set_target_plugin): the density is a HIP device function compiled into the engine's own fused
    trajectory kernels (include/ahmc_user_target.h).  tests/user_targets/iso_gauss.hpp repeats the built-in isotropic
    Gaussian's arithmetic: the plugin's chains must equal the built-in family's BIT FOR BIT — static HMC, NUTS, the fused
    warm-up with StanHMCAdaptor.  tests/user_targets/banana.hpp is no built-in family: it is held to the oracle, which takes
    the same density as a numpy callback through ask / tell;
  * target KERNEL (ahmc_set_target_kernel): the density is a device kernel (tests/user_targets/kernels.hip, compiled to a
    code object and bound as a hipFunction_t) that the step-synchronous engine launches itself; held to the oracle, whose
    form of the same call is a host function (AHMC_KERNEL_HOST);
and, CPU side: the plugin builds and describes itself; the oracle's host-kernel target equals its built-in family.
"""
import ctypes as C
import os

import numpy as np
import pytest

import ahmc_amd as A
from ahmc_amd import _capi as capi

HERE = os.path.dirname(os.path.abspath(__file__))
UT = os.path.join(HERE, "user_targets")
LOG2PI = 1.8378770664093454835606594728112


def banana_numpy(a, b):
    def fn(th):
        D = th.shape[0]
        lp = np.zeros(th.shape[1])
        g = np.zeros_like(th)
        m = D // 2
        x, y = th[0:2 * m:2], th[1:2 * m:2]
        u, w = x - a, y - x * x
        lp -= (u * u / 2 + b * w * w).sum(axis=0)
        g[0:2 * m:2] = -(u - 4 * b * w * x)
        g[1:2 * m:2] = -(2 * b * w)
        if D % 2:
            lp -= th[-1] ** 2 / 2
            g[-1] = -th[-1]
        return lp, g
    return fn


KFUNC = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int64, C.c_int32, C.c_int64,
                    C.c_void_p)


def host_kernel(fn):
    """a numpy log-density `fn(θ (D,n)) -> (ℓπ, ∇ℓπ)` as the CPU checker's form of a target kernel (AHMC_KERNEL_HOST)"""
    def f(theta, lp, grad_neg, cols, n_cols, D, N, user):
        for k in range(n_cols):
            c = cols[k] if cols else k
            th = np.ctypeslib.as_array(theta, shape=((c + 1) * D,))[c * D:(c + 1) * D]
            v, g = fn(th.reshape(D, 1).copy())
            lp[c] = float(v[0])
            out = np.ctypeslib.as_array(grad_neg, shape=((c + 1) * D,))
            out[c * D:(c + 1) * D] = -g[:, 0]
    return KFUNC(f)


# ---------------------------------------------------------------------------------------------------------------------
# CPU side
# ---------------------------------------------------------------------------------------------------------------------
def test_plugin_builds_and_describes_itself():
    """hipcc cross-compiles the engine's kernels with the user's device function inside (no GPU needed); the shared object
    exports the descriptor the engine checks before binding it"""
    from ahmc_amd.build import build_target_plugin

    so = build_target_plugin(os.path.join(UT, "banana.hpp"), np.float64, 64, 2, n_params=2)
    assert os.path.exists(so) and not os.path.realpath(so).startswith(os.path.realpath(os.path.dirname(HERE)) + os.sep)
    blob = open(so, "rb").read()
    assert b"ahmc_target_plugin_v1" in blob and b"hipv4-amdgcn-amd-amdhsa--gfx950" in blob
    import subprocess
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    assert " ahmc_target_plugin_v1" in syms
    assert so == build_target_plugin(os.path.join(UT, "banana.hpp"), np.float64, 64, 2, n_params=2)  # cached by content


@pytest.mark.parametrize("form", ["object", "bitcode"])
def test_object_plugin_links_and_inlines(form):
    """A density handed over as COMPILED device code (include/ahmc_user_target_object.h: one C symbol): a relocatable object of
    `hipcc -fgpu-rdc -c`, or raw amdgcn bitcode as GPUCompiler.jl emits it.  The engine's kernels are linked with it under
    device LTO (no GPU needed): the plugin exports the descriptor, every kernel of the geometry is there, and the density is
    INLINED — no call is left in the device code, the symbol is gone."""
    import json
    import subprocess
    from ahmc_amd.build import build_device_object, build_target_plugin_from_object

    obj = build_device_object(os.path.join(UT, "banana_object.hip"), bitcode=(form == "bitcode"))
    assert obj.endswith(".bc" if form == "bitcode" else ".o") and os.path.getsize(obj) > 0
    so = build_target_plugin_from_object(obj, np.float64, 64, 2, n_params=2)
    assert os.path.exists(so) and not os.path.realpath(so).startswith(os.path.realpath(os.path.dirname(HERE)) + os.sep)
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    assert " ahmc_target_plugin_v1" in syms
    for mode in range(5):
        assert f"k_nutsIdLi64ELi2ELi{mode}ELi4EE" in syms, mode      # the five NUTS instantiations of the geometry, family TK = 4
    meta = json.load(open(so + ".json"))
    assert meta["inlined"] is True and meta["G"] == 64 and meta["E"] == 2
    assert so == build_target_plugin_from_object(obj, np.float64, 64, 2, n_params=2)  # cached by content


def test_oracle_host_kernel_equals_builtin_family(oracle):
    """the checker's form of ahmc_set_target_kernel: the isotropic Gaussian as a host function == its built-in family,
    chain for chain (same scalar code path, only the evaluation of (ℓπ, ∇ℓπ) is the user's)"""
    D, N = 6, 24
    rs = np.random.default_rng(3)
    th0 = rs.normal(size=(D, N))

    def iso(th):  # the checker's own loop (oracle/ahmc_oracle.cpp, AHMC_TARGET_ISO_GAUSS): same operations in the same order
        lp = np.zeros(th.shape[1])
        for d in range(th.shape[0]):
            lp += -(LOG2PI + th[d] * th[d]) / 2
        return lp, -th

    cb = host_kernel(iso)
    lf = A.Leapfrog(np.full(N, 0.3))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=6)))
    res = []
    for target in (A.IsoGaussian(D), A.KernelTarget(D, cb, handle_kind=capi.KERNEL_HOST)):
        e = A.Engine(A.Hamiltonian(A.DiagEuclideanMetric((D, N)), target), N, rng=5, lib=oracle)
        e.set_integrator(lf)
        e.set_position(th0)
        e.adaptor_init(A.StepSizeAdaptor(0.8, lf))
        e.run(k, 12, 8)
        res.append((e.phasepoint(), e.stats(), e.get_stepsize()))
        e.close()
    (z0, s0, e0), (z1, s1, e1) = res
    np.testing.assert_array_equal(s0["n_steps"], s1["n_steps"])
    np.testing.assert_array_equal(z0.theta, z1.theta)
    np.testing.assert_array_equal(e0, e1)


def test_accumulators_are_part_of_the_checkpoint(oracle):
    """a run resumed in the sampling phase (ahmc_sample_from) used to zero Σθ, Σθ², Σ n_steps and the energy sums
    at its first kept iteration.  Now the accumulators travel with the checkpoint (ahmc_get/set_accum_state) and a resumed
    call continues them: get_accum / EBFMI of the resumed run == the uninterrupted run (oracle here, HIP in the gpu test)."""
    _checkpointed_accumulators(oracle)


def _checkpointed_accumulators(lib):
    D, N = 8, 40
    rs = np.random.default_rng(9)
    h = A.Hamiltonian(A.DiagEuclideanMetric(np.asfortranarray(0.5 + rs.random((D, N)))), A.Funnel(D))
    lf = A.Leapfrog(np.full(N, 0.2))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=7)))
    th0 = rs.normal(size=(D, N))
    n_adapts, n_total, cut = 20, 44, 31   # cut inside the sampling phase

    def fresh():
        e = A.Engine(h, N, rng=17, lib=lib)
        e.set_integrator(lf)
        e.set_position(th0)
        e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(h.metric), A.StepSizeAdaptor(0.8, lf), init_buffer=5, term_buffer=5, window_size=4))
        return e

    a = fresh()
    a.run(k, n_total, n_adapts, drop_warmup=True)
    ref_acc, ref_eb = a.accum(), a.ebfmi()
    b = fresh()
    b.run(k, cut, n_adapts, drop_warmup=True)
    st = b.get_state()
    assert st["accum"]["n_transitions"] == cut - n_adapts
    b.close()
    c = fresh()
    c.set_state(st)
    c.run(k, n_total, n_adapts, drop_warmup=True, i_first=cut + 1)
    acc, eb = c.accum(), c.ebfmi()
    assert acc["n_transitions"] == ref_acc["n_transitions"] == n_total - n_adapts
    assert acc["total_n_steps"] == ref_acc["total_n_steps"] and acc["n_divergent"] == ref_acc["n_divergent"]
    np.testing.assert_array_equal(acc["sum_theta"], ref_acc["sum_theta"])
    np.testing.assert_array_equal(acc["sumsq_theta"], ref_acc["sumsq_theta"])
    np.testing.assert_array_equal(eb, ref_eb)
    # and a resume that does not match the restored adaptor is refused, not silently mis-scheduled
    d = fresh()
    d.run(k, 7, n_adapts, drop_warmup=True)
    with pytest.raises(A.AHMCError):
        d.run(k, n_total, n_adapts, drop_warmup=True, i_first=12)
    for e in (a, c, d):
        e.close()


# ---------------------------------------------------------------------------------------------------------------------
# GPU side
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_accumulators_are_part_of_the_checkpoint(hip):
    _checkpointed_accumulators(hip)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("D,N", [(128, 512), (32, 1024), (600, 48)])
def test_plugin_equals_builtin_family_bit_for_bit(hip, dtype, D, N):
    """the isotropic Gaussian as a user device function inside k_nuts / k_hmc / k_find_eps / k_leapfrog == AHMC_TARGET_ISO_GAUSS:
    one chain per wave (D = 128), four chains per wave in lockstep (D = 32), a chain across two waves (D = 600)"""
    rs = np.random.default_rng(D)
    minv = np.asfortranarray(0.5 + rs.random((D, N)))
    th0 = rs.normal(size=(D, N))
    lf = A.Leapfrog(np.full(N, 0.1))
    nuts = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=8)))
    hmc = A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(5)))
    res = []
    for target in (A.IsoGaussian(D), A.PluginTarget(D, os.path.join(UT, "iso_gauss.hpp"))):
        metric = A.DiagEuclideanMetric(minv.copy(order="F"))
        e = A.Engine(A.Hamiltonian(metric, target), N, dtype=dtype, rng=A.PhiloxRNG(77), lib=hip)
        e.set_integrator(lf)
        e.set_position(th0)
        eps = e.find_good_stepsize()
        e.step(3)
        z_step = e.phasepoint()
        e.transition(hmc)
        s_hmc = e.stats()
        e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf), init_buffer=8, term_buffer=6, window_size=5))
        e.run(nuts, 40, 30)            # the fused warm-up (adapt! inside the kernel) and batched draws
        res.append((eps, z_step, s_hmc, e.phasepoint(), e.stats(), e.get_stepsize(), e.get_metric(), e.accum()))
        e.close()
    a, b = res
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1].theta, b[1].theta)
    np.testing.assert_array_equal(a[1].lp.value, b[1].lp.value)
    np.testing.assert_array_equal(a[2]["is_accept"], b[2]["is_accept"])
    np.testing.assert_array_equal(a[3].theta, b[3].theta)
    np.testing.assert_array_equal(a[3].r, b[3].r)
    for key in ("n_steps", "tree_depth", "acceptance_rate", "hamiltonian_energy", "step_size"):
        np.testing.assert_array_equal(a[4][key], b[4][key])
    np.testing.assert_array_equal(a[5], b[5])
    np.testing.assert_array_equal(a[6], b[6])
    np.testing.assert_array_equal(a[7]["sum_theta"], b[7]["sum_theta"])
    assert a[4]["tree_depth"].max() >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("form,dtype,D,N", [("object", np.float64, 128, 512), ("bitcode", np.float64, 128, 512), ("bitcode", np.float64, 50, 768),
                                            ("object", np.float32, 128, 256), ("bitcode", np.float64, 600, 48)])
def test_object_plugin_equals_header_plugin_bit_for_bit(hip, form, dtype, D, N):
    """The SAME density (banana) as a header plugin (compiled into the kernels from source) and as an object / bitcode plugin
    (linked into them under device LTO): static HMC, NUTS, find_good_stepsize and a fused Stan warm-up + draws must give the
    same chains BIT FOR BIT — the object form is the header form's arithmetic, inlined — on one-wave, shared-wave and multi-wave
    geometries, f64 and f32."""
    from ahmc_amd.build import build_device_object

    obj = build_device_object(os.path.join(UT, "banana_object.hip"), bitcode=(form == "bitcode"))
    params = np.array([0.5, 0.05])
    rs = np.random.default_rng(D + N)
    minv = np.asfortranarray(0.5 + rs.random((D, N)))
    th0 = np.asfortranarray(0.5 * rs.normal(size=(D, N)))
    out = []
    for target in (A.PluginTarget(D, os.path.join(UT, "banana.hpp"), params=params), A.ObjectTarget(D, obj, params=params)):
        metric = A.DiagEuclideanMetric(minv.copy(order="F"))
        e = A.Engine(A.Hamiltonian(metric, target), N, dtype=dtype, rng=A.PhiloxRNG(9), lib=hip)
        lf = A.Leapfrog(np.full(N, 0.1))
        e.set_integrator(lf)
        e.set_position(th0)
        z0 = e.phasepoint()
        eps = e.find_good_stepsize()
        hmc = A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(7)))
        e.transition(hmc)
        s_h = e.stats()
        k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=8, delta_max=1000.0)))
        e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf), init_buffer=9, term_buffer=6, window_size=15))
        draws = np.zeros((D, N, 10), order="F", dtype=dtype)
        e.run(k, 50, 40, drop_warmup=True, samples_out=draws)
        e.sync()
        out.append((z0.lp.value.copy(), z0.lp.gradient.copy(), eps.copy(), s_h["is_accept"].copy(), s_h["hamiltonian_energy"].copy(), draws,
                    e.get_stepsize().copy(), e.get_metric().copy(), e.stats()["n_steps"].copy(), e.accum()["total_n_steps"]))
        e.close()
    a, b = out
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    assert a[8].max() >= 7 and np.isfinite(a[5]).all()


@pytest.mark.gpu
def test_plugin_from_other_kernel_sources_is_refused(hip, tmp_path):
    """The scratch layout of k_nuts is a contract between the library's launch plan and the kernels a plugin carries that no struct size
    shows (round 5: a plugin compiled from newer sources than the loaded library wrote out of bounds).  The library knows the digest of the
    kernel sources it was built from; a plugin that names another one is refused with an ArgumentError, nothing is bound."""
    import subprocess
    from ahmc_amd import build as B

    so = str(tmp_path / "libstale_plugin.so")
    cmd = ["hipcc", *B.FLAGS, "-shared", "-DAHMC_INST_T=double", "-DAHMC_INST_TK=4", "-DAHMC_PLUGIN_G=64", "-DAHMC_PLUGIN_E=2",
           "-DAHMC_PLUGIN_NPARAMS=-1", f'-DAHMC_USER_TARGET_HEADER="{os.path.join(UT, "iso_gauss.hpp")}"', '-DAHMC_SOURCES_DIGEST="0123456789abcdef-not-this-library"',
           "-I", B.INCLUDE, os.path.join(B.CSRC, "ahmc_inst.hip"), "-o", so]
    subprocess.run(cmd, check=True, capture_output=True)
    D, N = 128, 64
    e = A.Engine(A.Hamiltonian(A.DiagEuclideanMetric(np.ones((D, N), order="F")), A.IsoGaussian(D)), N, rng=A.PhiloxRNG(1), lib=hip)
    with pytest.raises(A.ArgumentError, match="other kernel sources"):
        e._call("ahmc_set_target_plugin", so.encode(), None, 0)
    # the context still runs its built-in family
    lf = A.Leapfrog(np.full(N, 0.2))
    e.set_integrator(lf)
    e.set_position(np.zeros((D, N), order="F"))
    e.transition(A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=5))))
    assert np.isfinite(e.theta()).all()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("D", [10, 128, 300])
def test_plugin_banana_against_oracle(hip, oracle, D):
    """a density that is no built-in family, compiled into the fused kernels, vs the oracle evaluating the same density as a
    numpy callback through ask / tell: every transition, every chain (a chain may differ only at the oracle's own near-ties,
    tests/parity_util.py), with re-alignment"""
    N = 256
    a_, b_ = 0.5, 0.05
    rs = np.random.default_rng(D + 1)
    minv = np.asfortranarray(0.5 + rs.random((D, N)))
    th0 = 0.5 * rs.normal(size=(D, N))
    lf = A.Leapfrog(np.full(N, 0.15) * (0.7 + 0.6 * rs.random(N)))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=7)))
    g = A.Engine(A.Hamiltonian(A.DiagEuclideanMetric(minv), A.PluginTarget(D, os.path.join(UT, "banana.hpp"), params=np.array([a_, b_]))), N,
                 rng=A.PhiloxRNG(4), lib=hip)
    o = A.Engine(A.Hamiltonian(A.DiagEuclideanMetric(minv), A.ExternalTarget(D, banana_numpy(a_, b_))), N, rng=A.PhiloxRNG(4), lib=oracle)
    g.set_integrator(lf)
    o.set_integrator(lf)
    g.set_position(th0)
    lp0, g0 = banana_numpy(a_, b_)(th0)
    o.set_position(th0)
    zg, zo = g.phasepoint(), o.phasepoint()
    np.testing.assert_allclose(zg.lp.value, lp0, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(zg.lp.gradient, -g0, rtol=1e-10, atol=1e-10)
    from test_gpu_parity import compare_transition_stats
    depth = 0
    for it in range(4):
        g.transition(k)
        o.transition(k)
        sg, so = g.stats(), o.stats()
        same = compare_transition_stats(sg, so, np.float64, o, f"plugin banana D={D}")
        zg, zo = g.phasepoint(), o.phasepoint()
        np.testing.assert_allclose(zg.theta[:, same], zo.theta[:, same], rtol=1e-8, atol=1e-8)
        depth = max(depth, int(so["tree_depth"].max()))
        th = zo.theta
        g.set_position(th)
        o.set_position(th)
    assert depth >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("name,D,metric", [("iso_gauss_f64", 128, "diag"), ("banana_f64", 50, "diag"), ("banana_f64", 40, "dense")])
def test_kernel_target_against_oracle(hip, oracle, name, D, metric):
    """ahmc_set_target_kernel: a hipFunction_t from a separately compiled code object; the engine launches it between its tree
    kernels (phasepoint, step, static HMC, NUTS in batches through ahmc_sample, find_good_stepsize).  Oracle: the same
    density as a host function (AHMC_KERNEL_HOST)."""
    import torch
    from ahmc_amd.build import build_code_object
    from ahmc_amd.hipmod import Module

    N = 300
    a_, b_ = 0.3, 0.08
    rs = np.random.default_rng(D)
    mod = Module(build_code_object(os.path.join(UT, "kernels.hip")))
    user = torch.tensor([a_, b_], dtype=torch.float64, device="cuda")
    fn = (lambda th: (-(th * th).sum(axis=0) / 2 - th.shape[0] * LOG2PI / 2, -th)) if name.startswith("iso") else banana_numpy(a_, b_)
    cb = host_kernel(fn)
    if metric == "dense":  # the user's kernel behind a shared DenseEuclideanMetric: w′ = M⁻¹g′ on MFMA from the staged gradient
        Q, _ = np.linalg.qr(rs.normal(size=(D, D)))
        Mi = (Q * np.linspace(0.6, 2.0, D)) @ Q.T
        make_metric = lambda: A.DenseEuclideanMetric(np.asfortranarray((Mi + Mi.T) / 2))  # noqa: E731
    else:
        minv = np.asfortranarray(0.5 + rs.random((D, N)))
        make_metric = lambda: A.DiagEuclideanMetric(minv)  # noqa: E731
    th0 = 0.5 * rs.normal(size=(D, N))
    lf = A.Leapfrog(np.full(N, 0.2) * (0.7 + 0.6 * rs.random(N)))
    nuts = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=6)))
    hmc = A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(4)))
    tg = A.KernelTarget(D, mod.function(name), handle_kind=capi.KERNEL_HIP_FUNCTION, block_threads=256, chains_per_block=4, user=user.data_ptr())
    to = A.KernelTarget(D, cb, handle_kind=capi.KERNEL_HOST)
    g = A.Engine(A.Hamiltonian(make_metric(), tg), N, rng=A.PhiloxRNG(8), lib=hip)
    o = A.Engine(A.Hamiltonian(make_metric(), to), N, rng=A.PhiloxRNG(8), lib=oracle)
    for e in (g, o):
        e.set_integrator(lf)
        e.set_position(th0)
    zg, zo = g.phasepoint(), o.phasepoint()
    np.testing.assert_allclose(zg.lp.value, zo.lp.value, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(zg.lp.gradient, zo.lp.gradient, rtol=1e-10, atol=1e-10)
    for e in (g, o):
        e.step(3)
    np.testing.assert_allclose(g.phasepoint().theta, o.phasepoint().theta, rtol=1e-9, atol=1e-9)
    from test_gpu_parity import compare_transition_stats
    for k in (hmc, nuts, nuts):
        for e in (g, o):
            e.transition(k)
        same = compare_transition_stats(g.stats(), o.stats(), np.float64, o, f"kernel target {name} {metric}")
        np.testing.assert_allclose(g.phasepoint().theta[:, same], o.phasepoint().theta[:, same], rtol=1e-8, atol=1e-8)
        th = o.phasepoint().theta
        for e in (g, o):
            e.set_position(th)
    import parity_util as PU

    PU.reset_margin(o)
    eg, eo = g.find_good_stepsize(), o.find_good_stepsize()
    PU.check_equal_or_near_tie(eg, eo, PU.decision_margin(o), np.float64, f"kernel target {name} find_good_stepsize")
    # the bulk driver: batches of transitions, the engine launching the user's kernel once per global step, no host round trip
    g.set_integrator(A.Leapfrog(eo))
    o.set_integrator(A.Leapfrog(eo))
    for e in (g, o):
        e.set_position(th0)
        e.run(nuts, 3)
    on = np.isclose(g.phasepoint().theta, o.phasepoint().theta, rtol=1e-8, atol=1e-8).all(axis=0)
    PU.check_flips(on, PU.decision_margin(o), np.float64, f"kernel target {name} bulk run of 3")   # (free-running: a flip stays)
    assert g.accum()["n_transitions"] == 3
    g.close(); o.close()
