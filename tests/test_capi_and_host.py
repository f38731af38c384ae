"""CPU-only checks of the boundary: the product library exports the whole C ABI of
include/ahmc_hip.h (no compute call is made without a GPU), the header and the ctypes table agree,
and the host-side mirror validates arguments the way the reference does (exercised on the oracle,
which implements the same ABI).
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import ahmc_amd as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ahmc_hip.h")


def header_functions():
    src = open(HEADER, encoding="utf-8").read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(ahmc_[a-z_0-9]+)\s*\(", src)) - {"ahmc_ctx", "ahmc_kernel_cfg"}


def test_header_and_binding_table_agree():
    assert header_functions() == set(A.capi.SIGNATURES)


def test_hip_library_builds_and_exports_every_symbol():
    """`build()` cross-compiles for gfx950 without a GPU; every header symbol must resolve."""
    so = A.build_hip_library()
    assert os.path.exists(so)
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (ahmc_[a-z_0-9]+)$", out, flags=re.M))
    assert header_functions() <= exported
    # the code object is gfx950 and nothing else (no dual paths)
    # (read from the offload bundle's entry names in the file itself: `llvm-objdump --offloading` would unbundle tens of MB
    # of device images next to the .so, i.e. into the tree that travels to the GPU box)
    blob = open(so, "rb").read()
    archs = set(m.decode() for m in re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert archs == {"gfx950"}, archs


def test_hip_library_loads_without_gpu():
    """dlopen + ABI/version/backend queries only — no context is created"""
    import torch  # noqa: F401  (HIP runtime of the process must be torch's copy)

    lib = A.CLib(A.hip_library_path())
    assert lib.backend == "hip:gfx950"
    assert lib.dll.ahmc_abi_version() == A.capi.AHMC_ABI_VERSION
    ws, we, splits = A.stan_windows(1000, lib=lib)  # pure host logic of the product library
    assert (ws, we, splits) == (76, 950, [100, 150, 250, 450, 950])


def test_product_loader_has_no_fallback(monkeypatch, tmp_path):
    monkeypatch.setenv("AHMC_HIP_LIB", str(tmp_path / "missing.so"))
    monkeypatch.setattr(A.capi, "_HIP_LIB", None)
    with pytest.raises(ImportError, match="no CPU fallback"):
        A.load_hip_library()
    # the oracle is refused as a product library
    monkeypatch.setenv("AHMC_HIP_LIB", os.path.join(ROOT, "oracle", "libahmc_oracle.so"))
    monkeypatch.setattr(A.capi, "_HIP_LIB", None)
    with pytest.raises(ImportError, match="expected the HIP engine"):
        A.load_hip_library()
    monkeypatch.setattr(A.capi, "_HIP_LIB", None)


def test_dry_run_on_the_oracle_is_never_green():
    """AHMC_TEST_DRYRUN_ON_ORACLE=1 binds the `hip` fixture to the CPU checker (to debug test code without a GPU): such a run
    must not produce a single PASSED gpu test (tests/conftest.py: every one that ran through is a skip with the reason)."""
    import subprocess
    import sys

    env = dict(os.environ, AHMC_TEST_DRYRUN_ON_ORACLE="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-k",
                          "test_refresh_momentum", "-rs", "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    tail = out.stdout.strip().splitlines()[-1]
    assert "skipped" in tail and "passed" not in tail and "failed" not in tail, out.stdout[-2000:]
    assert "says nothing about the HIP engine" in out.stdout


def test_package_never_references_the_oracle():
    pkg = os.path.join(ROOT, "advancedhmc.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert "libahmc_oracle" not in text and "oracle/ahmc_oracle" not in text, f


# ---- argument validation mirrors the reference's errors (oracle-backed: same ABI, same host code) ----
def test_argument_errors(oracle):
    D, N = 5, 4
    h = A.Hamiltonian(A.UnitEuclideanMetric((D, N)), A.IsoGaussian(D))
    e = A.Engine(h, N, lib=oracle)
    with pytest.raises(A.ArgumentError):  # @argcheck length(θ) == length(r) … (src/hamiltonian.jl:94)
        e.set_position(np.zeros((D + 1, N)))
    with pytest.raises(A.ArgumentError):  # step-size vector of the wrong length
        e.set_integrator(A.Leapfrog(np.full(N + 1, 0.1)))
    with pytest.raises(A.AHMCError):  # transition before a phase point exists
        e.transition(A.HMCKernel(A.Trajectory(A.EndPointTS, A.Leapfrog(0.1), A.FixedNSteps(3))))
    with pytest.raises(A.ArgumentError):  # AxesMismatch (src/hamiltonian.jl:55-57)
        e.set_metric(A.DiagEuclideanMetric(np.ones(D + 2)))
    with pytest.raises(A.ArgumentError):  # metric / target dimension mismatch
        A.Hamiltonian(A.UnitEuclideanMetric((D + 1, N)), A.IsoGaussian(D))
    e.set_position(np.zeros((D, N)))
    with pytest.raises(A.ArgumentError):  # static kernels have no SliceTS
        e.transition(A.HMCKernel(A.Trajectory(A.SliceTS, A.Leapfrog(0.1), A.FixedNSteps(3))))
    with pytest.raises(A.ArgumentError):  # FixedIntegrationTime needs a scalar ϵ (src/trajectory.jl:241-243, Q6)
        e.set_integrator(A.Leapfrog(np.full(N, 0.1)))
        e.transition(A.HMCKernel(A.Trajectory(A.EndPointTS, A.Leapfrog(np.full(N, 0.1)), A.FixedIntegrationTime(1.0))))
    with pytest.raises(A.ArgumentError):  # length(rngs) == n_chains (src/utilities.jl:13)
        A.Engine(h, N, rng=[A.PhiloxRNG(1)] * (N + 1), lib=oracle)
    with pytest.raises(AssertionError):  # src/sampler.jl:172
        A.sample(1, h, A.HMCKernel(A.Trajectory(A.EndPointTS, A.Leapfrog(0.1), A.FixedNSteps(3))), np.zeros((D, N)), 10,
                 drop_warmup=True, lib=oracle)
    with pytest.raises(A.ArgumentError):
        A.Engine(h, N, dtype=np.float16, lib=oracle)


def test_vector_theta_and_stat_fields(oracle):
    """scalar-chain mode (θ a Vector) and the stat field names pinned by test/sampler.jl:12-46"""
    D = 5
    h = A.Hamiltonian(A.DiagEuclideanMetric(D), A.IsoGaussian(D))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, A.Leapfrog(0.2), A.GeneralisedNoUTurn()))
    samples, stats = A.sample(3, h, k, np.zeros(D), 30, A.StanHMCAdaptor(A.MassMatrixAdaptor(h.metric), A.StepSizeAdaptor(0.8, k.tau.integrator)), 20,
                              lib=oracle)
    assert samples[0].shape == (D,) and len(samples) == 30
    nuts_fields = {"n_steps", "is_accept", "acceptance_rate", "log_density", "hamiltonian_energy", "hamiltonian_energy_error",
                   "max_hamiltonian_energy_error", "tree_depth", "numerical_error", "step_size", "nom_step_size", "is_adapt"}
    assert set(stats[0]) == nuts_fields
    assert stats[0]["is_adapt"] and not stats[-1]["is_adapt"]
    eb = A.EBFMI([s["hamiltonian_energy"] for s in stats])
    assert np.all(np.isfinite(eb))


def test_float32_eltype_preserved(oracle):
    """test/constructors.jl:131-157: Float32 in → Float32 out"""
    D, N = 4, 3
    h = A.Hamiltonian(A.UnitEuclideanMetric(np.float32, (D, N)), A.IsoGaussian(D))
    k = A.HMCKernel(A.Trajectory(A.EndPointTS, A.Leapfrog(0.1), A.FixedNSteps(4)))
    samples, stats = A.sample(0, h, k, np.zeros((D, N), dtype=np.float32), 5, lib=oracle)
    assert samples[0].dtype == np.float32 and stats[0]["acceptance_rate"].dtype == np.float32


def test_external_target_split_step(oracle, rng):
    """lf_pre / lf_post around a caller-side gradient == the fused built-in step (src/integrator.jl:231-243)"""
    D, N = 6, 9
    Minv = np.asfortranarray(0.5 + rng.random((D, N)))
    fn = lambda th: (np.sum(-(np.log(2 * np.pi) + th ** 2) / 2, axis=0), -th)  # noqa: E731
    ext = A.Engine(A.Hamiltonian(A.DiagEuclideanMetric(Minv), A.ExternalTarget(D, fn)), N, lib=oracle)
    ref = A.Engine(A.Hamiltonian(A.DiagEuclideanMetric(Minv), A.IsoGaussian(D)), N, lib=oracle)
    th, r = rng.normal(size=(D, N)), rng.normal(size=(D, N))
    for e in (ext, ref):
        e.set_integrator(A.TemperedLeapfrog(np.full(N, 0.07), 1.1))
        e.set_position(th, r)
        e.step(6)
    za, zb = ext.phasepoint(), ref.phasepoint()
    np.testing.assert_allclose(za.theta, zb.theta, rtol=1e-13)
    np.testing.assert_allclose(za.r, zb.r, rtol=1e-13)
    np.testing.assert_allclose(za.lk.value, zb.lk.value, rtol=1e-13)
    ext.step(-3)
    ref.step(-3)
    np.testing.assert_allclose(ext.phasepoint().theta, ref.phasepoint().theta, rtol=1e-12, atol=1e-14)


def test_ref_compat_batch_early_exit(oracle):
    """Q1 (src/integrator.jl:252-258): in matrix mode the reference stops ALL chains at the first
    step where ANY chain is non-finite; per-chain semantics (the engine's, = the reference's scalar
    path) let the healthy chains finish.  The engine's default is the per-chain form; `ahmc_set_ref_compat` (ABI v6) switches to the
    reference's coupled form on both the oracle and the HIP engine."""
    D, N = 3, 4
    h = A.Hamiltonian(A.UnitEuclideanMetric((D, N)), A.IsoGaussian(D))
    th = np.zeros((D, N))
    th[:, 2] = 1e200  # chain 2 overflows on its first step
    r = np.ones((D, N))
    res = {}
    for compat in (0, 1):
        e = A.Engine(h, N, lib=oracle)
        e.set_ref_compat(bool(compat))     # (ABI v6: `ahmc_set_ref_compat`, implemented by the HIP engine too — tests/test_gpu_parity.py)
        e.set_integrator(A.Leapfrog(0.1))
        e.set_position(th, r)
        e.step(5)
        res[compat] = e.phasepoint().theta
    assert np.allclose(res[1][:, 0], 0.1, atol=1e-3)      # stopped after ONE step
    assert np.allclose(res[0][:, 0], np.sin(0.5), atol=1e-2)  # five steps of the unit oscillator


def test_ess_and_bundle_samples(oracle):
    """output side: ESS on an AR(1) process with known autocorrelation time,
    EBFMI, and the Chains-shaped bundle of `sample`'s outputs"""
    import ahmc_amd as A
    from ahmc_amd import diagnostics as dg

    rng = np.random.default_rng(0)
    n, m, phi = 20000, 8, 0.7
    x = np.zeros((n, m))
    e = rng.normal(size=(n, m))
    for t in range(1, n):
        x[t] = phi * x[t - 1] + e[t]
    ess = dg.ess(x)
    np.testing.assert_allclose(ess / n, (1 - phi) / (1 + phi), rtol=0.15)   # τ = (1+φ)/(1−φ)
    np.testing.assert_allclose(dg.ess(rng.normal(size=(4000, 3))) / 4000, 1.0, rtol=0.15)
    E = rng.normal(size=(500, 4))
    np.testing.assert_allclose(dg.EBFMI(E), A.EBFMI(E), rtol=1e-12)
    # bundle of a short sampler-vec run on the oracle
    D, N = 3, 5
    h = A.Hamiltonian(A.UnitEuclideanMetric((D, N)), A.IsoGaussian(D))
    k = A.HMCKernel(A.Trajectory(A.EndPointTS, A.Leapfrog(0.2), A.FixedNSteps(4)))
    ths, stats = A.sample(3, h, k, rng.normal(size=(D, N)), 12, lib=oracle)
    b = dg.bundle_samples(ths, stats, param_names=["a", "b", "c"], discard_initial=2)
    assert b["value"].shape == (10, 3 + len(b["internals"]), N)
    assert b["names"][:3] == ["a", "b", "c"] and "hamiltonian_energy" in b["internals"]
    np.testing.assert_array_equal(b["value"][:, :3, :], np.stack(ths)[2:])
    j = b["names"].index("n_steps")
    assert np.all(b["value"][:, j, :] == 4)


def test_header_is_plain_c_and_the_abi_is_usable_from_c(tmp_path, oracle):
    """include/ahmc_hip.h compiles as C99 and as C++17, and a plain-C program (tests/c_abi/ask_tell_demo.c) samples a
    user log-density through the ask / tell calls — linked here against the CPU checker's implementation of the ABI"""
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c"], ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++"]):
        subprocess.run(cmd + [HEADER], check=True, capture_output=True)
    exe = str(tmp_path / "ask_tell_demo")
    odir = os.path.dirname(oracle.path)
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", inc, os.path.join(ROOT, "tests", "c_abi", "ask_tell_demo.c"), "-o", exe,
                    "-L", odir, "-lahmc_oracle", "-lm", f"-Wl,-rpath,{odir}"], check=True, capture_output=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("ok ")
    acc = float(res.stdout.split()[1])
    assert 0.6 < acc < 0.95  # dual averaging towards δ = 0.8


def test_hip_engine_fails_loudly_without_a_gpu():
    """no device → ahmc_create returns a status and a message (nothing falls back to a CPU path, nothing aborts)"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = A.CLib(A.hip_library_path())
    ctx = C.c_void_p()
    code = lib.dll.ahmc_create(0, A.capi.F64, 4, 8, None, C.byref(ctx))
    assert code == A.capi.ERR_RUNTIME and not ctx.value
    assert b"device" in lib.dll.ahmc_last_error(None)
    with pytest.raises(A.AHMCError):
        A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(4), A.IsoGaussian(4)), 8, lib=lib)


def test_user_gradient_survives_the_call_boundary(oracle):
    """Round-1 regression (ADVICE high #1): `as_ptr(<temporary>)` handed the C call the address of an array that was
    already freed, so an ExternalTarget engine started from a garbage gradient once D*N*8 B outgrew numpy's
    small-block cache.  Run in a child with MALLOC_PERTURB_ so that freed memory is visibly poisoned."""
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
import ahmc_amd as A
lib = A.CLib(%r)
D, N = 10, 200
fn = lambda th: (np.sum(-th * th / 2, axis=0), [list(row) for row in -th])   # a list: the conversion makes a temporary
e = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.ExternalTarget(D, fn)), N, lib=lib)
e.set_integrator(A.Leapfrog(0.1))
th = np.random.default_rng(0).normal(size=(D, N))
e.set_position(th)
z = e.phasepoint()
assert np.array_equal(z.lp.gradient, th), np.abs(z.lp.gradient - th).max()
np.testing.assert_allclose(z.lp.value, np.sum(-th * th / 2, axis=0), rtol=1e-13)
e.step(3)   # lf_post reads a converted temporary as well
ref = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.IsoGaussian(D)), N, lib=lib)
ref.set_integrator(A.Leapfrog(0.1)); ref.set_position(th); ref.step(3)
np.testing.assert_allclose(e.phasepoint().theta, ref.phasepoint().theta, rtol=1e-13)
p = A.capi.as_ptr(np.arange(4.0) + 1)      # the pointer owns the temporary
import ctypes
assert list((ctypes.c_double * 4).from_address(p.value)) == [1.0, 2.0, 3.0, 4.0]
print("ok")
''' % (ROOT, oracle.path)
    env = dict(os.environ, MALLOC_PERTURB_="165")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout + res.stderr


def test_info_ids_agree_between_header_python_mirror_and_oracle(oracle):
    """every AHMC_INFO_* of include/ahmc_hip.h has the same id in the Python mirror (`Engine.INFO`) and is answered by the CPU
    checker's `ahmc_get_info` (0 for what only the device engine has) — the three places an introspection key is added in"""
    import re

    hdr = open(os.path.join(ROOT, "include", "ahmc_hip.h")).read()
    ids = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"AHMC_INFO_([A-Z0-9_]+)\s*=\s*(\d+)", hdr)}
    assert len(ids) >= 14 and sorted(ids.values()) == list(range(len(ids)))
    assert ids == A.Engine.INFO, (sorted(set(ids.items()) ^ set(A.Engine.INFO.items())))
    e = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(3), A.IsoGaussian(3)), 4, rng=1, lib=oracle)
    for key in ids:
        assert isinstance(e.info(key), int), key
    e.close()


def test_checkpoint_keeps_one_nominal_step_size_one(oracle):
    """(ABI v6, AHMC_INFO_STEPSIZE_SCALAR) `ahmc_get_stepsize` always fills N values; a checkpoint records whether the context holds ONE nominal
    step size and restores it as one — a FixedIntegrationTime kernel (which takes nothing else, src/trajectory.jl:241-243) resumes; round 6: it
    came back as a vector and the resumed run was refused (found by tests/test_random_configurations.py)"""
    D, N = 4, 6
    h = A.Hamiltonian(A.DiagEuclideanMetric((D, N)), A.IsoGaussian(D))
    lf = A.Leapfrog(0.2)
    k = A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedIntegrationTime(1.0)))
    th = np.random.default_rng(0).normal(size=(D, N))

    def fresh():
        e = A.Engine(h, N, rng=3, lib=oracle)
        e.set_integrator(lf)
        e.set_position(th)
        return e

    a = fresh()
    assert a.info("stepsize_scalar") == 1
    a.run(k, 3, 0)
    st = a.get_state()
    assert st["stepsize_scalar"] and st["stepsize"].shape == (N,)
    a.run(k, 6, 0, i_first=4)
    b = fresh()
    b.set_state(st)
    assert b.info("stepsize_scalar") == 1
    b.run(k, 6, 0, i_first=4)
    np.testing.assert_array_equal(a.theta(), b.theta())
    # per-chain step sizes stay per-chain (and FixedIntegrationTime refuses them, Q6)
    a.set_integrator(A.Leapfrog(np.full(N, 0.2)))
    assert a.info("stepsize_scalar") == 0 and not a.get_state()["stepsize_scalar"]
    with pytest.raises(A.ArgumentError):
        a.run(k, 7, 0, i_first=7)
    # … and so does a scalar one after adaptation has given every chain its own
    c = fresh()
    c.adaptor_init(A.StepSizeAdaptor(0.8, lf))
    c.run(A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(3))), 4, 4)
    assert c.info("stepsize_scalar") == 0
    for e in (a, b, c):
        e.close()
