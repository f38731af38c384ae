"""Shared fixtures.

`-m "not gpu"` tests run on CPU only: the oracle against the reference's known-answer tests,
host logic, and the C-ABI symbol check.  `-m gpu` tests are the parity tests proper: they call
the HIP engine through the C ABI and compare it with the CPU oracle (oracle/ — test
infrastructure, loaded ONLY from here, smoke() and bench.py's cpu_baseline leg).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ahmc_amd as A  # noqa: E402

ORACLE_SO = os.path.join(ROOT, "oracle", "libahmc_oracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def build_oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import build_oracle as bo
    finally:
        sys.path.pop(0)
    return bo.build()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle bound through the same ctypes class as the product library."""
    return A.CLib(build_oracle())


@pytest.fixture(scope="session")
def hip():
    """The HIP engine.  No fallback: a missing library or a missing GPU fails the test.
    (AHMC_TEST_DRYRUN_ON_ORACLE=1 — never set by the suite or the driver — binds the fixture to the CPU checker
    instead, to shake Python errors out of gpu-marked TEST code on a machine without a GPU; such a run proves nothing
    about the HIP engine.)"""
    if os.environ.get("AHMC_TEST_DRYRUN_ON_ORACLE") == "1":
        return A.CLib(build_oracle())
    import torch

    assert torch.cuda.is_available(), "gpu-marked test needs a visible MI355X"
    lib = A.load_hip_library()
    assert lib.backend == "hip:gfx950"
    return lib


@pytest.fixture
def rng():
    return np.random.default_rng(20260925)
