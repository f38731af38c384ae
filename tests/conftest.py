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
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # (tests/parity_util.py)

import ahmc_amd as A  # noqa: E402

ORACLE_SO = os.path.join(ROOT, "oracle", "libahmc_oracle.so")


DRYRUN = os.environ.get("AHMC_TEST_DRYRUN_ON_ORACLE") == "1"
DRYRUN_REASON = ("AHMC_TEST_DRYRUN_ON_ORACLE=1: the `hip` fixture was the CPU checker, so this run compared the oracle with itself "
                 "and says nothing about the HIP engine")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.hookimpl(wrapper=True)
def pytest_runtest_call(item):
    """A dry run can never be green: a gpu-marked test whose body ran through on the oracle is reported as SKIPPED with
    the reason above (a Python error in the test code still fails it, which is all a dry run is for)."""
    res = yield
    if DRYRUN and item.get_closest_marker("gpu") is not None:
        pytest.skip(DRYRUN_REASON)
    return res


def build_oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import build_oracle as bo
    finally:
        sys.path.pop(0)
    return bo.build()


def pytest_terminal_summary(terminalreporter):
    """the margin-aware parity record of the run (tests/parity_util.py): flips seen, near-ties, the largest margin a flip needed"""
    import json

    import parity_util as PU

    if not PU.RECORDS:
        return
    tot = PU.summary()
    terminalreporter.write_line("margin-aware parity: " + json.dumps(tot))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        flips = [dict(zip(("what", "dtype", "chains", "differ", "near_ties", "max_margin_of_differing", "min_margin_of_agreeing"), r))
                 for r in PU.RECORDS if r[3] or r[4]]
        with open(os.path.join(out, "parity_margins.json"), "w") as f:
            json.dump({"dry_run_on_oracle": DRYRUN, "totals": tot, "comparisons_with_flips_or_near_ties": flips}, f, indent=1)
    except OSError:
        pass


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle bound through the same ctypes class as the product library."""
    return A.CLib(build_oracle())


@pytest.fixture(scope="session")
def hip():
    """The HIP engine.  No fallback: a missing library or a missing GPU fails the test.
    (AHMC_TEST_DRYRUN_ON_ORACLE=1 — never set by the suite or the driver — binds the fixture to the CPU checker
    instead, to shake Python errors out of gpu-marked TEST code on a machine without a GPU; such a run proves nothing
    about the HIP engine and NO test of it passes: `pytest_runtest_call` above turns every one that ran through into a
    skip with that reason.)"""
    if DRYRUN:
        return A.CLib(build_oracle())
    import torch

    assert torch.cuda.is_available(), "gpu-marked test needs a visible MI355X"
    lib = A.load_hip_library()
    assert lib.backend == "hip:gfx950"
    return lib


@pytest.fixture
def rng():
    return np.random.default_rng(20260925)
