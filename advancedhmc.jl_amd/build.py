"""Build the HIP engine in-tree: hipcc --offload-arch=gfx950 → csrc/libahmc_hip.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(CSRC, "libahmc_hip.so")
SOURCES = ["ahmc_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-Wno-unused-parameter"]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".hpp", ".cuh"))]
    deps.append(os.path.join(_HERE, "..", "include", "ahmc_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP engine (and there is no fallback)")
    cmd = [hipcc, *FLAGS, "-I", os.path.join(_HERE, "..", "include"), *[os.path.join(CSRC, s) for s in SOURCES],
           "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr:
        print(res.stderr)
    return OUT
