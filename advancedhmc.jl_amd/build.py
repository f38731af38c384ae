"""Build the HIP engine in-tree: hipcc --offload-arch=gfx950 → csrc/libahmc_hip.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels with the repo snapshot to the GPU box.

Translation units (compiled in parallel; objects cached OUTSIDE the repository — `$AHMC_BUILD_CACHE`, else
`$XDG_CACHE_HOME/ahmc_build`, else `~/.cache/ahmc_build` — so that nothing but the linked .so and its two digest stamps
ever lands in the tree that `gpurun` snapshots; each object carries the digest of the files it was compiled from — taken from the compiler's own dependency list — so that a change to the host
side or to the dense engine does not recompile the eight log-density instantiations):
  ahmc_api.hip                      host side of the C ABI + the target-independent kernels
  ahmc_inst.hip  × {f32,f64} × {iso,diag,funnel,hier}
                                    the kernels that evaluate a built-in log-density family, which is
                                    a compile-time parameter (see ahmc_inst.hpp)
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

try:
    from . import isa_check
except ImportError:  # run as a plain script / imported by path
    import isa_check

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")


def _cache_dir() -> str:
    """object cache outside the repository (the tree that travels to the GPU box holds sources, the .so and its stamps only)"""
    base = os.environ.get("AHMC_BUILD_CACHE")
    if not base:
        base = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "ahmc_build")
    return base


OBJ = _cache_dir()
OUT = os.path.join(CSRC, "libahmc_hip.so")
INCLUDE = os.path.join(_HERE, "..", "include")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-unused-parameter"]


def _units():
    units = [("api", "ahmc_api.hip", [])]
    for tname, tdef in (("f32", "float"), ("f64", "double")):
        for tk in range(4):
            units.append((f"inst_{tname}_t{tk}", "ahmc_inst.hip", [f"-DAHMC_INST_T={tdef}", f"-DAHMC_INST_TK={tk}"]))
    return units


def _sources_digest() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(INCLUDE, "ahmc_hip.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


KERNEL_SOURCES = ("ahmc_device.hpp", "ahmc_kernels.hpp", "ahmc_nuts.hpp", "ahmc_inst.hpp", "ahmc_inst.hip")


def kernel_digest() -> str:
    """sha256 over the DEVICE code of the trajectory kernels (k_nuts, k_hmc, k_leapfrog, …) and the compiler flags: what
    decides their instruction counts.  bench.py matches it against profiles/counters_at_head.json — host-side edits
    (ahmc_api.hip …) do not invalidate PMC counters taken on the same kernels."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_hip_library(force: bool = False, verbose: bool = False) -> str:
    digest = _sources_digest()
    stamp = OUT + ".digest"  # next to the .so (the object cache is outside the repo and does not travel to the GPU box)
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == digest:
        if not os.path.exists(OUT + ".kdigest"):
            with open(OUT + ".kdigest", "w") as f:
                f.write(kernel_digest())
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP engine (and there is no fallback)")
    os.makedirs(OBJ, exist_ok=True)

    def unit_digest(depfile, defs):
        """sha256 over the unit's flags and the project files the compiler read for it (None if unknown)"""
        try:
            words = open(depfile).read().replace("\\\n", " ").split()
        except OSError:
            return None
        roots = (os.path.realpath(CSRC), os.path.realpath(INCLUDE))
        files = sorted({os.path.realpath(w) for w in words[1:] if os.path.realpath(w).startswith(roots)})
        if not files:
            return None
        h = hashlib.sha256(" ".join(FLAGS + defs).encode())
        for f in files:
            if not os.path.exists(f):
                return None
            h.update(f.encode())
            h.update(open(f, "rb").read())
        return h.hexdigest()

    def compile_one(unit):
        name, src, defs = unit
        obj = os.path.join(OBJ, name + ".o")
        dep, dig = obj + ".d", obj + ".digest"
        if not force and os.path.exists(obj) and os.path.exists(dig):
            d = unit_digest(dep, defs)
            if d is not None and d == open(dig).read():
                if verbose:
                    print("up to date", name)
                return obj
        cmd = [hipcc, *FLAGS, *defs, "-I", INCLUDE, "-MD", "-MF", dep, "-c", os.path.join(CSRC, src), "-o", obj]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name}:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
        # static check of the code object just made: a VGPR spill stored only under a narrowed exec mask and reloaded
        # under the full one is a register-allocation artefact that returns wrong results on the device (isa_check.py,
        # DESIGN §7.3).  The build fails on it — it is not a skippable test.
        if isa_check.available() and not os.environ.get("AHMC_SKIP_ISA_CHECK"):
            nbad = isa_check.check_object(obj, name)
            if nbad:
                os.remove(obj)
                raise RuntimeError(f"{name}: {nbad} spill store(s) under a narrowed exec mask with outside reloads (see above); "
                                   "the object was discarded")
        d = unit_digest(dep, defs)
        if d is not None:
            with open(dig, "w") as f:
                f.write(d)
        if verbose:
            print("compiled", name)
        return obj

    units = _units()
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, units))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-Wl,-Bsymbolic-functions", *objs, "-o", OUT]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    for f in os.listdir(CSRC):  # unbundled device images an interrupted link may leave behind
        if f.startswith("libahmc_hip.so.") and f != "libahmc_hip.so.digest":
            os.remove(os.path.join(CSRC, f))
    with open(stamp, "w") as f:
        f.write(digest)
    with open(OUT + ".kdigest", "w") as f:  # (travels with the .so: the GPU box never sees a stale pairing)
        f.write(kernel_digest())
    if verbose:
        print("linked", OUT)
    return OUT
