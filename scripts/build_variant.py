#!/usr/bin/env python
"""Build a VARIANT of the HIP engine next to the shipped one, for A/B measurements on the GPU box:

    python scripts/build_variant.py mfma -DAHMC_MFMA_REDUCE=1
    AHMC_HIP_LIB=advancedhmc.jl_amd/csrc/variants/libahmc_hip_mfma.so python bench.py ...

Same sources, extra -D flags; objects under <object cache>/variants/<name>/ (outside the repository), the .so under csrc/variants/
(travels: *.so is git-ignored, not gpurun-ignored).  The shipped library is never touched."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ahmc_amd as A  # noqa: E402
from ahmc_amd import build as B  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    obj = os.path.join(B.OBJ, "variants", name)
    out_dir = os.path.join(B.CSRC, "variants")
    os.makedirs(obj, exist_ok=True)
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"libahmc_hip_{name}.so")
    only = os.environ.get("VARIANT_UNITS")  # e.g. "api,inst_f64_t0": the other units are taken from the shipped build's objects

    def one(u):
        uname, src, defs = u
        o = os.path.join(obj, uname + ".o")
        if only and uname not in only.split(","):
            return os.path.join(B.OBJ, uname + ".o")
        cmd = ["hipcc", *B.FLAGS, *defs, *extra, "-I", B.INCLUDE, "-c", os.path.join(B.CSRC, src), "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stderr[-3000:])
        # VARIANT_ALLOW_MASKED=1: a MEASUREMENT build whose flagged kernels are not the ones it runs (never the shipped library)
        if B.isa_check.available() and B.isa_check.check_object(o, f"{name}/{uname}") and not os.environ.get("VARIANT_ALLOW_MASKED"):
            raise SystemExit(f"{name}/{uname}: masked-spill pattern in the code object (isa_check.py) - variant not built")
        return o

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        objs = list(pool.map(one, B._units()))
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-Wl,-Bsymbolic-functions", *objs, "-o", out],
                       capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr[-3000:])
    for f in os.listdir(out_dir):
        if ".so." in f:
            os.remove(os.path.join(out_dir, f))
    print(out)


if __name__ == "__main__":
    main()
