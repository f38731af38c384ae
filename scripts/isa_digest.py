#!/usr/bin/env python
"""Per-kernel ISA digests of the built HIP engine (the object cache of build.py, outside the repository).

    python scripts/isa_digest.py out.json            # write {unit: {kernel: sha1 of its gfx950 instructions}}
    python scripts/isa_digest.py out.json base.json  # ... and report kernels whose code differs from base.json

Used when host code or a new kernel is added without a GPU at hand: an existing kernel whose digest is unchanged
runs the same instructions as the build that passed the GPU parity tests."""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ahmc_amd import build as _B  # noqa: E402

OBJ = os.environ.get("AHMC_OBJ_DIR") or _B.OBJ  # the build's object cache (outside the repository); AHMC_OBJ_DIR: another build's objects
LLVM = "/opt/rocm/lib/llvm/bin"


def unit_digests(obj, tmp):
    import shutil

    cp = os.path.join(tmp, os.path.basename(obj))
    shutil.copyfile(obj, cp)  # llvm-objdump --offloading writes the device image next to its input
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", cp], capture_output=True, check=True)
    co = cp + ".0.hipv4-amdgcn-amd-amdhsa--gfx950"
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", co], capture_output=True, text=True, check=True).stdout
    out, name, h, n = {}, None, None, 0
    for line in txt.splitlines():
        m = re.match(r"^<(.+)>:$", line.strip()) if line and not line.startswith(" ") and not line.startswith("\t") else None
        if m:
            if name:
                out[name] = (h.hexdigest(), n)
            name, h, n = m.group(1), hashlib.sha1(), 0
            continue
        if name and line.strip() and line.strip() != "...":  # ("..." = zero padding after the last instruction)
            ins = re.sub(r"\s*//.*$", "", line.strip())
            h.update(ins.encode() + b"\n")
            n += 1
    if name:
        out[name] = (h.hexdigest(), n)
    demangled = {}
    names = list(out)
    dm = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for k, d in zip(names, dm):
        demangled[d if d else k] = out[k]
    return demangled


def main():
    tmp = "/tmp/isa"
    os.makedirs(tmp, exist_ok=True)
    res = {}
    for f in sorted(os.listdir(OBJ)):
        if f.endswith(".o"):
            res[f[:-2]] = unit_digests(os.path.join(OBJ, f), tmp)
    with open(sys.argv[1], "w") as fh:
        json.dump(res, fh, indent=0)
    print({u: len(k) for u, k in res.items()})
    if len(sys.argv) > 2:
        base = json.load(open(sys.argv[2]))
        changed = added = removed = 0
        for u in sorted(set(res) | set(base)):
            a, b = res.get(u, {}), base.get(u, {})
            for k in sorted(set(a) | set(b)):
                if k not in b:
                    added += 1
                    print(f"+ {u}: {k[:140]} ({a[k][1]} instructions)")
                elif k not in a:
                    removed += 1
                    print(f"- {u}: {k[:140]}")
                elif a[k][0] != b[k][0]:
                    changed += 1
                    print(f"* {u}: {k[:140]} ({b[k][1]} -> {a[k][1]} instructions)")
        print(f"changed {changed}, added {added}, removed {removed}")


if __name__ == "__main__":
    main()
