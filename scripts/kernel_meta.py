#!/usr/bin/env python
"""Register / scratch / LDS footprint of the kernels in an object file of the build (the code object's metadata notes):

    python scripts/kernel_meta.py <unit.o> [name filter]      e.g.  ~/.cache/ahmc_build/inst_f64_t2.o 'k_nuts<double, 16, 2'
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_meta(obj):
    with tempfile.TemporaryDirectory(prefix="ahmc_meta_") as tmp:
        cp = os.path.join(tmp, os.path.basename(obj))
        shutil.copyfile(obj, cp)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", cp], capture_output=True, check=True)
        co = cp + ".0.hipv4-amdgcn-amd-amdhsa--gfx950"
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count":
            cur = {"agpr_count": int(v)}
            out.append(cur)
        elif cur is not None and k in ("name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count",
                                       "sgpr_spill_count", "max_flat_workgroup_size"):
            cur[k] = v if k == "name" else int(v)
    return [k for k in out if "name" in k]


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernel_meta(sys.argv[1]):
        name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
        if flt in name:
            print(f"{name[:90]:90s} vgpr {k.get('vgpr_count'):4d} agpr {k.get('agpr_count'):4d} sgpr {k.get('sgpr_count'):4d} scratch {k.get('private_segment_fixed_size'):5d} "
                  f"spill v{k.get('vgpr_spill_count')} s{k.get('sgpr_spill_count')} lds {k.get('group_segment_fixed_size')}")
