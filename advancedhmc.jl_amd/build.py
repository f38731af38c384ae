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
# -ffp-contract=on (round 4): a*b+c fuses where the SOURCE writes it in one expression, and nowhere else.  hipcc's default
# (`fast`) lets the optimiser fuse across statements, and it decided differently in different instantiations of the same
# template: the warm-up kernel (MODE 3) and the sampling kernel (MODE 0) of the geometries with E >= 4 differed in the last bit
# of a leaf's energy now and then — harmless for one transition, doubled by every dual-averaging step after it, so a warm-up
# run in one launch was not bit-identical to the same warm-up run one iteration per call (scripts/sweep_inst.py).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=on", "-Wall", "-Wno-unused-function",
         "-Wno-unused-parameter"]


PART_B_FLAGS = ["-mllvm", "-disable-machine-licm"]


def _units():
    # (the host side knows the digest of the kernel sources it was built with: ahmc_set_target_plugin refuses a plugin compiled from OTHER
    # kernel sources — a plugin whose kernels expect another scratch layout than this library allocates would write out of bounds)
    units = [("api", "ahmc_api.hip", [f'-DAHMC_KERNEL_SOURCES_DIGEST="{source_digest()}"'])]
    for tname, tdef in (("f32", "float"), ("f64", "double")):
        for tk in range(4):
            # part A: everything but …; part B: the warm-up instantiations of k_nuts and all of the multi-wave geometries', with the
            # machine-level LICM off (ahmc_inst.hpp: nuts_in_part_b — they are at their register cap and spill what it hoists)
            units.append((f"inst_{tname}_t{tk}", "ahmc_inst.hip", [f"-DAHMC_INST_T={tdef}", f"-DAHMC_INST_TK={tk}", "-DAHMC_INST_PART=0"]))
            units.append((f"inst_{tname}_t{tk}b", "ahmc_inst.hip", [f"-DAHMC_INST_T={tdef}", f"-DAHMC_INST_TK={tk}", "-DAHMC_INST_PART=1", *PART_B_FLAGS]))
    return units


def _sources_digest() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(INCLUDE, "ahmc_hip.h"), "rb").read())
    h.update(" ".join(FLAGS + PART_B_FLAGS).encode())
    return h.hexdigest()


KERNEL_SOURCES = ("ahmc_device.hpp", "ahmc_kernels.hpp", "ahmc_nuts.hpp", "ahmc_inst.hpp", "ahmc_inst.hip")


def source_digest() -> str:
    """sha256 over the SOURCES of the trajectory kernels and the compiler flags (what a target plugin is keyed on)"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def kernel_digest() -> str:
    """Digest of the DEVICE CODE of the trajectory kernels as built: sha256 over the gfx950 instructions of the eight
    log-density units (k_nuts, k_hmc, k_leapfrog, … of every geometry), taken from the disassembly the build's ISA scan
    makes anyway (`<unit>.o.isa` in the object cache).  It is what decides the kernels' instruction counts — so it is what
    profiles/counters_at_head.json is keyed on: comments, host-side edits and unrelated templates do not invalidate PMC
    counters, a different compiler or ROCm does.  Falls back to the stamp next to the .so (the GPU box has no object cache)."""
    h = hashlib.sha256()
    for name, _, _ in _units():
        if not name.startswith("inst_"):
            continue
        f = os.path.join(OBJ, name + ".o.isa")
        if not os.path.exists(f):
            try:
                return open(OUT + ".kdigest").read().strip()
            except OSError:
                return source_digest()
        h.update(name.encode())
        h.update(open(f).read().strip().encode())
    return h.hexdigest()


def unit_digests() -> dict:
    """digest of the device code of every instantiation unit, by unit name (`inst_f64_t0`, `inst_f64_t0b`, …): what PMC counters
    of ONE config are keyed on — the config's log-density family lives in two units (parts A and B), and an edit that changes
    another family's code does not invalidate them.  From the object cache, else from the stamp next to the .so."""
    import json

    out = {}
    for name, _, _ in _units():
        # (`api`: the unit that holds the dense engine's kernels — k_dense_epoch, k_dgemm, k_d_tree2 — and the target-independent
        # ones: what cfg4's counters are keyed on)
        if not (name.startswith("inst_") or name == "api"):
            continue
        f = os.path.join(OBJ, name + ".o.isa")
        if not os.path.exists(f):
            try:
                return json.load(open(OUT + ".kdigests"))
            except (OSError, ValueError):
                return {}
        out[name] = hashlib.sha256(open(f).read().strip().encode()).hexdigest()
    return out


def config_digest(tk: int, dtype: str = "f64", digests=None) -> str:
    """digest of the two units that hold the kernels of log-density family `tk` (0 iso, 1 diag, 2 funnel, 3 hier) for `dtype`"""
    d = unit_digests() if digests is None else digests
    a, b = d.get(f"inst_{dtype}_t{tk}"), d.get(f"inst_{dtype}_t{tk}b")
    return hashlib.sha256(f"{a}|{b}".encode()).hexdigest() if a and b else ""


def _write_stamps():
    import json

    with open(OUT + ".kdigest", "w") as f:  # (travels with the .so: the GPU box never sees a stale pairing)
        f.write(kernel_digest())
    with open(OUT + ".kdigests", "w") as f:
        json.dump(unit_digests(), f)


def build_hip_library(force: bool = False, verbose: bool = False) -> str:
    digest = _sources_digest()
    stamp = OUT + ".digest"  # next to the .so (the object cache is outside the repo and does not travel to the GPU box)
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == digest:
        if not os.path.exists(OUT + ".kdigest") or not os.path.exists(OUT + ".kdigests") or '"api"' not in open(OUT + ".kdigests").read():
            _write_stamps()
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
        if not force and os.path.exists(obj) and os.path.exists(dig) and (os.path.exists(obj + ".isa") or not isa_check.available()):
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
            nbad, code = isa_check.analyse_object(obj, name)
            with open(obj + ".isa", "w") as f:  # digest of the unit's device code (kernel_digest)
                f.write(code)
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
    _write_stamps()
    if verbose:
        print("linked", OUT)
    return OUT


def _include_closure(source: str, dirs=()) -> bytes:
    """The bytes of `source` and of every file it #includes with quotes, transitively (resolved next to the including file,
    then in `dirs`): what a content-keyed cache of a user's density must cover — editing a header the user's file includes
    has to rebuild the plugin / code object, not silently bind the stale one.  System headers (<…>) are the toolchain's."""
    import re

    seen, out, todo = set(), [], [os.path.abspath(source)]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.add(f)
        data = open(f, "rb").read()
        out.append(f.encode() + b"\0" + data)
        for inc in re.findall(rb'^[ \t]*#[ \t]*include[ \t]*"([^"]+)"', data, flags=re.M):
            name = inc.decode(errors="replace")
            for base in (os.path.dirname(f), *dirs):
                cand = os.path.abspath(os.path.join(base, name))
                if os.path.exists(cand):
                    todo.append(cand)
                    break
    return b"\0".join(sorted(out))


def build_target_plugin(source: str, dtype, G: int, E: int, n_params: int = -1, force: bool = False, verbose: bool = False) -> str:
    """Compile a user log-density (a header defining `ahmc_user::logdensity<T, G, E>`, contract in include/ahmc_user_target.h)
    INTO the engine's trajectory kernels for one element type and one thread geometry: ahmc_inst.hip with TK = 4 → a shared
    object that `ahmc_set_target_plugin` binds.  Cached outside the repository under <object cache>/plugins/, keyed on the
    header's content, the engine's kernel sources, the flags, dtype and geometry; scanned by isa_check like every unit of
    the engine.  ≈ 20–60 s the first time (one geometry, all kernel families and NUTS modes)."""
    import numpy as np

    source = os.path.abspath(source)
    if not os.path.exists(source):
        raise FileNotFoundError(source)
    tname = {"float32": "float", "float64": "double"}[np.dtype(dtype).name]
    kd = source_digest()
    h = hashlib.sha256()
    for part in (_include_closure(source, (INCLUDE, CSRC)), open(os.path.join(INCLUDE, "ahmc_user_target.h"), "rb").read(), kd.encode(), tname.encode(), f"{G},{E},{n_params}".encode(), " ".join(FLAGS).encode(),
                 open(os.path.join(CSRC, "ahmc_kernels.hpp"), "rb").read(), open(os.path.join(CSRC, "ahmc_inst.hpp"), "rb").read()):
        h.update(part)
    out_dir = os.path.join(OBJ, "plugins")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"libahmc_target_{h.hexdigest()[:20]}.so")
    if os.path.exists(out) and not force:
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: a target plugin is compiled from the engine's kernel sources")
    tmp = out + f".tmp{os.getpid()}"
    cmd = [hipcc, *FLAGS, "-shared", f"-DAHMC_INST_T={tname}", "-DAHMC_INST_TK=4", f"-DAHMC_PLUGIN_G={int(G)}", f"-DAHMC_PLUGIN_E={int(E)}",
           f"-DAHMC_PLUGIN_NPARAMS={int(n_params)}", f'-DAHMC_USER_TARGET_HEADER="{source}"', f'-DAHMC_SOURCES_DIGEST="{kd}"',
           "-I", INCLUDE, os.path.join(CSRC, "ahmc_inst.hip"), "-o", tmp]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on the target plugin {source}:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    if isa_check.available() and not os.environ.get("AHMC_SKIP_ISA_CHECK"):
        if isa_check.check_object(tmp, os.path.basename(source)):
            os.remove(tmp)
            raise RuntimeError(f"{source}: the compiled plugin holds a spill stored under a narrowed exec mask (isa_check.py); discarded")
    os.replace(tmp, out)
    if verbose:
        print("built target plugin", out)
    return out


def build_device_object(source: str, bitcode: bool = False, force: bool = False) -> str:
    """Compile a file that DEFINES `ahmc_user_logdensity_f64 / _f32` (include/ahmc_user_target_object.h) to what a user without
    the engine's headers would hand over: a relocatable device object (`hipcc -fgpu-rdc -c`, host ELF + bundled device
    bitcode) or — `bitcode=True` — raw device LLVM bitcode for amdgcn-amd-amdhsa (the form GPUCompiler.jl / AMDGPU.jl emit for a
    Julia function).  For tests and as the recipe INTEGRATION.md quotes; cached outside the repository by content."""
    source = os.path.abspath(source)
    h = hashlib.sha256(_include_closure(source, (INCLUDE,)) + (b"bc" if bitcode else b"o") + " ".join(FLAGS).encode()).hexdigest()[:20]
    out_dir = os.path.join(OBJ, "objects")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"{os.path.splitext(os.path.basename(source))[0]}_{h}.{'bc' if bitcode else 'o'}")
    if os.path.exists(out) and not force:
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = [f for f in FLAGS if f != "-fno-gpu-rdc"] + ["-fgpu-rdc"]
    cmd = [hipcc, *flags, "-I", INCLUDE, "-c", source, "-o", out + ".tmp"] + (["--cuda-device-only", "-emit-llvm"] if bitcode else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {source}:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")
    os.replace(out + ".tmp", out)
    return out


def build_target_plugin_from_object(obj: str, dtype, G: int, E: int, n_params: int = -1, force: bool = False, verbose: bool = False) -> str:
    """A user log-density that exists only as COMPILED device code (include/ahmc_user_target_object.h: one C symbol,
    `ahmc_user_logdensity_f64 / _f32`) → a target plugin for `ahmc_set_target_plugin`: the engine's trajectory kernels are
    compiled under -fgpu-rdc with the symbol as their log-density family (TK = 4) and LINKED with the object; the device link
    runs LTO over both, so a bitcode object is inlined into the leaf loop like a header plugin's function.

    `obj`: a relocatable object of `hipcc -fgpu-rdc -c` (host ELF with bundled device bitcode), or raw device bitcode / IR
    for amdgcn-amd-amdhsa (`.bc` / `.ll`: what GPUCompiler.jl emits) — wrapped with an empty host object by
    clang-offload-bundler.  Returns the plugin's path; `<path>.json` says whether the density was inlined (no call left in the
    kernels).  Cached by the object's content, the engine's kernel sources, flags, dtype and geometry; ISA-scanned."""
    import json

    import numpy as np

    obj = os.path.abspath(obj)
    if not os.path.exists(obj):
        raise FileNotFoundError(obj)
    tname = {"float32": "float", "float64": "double"}[np.dtype(dtype).name]
    kd = source_digest()
    shim = os.path.join(INCLUDE, "ahmc_user_target_object.h")
    h = hashlib.sha256()
    for part in (open(obj, "rb").read(), open(shim, "rb").read(), kd.encode(), tname.encode(), f"obj,{G},{E},{n_params}".encode(), " ".join(FLAGS).encode(),
                 open(os.path.join(CSRC, "ahmc_kernels.hpp"), "rb").read(), open(os.path.join(CSRC, "ahmc_inst.hpp"), "rb").read()):
        h.update(part)
    out_dir = os.path.join(OBJ, "plugins")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"libahmc_target_obj_{h.hexdigest()[:20]}.so")
    if os.path.exists(out) and os.path.exists(out + ".json") and not force:
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: a target plugin is linked from the engine's kernel sources")
    llvm = isa_check.LLVM
    tmp = out + f".tmp{os.getpid()}"
    work = []

    def run(cmd, what):
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            for f in work:
                if os.path.exists(f):
                    os.remove(f)
            raise RuntimeError(f"{what} failed for the target object {obj}:\n{' '.join(cmd)}\n{res.stdout}\n{res.stderr}")

    user_o = obj
    if obj.endswith((".bc", ".ll")):
        # raw device bitcode: bundle it with an empty host object, as `hipcc -fgpu-rdc -c` would have
        host_c, host_o, user_o = tmp + ".host.c", tmp + ".host.o", tmp + ".user.o"
        work += [host_c, host_o, user_o]
        open(host_c, "w").write("\n")
        run([f"{llvm}/clang", "-c", "-fPIC", host_c, "-o", host_o], "compiling the empty host half")
        run([f"{llvm}/clang-offload-bundler", "--type=o", "--targets=host-x86_64-unknown-linux-gnu,hip-amdgcn-amd-amdhsa--gfx950",
             f"--input={host_o}", f"--input={obj}", f"--output={user_o}"], "bundling the device bitcode")
    rdc = [f for f in FLAGS if f != "-fno-gpu-rdc"] + ["-fgpu-rdc"]
    inst_o = tmp + ".inst.o"
    work.append(inst_o)
    run([hipcc, *rdc, f"-DAHMC_INST_T={tname}", "-DAHMC_INST_TK=4", f"-DAHMC_PLUGIN_G={int(G)}", f"-DAHMC_PLUGIN_E={int(E)}",
         f"-DAHMC_PLUGIN_NPARAMS={int(n_params)}", "-DAHMC_USER_TARGET_FROM_OBJECT=1", f'-DAHMC_USER_TARGET_HEADER="{shim}"',
         f'-DAHMC_SOURCES_DIGEST="{kd}"', "-I", INCLUDE, "-c", os.path.join(CSRC, "ahmc_inst.hip"), "-o", inst_o], "compiling the engine's kernels (-fgpu-rdc)")
    run([hipcc, "--offload-arch=gfx950", "-fgpu-rdc", "--hip-link", "-shared", "-fPIC", inst_o, user_o, "-o", tmp], "the device link")
    for f in work:
        if os.path.exists(f):
            os.remove(f)
    inlined = None
    if isa_check.available():
        import tempfile

        with tempfile.TemporaryDirectory(prefix="ahmc_isa_") as td:
            text = isa_check.disassemble(tmp, td)
        inlined = ("s_swappc_b64" not in text) and ("ahmc_user_logdensity" not in text)
        if not os.environ.get("AHMC_SKIP_ISA_CHECK") and isa_check.check_object(tmp, os.path.basename(obj)):
            os.remove(tmp)
            raise RuntimeError(f"{obj}: the linked plugin holds a spill stored under a narrowed exec mask (isa_check.py); discarded")
    os.replace(tmp, out)
    with open(out + ".json", "w") as f:
        json.dump({"object": obj, "dtype": tname, "G": int(G), "E": int(E), "inlined": inlined}, f)
    if verbose:
        print("built target plugin", out, "(density inlined)" if inlined else "(density CALLED per leapfrog: no bitcode in the object?)")
    return out


def build_code_object(source: str, force: bool = False) -> str:
    """hipcc --genco of a file of user KERNELS (ahmc_set_target_kernel) → a gfx950 code object for hipModuleLoad; cached
    outside the repository by content."""
    source = os.path.abspath(source)
    h = hashlib.sha256(_include_closure(source, (INCLUDE,))).hexdigest()[:20]
    out_dir = os.path.join(OBJ, "kernels")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"{os.path.splitext(os.path.basename(source))[0]}_{h}.hsaco")
    if os.path.exists(out) and not force:
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "--genco", source, "-o", out + ".tmp"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc --genco failed on {source}:\n{res.stdout}\n{res.stderr}")
    os.replace(out + ".tmp", out)
    return out
