"""hipModuleLoad / hipModuleGetFunction through ctypes: a hipFunction_t for `KernelTarget` (ahmc_set_target_kernel) from a
code object file — the Python stand-in for what a host language with its own GPU compiler (AMDGPU.jl) already holds.
The HIP runtime used is the copy already mapped into the process (torch's bundled libamdhip64 when torch is imported, the
one libahmc_hip.so pulled in otherwise), so the handle belongs to the runtime the engine launches with."""
from __future__ import annotations

import ctypes as C
import os

_RUNTIME = None


def hip_runtime() -> C.CDLL:
    global _RUNTIME
    if _RUNTIME is None:
        path = None
        try:
            with open("/proc/self/maps") as f:
                for line in f:
                    if "libamdhip64" in line:
                        path = line.split()[-1]
                        break
        except OSError:
            pass
        _RUNTIME = C.CDLL(path or "libamdhip64.so")
    return _RUNTIME


class Module:
    """a loaded code object; keeps the hipModule_t alive as long as its functions are in use"""

    def __init__(self, path: str):
        rt = hip_runtime()
        self._mod = C.c_void_p()
        rc = rt.hipModuleLoad(C.byref(self._mod), os.fsencode(path))
        if rc != 0:
            raise RuntimeError(f"hipModuleLoad({path}) failed with hipError {rc}")

    def function(self, name: str) -> int:
        rt = hip_runtime()
        fn = C.c_void_p()
        rc = rt.hipModuleGetFunction(C.byref(fn), self._mod, name.encode())
        if rc != 0:
            raise RuntimeError(f"hipModuleGetFunction({name}) failed with hipError {rc}")
        return fn.value

    def close(self):
        """hipModuleUnload — only once no engine holds a function of this module any more (the engine launches the
        hipFunction_t it was given; it does not own the module)"""
        if self._mod is not None and self._mod.value:
            hip_runtime().hipModuleUnload(self._mod)
        self._mod = None
