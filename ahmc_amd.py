"""Import shim: the package directory `advancedhmc.jl_amd/` (name fixed by the project layout)
contains a dot, so it cannot be imported by name.  `import ahmc_amd` loads it under this alias."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "advancedhmc.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "ahmc_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ahmc_amd"] = _mod
_spec.loader.exec_module(_mod)
