"""Import shim: the package lives in the directory `waves.jl_amd/` (the name the project layout prescribes), which is
not a valid Python identifier.  `import waves_jl_amd` loads that directory as the package `waves_jl_amd`."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "waves.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "waves_jl_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["waves_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
