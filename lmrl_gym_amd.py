"""Import shim: exposes the package that lives in the directory `lmrl-gym_amd/` (not a valid Python
identifier) under the importable name `lmrl_gym_amd`."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lmrl-gym_amd")
_spec = _ilu.spec_from_file_location(
    "lmrl_gym_amd", _os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["lmrl_gym_amd"] = _mod
_spec.loader.exec_module(_mod)
