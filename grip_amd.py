"""Import shim: loads the package directory `menghini-neurips23-code_amd/`
under the importable name `grip_amd` (the directory name has hyphens)."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.join(_here, "menghini-neurips23-code_amd")
_spec = importlib.util.spec_from_file_location(
    "grip_amd", os.path.join(_pkg, "__init__.py"), submodule_search_locations=[_pkg]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["grip_amd"] = _mod
_spec.loader.exec_module(_mod)
