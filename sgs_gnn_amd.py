"""Import shim: the package lives in the directory `sgs-gnn_amd/` (not a valid Python
identifier), so `import sgs_gnn_amd` loads it from there under this name."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "sgs-gnn_amd")
_spec = importlib.util.spec_from_file_location(
    "sgs_gnn_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sgs_gnn_amd"] = _mod
_spec.loader.exec_module(_mod)
