"""Import shim: the package directory is named `scan-rs_amd/` (not a valid Python
identifier), so `import scanrs_amd` loads it from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scan-rs_amd")
_spec = importlib.util.spec_from_file_location(
    "scanrs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["scanrs_amd"] = _mod
_spec.loader.exec_module(_mod)
