"""Import shim: makes the hyphenated directory `reze-engine_amd/` importable as `reze_engine_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reze-engine_amd")
_spec = importlib.util.spec_from_file_location(
    "reze_engine_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["reze_engine_amd"] = _mod
_spec.loader.exec_module(_mod)
