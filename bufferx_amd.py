"""Import alias: the product package lives in the directory ``buffer-x_amd/`` (not a valid Python
identifier), this shim registers it under the importable name ``bufferx_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "buffer-x_amd")
_spec = importlib.util.spec_from_file_location(
    "bufferx_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bufferx_amd"] = _mod
_spec.loader.exec_module(_mod)
