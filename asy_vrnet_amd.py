"""Import alias: ``import asy_vrnet_amd`` loads the package kept in the directory
``asy-vrnet_amd/`` (a hyphen is not importable as a Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "asy-vrnet_amd")
_spec = importlib.util.spec_from_file_location(
    "asy_vrnet_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["asy_vrnet_amd"] = _mod
_spec.loader.exec_module(_mod)
