"""`import capnet` -> the package in ./image-caption-emotion-indonesia_amd (the directory name
is not a Python identifier, so it is loaded by path and registered under this name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image-caption-emotion-indonesia_amd")
_spec = importlib.util.spec_from_file_location(
    "capnet", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["capnet"] = _mod
_spec.loader.exec_module(_mod)
