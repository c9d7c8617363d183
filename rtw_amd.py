"""Import alias for the package directory ``raytracingweekend.jl_amd/`` (a dot in a directory
name cannot appear in an ``import`` statement).  ``import rtw_amd`` yields that package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raytracingweekend.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "rtw_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rtw_amd"] = _mod
_spec.loader.exec_module(_mod)
