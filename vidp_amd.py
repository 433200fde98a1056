"""
Import shim: the product package lives in the directory ``vi-diffusion-processes_amd/`` whose name is
not a valid Python identifier, so it is registered here under the importable name ``vidp_amd``.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "vi-diffusion-processes_amd")
_spec = importlib.util.spec_from_file_location(
    "vidp_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vidp_amd"] = _mod
_spec.loader.exec_module(_mod)
