"""Import alias for the package directory ``3d-gaussian-splatting_amd/``.

The directory name is not a valid Python identifier, so ``import gsplat_amd``
executes this file, which loads the directory as a regular package and
installs it in ``sys.modules`` under the name ``gsplat_amd``.  After that
``from gsplat_amd.renderer import GaussianRenderer`` etc. work as usual.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "3d-gaussian-splatting_amd")
_spec = importlib.util.spec_from_file_location(
    "gsplat_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gsplat_amd"] = _mod
_spec.loader.exec_module(_mod)
