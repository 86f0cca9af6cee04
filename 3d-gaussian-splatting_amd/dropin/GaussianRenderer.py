"""Module shim: put this directory on ``sys.path`` ahead of the reference checkout and
``from GaussianRenderer import GaussianRenderer`` (``Trainer.py:10``) resolves to the HIP renderer."""
import importlib.util
import os
import sys

if "gsplat_amd" not in sys.modules:
    _pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _spec = importlib.util.spec_from_file_location("gsplat_amd", os.path.join(_pkg, "__init__.py"),
                                                   submodule_search_locations=[_pkg])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["gsplat_amd"] = _mod
    _spec.loader.exec_module(_mod)

from gsplat_amd.renderer import GaussianRenderer  # noqa: E402,F401
