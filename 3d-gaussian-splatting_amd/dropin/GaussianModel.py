"""Module shim: with this directory on ``sys.path`` ahead of the reference checkout,
``from GaussianModel import GaussianModel`` (``main.py:6``, ``Trainer.py:9``) resolves to the model class
of this package.  The reference's own module imports ``simple_knn``, a CUDA extension, and ``plyfile``;
this one needs neither (``gsr_knn_dist2`` and ``ply.py`` stand in for them)."""
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

from gsplat_amd.model import GaussianModel  # noqa: E402,F401
