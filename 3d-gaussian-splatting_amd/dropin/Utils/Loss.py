"""Module shim for ``from Utils.Loss import l1_loss, ssim_loss`` (``Trainer.py:7``): the two functions of
``Utils/Loss.py:4-5,59-97`` on the HIP loss kernel, differentiable, with the reference's arguments."""
import importlib.util
import os
import sys

if "gsplat_amd" not in sys.modules:
    _pkg = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _spec = importlib.util.spec_from_file_location("gsplat_amd", os.path.join(_pkg, "__init__.py"),
                                                   submodule_search_locations=[_pkg])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["gsplat_amd"] = _mod
    _spec.loader.exec_module(_mod)

from gsplat_amd.loss import l1_loss, ssim_loss  # noqa: E402,F401
