"""CPU: the drop-in GaussianModel module (dropin/GaussianModel.py) - what main.py and Trainer.py use of the
reference's class that needs no GPU: the reference's constructor forms, the activated properties, the
3-D covariance against the reference's own (fixture), the all-false broadcast remove mask of
Trainer.py:68."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import load_golden  # noqa: E402


def dropin_class():
    path = os.path.join(ROOT, "3d-gaussian-splatting_amd", "dropin")
    sys.path.insert(0, path)
    try:
        sys.modules.pop("GaussianModel", None)
        from GaussianModel import GaussianModel
    finally:
        sys.path.remove(path)
        sys.modules.pop("GaussianModel", None)
    return GaussianModel


def test_reference_constructor_forms():
    GaussianModel = dropin_class()
    m = GaussianModel()                      # main.py:31
    assert m._size == 0 and m.capacity == 0 and m.max_sh_degree == 3 and m.curr_sh_degree == 0
    assert m._sh.shape == (0, 3, 16) and len(list(m.parameters())) == 5
    assert GaussianModel(2)._sh.shape == (0, 3, 9) and GaussianModel(max_sh_degree=1).max_sh_degree == 1
    assert len(m.construct_list_of_attributes()) == 62
    import torch
    assert m.scale_activation is torch.exp and m.scale_inv_activation is torch.log and m.opactiy_activation is torch.sigmoid   # GaussianModel.py:29-32


def test_covariance_and_properties_match_the_reference():
    GaussianModel = dropin_class()
    g = load_golden("syn_n256_64x48_deg1_t64")
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k]))  # noqa: E731
    m = GaussianModel(t("coords"), t("scale"), t("rotation"), t("opacity").reshape(-1, 1), t("sh"), sh_degree=int(g["sh_degree"]))
    np.testing.assert_allclose(m.covariance.detach().numpy(), g["ref_cov3d"], rtol=1e-5, atol=1e-7)
    mask = torch.from_numpy(g["ref_mask"])
    np.testing.assert_allclose(m.get_covariance(mask).detach().numpy(), g["ref_cov3d"][g["ref_mask"]], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(m.opacity.detach().numpy()[g["ref_mask"]].reshape(-1), g["ref_opacity_act"].reshape(-1), rtol=1e-6)
    assert m.sh.shape == (256, 3, 4) and m.size == 256 and m.sh_degree == 1
    np.testing.assert_allclose(m.rotation.detach().norm(dim=1).numpy(), 1.0, rtol=1e-6)


def test_broadcast_remove_mask_of_the_reference_trainer():
    GaussianModel = dropin_class()
    g = torch.Generator().manual_seed(0)
    m = GaussianModel(torch.randn(6, 3, generator=g), torch.randn(6, 3, generator=g), torch.randn(6, 4, generator=g),
                      torch.randn(6, 1, generator=g) + 5.0, torch.randn(6, 3, 16, generator=g))
    radius = torch.rand(4) * 10
    remove_mask = (m.opacity < 0.005) & (radius > 1000)          # Trainer.py:68: [N,1] & [M] -> [N,M]
    assert remove_mask.shape == (6, 4) and m.remove(remove_mask) == 0 and m._size == 6
    with pytest.raises(IndexError):
        m.remove(torch.ones(6, 4, dtype=torch.bool))


def test_utils_shim_overrides_the_loss_module_only(tmp_path):
    """dropin/Utils: Utils.Loss is this package's, every other Utils module still the checkout's (here a
    stand-in checkout with one module)."""
    import subprocess
    import sys
    fake = tmp_path / "checkout"
    (fake / "Utils").mkdir(parents=True)
    (fake / "Utils" / "__init__.py").write_text("")
    (fake / "Utils" / "ContainerUtils.py").write_text("def torch2numpy(t):\n    return 'from the checkout'\n")
    (fake / "Utils" / "Loss.py").write_text("def l1_loss(a, b):\n    return 'from the checkout'\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dropin = os.path.join(root, "3d-gaussian-splatting_amd", "dropin")
    code = (
        "import sys\n"
        f"sys.path[:0] = [{dropin!r}, {str(fake)!r}]\n"
        "from Utils.ContainerUtils import torch2numpy\n"
        "from Utils.Loss import l1_loss, ssim_loss\n"
        "from GaussianRenderer import GaussianRenderer\n"
        "from GaussianModel import GaussianModel\n"
        "print(torch2numpy(None), l1_loss.__module__, ssim_loss.__module__, GaussianRenderer.__module__, GaussianModel.__module__)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split("\n")[0] == "from the checkout gsplat_amd.loss gsplat_amd.loss gsplat_amd.renderer gsplat_amd.model"
