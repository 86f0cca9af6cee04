"""GPU tests of the drop-in GaussianRenderer class (the reference's call surface)."""
import numpy as np
import pytest
import torch

from util import assert_close, grad_tolerance

pytestmark = pytest.mark.gpu


def _setup(scene):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.model import GaussianParams
    from gsplat_amd.renderer import GaussianRenderer
    dev = torch.device("cuda:0")
    model = GaussianParams.from_scene(scene, dev)
    return model, GaussianRenderer(model), scene.camera(device=dev)


def test_call_surface_and_autograd(oracle_mod):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.scenes import synthetic_scene
    sc = synthetic_scene(5000, 160, 96, 2, 21)
    model, renderer, cam = _setup(sc)
    assert renderer.device == model._coords.device
    img, mask, screen_coords, radius = renderer(cam)            # default tile edge 64
    assert img.shape == (160, 96, 3) and img.requires_grad
    assert mask.dtype == torch.bool and mask.shape == (5000,)
    assert radius.shape == (int(mask.sum()),) and not radius.requires_grad
    assert screen_coords.shape == (5000, 3) and screen_coords.requires_grad and screen_coords.is_leaf
    cot = torch.rand(160, 96, 3, generator=torch.Generator().manual_seed(3))
    (img * cot.to(img.device)).sum().backward()
    o = oracle_mod.render_scene(sc, 64, dimg=cot.numpy())
    assert_close(img.detach().cpu().numpy(), o["img"], "img")
    assert_close(model._coords.grad.cpu().numpy(), o["g_coords"], "g_coords", **grad_tolerance(len(o["g_coords"])))
    assert_close(model._scale.grad.cpu().numpy(), o["g_scale"], "g_scale", **grad_tolerance(len(o["g_scale"])))
    assert_close(model._rotation.grad.cpu().numpy(), o["g_rotation"], "g_rotation", **grad_tolerance(len(o["g_rotation"])))
    assert_close(model._opacity.grad.cpu().numpy(), o["g_opacity"], "g_opacity", **grad_tolerance(len(o["g_opacity"])))
    assert_close(model._sh.grad.cpu().numpy(), o["g_sh"], "g_sh", **grad_tolerance(len(o["g_sh"])))
    assert_close(screen_coords.grad.cpu().numpy(), o["g_screen"], "g_screen", **grad_tolerance(len(o["g_screen"])))
    # what GaussianModel.update_densify_stats reads (GaussianModel.py:231)
    assert torch.isfinite(torch.norm(screen_coords.grad[mask, :2], dim=-1)).all()


def test_capacity_larger_than_size(oracle_mod):
    """The model stores capacity rows; only the first _size are live (GaussianModel.py:42-65)."""
    import gsplat_amd  # noqa: F401
    from gsplat_amd.scenes import synthetic_scene
    sc = synthetic_scene(1000, 64, 64, 1, 22)
    model, renderer, cam = _setup(sc)
    model._size = 600
    img, mask, screen_coords, radius = renderer.render(cam, 16)
    assert mask.shape == (600,)
    img.sum().backward()
    assert not model._coords.grad[600:].any() and not model._sh.grad[600:].any()
    sub = synthetic_scene(1000, 64, 64, 1, 22)
    o = oracle_mod.render(sub.coords.numpy()[:600], sub.scale.numpy()[:600], sub.rotation.numpy()[:600],
                          sub.opacity.numpy()[:600], sub.sh.numpy()[:600], 1, 64, 64, 16,
                          cam.w2m_host, cam.proj_host, np.zeros(3, np.float32), dimg=np.ones((64, 64, 3), np.float32))
    assert_close(img.detach().cpu().numpy(), o["img"], "img")
    assert_close(model._coords.grad[:600].cpu().numpy(), o["g_coords"], "g_coords", **grad_tolerance(len(o["g_coords"])))


def test_nothing_visible_returns_four_nones():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.scenes import all_culled_scene
    sc = all_culled_scene(300, 64, 48, 0, 5)
    _, renderer, cam = _setup(sc)
    assert renderer(cam) == (None, None, None, None)


def test_no_grad_evaluation():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.scenes import synthetic_scene
    sc = synthetic_scene(2000, 64, 48, 3, 6)
    _, renderer, cam = _setup(sc)
    with torch.no_grad():
        img, mask, sc_, radius = renderer(cam, 16)
    assert img is not None and not img.requires_grad
