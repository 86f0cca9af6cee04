"""Runs the *reference* renderer (imported from /root/reference, CPU) on a Scene.

Only usable where /root/reference exists (the build container).  Nothing here
travels to the GPU box except this text; the fixtures it produces do.

* Two empty stub modules stand in for ``plyfile`` and ``simple_knn._C``
  (imported by ``GaussianModel.py:4,7`` but not used by the render path).
* ``torch.argsort`` is wrapped to ``stable=True`` while the reference runs
  (``GaussianRenderer.py:89`` uses the default, unstable sort whose tie order
  is implementation-defined; stable = ascending original index is the
  contract the fixtures pin).
* No reference file is modified or copied.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def reference_available() -> bool:
    return os.path.isdir(REF) and os.path.isfile(os.path.join(REF, "GaussianRenderer.py"))


def _import_reference():
    sys.dont_write_bytecode = True
    if "plyfile" not in sys.modules:
        m = types.ModuleType("plyfile")
        m.PlyData = type("PlyData", (), {})
        m.PlyElement = type("PlyElement", (), {})
        sys.modules["plyfile"] = m
    if "simple_knn" not in sys.modules:
        pkg = types.ModuleType("simple_knn")
        sub = types.ModuleType("simple_knn._C")
        sub.distCUDA2 = None
        pkg._C = sub
        sys.modules["simple_knn"] = pkg
        sys.modules["simple_knn._C"] = sub
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import GaussianModel as GM  # noqa: N811
    import GaussianRenderer as GR  # noqa: N811
    from Utils.Camera import Camera as RefCamera
    from Utils.Screen import Screen as RefScreen
    return GM, GR, RefCamera, RefScreen


class _StableArgsort:
    def __enter__(self):
        self._orig = torch.argsort

        def stable_argsort(x, *a, **k):
            k["stable"] = True
            return self._orig(x, *a, **k)

        torch.argsort = stable_argsort
        return self

    def __exit__(self, *exc):
        torch.argsort = self._orig


def build_reference(scene):
    """-> (model, camera, renderer) reference objects holding the scene's tensors."""
    GM, GR, RefCamera, _ = _import_reference()
    model = GM.GaussianModel()
    model._size = scene.n
    model._device = "cpu"
    model._coords = torch.nn.Parameter(scene.coords.clone())
    model._scale = torch.nn.Parameter(scene.scale.clone())
    model._rotation = torch.nn.Parameter(scene.rotation.clone())
    model._opacity = torch.nn.Parameter(scene.opacity.clone())
    model._sh = torch.nn.Parameter(scene.sh.clone())
    model.curr_sh_degree = scene.sh_degree
    cam = RefCamera(scene.cam_position, scene.cam_rotation, scene.fx, scene.fy,
                    scene.width / 2, scene.height / 2, scene.width, scene.height)
    renderer = GR.GaussianRenderer(model)
    return model, cam, renderer


def run_reference(scene, tile_length, cotangent=None, want_intermediates=True):
    """Forward (+ backward when ``cotangent`` is given) through the reference.

    Returns a dict of numpy arrays.  ``img`` is None when nothing is visible.
    Loss for the backward is ``sum(img * cotangent)``.
    """
    _, _, _, RefScreen = _import_reference()
    model, cam, renderer = build_reference(scene)
    out = {}
    with _StableArgsort():
        img, mask, screen_coords, radius = renderer.render(cam, tile_length)
        if img is None:
            out["img"] = None
            return out
        out["img"] = img.detach().numpy().copy()
        out["mask"] = mask.numpy().copy()
        out["radius"] = radius.detach().numpy().copy()
        if cotangent is not None:
            loss = (img * torch.as_tensor(cotangent)).sum()
            loss.backward()
            out["g_coords"] = model._coords.grad.numpy().copy()
            out["g_scale"] = model._scale.grad.numpy().copy()
            out["g_rotation"] = model._rotation.grad.numpy().copy()
            out["g_opacity"] = model._opacity.grad.numpy().copy()
            out["g_sh"] = model._sh.grad.numpy().copy()
            out["g_screen"] = screen_coords.grad.numpy().copy()
        if want_intermediates:
            with torch.no_grad():
                from Utils.SphericalHarmonic import eval_sh
                sc0 = torch.zeros_like(model.coords)
                cov3d = model.covariance
                pos2d, cov2d, mask2 = cam.project_gaussian(sc0, model.coords, cov3d)
                assert torch.equal(mask2, mask)
                color = eval_sh(model.sh_degree, model.sh[mask],
                                model.coords[mask] - torch.tensor(cam.transform.position, dtype=torch.float32))
                color = torch.clamp(color + 0.5, min=0.0)
                opac = model.opacity[mask]
                scr = RefScreen(cam.width, cam.height, tile_length, device="cpu")
                scr.create_tiles(pos2d, radius)
                out["cov3d"] = cov3d.numpy().copy()
                out["pos2d"] = pos2d.numpy().copy()
                out["cov2d"] = cov2d.numpy().copy()
                out["color"] = color.numpy().copy()
                out["opacity_act"] = opac.numpy().copy()
                out["bbox"] = torch.stack([scr.xmin, scr.ymin, scr.xmax, scr.ymax], dim=1).numpy().copy()
                vis_index = torch.nonzero(mask).squeeze(1)
                lists, offsets = [], [0]
                for tid in range(scr.num_block):
                    left, top = scr.get_tl(tid)
                    right, bottom = scr.get_br(tid)
                    inmask = scr.get_inmask(left, right, top, bottom)
                    order = torch.argsort(pos2d[inmask][:, 2])
                    lists.append(vis_index[inmask][order])
                    offsets.append(offsets[-1] + int(order.numel()))
                out["tile_lists"] = (torch.cat(lists) if lists else torch.zeros(0, dtype=torch.int64)).numpy().astype(np.int32)
                out["tile_offsets"] = np.asarray(offsets, dtype=np.int64)
    return out
