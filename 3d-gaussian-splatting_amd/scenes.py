"""Synthetic scenes: random Gaussian clouds + a camera, seeded and reproducible.

The generator is the workload definition used by the parity tests, the golden
fixtures and ``bench.py``.  It produces the *raw* (pre-activation) parameter
tensors that ``GaussianModel`` stores (``GaussianModel.py:87-91``):
``_coords [N,3]``, ``_scale [N,3]`` (log), ``_rotation [N,4]`` (un-normalised
quaternion), ``_opacity [N,1]`` (logit), ``_sh [N,3,16]``.
"""
from dataclasses import dataclass

import numpy as np
import torch

from .camera import Camera


@dataclass
class Scene:
    coords: torch.Tensor      # [N,3]
    scale: torch.Tensor       # [N,3] log-scale
    rotation: torch.Tensor    # [N,4] raw quaternion (w,x,y,z)
    opacity: torch.Tensor     # [N,1] logit
    sh: torch.Tensor          # [N,3,16]
    sh_degree: int
    width: int
    height: int
    cam_position: np.ndarray  # [3]
    cam_rotation: np.ndarray  # [4] (w,x,y,z)
    fx: float
    fy: float

    @property
    def n(self) -> int:
        return self.coords.shape[0]

    def camera(self, device=None) -> Camera:
        return Camera(self.cam_position, self.cam_rotation, self.fx, self.fy,
                      self.width / 2, self.height / 2, self.width, self.height, device=device)

    def params(self):
        return self.coords, self.scale, self.rotation, self.opacity, self.sh

    def to(self, device):
        return Scene(self.coords.to(device), self.scale.to(device), self.rotation.to(device),
                     self.opacity.to(device), self.sh.to(device), self.sh_degree, self.width,
                     self.height, self.cam_position, self.cam_rotation, self.fx, self.fy)


def synthetic_scene(n: int, width: int, height: int, sh_degree: int = 3, seed: int = 0,
                    max_radius_px: float = 12.0) -> Scene:
    """Frustum-filling cloud in front of a camera at the origin looking down -z.

    Draw order (one ``torch.Generator`` seeded with ``seed``): depth, x, y,
    target 3-sigma radius, per-axis scale jitter, opacity, SH, rotation.
    Depth is uniform in [2, 8]; x/y fill the view frustum at that depth; the
    projected 3-sigma radius is log-uniform in [1, ``max_radius_px``] pixels.
    The 2-D covariance of the path under test is expressed in (world units /
    world z), so a world scale ``s`` at depth ``d`` spans ``s/d`` pixels.
    """
    g = torch.Generator().manual_seed(seed)

    def u(*shape):
        return torch.rand(*shape, generator=g, dtype=torch.float32)

    fx = fy = 0.9 * width
    z = -(2.0 + 6.0 * u(n))
    x = (2.0 * u(n) - 1.0) * (-z) * width / (2.0 * fx)
    y = (2.0 * u(n) - 1.0) * (-z) * height / (2.0 * fy)
    r = torch.pow(torch.tensor(max_radius_px, dtype=torch.float32), u(n))
    s = (r / 3.0) * (-z)
    scale = torch.log(s[:, None] * (0.5 + u(n, 3)))
    opacity = torch.randn(n, 1, generator=g, dtype=torch.float32)
    sh = 0.3 * torch.randn(n, 3, 16, generator=g, dtype=torch.float32)
    rotation = torch.randn(n, 4, generator=g, dtype=torch.float32)
    coords = torch.stack([x, y, z], dim=1).contiguous()
    return Scene(coords, scale.contiguous(), rotation, opacity, sh, sh_degree, width, height,
                 np.zeros(3), np.array([1.0, 0.0, 0.0, 0.0]), fx, fy)


def adversarial_scene(n: int, width: int, height: int, sh_degree: int = 2, seed: int = 0) -> Scene:
    """``synthetic_scene`` plus the cases that decide visibility and tile membership:

    * a block of Gaussians mirrored behind the camera (they pass the cull test),
    * a block pushed outside the +-1.3 NDC guard band (culled),
    * a block with sub-pixel extent (integer bounding box is empty: in no tile),
    * a block with duplicated depths (ties in the sort key),
    * a few very large Gaussians that cover many tiles.
    """
    sc = synthetic_scene(n, width, height, sh_degree, seed)
    k = max(n // 8, 1)
    c = sc.coords.clone()
    s = sc.scale.clone()
    c[0:k, 2] = -c[0:k, 2]                      # behind the camera
    c[k:2 * k, 0] = c[k:2 * k, 0] * 4.0 + 0.5   # off-screen in x (most of them)
    s[2 * k:3 * k] = s[2 * k:3 * k] - 6.0       # sub-pixel
    c[3 * k:4 * k, 2] = c[3 * k, 2]             # identical world z ...
    c[3 * k:4 * k, :2] *= 0.25                  # ... near the axis: many equal NDC depths
    s[4 * k:4 * k + 3] = s[4 * k:4 * k + 3] + 2.5   # huge
    return Scene(c, s, sc.rotation, sc.opacity, sc.sh, sh_degree, width, height,
                 sc.cam_position, sc.cam_rotation, sc.fx, sc.fy)


def all_culled_scene(n: int, width: int, height: int, sh_degree: int = 0, seed: int = 0) -> Scene:
    """Every Gaussian is far outside the guard band: the render returns four ``None``s."""
    sc = synthetic_scene(n, width, height, sh_degree, seed)
    c = sc.coords.clone()
    c[:, 0] = 50.0 + c[:, 0].abs()
    return Scene(c, sc.scale, sc.rotation, sc.opacity, sc.sh, sh_degree, width, height,
                 sc.cam_position, sc.cam_rotation, sc.fx, sc.fy)


def orbit_scene(n: int, width: int, height: int, sh_degree: int, view: int, n_views: int = 8,
                seed: int = 0) -> Scene:
    """One Gaussian cloud seen from ``n_views`` cameras on a ring (view minibatch).

    The cloud is the ``synthetic_scene`` of ``seed`` re-centred on the origin;
    camera ``view`` sits on a circle of radius 5 in the x-z plane and looks at
    the centre.
    """
    sc = synthetic_scene(n, width, height, sh_degree, seed)
    c = sc.coords.clone()
    c[:, 2] += 5.0
    ang = 2.0 * np.pi * view / n_views
    pos = np.array([5.0 * np.sin(ang), 0.0, 5.0 * np.cos(ang)])
    quat = np.array([np.cos(ang / 2.0), 0.0, np.sin(ang / 2.0), 0.0])  # rotation about +y
    return Scene(c, sc.scale, sc.rotation, sc.opacity, sc.sh, sh_degree, width, height,
                 pos, quat, sc.fx, sc.fy)
