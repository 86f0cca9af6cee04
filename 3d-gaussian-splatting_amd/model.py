"""The model containers.

``GaussianParams`` holds exactly the attributes the renderer reads (``GaussianModel.py:19-26,42-69``):
raw ``nn.Parameter``s of capacity rows, ``_size`` live rows and ``curr_sh_degree``.  ``GaussianModel``
adds the reference class's training-state surface on the GPU (statistics, densification, removal,
PLY files, ``from_pcd``).  The reference's own ``GaussianModel`` works with the renderer and with
``densify.py`` unchanged.
"""
import torch
import torch.nn as nn


class GaussianParams(nn.Module):
    def __init__(self, coords, scale, rotation, opacity, sh, sh_degree: int, max_sh_degree: int = 3):
        super().__init__()
        self.max_sh_degree = max_sh_degree
        self.curr_sh_degree = int(sh_degree)
        self._coords = nn.Parameter(coords.clone())
        self._scale = nn.Parameter(scale.clone())
        self._rotation = nn.Parameter(rotation.clone())
        self._opacity = nn.Parameter(opacity.clone())
        self._sh = nn.Parameter(sh.clone())
        self._size = coords.shape[0]

    @staticmethod
    def from_scene(scene, device):
        return GaussianParams(scene.coords.to(device), scene.scale.to(device), scene.rotation.to(device),
                              scene.opacity.to(device), scene.sh.to(device), scene.sh_degree)

    @property
    def size(self):
        return self._size

    @property
    def coords(self):
        return self._coords[:self._size]

    @property
    def sh_degree(self):
        return self.curr_sh_degree

    def raw(self):
        n = self._size
        return self._coords[:n], self._scale[:n], self._rotation[:n], self._opacity[:n], self._sh[:n]


class GaussianModel(GaussianParams):
    """The reference ``GaussianModel``'s training-state surface (``GaussianModel.py:14-245``) on the
    GPU: capacity rows of which ``_size`` are live, the densification statistics and the methods the
    trainer calls around a render (``Trainer.py:60-74``).  Each method does what the reference's
    method of the same name does, through the kernels behind ``densify.py`` (PLY I/O: ``ply.py``).
    ``GaussianModel()`` / ``GaussianModel(max_sh_degree)`` is the reference's constructor: an empty model, to
    be filled by ``load_ply`` (``main.py:31-32``)."""

    def __init__(self, coords=None, scale=None, rotation=None, opacity=None, sh=None, sh_degree: int = 0,
                 max_sh_degree: int = 3, size=None):
        if coords is None or isinstance(coords, int):
            if isinstance(coords, int):
                max_sh_degree = coords
            dev = "cuda" if torch.cuda.is_available() else "cpu"
            f32 = dict(dtype=torch.float32, device=dev)
            coords, scale, rotation = torch.empty((0, 3), **f32), torch.empty((0, 3), **f32), torch.empty((0, 4), **f32)
            opacity, sh = torch.empty((0, 1), **f32), torch.empty((0, 3, (max_sh_degree + 1) ** 2), **f32)
        super().__init__(coords, scale, rotation, opacity, sh, sh_degree, max_sh_degree)
        cap = coords.shape[0]
        self._size = cap if size is None else int(size)
        self._device = coords.device
        self._coords_grads = torch.zeros((cap, 1), dtype=torch.float32, device=coords.device)
        self._denom = torch.zeros((cap, 1), dtype=torch.int32, device=coords.device)
        # the activation attributes of GaussianModel.py:29-32, under the reference's names (its spelling)
        self.opactiy_activation = torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        self.scale_activation = torch.exp
        self.scale_inv_activation = torch.log

    @staticmethod
    def from_pcd(pcd, device="cuda", max_sh_degree: int = 3) -> "GaussianModel":
        """``GaussianModel.from_pcd`` (``GaussianModel.py:71-97``): one Gaussian per point of a point cloud
        (an object with ``coords`` ``[N,3]`` and ``colors`` ``[N,>=3]`` in 0..255, like the reference's
        ``PointCloud``): raw opacity 1, the colour in the first sh coefficient, an isotropic scale of the
        root of the mean squared distance to the three nearest points (``gsr_knn_dist2`` in place of
        ``simple_knn``'s ``distCUDA2``), the identity rotation, SH degree 0."""
        import numpy as np
        from . import _lib
        from .rasterizer import _call, _stream
        dev = torch.device(device)
        coords = torch.as_tensor(np.asarray(pcd.coords), dtype=torch.float32).to(dev).contiguous()
        n = coords.shape[0]
        colors = torch.as_tensor(np.asarray(pcd.colors)[:, :3].astype(np.float32)).to(dev)
        sh = torch.zeros((n, 3, (max_sh_degree + 1) ** 2), dtype=torch.float32, device=dev)
        sh[:, :, 0] = (colors / 255 - 0.5) / 0.28209479177387814          # RGB2SH, Utils/SphericalHarmonic.py:6-7,14
        dist2 = torch.empty(n, dtype=torch.float32, device=dev)
        lib = _lib.load()
        _call("gsr_knn_dist2", lib.gsr_knn_dist2, n, coords.data_ptr(), dist2.data_ptr(), _stream(dev))
        scale = torch.log(torch.sqrt(torch.clamp_min(dist2, 0.0000001)))[..., None].repeat(1, 3)
        rotation = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        rotation[:, 0] = 1
        opacity = torch.ones((n, 1), dtype=torch.float32, device=dev)
        return GaussianModel(coords, scale, rotation, opacity, sh, sh_degree=0, max_sh_degree=max_sh_degree)

    @property
    def capacity(self):
        return self._coords.shape[0]

    # activated views, GaussianModel.py:50-65 (the renderer itself reads the raw tensors)
    @property
    def scale(self):
        return torch.exp(self._scale[:self._size])

    @property
    def rotation(self):
        return torch.nn.functional.normalize(self._rotation[:self._size])

    @property
    def opacity(self):
        return torch.sigmoid(self._opacity[:self._size])

    @property
    def sh(self):
        return self._sh[:self._size, :, :(self.curr_sh_degree + 1) ** 2]

    @property
    def covariance(self):
        return self.get_covariance()

    def get_covariance(self, mask: torch.Tensor = None):
        """``GaussianModel.get_covariance`` (``GaussianModel.py:99-108``): ``(R S)(R S)^T`` per Gaussian, in
        plain tensor operations (the renderer computes it inside its first kernel; this is for callers
        such as ``Trainer.test``)."""
        q, s = self.rotation, self.scale
        if mask is not None:
            q, s = q[mask], s[mask]
        w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack((2 * (w * w + x * x) - 1, 2 * (x * y - w * z), 2 * (x * z + w * y),
                         2 * (x * y + w * z), 2 * (w * w + y * y) - 1, 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), 2 * (w * w + z * z) - 1), dim=1).reshape(-1, 3, 3)
        RS = R * s[:, None, :]
        return torch.bmm(RS, RS.transpose(1, 2))

    def construct_list_of_attributes(self):
        """``GaussianModel.py:256-266``."""
        from . import ply
        return ply.attribute_names(self._sh.shape[1] * self._sh.shape[2], self._scale.shape[1], self._rotation.shape[1])

    def debug_memory(self):
        """``GaussianModel.py:247-254``."""
        for name in ("coords", "opacity", "scale", "rotation", "sh", "coords_grads", "denom"):
            t = getattr(self, "_" + name)
            print(f"{name} takes memory {t.element_size() * t.nelement()}")

    def reset_opacity(self):
        from . import densify as D
        D.reset_opacity(self)

    def add_sh_degree(self):
        from . import densify as D
        D.add_sh_degree(self)

    def expand_capacity(self, new_capacity: int):
        from . import densify as D
        D.expand_capacity(self, new_capacity)

    def update_densify_stats(self, pos2d, mask):
        from . import densify as D
        D.update_densify_stats(self, pos2d, mask)

    def densify(self, scene_radius, noise=None, generator=None):
        from . import densify as D
        return D.densify(self, scene_radius, noise=noise, generator=generator)

    def clone(self, mask, expand_factor: int = 2):
        from . import densify as D
        return D.clone(self, mask, expand_factor)

    def split(self, mask, expand_factor: int = 2, split_factor: int = 2, noise=None, generator=None):
        from . import densify as D
        return D.split(self, mask, expand_factor, split_factor, noise=noise, generator=generator)

    def remove(self, mask, inplace: bool = True):
        from . import densify as D
        return D.remove(self, mask, inplace)

    def save_ply(self, path: str):
        """``GaussianModel.save_ply`` (``GaussianModel.py:268-281``)."""
        from . import ply
        ply.save_ply(self, path)

    def load_ply(self, path: str):
        """``GaussianModel.load_ply`` (``GaussianModel.py:283-318``)."""
        from . import ply
        ply.load_ply(self, path, device=self._coords.device)

    def capture(self, optimizer):
        """``GaussianModel.capture`` (``GaussianModel.py:320-331``)."""
        return (self.curr_sh_degree, self._coords, self._sh, self._scale, self._rotation, self._opacity,
                self._coords_grads, self._denom, optimizer.state_dict())
