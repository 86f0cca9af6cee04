"""Minimal stand-in for the reference ``GaussianModel`` parameter container.

Holds exactly the attributes the renderer reads (``GaussianModel.py:19-26,42-69``): raw
``nn.Parameter``s of capacity rows, ``_size`` live rows and ``curr_sh_degree``.  Used by tests and
``bench.py`` on machines where the reference checkout is absent; the reference's own
``GaussianModel`` works with the renderer unchanged.
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
