"""Drop-in ``GaussianRenderer`` (same constructor, attributes and call surface as
``GaussianRenderer.py:13-113`` of the reference) on top of the HIP rasterizer.

``Trainer`` uses: ``GaussianRenderer(model, debug)`` (``Trainer.py:32``), ``.device``
(``Trainer.py:33``), ``renderer(cam)`` returning ``(image, visible_mask, screen_coords, radius)``
(``Trainer.py:55``), the image's ``requires_grad`` (``Trainer.py:52``) and, after ``backward``,
``screen_coords.grad`` (``GaussianModel.py:231``).  All of that is preserved.  The model is read
through its raw parameter tensors ``_coords, _scale, _rotation, _opacity, _sh`` sliced to
``_size`` and ``sh_degree`` (``GaussianModel.py:42-69``); the activations are fused into the
first kernel.
"""
import torch
import torch.nn as nn

from .rasterizer import ViewSpec, rasterize


class GaussianRenderer(nn.Module):

    def __init__(self, model, debug: bool = False):
        super().__init__()
        self.model = model
        self.device = self.model.coords.device
        self.debug = debug
        self.prof = None   # the reference builds a torch profiler here when debug is set and never starts it

    @staticmethod
    def get_radius(cov2d: torch.Tensor) -> torch.Tensor:
        """3 * sqrt(larger eigenvalue) of [M,2,2] covariances (``GaussianRenderer.py:37-44``)."""
        trace = cov2d[:, 0, 0] + cov2d[:, 1, 1]
        det = cov2d[:, 0, 0] * cov2d[:, 1, 1] - cov2d[:, 0, 1] * cov2d[:, 1, 0]
        half = 0.5 * trace
        root = 0.5 * torch.sqrt(torch.clamp(trace * trace - 4 * det, min=0))
        return 3 * torch.sqrt(torch.maximum(half - root, half + root))

    def get_empty_image(self, cam):
        return torch.zeros((cam.width, cam.height, 3), dtype=torch.float, device=self.device)

    def forward(self, cam, num_tiles: int = 64):
        # the second argument is the tile edge in pixels, default 64 (GaussianRenderer.py:49-50)
        return self.render(cam, num_tiles)

    def _raw_params(self):
        m = self.model
        n = m._size
        return m._coords[:n], m._scale[:n], m._rotation[:n], m._opacity[:n], m._sh[:n]

    def render(self, cam, tile_length: int = 64):
        coords, scale, rotation, opacity, sh = self._raw_params()
        if not coords.is_cuda:
            raise RuntimeError("GaussianRenderer needs the model on the GPU: the render path has no CPU implementation")
        view = ViewSpec.from_camera(cam, self.model.sh_degree, tile_length)
        screen_coords = torch.zeros_like(coords, dtype=torch.float32, requires_grad=True)
        img, mask, radius = rasterize(coords, scale, rotation, opacity, sh, screen_coords, view)
        if img is None:
            return None, None, None, None
        return img, mask, screen_coords, radius
