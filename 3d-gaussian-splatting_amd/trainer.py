"""``Trainer``: the reference's training loop (``Trainer.py:13-89``) on the device path - HIP renderer,
loss kernel, fused Adam, densification kernels - with the reference's constructor arguments, schedule
and method names.

What differs from the reference, and why:

* The data loader is any object with ``sample(is_train) -> (camera, target[W,H,3])`` and
  ``scene_radius`` (the reference's ``DataLoader`` needs ``pycolmap`` and ``cv2``); ``evaluate`` returns
  the validation loss instead of writing JPEGs with ``cv2``.
* ``Trainer.py:68`` builds the remove mask as ``(opacity[N,1] < 0.005) & (radius[M] > threshold)``, which
  broadcasts to ``[N,M]`` and cannot index the model unless it is empty; here the radius of the visible
  Gaussians is scattered to ``[N]`` first, which gives the 1-D mask ``GaussianModel.remove`` handles.
* After ``densify`` grew the capacity the optimiser adopts the new parameter tensors
  (``Adam.resize``); the reference's keeps the old ones and stops training them.
"""
import os

import torch

from .loss import compute_loss as _compute_loss
from .optimizer import Adam
from .renderer import GaussianRenderer


class Trainer:
    def __init__(self, model, dataloader, lr: float = 1e-3, betas=(0.5, 0.999), train_steps: int = 2000,
                 refine_cycle: int = 500, loss_lambda: float = 0.2, result_dir: str = "result/", debug: bool = False,
                 checkpoint_dir: str = None, tile_length: int = 64):
        self.model = model
        self.dataloader = dataloader
        self.optimizer = Adam(self.model.parameters(), lr=lr, betas=betas)
        self.train_steps = train_steps
        self.refine_cycle = refine_cycle
        self.loss_lambda = loss_lambda
        self.result_dir = result_dir
        self.debug = debug
        self.checkpoint_dir = checkpoint_dir      # None: no checkpoint files (the reference writes checkpoints/)
        self.tile_length = tile_length            # the reference renders with the renderer's default, 64
        self.loss_value = []
        self.renderer = GaussianRenderer(self.model, self.debug)
        self.scene_radius = torch.as_tensor(self.dataloader.scene_radius, dtype=torch.float32).max()

    def compute_loss(self, pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        """``Trainer.py:35-38``."""
        return _compute_loss(pred, gt, self.loss_lambda)

    def get_image_scale(self, step: int):
        """``Trainer.py:40-45`` (defined by the reference, used nowhere)."""
        if step < 250:
            return 0.25
        elif step < 500:
            return 0.5
        return 1

    def train_step(self, train_step: int):
        """One iteration of ``Trainer.train`` (``Trainer.py:51-84``); returns the loss tensor."""
        rendered_image = None
        while rendered_image is None or not rendered_image.requires_grad:
            self.optimizer.zero_grad(set_to_none=True)
            cam, target_image = self.dataloader.sample(is_train=True)
            rendered_image, visible_mask, screen_coords, radius = self.renderer(cam, self.tile_length)
        loss = self.compute_loss(rendered_image, target_image)
        loss.backward()

        with torch.no_grad():
            self.model.update_densify_stats(screen_coords, visible_mask)
            if train_step > 0 and train_step % 500 == 0:
                self.model.add_sh_degree()
            grew = False
            if train_step > 0 and train_step % 50 == 0:
                capacity = self.model.capacity
                size = self.model._size
                self.model.densify(self.scene_radius)
                size_threshold = 20 if train_step > 1500 else 1000
                if self.model._size == size:
                    # (after a clone or split the mask of this render no longer has the model's size: the
                    # reference's remove then fails on the shapes unless nothing is to be removed)
                    radius_all = torch.zeros(size, dtype=torch.float32, device=radius.device)
                    radius_all[visible_mask] = radius
                    remove_mask = (self.model.opacity[:, 0] < 0.005) & (radius_all > size_threshold)
                    self.model.remove(remove_mask)
                grew = self.model.capacity != capacity
            if train_step > 0 and train_step % 1500 == 0:
                self.model.reset_opacity()
            if train_step > 0 and train_step % 500 == 0:
                self.evaluate(train_step)
                if self.checkpoint_dir is not None:
                    os.makedirs(self.checkpoint_dir, exist_ok=True)
                    torch.save((self.model.capture(self.optimizer), train_step),
                               os.path.join(self.checkpoint_dir, f"checkpoint{train_step}.pth"))
        if grew:
            # new parameter tensors: no gradient this iteration, optimiser state padded for the new rows
            self.optimizer.resize(self.model.parameters())
        else:
            self.optimizer.step()
        return loss.detach()

    def train(self):
        self.loss_value = []
        for train_step in range(1, self.train_steps + 1):
            loss = self.train_step(train_step)
            if train_step % 10 == 0:
                self.loss_value.append(loss.item())
        return self.loss_value

    @torch.no_grad()
    def evaluate(self, train_step: int = 0):
        """``Trainer.py:91-104`` without the image files: the validation loss, or ``None`` when nothing is
        visible."""
        cam, target_image = self.dataloader.sample(is_train=False)
        image, _, _, _ = self.renderer(cam, self.tile_length)
        if image is None:
            return None
        return float(self.compute_loss(image, target_image))
