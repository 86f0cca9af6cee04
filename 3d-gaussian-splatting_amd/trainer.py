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


class MinibatchTrainer(Trainer):
    """``Trainer`` with an optimisation step over ``views_per_step`` views that are split over the ranks
    of a ``torch.distributed`` process group (one process per GPU, RCCL): the loss of a step is the mean
    of the views' training losses.  The reference trains on one view per step on one device
    (``Trainer.py:54-57``); with ``views_per_step=1`` and one rank this is that loop.

    Every rank holds the whole model and must draw the SAME views: ``dataloader.sample`` is called
    ``views_per_step`` times per step on every rank (give the ranks equally seeded loaders) and a rank
    renders the block ``minibatch.shard_views`` assigns to it.  Per step the ranks exchange the
    parameter gradients (``minibatch.reduce_gradients``, or one all-reduce of the bucket when the
    views do not split evenly), the densification statistics of their views
    (``minibatch.reduce_densify_stats``) and, on the steps that remove Gaussians, the largest screen
    radius of every Gaussian over the step's views (the reference tests the radius in its one view,
    ``Trainer.py:68``).  The split's random draws come from a generator seeded with ``seed`` and the step
    number, so all ranks draw the same; parameters, optimiser state and sizes stay identical on all ranks
    without ever being broadcast.

    Call ``minibatch.reserve_streams()`` before ``torch.distributed.init_process_group``: the views of a rank are
    pipelined over two HIP streams, which have to be on hardware queues of their own (RCCL takes queues when the
    group is initialised).  Gradients are bit-reproducible run to run for tile edges up to 64."""

    def __init__(self, model, dataloader, views_per_step: int = 8, group=None, seed: int = 0, **kwargs):
        super().__init__(model, dataloader, **kwargs)
        if views_per_step < 1:
            raise ValueError("views_per_step must be at least 1")
        self.views_per_step = int(views_per_step)
        self.group = group
        self.seed = int(seed)
        self._bucket = None
        self._bucket_rows = 0

    def _ranks(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def _gradient_bucket(self, size: int):
        """A bucket of capacity rows (what Adam updates), of which the first ``size`` receive gradients."""
        from .minibatch import GradBucket
        cap = self.model.capacity
        if self._bucket is None or self._bucket.n != cap:
            self._bucket = GradBucket(cap, self.model._coords.device)
        elif size < self._bucket_rows:
            self._bucket.flat.zero_()          # rows that were live in the last step and are not any more
        self._bucket_rows = size
        return self._bucket

    def train_step(self, train_step: int):
        import torch.distributed as dist
        from . import densify as D
        from .loss import loss_and_grad
        from .minibatch import (can_factor, factored_moves_less, gather_colors, reduce_densify_stats, reduce_gradients, render_views,
                                shard_views)
        from .rasterizer import ViewSpec, sh_grad_views
        model, V = self.model, self.views_per_step
        world, rank = self._ranks()
        dev = model._coords.device
        while True:
            samples = [self.dataloader.sample(is_train=True) for _ in range(V)]
            size = int(model._size)
            specs = [ViewSpec.from_camera(cam, model.curr_sh_degree, self.tile_length) for cam, _ in samples]
            mine = shard_views(V, world, rank)
            params = [p.data[:size] for p in (model._coords, model._scale, model._rotation, model._opacity, model._sh)]
            bucket = self._gradient_bucket(size)
            grads = tuple(v[:size] for v in bucket.views)
            factored = world > 1 and can_factor(V, world) and factored_moves_less(V)
            g_colors = torch.empty((len(mine), size, 3), dtype=torch.float32, device=dev) if factored else None
            losses = []

            def cotangent(i, st):
                out, dpred = loss_and_grad(st.img, samples[mine[i]][1], self.loss_lambda, 1.0 / V)
                losses.append(out[0])
                return dpred

            # (the gather of the colour gradients starts as soon as they exist, beside the pass that chains the views
            # to the parameters)
            early = []
            states, _, g_screens = render_views(params, [specs[v] for v in mine], cotangent, grads=grads, g_colors=g_colors,
                                                colors_ready=lambda: early.append(gather_colors(g_colors, V, group=self.group)))
            live = [st for st in states if st.num_visible]
            # the reference draws again when nothing is visible (Trainer.py:52-55): here, when no view of
            # the step on any rank shows anything
            if world > 1:
                seen = torch.tensor([float(len(live))], device=dev)
                dist.all_reduce(seen, group=self.group)
                any_seen = float(seen) > 0
            else:
                any_seen = bool(live)       # (no device round trip: the host stays ahead of the GPU from step to step)
            if any_seen:
                break
            if early and early[0] is not None:
                early[0][1].wait()          # (every rank draws again: the gather of this attempt is not used)
        loss = torch.stack(losses).sum() / V if losses else torch.zeros((), device=dev)
        if world > 1:
            if factored:
                reduce_gradients(bucket, g_colors, V, lambda g_all: sh_grad_views(specs, params[0], params[4], list(g_all), grads[4]),
                                 group=self.group, gathered=early[0] if early else None)
            else:
                bucket.all_reduce(self.group)
            dist.all_reduce(loss, group=self.group)

        with torch.no_grad():
            if world > 1:
                if live:
                    dg, dd = D.stats_delta(size, g_screens, [st.mask_u8 for st in live])
                else:
                    dg = torch.zeros((size, 1), dtype=torch.float32, device=dev)
                    dd = torch.zeros((size, 1), dtype=torch.int32, device=dev)
                reduce_densify_stats(model, dg, dd, group=self.group)
            else:
                D.accumulate_stats_views(model, g_screens, [st.mask_u8 for st in live])
            if train_step > 0 and train_step % 500 == 0:
                model.add_sh_degree()
            grew = False
            if train_step > 0 and train_step % 50 == 0:
                capacity = model.capacity
                gen = torch.Generator(device=dev)
                gen.manual_seed(self.seed * 1000003 + train_step)
                model.densify(self.scene_radius, generator=gen)
                size_threshold = 20 if train_step > 1500 else 1000
                if model._size == size:
                    radius_all = torch.zeros(size, dtype=torch.float32, device=dev)
                    for st in live:
                        torch.maximum(radius_all, st.radius_full * st.mask_u8, out=radius_all)
                    if world > 1:
                        dist.all_reduce(radius_all, op=dist.ReduceOp.MAX, group=self.group)
                    model.remove((model.opacity[:, 0] < 0.005) & (radius_all > size_threshold))
                grew = model.capacity != capacity
            if train_step > 0 and train_step % 1500 == 0:
                model.reset_opacity()
            if train_step > 0 and train_step % 500 == 0:
                self.evaluate(train_step)
                if self.checkpoint_dir is not None and rank == 0:
                    os.makedirs(self.checkpoint_dir, exist_ok=True)
                    torch.save((model.capture(self.optimizer), train_step),
                               os.path.join(self.checkpoint_dir, f"checkpoint{train_step}.pth"))
        if grew:
            self.optimizer.resize(model.parameters())
        else:
            self.optimizer.step(grads=bucket.views)
        return loss.detach()
