"""GPU: the full training iteration at the size of the playroom configuration - 2.5 M Gaussians, SH degree 3,
1920x1080, two views per step - through ``MinibatchTrainer.train_step`` on a densification step
(Trainer.py:51-80, GaussianModel.py:230-245), checked piece by piece:

* one view's gradients (cotangent = the training loss's dL/dimg) against the render oracle;
* the densification statistics the step leaves against ``train_oracle.densify_stats`` over both views;
* the densification itself (selection, clone, split with the step's own random draws, capacity doubling,
  row placement) against ``train_oracle.densify``;
* an ordinary iteration before it (fused Adam over the bucket: every live row moves, by at most the learning rate) and
  the end of the densification step (the capacity doubled: no update, optimiser state resized).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from test_gpu_train_state import KEYS, assert_model, model_dict  # noqa: E402
from util import GRAD_OUTLIERS, assert_close  # noqa: E402

pytestmark = pytest.mark.gpu

N = 2_500_000
W, H, DEG, TILE = 1920, 1080, 3, 16
# The reference compares the RAW (log) scale of a Gaussian with 1 % of the scene radius (GaussianModel.py:240,244).
# The log-scales of the rows this step selects lie between 1 and 3: a radius of 200 puts the threshold (2) among
# them, so both branches, clone and split, run.
SCENE_RADIUS = 200.0


class _Loader:
    """Stand-in for the reference's DataLoader (Utils/DataLoader.py:72-106): sample() and scene_radius."""

    def __init__(self, cams, targets):
        self.cams, self.targets, self.k = cams, targets, -1
        self.scene_radius = np.array([SCENE_RADIUS, SCENE_RADIUS], np.float32)

    def sample(self, is_train=True):
        self.k = (self.k + 1) % len(self.cams)
        return self.cams[self.k], self.targets[self.k]


def test_config4_training_iteration_with_densification_at_size(oracle_mod):
    import gsplat_amd  # noqa: F401
    import train_oracle as to
    from gsplat_amd.camera import Camera
    from gsplat_amd.loss import loss_and_grad
    from gsplat_amd.minibatch import render_views
    from gsplat_amd.model import GaussianModel
    from gsplat_amd.rasterizer import ViewSpec, backward_pipeline, forward_pipeline
    from gsplat_amd.scenes import synthetic_scene
    from gsplat_amd.trainer import MinibatchTrainer
    dev = torch.device("cuda", 0)
    sc = synthetic_scene(N, W, H, DEG, seed=4)
    host = [sc.coords, sc.scale, sc.rotation, sc.opacity, sc.sh]
    p = [t.to(dev) for t in host]
    cams = []
    for v in range(2):
        pos = np.array([0.15 * np.cos(np.pi * v), 0.15 * np.sin(np.pi * v), 0.0])
        cams.append(Camera(pos, np.array([1.0, 0.0, 0.0, 0.0]), sc.fx, sc.fy, W / 2, H / 2, W, H, device=dev))
    gen = torch.Generator().manual_seed(11)
    targets = [torch.floor(torch.rand(W, H, 3, generator=gen) * 256).to(dev) for _ in cams]
    model = GaussianModel(*[t.clone() for t in p], sh_degree=DEG)
    assert model.capacity == N and int(model._size) == N

    # ---- one view, forward + backward, against the render oracle (cotangent from the loss kernel)
    specs = [ViewSpec.from_camera(c, DEG, TILE) for c in cams]
    st = forward_pipeline(*p, None, specs[0])
    assert st.num_visible > 0.9 * N and st.num_pairs > 10_000_000
    dpred = loss_and_grad(st.img, targets[0], 0.2, 0.5)[1]
    g = backward_pipeline(st, *p, dpred)
    torch.cuda.synchronize()
    ora = oracle_mod.render(*[t.numpy() for t in host], DEG, W, H, TILE, cams[0].w2m_host, cams[0].proj_host,
                            np.asarray(cams[0].transform.position, np.float32), dimg=dpred.cpu().numpy())
    assert np.array_equal(st.mask_u8.bool().cpu().numpy(), ora["mask"])
    assert np.array_equal(st.sorted_gauss.cpu().numpy()[:st.num_pairs], ora["tile_lists"])
    assert_close(st.img.cpu().numpy(), ora["img"], "img")
    for name, got in zip(("g_coords", "g_scale", "g_rotation", "g_opacity", "g_sh", "g_screen"), g):
        assert_close(got.cpu().numpy(), ora[name], name, **GRAD_OUTLIERS)
    del st, g, ora

    def cotangent(i, state):
        return loss_and_grad(state.img, targets[i], 0.2, 0.5)[1]

    # ---- an ordinary iteration first (Trainer.py:53-61,80): render of both views, statistics, fused Adam on the
    # gradient bucket.  Every live row moves, by at most the learning rate per coordinate (Adam's first step)
    tr = MinibatchTrainer(model, _Loader(cams, targets), views_per_step=2, tile_length=TILE, lr=1e-3)
    start = model_dict(model)
    loss1 = tr.train_step(49)
    torch.cuda.synchronize()
    assert np.isfinite(float(loss1))
    md = model_dict(model)
    step = np.abs(md["coords"] - start["coords"])
    assert (step.max(axis=1) > 0).mean() > 0.5 and step.max() <= 1.01e-3   # (rows hidden behind thousands of others get a zero gradient)
    assert int(md["denom"].max()) == 2 and md["size"] == N

    # ---- what the next step's two views add to the statistics (that step renders with the updated parameters)
    now = [getattr(model, "_" + k).data[:N] for k in KEYS]
    states, _, g_screens = render_views(now, specs, cotangent)
    for state, gs in zip(states, g_screens):
        md = to.densify_stats(md, gs.cpu().numpy(), state.mask_u8.bool().cpu().numpy())
    want_stats = md
    del states, g_screens

    # ---- the training step, with the densification's inputs and outputs captured on the way
    seen = {}
    real_densify = model.densify

    def densify_spy(scene_radius, generator=None, **kw):
        seen["before"] = model_dict(model)
        seen["generator_state"] = generator.get_state().clone()
        seen["counts"] = real_densify(scene_radius, generator=generator, **kw)
        seen["after"] = model_dict(model)
        return seen["counts"]

    model.densify = densify_spy
    loss = tr.train_step(50)
    torch.cuda.synchronize()
    assert np.isfinite(float(loss)) and "before" in seen
    final = model_dict(model)

    # statistics: counts exact, gradient norms to rounding
    assert_model(seen["before"], lambda k: want_stats[k], "statistics before the densification")

    # densification against the oracle, with the step's own draws
    n_clone, n_split = seen["counts"]
    b = seen["before"]
    with np.errstate(divide="ignore", invalid="ignore"):
        sel = np.nan_to_num(b["coords_grads"][:N, 0] / b["denom"][:N, 0]) >= 0.0002
    assert n_clone > 100 and n_split > 100, (n_clone, n_split, np.quantile(b["scale"][:N].max(axis=1)[sel], [0.01, 0.1, 0.5, 0.9]).tolist())
    g2 = torch.Generator(device=dev)
    g2.set_state(seen["generator_state"])
    noise = torch.randn((2 * n_split, 3), dtype=torch.float32, device=dev, generator=g2)
    want, wc, ws = to.densify(seen["before"], tr.scene_radius, noise.cpu().numpy())
    assert (wc, ws) == (n_clone, n_split)
    assert seen["after"]["coords"].shape[0] == 2 * N            # the capacity doubled once
    assert_model(seen["after"], lambda k: want[k], "densification at 2.5 M rows")

    # The capacity grew in this step, so - as in the reference, whose optimiser loses the re-created parameters
    # (GaussianModel.py:122-136 against Trainer.py:24) - the step ends without an update: the state is resized
    size_now = N + n_clone + n_split
    assert final["size"] == size_now
    for k in KEYS:
        assert np.array_equal(final[k], seen["after"][k]), k
    assert tr.optimizer.exp_avg[0].shape[0] == 2 * N

