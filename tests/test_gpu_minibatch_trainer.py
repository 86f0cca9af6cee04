"""GPU: the data-parallel training step (trainer.MinibatchTrainer) and the densification statistics
of views that are spread over ranks (densify.stats_delta + minibatch.reduce_densify_stats), in one
process.  The run with two real ranks is tests/test_zy_two_rank_trainer.py (child processes)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


class _RingLoader:
    """Stand-in for the reference's DataLoader (Utils/DataLoader.py:72-106): sample() and scene_radius."""

    def __init__(self, targets, cams, radius):
        self.targets, self.cams, self.k = targets, cams, 0
        self.scene_radius = np.array([radius, 0.5 * radius], np.float32)

    def sample(self, is_train=True):
        self.k = (self.k + 1) % len(self.cams)
        return self.cams[self.k], self.targets[self.k]


def _setup(n=2000, n_cams=4):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.camera import Camera
    from gsplat_amd.model import GaussianModel
    from gsplat_amd.renderer import GaussianRenderer
    from gsplat_amd.scenes import synthetic_scene
    sc = synthetic_scene(n, 96, 64, 1, seed=8)
    p = [t.to(dev()) for t in sc.params()]
    cams = []
    for v in range(n_cams):
        ang = 2.0 * np.pi * v / n_cams
        cams.append(Camera(np.array([0.1 * np.cos(ang), 0.1 * np.sin(ang), 0.0]), np.array([1.0, 0.0, 0.0, 0.0]), sc.fx, sc.fy,
                           sc.width / 2, sc.height / 2, sc.width, sc.height, device=dev()))
    truth = GaussianRenderer(GaussianModel(*p, sh_degree=1))
    with torch.no_grad():
        targets = [truth(c, 16)[0].clone() for c in cams]
    gen = torch.Generator().manual_seed(2)
    start = [p[0] + 0.02 * torch.randn(p[0].shape, generator=gen).to(dev()), p[1], p[2], p[3] - 0.5,
             p[4] + 0.1 * torch.randn(p[4].shape, generator=gen).to(dev())]
    return start, cams, targets


def test_one_view_per_step_is_the_single_view_trainer():
    """views_per_step=1 on one rank walks the same path as Trainer.train_step (Trainer.py:51-84) through
    the minibatch machinery (gradient bucket, one chain pass, statistics from g_screen): after 40
    iterations parameters, optimiser state and statistics agree to rounding."""
    from gsplat_amd.model import GaussianModel
    from gsplat_amd.trainer import MinibatchTrainer, Trainer
    start, cams, targets = _setup()
    a = GaussianModel(*start, sh_degree=1)
    b = GaussianModel(*start, sh_degree=1)
    ta = Trainer(a, _RingLoader(targets, cams, 0.05), lr=2e-3, tile_length=16)
    tb = MinibatchTrainer(b, _RingLoader(targets, cams, 0.05), views_per_step=1, lr=2e-3, tile_length=16)
    la = [float(ta.train_step(k)) for k in range(1, 41)]
    lb = [float(tb.train_step(k)) for k in range(1, 41)]
    np.testing.assert_allclose(lb, la, rtol=1e-4)
    assert la[-1] < 0.7 * la[0]
    for pa, pb in zip(a.parameters(), b.parameters()):
        scale = float(pa.detach().abs().max())
        np.testing.assert_allclose(pb.detach().cpu().numpy(), pa.detach().cpu().numpy(), rtol=1e-3, atol=1e-4 * scale)
    assert torch.equal(a._denom, b._denom)
    np.testing.assert_allclose(b._coords_grads.cpu().numpy(), a._coords_grads.cpu().numpy(), rtol=1e-3,
                               atol=1e-5 * float(a._coords_grads.max()))
    assert ta.optimizer.steps == tb.optimizer.steps == [40] * 5


def test_minibatch_trainer_trains_and_densifies():
    """Four views per step for 110 iterations: the loss of the minibatch falls, two densification rounds
    grow the model and its capacity, the gradient bucket and the optimiser follow the capacity."""
    from gsplat_amd.model import GaussianModel
    from gsplat_amd.trainer import MinibatchTrainer
    start, cams, targets = _setup()
    model = GaussianModel(*start, sh_degree=1)
    tr = MinibatchTrainer(model, _RingLoader(targets, cams, 0.05), views_per_step=4, lr=2e-3, train_steps=110, tile_length=16)
    losses = tr.train()
    assert len(losses) == 11 and all(np.isfinite(losses))
    assert losses[3] < 0.6 * losses[0]
    assert model._size > 2000 and model.capacity >= model._size
    assert tr.optimizer.exp_avg[0].shape[0] == model.capacity == tr._bucket.n
    assert [id(q) for q in tr.optimizer.params] == [id(q) for q in model.parameters()]
    assert int(model._denom.max()) <= 4 * 10          # restarted at iteration 100, four views per step since

    # the same run again from the same start: the seeded split noise makes it reproducible
    model2 = GaussianModel(*start, sh_degree=1)
    tr2 = MinibatchTrainer(model2, _RingLoader(targets, cams, 0.05), views_per_step=4, lr=2e-3, train_steps=110, tile_length=16)
    tr2.train()
    assert model2._size == model._size
    for pa, pb in zip(model.parameters(), model2.parameters()):
        assert torch.equal(pa, pb)


def test_stats_delta_of_two_halves_is_the_statistics_of_all_views():
    """Two emulated ranks: the deltas of views {0,1} and {2,3}, summed as the all-reduce sums them and
    added by reduce_densify_stats, against accumulate_stats over the four views on one model."""
    import gsplat_amd  # noqa: F401
    from gsplat_amd import densify as D
    from gsplat_amd.minibatch import reduce_densify_stats
    from gsplat_amd.model import GaussianModel
    start, _, _ = _setup(n=5000)
    cap, size = 5000, 4321
    one = GaussianModel(*start, sh_degree=1, size=size)
    two = GaussianModel(*start, sh_degree=1, size=size)
    for m in (one, two):
        m._coords_grads.fill_(0.5)
        m._denom.fill_(2)
    gen = torch.Generator().manual_seed(77)
    gs = [(torch.randn(size, 3, generator=gen) * 1e-3).to(dev()) for _ in range(4)]
    masks = [(torch.rand(size, generator=gen) < 0.5).to(dev()) for _ in range(4)]
    for g, m in zip(gs, masks):
        D.accumulate_stats(one, g, m)
    d0 = D.stats_delta(size, gs[:2], masks[:2])
    d1 = D.stats_delta(size, gs[2:], masks[2:])
    reduce_densify_stats(two, d0[0] + d1[0], d0[1] + d1[1])
    assert torch.equal(one._denom, two._denom)
    np.testing.assert_allclose(two._coords_grads.cpu().numpy(), one._coords_grads.cpu().numpy(), rtol=1e-6)
    assert float(two._coords_grads[size:].min()) == 0.5 == float(two._coords_grads[size:].max())
    assert int(two._denom[size:].min()) == 2 == int(two._denom[size:].max()) and cap == two.capacity
    with pytest.raises(ValueError):
        D.stats_delta(size, gs[:2], masks[:1])
    with pytest.raises(ValueError):
        D.stats_delta(size + 1, gs[:1], masks[:1])


@pytest.mark.parametrize("nviews", [1, 3, 8, 11])
def test_statistics_of_a_step_in_one_launch_equal_one_launch_per_view(nviews):
    """accumulate_stats_views (gsr_densify_stats_views, groups of 8 views) leaves, bit for bit, what accumulate_stats
    called view after view leaves: the same float additions in the same order; rows beyond the size are not touched."""
    import gsplat_amd  # noqa: F401
    from gsplat_amd import densify as D
    from gsplat_amd.model import GaussianModel
    start, _, _ = _setup(n=5000)
    size = 4321
    one = GaussianModel(*start, sh_degree=1, size=size)
    two = GaussianModel(*start, sh_degree=1, size=size)
    gen = torch.Generator().manual_seed(78)
    for m in (one, two):
        m._coords_grads.copy_(torch.rand(5000, 1, generator=torch.Generator().manual_seed(3)).to(dev()))
        m._denom.fill_(2)
    gs = [(torch.randn(size, 3, generator=gen) * 1e-3).to(dev()) for _ in range(nviews)]
    masks = [(torch.rand(size, generator=gen) < 0.5).to(dev()) for _ in range(nviews)]
    for g, m in zip(gs, masks):
        D.accumulate_stats(one, g, m)
    D.accumulate_stats_views(two, gs, masks)
    assert torch.equal(one._denom, two._denom)
    assert torch.equal(one._coords_grads.view(torch.int32), two._coords_grads.view(torch.int32))
    with pytest.raises(ValueError):
        D.accumulate_stats_views(two, gs, masks[:-1] if nviews > 1 else [])
