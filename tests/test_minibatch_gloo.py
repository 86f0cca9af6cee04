"""The N>1 path on CPU: two gloo ranks split a 4-view minibatch, accumulate per-view gradients in
their flat buckets and all-reduce; the result must equal the single-process sum over all views.
Per-view gradients come from the CPU oracle (the HIP kernels need a GPU); the code under test is
the sharding and the bucket."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRADS = ("g_coords", "g_scale", "g_rotation", "g_opacity", "g_sh")
N_VIEWS, N_GAUSS = 4, 600


def _view_grads(view):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gsplat_amd  # noqa: F401
    import oracle
    from gsplat_amd.camera import Camera
    from gsplat_amd.scenes import synthetic_scene
    sc = synthetic_scene(N_GAUSS, 96, 64, 2, seed=5)
    ang = 2.0 * np.pi * view / N_VIEWS
    cam = Camera(np.array([0.15 * np.cos(ang), 0.15 * np.sin(ang), 0.0]), np.array([1.0, 0, 0, 0]), sc.fx, sc.fy,
                 48, 32, 96, 64, device="cpu")
    cot = torch.rand(96, 64, 3, generator=torch.Generator().manual_seed(100 + view)).numpy() / N_VIEWS
    o = oracle.render(sc.coords.numpy(), sc.scale.numpy(), sc.rotation.numpy(), sc.opacity.numpy(), sc.sh.numpy(), 2,
                      96, 64, 16, cam.w2m_host, cam.proj_host, np.asarray(cam.transform.position, np.float32),
                      dimg=cot, threads=1)
    return [torch.from_numpy(o[k]) for k in GRADS]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import gsplat_amd  # noqa: F401
    from gsplat_amd.minibatch import GradBucket, shard_views
    bucket = GradBucket(N_GAUSS, "cpu")
    for v in shard_views(N_VIEWS, world, rank):
        bucket.add_(_view_grads(v))
    bucket.all_reduce()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), bucket.flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_views_partitions_the_minibatch():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.minibatch import shard_views
    for n_views in (8, 7, 1):
        for world in (1, 2, 4, 8):
            got = [v for r in range(world) for v in shard_views(n_views, world, r)]
            assert got == list(range(n_views))
    assert shard_views(8, 8, 3) == [3] and shard_views(8, 2, 1) == [4, 5, 6, 7]
    with pytest.raises(ValueError):
        shard_views(8, 2, 2)


def test_bucket_views_alias_the_flat_buffer():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.minibatch import FLOATS_PER_GAUSSIAN, GradBucket
    b = GradBucket(10, "cpu")
    assert FLOATS_PER_GAUSSIAN == 59 and b.flat.numel() == 590
    b.sh[3, 1, 5] = 2.0
    b.coords[9, 2] = 1.0
    assert b.flat.sum().item() == 3.0
    assert [tuple(v.shape) for v in b.views] == [(10, 3), (10, 3), (10, 4), (10, 1), (10, 3, 16)]


def test_two_rank_allreduce_matches_single_process(tmp_path):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.minibatch import GradBucket
    ref = GradBucket(N_GAUSS, "cpu")
    for v in range(N_VIEWS):
        ref.add_(_view_grads(v))
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npy"))
        # fp32 reassociation only: (v0+v1)+(v2+v3) against ((v0+v1)+v2)+v3
        np.testing.assert_allclose(got, ref.flat.numpy(), rtol=1e-5, atol=1e-6 * float(np.abs(ref.flat.numpy()).max()))


def _view_small_and_colour(view):
    """(the 11 small gradient floats per Gaussian, flat in bucket order; a stand-in colour gradient [N,3])"""
    g = _view_grads(view)
    small = torch.cat([t.reshape(-1) for t in g[:4]])
    colour = torch.rand(N_GAUSS, 3, generator=torch.Generator().manual_seed(900 + view)) - 0.5
    return small, colour


def _factored_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import gsplat_amd  # noqa: F401
    from gsplat_amd.minibatch import SMALL_FLOATS, GradBucket, reduce_gradients, shard_views
    bucket = GradBucket(N_GAUSS, "cpu")
    mine = shard_views(N_VIEWS, world, rank)
    cols = torch.empty(len(mine), N_GAUSS, 3)
    for k, v in enumerate(mine):
        small, colour = _view_small_and_colour(v)
        bucket.flat[:N_GAUSS * SMALL_FLOATS] += small
        cols[k] = colour
    bucket.sh.fill_(-7.0)       # the collectives must not touch the sh part
    seen = {}
    g_all = reduce_gradients(bucket, cols, N_VIEWS, lambda g: seen.setdefault("g", g.clone()))
    assert torch.equal(seen["g"], g_all) and float(bucket.sh.min()) == -7.0 == float(bucket.sh.max())
    np.save(os.path.join(out_dir, f"small{rank}.npy"), bucket.flat[:N_GAUSS * SMALL_FLOATS].numpy())
    np.save(os.path.join(out_dir, f"colour{rank}.npy"), g_all.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_factored_exchange_moves_the_right_blocks(tmp_path):
    """reduce_gradients: the small gradients are summed over the ranks, the colour gradients of all
    views arrive on every rank in minibatch order, the sh part of the bucket is left to the rebuild."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_factored_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    parts = [_view_small_and_colour(v) for v in range(N_VIEWS)]
    want_small = sum(p[0] for p in parts).numpy()
    want_colour = torch.stack([p[1] for p in parts]).numpy()
    for r in range(2):
        np.testing.assert_allclose(np.load(os.path.join(str(tmp_path), f"small{r}.npy")), want_small, rtol=1e-5,
                                   atol=1e-6 * float(np.abs(want_small).max()))
        np.testing.assert_array_equal(np.load(os.path.join(str(tmp_path), f"colour{r}.npy")), want_colour)


def test_factored_exchange_needs_an_even_split():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.minibatch import GradBucket, can_factor, factored_moves_less, reduce_gradients
    assert can_factor(8, 1) and can_factor(8, 8) and not can_factor(8, 3)
    assert factored_moves_less(8) and factored_moves_less(31) and not factored_moves_less(32)
    with pytest.raises(ValueError):
        reduce_gradients(GradBucket(5, "cpu"), torch.zeros(2, 6, 3), 2, lambda g: None)    # more Gaussians than bucket rows
    # a bucket of capacity rows takes the colour gradients of fewer live Gaussians
    seen = []
    reduce_gradients(GradBucket(5, "cpu"), torch.zeros(2, 4, 3), 2, seen.append)
    assert tuple(seen[0].shape) == (2, 4, 3)


class _Stats:
    """The three attributes reduce_densify_stats touches of a GaussianModel (GaussianModel.py:19-27)."""

    def __init__(self, cap, size):
        self._size = size
        self._coords_grads = torch.zeros(cap, 1)
        self._denom = torch.zeros(cap, 1, dtype=torch.int32)


def _stats_inputs(view, size):
    gen = torch.Generator().manual_seed(4000 + view)
    return torch.randn(size, 3, generator=gen) * 1e-3, torch.rand(size, generator=gen) < 0.6


def _stats_delta_cpu(views, size):
    """What densify.stats_delta computes on the GPU, from the numpy restatement of update_densify_stats."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import train_oracle as to
    m = {"size": size, "coords_grads": np.zeros((size, 1), np.float32), "denom": np.zeros((size, 1), np.int32)}
    for v in views:
        g, mask = _stats_inputs(v, size)
        m = to.densify_stats(m, g.numpy(), mask.numpy())
    return torch.from_numpy(m["coords_grads"].copy()), torch.from_numpy(m["denom"].copy())


def _stats_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import gsplat_amd  # noqa: F401
    from gsplat_amd.minibatch import reduce_densify_stats, shard_views
    cap, size = 50, 41
    model = _Stats(cap, size)
    model._coords_grads[:] = 0.25            # what earlier steps left, identical on the ranks
    model._denom[:] = 3
    for step in range(2):
        dg, dd = _stats_delta_cpu([10 * step + v for v in shard_views(N_VIEWS, world, rank)], size)
        reduce_densify_stats(model, dg, dd)
    np.save(os.path.join(out_dir, f"g{rank}.npy"), model._coords_grads.numpy())
    np.save(os.path.join(out_dir, f"d{rank}.npy"), model._denom.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_densify_stats_equal_the_single_process(tmp_path):
    """reduce_densify_stats: after two steps of four views split over two ranks, both ranks hold the
    statistics a single process gathers from all eight views; rows at and above the size are untouched."""
    import gsplat_amd  # noqa: F401
    from gsplat_amd.minibatch import reduce_densify_stats
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_stats_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    cap, size = 50, 41
    dg, dd = _stats_delta_cpu([10 * step + v for step in range(2) for v in range(N_VIEWS)], size)
    for r in range(2):
        g = np.load(os.path.join(str(tmp_path), f"g{r}.npy"))
        d = np.load(os.path.join(str(tmp_path), f"d{r}.npy"))
        np.testing.assert_allclose(g[:size], 0.25 + dg.numpy(), rtol=1e-6)
        np.testing.assert_array_equal(d[:size], 3 + dd.numpy())
        assert (g[size:] == 0.25).all() and (d[size:] == 3).all()
    assert int(dd.max()) >= 4
    with pytest.raises(ValueError):
        reduce_densify_stats(_Stats(cap, size), torch.zeros(cap, 1), torch.zeros(cap, 1, dtype=torch.int32))
    with pytest.raises(TypeError):
        reduce_densify_stats(_Stats(cap, size), torch.zeros(size, 1), torch.zeros(size, 1))
