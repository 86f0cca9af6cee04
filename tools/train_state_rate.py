#!/usr/bin/env python3
"""Device time of the training-state kernels at 1 M Gaussians (HIP events on the launch stream):
the fused Adam step over the 59 floats per Gaussian, the densification statistics, a densify pass
and a remove.  Prints achieved HBM GB/s from the algorithmic bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsplat_amd  # noqa: E402,F401
from gsplat_amd.minibatch import GradBucket  # noqa: E402
from gsplat_amd.model import GaussianModel  # noqa: E402
from gsplat_amd.optimizer import Adam  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)   # noqa: E731
    model = GaussianModel(mk(n, 3), mk(n, 3) * 0.5 + 0.3, mk(n, 4), mk(n, 1), mk(n, 3, 16), sh_degree=3)
    params = [model._coords, model._scale, model._rotation, model._opacity, model._sh]
    opt = Adam(params, lr=1e-3, betas=(0.5, 0.999))
    bucket = GradBucket(n, dev)
    bucket.flat.normal_()
    ms = timed(lambda: opt.step(grads=bucket.views))
    nbytes = 59 * n * 4 * 7
    print(f"adam_step: {ms:.4f} ms, {nbytes / ms / 1e6:.0f} GB/s ({nbytes / ms / 1e6 / 8000:.1%} of 8 TB/s), 28 B per element")
    sc = torch.zeros(n, 3, device=dev, requires_grad=True)
    sc.grad = mk(n, 3) * 1e-3
    mask = torch.rand(n, generator=g).to(dev) < 0.9
    m8 = mask.to(torch.uint8)
    ms = timed(lambda: model.update_densify_stats(sc, m8))
    nbytes = n * (1 + 12 + 16)
    print(f"densify_stats: {ms:.4f} ms, {nbytes / ms / 1e6:.0f} GB/s")
    model._coords_grads.uniform_(0, 0.0006)
    model._denom.fill_(1)
    noise = torch.randn(2 * n, 3, device=dev)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    nc, ns = model.densify(60.0, noise=noise)
    b.record()
    torch.cuda.synchronize()
    print(f"densify: {a.elapsed_time(b):.3f} ms incl. count read-back and capacity growth: {nc} cloned, {ns} split, size {model._size}, capacity {model.capacity}")
    mask = torch.rand(model._size, device=dev) < 0.2
    a.record()
    r = model.remove(mask)
    b.record()
    torch.cuda.synchronize()
    print(f"remove: {a.elapsed_time(b):.3f} ms incl. count read-back: {r} rows")


if __name__ == "__main__":
    main()
