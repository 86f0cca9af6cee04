"""Per-block lists against the tile-group kernels on one view: image, final transmittance and the six gradients, the
structure of the block streams, and the time of each stage.  Usage: python tools/block_lists_check.py [N W H tile]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsplat_amd  # noqa: E402,F401
from gsplat_amd import rasterizer as R  # noqa: E402
from gsplat_amd.scenes import synthetic_scene  # noqa: E402


def run(p, view, cot, blocks):
    os.environ["GSR_BLOCK_LISTS"] = "1" if blocks else "0"
    st = R.forward_pipeline(*p, None, view)
    grads = R.backward_pipeline(st, *p, cot)
    torch.cuda.synchronize()
    return st, grads


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    n, W, H, tile = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (30000, 330, 210, 64)
    deg = 3
    dev = torch.device("cuda:0")
    sc = synthetic_scene(n, W, H, deg, seed=0)
    p = [t.to(dev) for t in sc.params()]
    view = R.ViewSpec.from_camera(sc.camera(device=dev), deg, tile)
    cot = torch.rand(W, H, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    a, ga = run(p, view, cot, False)
    b, gb = run(p, view, cot, True)
    assert b.blocks is not None, "the view did not qualify for per-block lists"
    S2 = (tile // 16) ** 2
    off = b.blocks.offsets.cpu().long()
    R_total = int(off[-1])
    combo = b.blocks.combo.cpu().long() & 0xFFFFFFFF
    ev, outside = int(((combo != 0xFFFFFFFF) & ((combo & 32) != 0)).sum()), int((combo == 0xFFFFFFFF).sum())
    print(f"{n} Gaussians {W}x{H} tile {tile}: {b.num_pairs} pairs, {len(off) - 1} blocks, {R_total} block records "
          f"({R_total / max(b.num_pairs * S2 - outside, 1):.3f} of the pair-block combinations), evaluated {ev}, outside {outside}")
    assert bool((off[1:] >= off[:-1]).all())
    names = ("img", "final_T", "g_coords", "g_scale", "g_rotation", "g_opacity", "g_sh", "g_screen")
    for name, x, y in zip(names, (a.img, a.final_T) + tuple(ga), (b.img, b.final_T) + tuple(gb)):
        scale = float(x.abs().max())
        d = float((x - y).abs().max())
        rel = float(((x - y).abs() / (x.abs() + 2e-6 * scale)).max())
        print(f"  {name:10s} max |diff| / max |ref| = {d / max(scale, 1e-30):.2e}   worst relative (floor 2e-6 max) = {rel:.2e}")
    os.environ["GSR_BLOCK_LISTS"] = "1"
    lib = gsplat_amd._lib.load()
    t_lists = timed(lambda: R._block_lists_of(b))
    t_fwd = timed(lambda: R.forward_composite(b))
    t_bwd = timed(lambda: R.backward_composite(b, cot))
    os.environ["GSR_BLOCK_LISTS"] = "0"
    t_fwd0 = timed(lambda: R.forward_composite(a))
    t_bwd0 = timed(lambda: R.backward_composite(a, cot))
    print(f"  block lists {t_lists:.3f} ms, forward {t_fwd:.3f} (tile groups {t_fwd0:.3f}), backward + row sums {t_bwd:.3f} (tile groups {t_bwd0:.3f})")


if __name__ == "__main__":
    main()
