"""How long are the runs of consecutive list entries whose alpha is pinned at 0.01 on the whole block
(both hint bits set)?  Config 3 scene (1 M Gaussians, 1080p, tile 16), one view.  Prints the share of such
entries, the distribution of run lengths, and the entries that would remain if every run became one."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib
    from gsplat_amd.rasterizer import ViewSpec, forward_pipeline
    from gsplat_amd.scenes import synthetic_scene
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    dev = torch.device("cuda", 0)
    sc = synthetic_scene(n, 1920, 1080, 3, seed=0)
    p = [t.to(dev) for t in sc.params()]
    view = ViewSpec.from_camera(sc.camera(device=dev), 3, 16)
    st = forward_pipeline(*p, None, view)
    flags = (st.stream_rec[:st.num_pairs].view(torch.int32)[:, 9].cpu().numpy() >> 30) & 3
    off = st.tile_offsets.cpu().numpy().astype(np.int64)
    low = flags == 3
    # run starts: a low entry whose predecessor is not low or belongs to another tile
    first_of_tile = np.zeros(len(low), bool)
    first_of_tile[off[:-1][off[:-1] < len(low)]] = True
    prev_low = np.concatenate(([False], low[:-1])) & ~first_of_tile
    starts = low & ~prev_low
    runs = int(starts.sum())
    ids = np.cumsum(starts)[low]
    lengths = np.bincount(ids)[1:]
    print(f"pairs {len(low)}, fully clamped {low.mean():.3f}, runs {runs}, mean run {lengths.mean():.2f}, "
          f"median {np.median(lengths):.0f}, p90 {np.percentile(lengths, 90):.0f}, max {lengths.max()}")
    remaining = len(low) - int(low.sum()) + runs
    print(f"entries after merging every run into one: {remaining} = {remaining / len(low):.3f} of the list entries")
    one = flags == 1
    two = flags == 2
    print(f"one half clamped: top {one.mean():.3f}, bottom {two.mean():.3f}; none {np.mean(flags == 0):.3f}")
    assert _lib is not None


if __name__ == "__main__":
    main()
