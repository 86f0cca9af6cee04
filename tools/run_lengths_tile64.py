"""Per-block shares and run lengths of list entries whose alpha is pinned at 0.01 on a whole 16x16 block, for
tiles of several blocks (config 3 scene at tile edge 64 by default: sixteen blocks per tile, two hint bits each)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.rasterizer import ViewSpec, forward_pipeline
    from gsplat_amd.scenes import synthetic_scene
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    S = (L + 15) // 16
    dev = torch.device("cuda", 0)
    sc = synthetic_scene(n, 1920, 1080, 3, seed=0)
    p = [t.to(dev) for t in sc.params()]
    st = forward_pipeline(*p, None, ViewSpec.from_camera(sc.camera(device=dev), 3, L))
    flags = st.stream_rec[:st.num_pairs].view(torch.int32)[:, 11].cpu().numpy().astype(np.uint32)
    off = st.tile_offsets.cpu().numpy().astype(np.int64)
    first_of_tile = np.zeros(len(flags), bool)
    first_of_tile[off[:-1][off[:-1] < len(flags)]] = True
    shares = np.zeros(4)
    all_len = []
    for c in range(S * S):
        f = (flags >> (2 * c)) & 3
        shares += np.bincount(f, minlength=4)
        low = f == 3
        prev_low = np.concatenate(([False], low[:-1])) & ~first_of_tile
        starts = low & ~prev_low
        ids = np.cumsum(starts)[low]
        all_len.append(np.bincount(ids)[1:])
    lengths = np.concatenate(all_len)
    shares /= shares.sum()
    print(f"tile {L}: pairs {len(flags)}, block entries {len(flags) * S * S}; none {shares[0]:.3f} top-low {shares[1]:.3f} "
          f"bottom-low {shares[2]:.3f} both {shares[3]:.3f}")
    print(f"runs of both-low entries per block: {len(lengths)}, mean {lengths.mean():.2f}, median {np.median(lengths):.0f}, "
          f"p90 {np.percentile(lengths, 90):.0f}, max {lengths.max()}")


if __name__ == "__main__":
    main()
