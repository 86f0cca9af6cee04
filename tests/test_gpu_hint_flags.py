"""GPU: the hint flags of the record stream against a direct evaluation.  A half block flagged "clamped low" must
have opacity * G below 0.01 on every one of its pixels (the compositing kernels skip the exponent there: a wrong flag
is a wrong image), and the test the list builder applies - the minimum of the quadratic form over the half's
rectangle, on the edges that face the centre - should miss few of the halves that are clamped (a missed half only
costs time).  Tile edges 16 (one block per tile, bits 0 and 1) and 64 (sixteen blocks, two bits each) and 40 (nine
blocks, the last row and column of blocks cut by the tile's edge)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tile", [16, 64, 40])
def test_flagged_halves_are_clamped_and_few_clamped_halves_are_missed(tile):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.rasterizer import ViewSpec, forward_pipeline
    from gsplat_amd.scenes import adversarial_scene
    W, H = 330, 210                      # (neither a multiple of 16 nor of the tile edges)
    dev = torch.device("cuda:0")
    sc = adversarial_scene(30000, W, H, 2, seed=5)
    p = [t.to(dev) for t in sc.params()]
    st = forward_pipeline(*p, None, ViewSpec.from_camera(sc.camera(device=dev), 2, tile))
    torch.cuda.synchronize()
    S = (tile + 15) // 16
    nbh = (H + tile - 1) // tile
    off = st.tile_offsets.cpu().long()
    rec = st.stream_rec[:st.num_pairs]
    # (one-block tiles: the two hint bits ride above the emission slot in dword 9; larger tiles: dword 11)
    flags_all = (rec[:, 9].contiguous().view(torch.int32) >> 30) & 3 if tile <= 16 else rec[:, 11].contiguous().view(torch.int32)
    gen = torch.Generator().manual_seed(0)
    tiles = torch.randperm(off.numel() - 1, generator=gen)[:60].tolist()
    flagged = clamped = wrong = halves = 0
    for t in tiles:
        b, e = int(off[t]), int(off[t + 1])
        if e <= b:
            continue
        r, fl = rec[b:e], flags_all[b:e]
        xi, yi = t // nbh, t % nbh
        for bx in range(S):
            for by in range(S):
                x0, y0 = xi * tile + 16 * bx, yi * tile + 16 * by
                x1, y1 = min(x0 + 16, (xi + 1) * tile, W), min(y0 + 16, (yi + 1) * tile, H)
                for half, (ya, yb) in enumerate(((y0, min(y0 + 8, y1)), (y0 + 8, y1))):
                    bit = (fl >> (2 * (bx * S + by) + half)) & 1
                    if x1 <= x0 or yb <= ya:
                        assert bool((bit == 1).all()), "a half without pixels counts as clamped"
                        continue
                    xs = torch.arange(x0, x1, device=dev).float()
                    ys = torch.arange(ya, yb, device=dev).float()
                    # stream layout: 0 y, 1 x, 2 D', 3 A', 4 opacity, 5 B'; the kernels' own evaluation order
                    dx = r[:, 1, None, None] - xs[None, :, None]
                    dy = r[:, 0, None, None] - ys[None, None, :]
                    q = torch.addcmul(r[:, 3, None, None] * dx * dx, torch.addcmul(r[:, 5, None, None] * dx, r[:, 2, None, None], dy), dy)
                    raw = r[:, 4, None, None] * torch.exp2(q)
                    low = (raw < 0.01).flatten(1).all(1)
                    halves += low.numel()
                    flagged += int((bit == 1).sum())
                    clamped += int(low.sum())
                    wrong += int(((bit == 1) & ~low).sum())
    assert halves > 20000
    assert wrong == 0, f"{wrong} flagged halves have a pixel with opacity * G >= 0.01"
    assert flagged >= 0.97 * clamped, f"only {flagged} of {clamped} clamped halves are flagged"
