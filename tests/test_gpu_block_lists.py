"""GPU: tiles of 2x2, 3x3 and 4x4 blocks composited from one compact list per 16x16 block (``gsr_block_lists``,
``gsr_composite_blocks_fwd`` / ``_bwd``, ``gsr_block_rows_reduce``) against the tile-group kernels that walk the tile's
whole list in every block (``GSR_BLOCK_LISTS=0``), which the oracle and the reference's fixtures pin: same image and
transmittance to rounding (the product 0.99^n of a run is formed once instead of factor by factor), gradients within the
suite's tolerance; the structure of the block streams against the hint flags of the tile stream; the cases that stay on
the tile-group kernels."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import assert_close, assert_close_nan, grad_tolerance, load_golden, nan_colour_golden_names, nonfinite_golden_names  # noqa: E402

pytestmark = pytest.mark.gpu
GRADS = ("g_coords", "g_scale", "g_rotation", "g_opacity", "g_sh", "g_screen")
OUTSIDE, EVALUATED_BIT = 0xFFFFFFFF, 32


@pytest.fixture(autouse=True)
def _restore_switches():
    old = {k: os.environ.get(k) for k in ("GSR_BLOCK_LISTS", "GSR_BLOCK_WIDE")}
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _render(p, view, cot, blocks):
    from gsplat_amd import rasterizer as R
    os.environ["GSR_BLOCK_LISTS"] = "1" if blocks else "0"
    st = R.forward_pipeline(*p, None, view)
    grads = R.backward_pipeline(st, *p, cot)
    torch.cuda.synchronize()
    return st, grads


def _scene(kind, n, W, H, deg, seed):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.scenes import adversarial_scene, synthetic_scene
    return (adversarial_scene if kind == "adv" else synthetic_scene)(n, W, H, deg, seed=seed)   # ("syn", "opaque")


@pytest.mark.parametrize("kind,n,W,H,deg,tile", [("adv", 30000, 330, 210, 2, 64), ("adv", 30000, 330, 210, 2, 48),
                                                 ("adv", 30000, 330, 210, 2, 32), ("syn", 200000, 800, 800, 3, 64),
                                                 ("syn", 5, 64, 48, 0, 64), ("syn", 3000, 17, 130, 1, 32),
                                                 ("opaque", 6000, 200, 160, 1, 64), ("opaque", 6000, 200, 160, 1, 32)])
@pytest.mark.parametrize("wide", [False, True])
def test_block_lists_equal_the_tile_groups(kind, n, W, H, deg, tile, wide):
    """(``wide``: the 64-bit combination words that views with 2^26 - 1 or more pair-block combinations get)"""
    from gsplat_amd.rasterizer import ViewSpec
    if wide and n > 50000:
        pytest.skip("the wide words are exercised on the smaller scenes")
    os.environ["GSR_BLOCK_WIDE"] = "1" if wide else "0"
    dev = torch.device("cuda:0")
    sc = _scene(kind, n, W, H, deg, 5)
    if kind == "opaque":            # opaque splats of some size on top of each other
        sc.opacity = torch.full_like(sc.opacity, 6.0)
        sc.scale = sc.scale + 1.0
    p = [t.to(dev) for t in sc.params()]
    view = ViewSpec.from_camera(sc.camera(device=dev), deg, tile)
    cot = torch.rand(W, H, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    a, ga = _render(p, view, cot, blocks=False)
    b, gb = _render(p, view, cot, blocks=True)
    assert a.blocks is None and b.blocks is not None
    assert b.blocks.combo.dtype == (torch.int64 if wide else torch.int32)
    if kind == "opaque":
        # pixels whose transmittance underflows: their waves take the checked sweep of the backward, through runs too
        combo = b.blocks.combo.cpu().numpy()
        combo = combo.view(np.uint64 if wide else np.uint32)
        out = np.uint64(0xFFFFFFFFFFFFFFFF) if wide else np.uint32(OUTSIDE)
        assert bool((b.final_T < 1.1754944e-38).any()) and bool(((combo != out) & ((combo & EVALUATED_BIT) == 0)).any())
    # forward: only the order of the roundings differs
    assert_close(b.img.cpu().numpy(), a.img.cpu().numpy(), "img", rtol=2e-5)
    assert_close(b.final_T.cpu().numpy(), a.final_T.cpu().numpy(), "final_T", rtol=2e-5)
    for name, x, y in zip(GRADS, gb, ga):
        assert_close(x.cpu().numpy(), y.cpu().numpy(), name, **grad_tolerance(n))


@pytest.mark.parametrize("tile,W,H", [(64, 330, 210), (48, 330, 210), (32, 200, 120)])
def test_block_streams_follow_the_hint_flags(tile, W, H):
    """Every (list position, block) combination is either evaluated - then the block's stream holds the entry's record,
    in list order, with its index in the block's stream - or inside a run of entries whose two hint bits for that block are set; a run
    record carries 0.99^n and the run's colour; blocks beyond the image's edge have no stream."""
    from gsplat_amd import rasterizer as R
    dev = torch.device("cuda:0")
    sc = _scene("adv", 20000, W, H, 1, 9)
    p = [t.to(dev) for t in sc.params()]
    os.environ["GSR_BLOCK_LISTS"] = "1"
    st = R.forward_pipeline(*p, None, R.ViewSpec.from_camera(sc.camera(device=dev), 1, tile))
    torch.cuda.synchronize()
    assert st.blocks is not None
    S = tile // 16
    S2 = S * S
    I = st.num_pairs
    nbh, nbw16, nbh16 = (H + tile - 1) // tile, (W + 15) // 16, (H + 15) // 16
    toff = st.tile_offsets.cpu().numpy().astype(np.int64)
    tile_rec = st.stream_rec[:I].cpu().numpy()
    flags = tile_rec[:, 11].copy().view(np.uint32)
    boff = st.blocks.offsets.cpu().numpy().astype(np.int64)
    assert len(boff) == nbw16 * nbh16 + 1 and boff[0] == 0 and np.all(np.diff(boff) >= 0)
    R_total = int(boff[-1])
    brec = st.blocks.stream[:R_total].cpu().numpy()
    combo = st.blocks.combo[:I * S2].cpu().numpy().view(np.uint32).reshape(I, S2)
    word9 = brec[:, 9].copy().view(np.uint32)
    is_run = (word9 >> 30) == 3
    seen_records = 0
    for t in range(len(toff) - 1):
        b, e = int(toff[t]), int(toff[t + 1])
        xi, yi = t // nbh, t % nbh
        for c in range(S2):
            bx, by = c // S, c % S
            X16, Y16 = xi * S + bx, yi * S + by
            col = combo[b:e, c]
            if X16 >= nbw16 or Y16 >= nbh16:
                assert np.all(col == OUTSIDE)
                continue
            blk = Y16 + nbh16 * X16
            lo, hi = int(boff[blk]), int(boff[blk + 1])
            seen_records += hi - lo
            clamped = ((flags[b:e] >> (2 * c)) & 3) == 3
            assert np.all(col != OUTSIDE)
            assert np.array_equal((col & EVALUATED_BIT) != 0, ~clamped), "evaluated exactly where the hint bits are not both set"
            # walk the list: expected records of this block
            expect = []          # ("e", position) or ("r", first position, length)
            k = b
            while k < e:
                if not clamped[k - b]:
                    expect.append(("e", k))
                    k += 1
                    continue
                end = k
                chunk_end = b + ((k - b) // 32 + 1) * 32          # runs end at multiples of 32 list positions of the tile
                while end < e and end < chunk_end and clamped[end - b]:
                    end += 1
                expect.append(("r", k, end - k))
                k = end
            assert len(expect) == hi - lo
            for r, x in zip(range(lo, hi), expect):
                if x[0] == "e":
                    pos = x[1]
                    assert not is_run[r]
                    assert (word9[r] & 0x3FFFFFFF) == r - lo and (word9[r] >> 30) == ((flags[pos] >> (2 * c)) & 3)
                    assert col[pos - b] == ((r << 6) | EVALUATED_BIT)
                    assert np.array_equal(brec[r, :9].view(np.uint32), tile_rec[pos, :9].view(np.uint32))   # y x D' A' o B' r ca g
                    assert brec[r, 10] == tile_rec[pos, 10] and brec[r, 12] == tile_rec[pos, 12] and brec[r, 14] == tile_rec[pos, 14]
                    assert brec[r, 11] == np.float32(0.01) * tile_rec[pos, 10]
                    assert brec[r, 13] == np.float32(0.01) * tile_rec[pos, 6]
                    assert brec[r, 15] == np.float32(0.01) * tile_rec[pos, 8]
                else:
                    first, length = x[1], x[2]
                    assert is_run[r] and (word9[r] & 0x3FFFFFFF) == r - lo
                    assert np.array_equal(col[first - b:first - b + length], (r << 6) | np.arange(length, dtype=np.uint32))
                    P = np.float32(np.float64(np.float32(0.99)) ** length)
                    assert brec[r, 4] == P and abs(brec[r, 5] * P - 1.0) < 1e-6
                    w = np.float64(np.float32(0.99)) ** np.arange(length)
                    for dword, src in ((13, 6), (15, 8), (11, 10)):       # red, green, blue
                        want = float(np.sum(w * 0.01 * tile_rec[first:first + length, src].astype(np.float64)))
                        assert abs(brec[r, dword] - want) <= 1e-5 * max(abs(want), 1e-3)
    assert seen_records == R_total


def test_views_that_keep_the_tile_groups():
    """Tile edges that are not 32, 48 or 64 and ``GSR_BLOCK_LISTS=0`` stay on the tile lists; the library says which."""
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib, rasterizer as R
    dev = torch.device("cuda:0")
    sc = _scene("syn", 2000, 160, 120, 1, 2)
    p = [t.to(dev) for t in sc.params()]
    lib = _lib.load()
    for tile, want in ((16, False), (40, False), (80, False), (32, True), (48, True), (64, True)):
        os.environ["GSR_BLOCK_LISTS"] = "1"
        st = R.forward_pipeline(*p, None, R.ViewSpec.from_camera(sc.camera(device=dev), 1, tile))
        assert (st.blocks is not None) == want
        assert bool(lib.gsr_block_lists_supported(st.cview, st.num_pairs)) == want
        assert R.pairgrad_parts(st) == (1 if want else int(lib.gsr_pairgrad_parts(tile)))
    os.environ["GSR_BLOCK_LISTS"] = "0"
    st = R.forward_pipeline(*p, None, R.ViewSpec.from_camera(sc.camera(device=dev), 1, 64))
    assert st.blocks is None
    assert not lib.gsr_block_lists_supported(st.cview, 1 << 24)        # 2^24 pairs x 16 blocks = 2^28: too many records
    assert lib.gsr_block_lists_supported(st.cview, (1 << 24) - 1)


def test_minibatch_at_the_default_tile_edge_equals_the_one_view_calls():
    """``render_views`` (lists on a side stream, record sums on a third) at a tile edge of 64 against the views rendered
    one by one: bit for bit."""
    import gsplat_amd  # noqa: F401
    from gsplat_amd import rasterizer as R
    from gsplat_amd.minibatch import render_views
    from gsplat_amd.scenes import orbit_scene
    dev = torch.device("cuda:0")
    n, W, H, deg, nv = 20000, 320, 200, 2, 3
    scenes = [orbit_scene(n, W, H, deg, view=v, n_views=nv, seed=4) for v in range(nv)]
    p = [t.to(dev) for t in scenes[0].params()]
    views = [R.ViewSpec.from_camera(s.camera(device=dev), deg, 64) for s in scenes]
    cots = [torch.rand(W, H, 3, generator=torch.Generator().manual_seed(10 + v)).to(dev) for v in range(nv)]
    os.environ["GSR_BLOCK_LISTS"] = "1"
    f32 = dict(dtype=torch.float32, device=dev)
    grads = (torch.empty(n, 3, **f32), torch.empty(n, 3, **f32), torch.empty(n, 4, **f32), torch.empty(n, 1, **f32),
             torch.empty(n, 3, 16, **f32))
    states, _, g_screens = render_views(p, views, lambda i, st: cots[i], grads=grads)
    torch.cuda.synchronize()
    assert all(st.blocks is not None for st in states if st.num_visible)
    acc = None
    for v in range(nv):
        st = R.forward_pipeline(*p, None, views[v])
        assert torch.equal(st.img, states[v].img)
        six = R.backward_pipeline(st, *p, cots[v])
        assert torch.equal(six[5], g_screens[v])
        acc = [g.clone() for g in six[:5]] if acc is None else [a + g for a, g in zip(acc, six[:5])]
    for name, a, g in zip(GRADS, acc, grads):
        assert_close(g.cpu().numpy(), a.cpu().numpy(), name, rtol=2e-5)


def test_block_streams_beyond_4_gib():
    """Splats that cover the whole image: every block evaluates every entry of its tile, the block streams hold as many
    records as there are pair-block combinations - more than 2^26 of them, more than 4 GiB, with 64-bit combination words -
    and the compositing kernels address them from each list's own first record."""
    from gsplat_amd import rasterizer as R
    dev = torch.device("cuda:0")
    W, H, n = 1920, 1080, 12000
    sc = _scene("syn", n, W, H, 0, 11)
    sc.scale = sc.scale + 8.0
    p = [t.to(dev) for t in sc.params()]
    view = R.ViewSpec.from_camera(sc.camera(device=dev), 0, 64)
    cot = torch.rand(W, H, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    b, gb = _render(p, view, cot, blocks=True)
    assert b.blocks is not None and b.blocks.combo.dtype == torch.int64
    records = int(b.blocks.offsets[-1].item())
    assert b.num_pairs * 16 >= (1 << 26) and records * 64 > (1 << 32), (b.num_pairs, records)
    img, final_T = b.img.clone(), b.final_T.clone()
    del b
    torch.cuda.empty_cache()
    a, ga = _render(p, view, cot, blocks=False)
    assert_close(img.cpu().numpy(), a.img.cpu().numpy(), "img", rtol=2e-5)
    assert_close(final_T.cpu().numpy(), a.final_T.cpu().numpy(), "final_T", rtol=2e-5)
    for name, x, y in zip(GRADS, gb, ga):
        assert_close(x.cpu().numpy(), y.cpu().numpy(), name, **grad_tolerance(n))


@pytest.mark.parametrize("tile", [32, 64])
def test_nan_and_infinite_inputs_spread_as_with_the_tile_groups(tile):
    """The inputs of the fixtures with a NaN sh coefficient, a NaN opacity, NaN / infinite coordinates and degenerate
    quaternions (generated from the reference at a tile edge of 16), rendered at tile edges of several blocks: the
    per-block lists turn the same pixels and the same gradient entries NaN as the tile groups, the rest agrees."""
    from gsplat_amd import rasterizer as R
    dev = torch.device("cuda:0")
    names = nan_colour_golden_names() + nonfinite_golden_names()
    assert names
    for name in names:
        g = load_golden(name)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(dev)  # noqa: E731
        p = [t(g["coords"]), t(g["scale"]), t(g["rotation"]), t(np.asarray(g["opacity"]).reshape(-1, 1)), t(g["sh"])]
        f = lambda a, k: tuple(np.asarray(a, np.float32).reshape(k).tolist())  # noqa: E731
        view = R.ViewSpec(int(g["sh_degree"]), int(g["width"]), int(g["height"]), tile, f(g["w2m"], 16), f(g["proj"], 16), f(g["cam_pos"], 3))
        cot = t(g["cotangent"])
        a, ga = _render(p, view, cot, blocks=False)
        b, gb = _render(p, view, cot, blocks=True)
        assert a.blocks is None and b.blocks is not None
        assert_close_nan(b.img.cpu().numpy(), a.img.cpu().numpy(), name + " img", rtol=2e-5)
        for key, x, y in zip(GRADS, gb, ga):
            assert_close_nan(x.cpu().numpy(), y.cpu().numpy(), name + " " + key, **grad_tolerance(len(g["coords"])))
