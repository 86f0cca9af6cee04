"""GPU: stage 1 and the depth order of several views in one pass (gsr_preprocess_fwd_views, gsr_bin_offsets_views)
leave, for every view, exactly what the one-view entry points leave - every array bit for bit, counters included -
and a minibatch rendered through them gives bit-identical images and gradients."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

FIELDS = ("mask_u8", "pos2d", "cov2d", "radius_full", "bbox", "ntiles", "rec", "order", "slot_base", "slot_seq")


def _scene(n, w, h, deg, nviews, seed=0):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.camera import Camera
    from gsplat_amd.rasterizer import ViewSpec
    from gsplat_amd.scenes import synthetic_scene
    dev = torch.device("cuda", 0)
    sc = synthetic_scene(n, w, h, deg, seed=seed)
    params = [t.to(dev) for t in sc.params()]
    views = []
    for v in range(nviews):
        ang = 2.0 * np.pi * v / max(nviews, 1)
        pos = np.array([0.15 * np.cos(ang), 0.15 * np.sin(ang), 0.0])
        views.append(ViewSpec.from_camera(Camera(pos, np.array([1.0, 0.0, 0.0, 0.0]), sc.fx, sc.fy, w / 2, h / 2, w, h, device=dev), deg, 16))
    return dev, params, views


def _same(a, b, what):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert a.shape == b.shape, what
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"{what} differs"


@pytest.mark.parametrize("lists", ["all views at once", "view by view"])
@pytest.mark.parametrize("n,w,h,deg,nviews", [(3000, 200, 120, 3, 3), (70_001, 640, 360, 2, 8), (5, 64, 48, 0, 2),
                                               (200_000, 800, 800, 3, 11)])
def test_batched_front_equals_the_one_view_calls(n, w, h, deg, nviews, lists):
    from gsplat_amd import _lib
    from gsplat_amd import rasterizer as R
    dev, params, views = _scene(n, w, h, deg, nviews)
    # pos2d and cov2d are optional outputs of the batched pass: asked for in one of the two cases
    projection = lists == "all views at once"
    batched = R.forward_begin_views(*params, views, projection=projection)
    assert len(batched) == nviews
    if lists == "all views at once":
        for k in range(0, nviews, _lib.MAX_VIEWS):
            R.forward_lists_views(batched[k:k + _lib.MAX_VIEWS])
    for k, view in enumerate(views):
        single = R.forward_begin(*params, None, view)
        R.forward_lists(single)
        if lists == "view by view":
            R.forward_lists(batched[k])
        assert (batched[k].num_visible, batched[k].num_pairs) == (single.num_visible, single.num_pairs), k
        for f in FIELDS:
            if f in ("pos2d", "cov2d") and not projection and nviews > 1:
                assert getattr(batched[k], f) is None
                continue
            _same(getattr(batched[k], f), getattr(single, f), f"view {k}: {f}")
        if single.num_visible:
            _same(batched[k].tile_offsets, single.tile_offsets, f"view {k}: tile_offsets")
            _same(batched[k].sorted_gauss[:single.num_pairs], single.sorted_gauss[:single.num_pairs], f"view {k}: lists")
            _same(batched[k].stream_rec[:single.num_pairs], single.stream_rec[:single.num_pairs], f"view {k}: stream")
            if lists == "all views at once":
                # (the order inside a length class is unspecified: compare the lengths the two orders visit)
                off = single.tile_offsets.cpu().long()
                lens = off[1:] - off[:-1]
                a, b = lens[batched[k].tile_order.cpu().long()], lens[single.tile_order.cpu().long()]
                assert sorted(batched[k].tile_order.cpu().tolist()) == list(range(lens.numel()))
                slack = int(lens.max()) // 1023 + 1
                assert bool((a[:-1] + slack >= a[1:]).all()) and bool((b[:-1] + slack >= b[1:]).all())


def test_minibatch_with_batched_front_is_bit_identical():
    from gsplat_amd.minibatch import render_views
    dev, params, views = _scene(150_000, 640, 480, 3, 5, seed=3)
    gen = torch.Generator().manual_seed(5)
    cots = [torch.rand(640, 480, 3, generator=gen).to(dev) for _ in views]
    out = {}
    for batch in (False, True):
        states, grads, g_screens = render_views(params, views, lambda i, st: cots[i], batch_front=batch)
        torch.cuda.synchronize()
        out[batch] = ([st.img.clone() for st in states], [g.clone() for g in grads], [g.clone() for g in g_screens])
    for a, b in zip(out[False][0] + out[False][1] + out[False][2], out[True][0] + out[True][1] + out[True][2]):
        _same(a, b, "minibatch output")


def test_batched_front_refuses_mixed_views():
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib
    dev, params, views = _scene(100, 64, 48, 1, 2)
    lib = _lib.load()
    cv = (_lib.GsrView * 2)(_lib.make_view(100, 1, 64, 48, 16, views[0].w2m, views[0].proj, views[0].cam_pos),
                             _lib.make_view(100, 2, 64, 48, 16, views[1].w2m, views[1].proj, views[1].cam_pos))
    t = torch.zeros(4096, dtype=torch.float32, device=dev)
    assert lib.gsr_preprocess_fwd_views(cv, 2, *params_ptrs(params), *[t.data_ptr()] * 9, torch.cuda.current_stream().cuda_stream) != 0
    assert b"differs" in lib.gsr_last_error()
    assert lib.gsr_preprocess_fwd_views(cv, 9, *params_ptrs(params), *[t.data_ptr()] * 9, torch.cuda.current_stream().cuda_stream) != 0


def params_ptrs(params):
    return [p.contiguous().data_ptr() for p in params]


@pytest.mark.parametrize("n,w,h,deg,tile", [(60_000, 640, 360, 1, 16), (20_000, 300, 200, 2, 64)])
def test_packed_pair_words_and_wide_pairs_build_the_same_lists(n, w, h, deg, tile):
    """The list builder sorts one 64-bit word per pair when tile, Gaussian and slot fit one (the normal case) and a
    32-bit key beside a 64-bit value otherwise; GSR_WIDE_PAIRS=1 forces the latter.  Both give the same lists,
    record stream and tile offsets, bit for bit."""
    from gsplat_amd import rasterizer as R
    dev, params, views = _scene(n, w, h, deg, 1)
    view = R.ViewSpec(views[0].sh_degree, w, h, tile, views[0].w2m, views[0].proj, views[0].cam_pos)
    out = {}
    for wide in ("0", "1"):
        os.environ["GSR_WIDE_PAIRS"] = wide
        try:
            st = R.forward_pipeline(*params, None, view)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("GSR_WIDE_PAIRS", None)
        out[wide] = st
    a, b = out["0"], out["1"]
    assert a.num_pairs == b.num_pairs > 0
    for f in ("tile_offsets", "sorted_gauss", "stream_rec", "img", "final_T"):
        _same(getattr(a, f)[:a.num_pairs] if f in ("sorted_gauss", "stream_rec") else getattr(a, f),
              getattr(b, f)[:a.num_pairs] if f in ("sorted_gauss", "stream_rec") else getattr(b, f), f)


def test_batched_front_with_a_view_that_sees_nothing():
    """One of the views of a minibatch looks away from the cloud: its state reports nothing visible, the others are
    unaffected, and render_views returns gradients equal to those of the views that see something."""
    from gsplat_amd import rasterizer as R
    from gsplat_amd.camera import Camera
    from gsplat_amd.minibatch import render_views
    from gsplat_amd.scenes import synthetic_scene
    dev, params, views = _scene(20_000, 320, 200, 2, 3)
    sc = synthetic_scene(20_000, 320, 200, 2, seed=0)
    # ten thousand units to the side: the whole cloud falls outside the clip window of this camera
    away = Camera(np.array([1.0e4, 0.0, 0.0]), np.array([1.0, 0.0, 0.0, 0.0]), sc.fx, sc.fy, 160, 100, 320, 200, device=dev)
    blind = R.ViewSpec.from_camera(away, 2, 16)
    mixed = [views[0], blind, views[1]]
    states = R.forward_begin_views(*params, mixed)
    for st in states:
        R.forward_lists(st)
    if states[1].num_visible != 0:
        pytest.skip("the displaced camera still sees part of the synthetic cloud")
    assert states[0].num_visible > 0 and states[2].num_visible > 0
    cot = torch.rand(320, 200, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    _, g_mixed, s_mixed = render_views(params, mixed, lambda i, st: cot)
    _, g_seen, s_seen = render_views(params, [views[0], views[1]], lambda i, st: cot)
    torch.cuda.synchronize()
    assert len(s_mixed) == len(s_seen) == 2
    for a, b in zip(list(g_mixed) + list(s_mixed), list(g_seen) + list(s_seen)):
        _same(a, b, "gradients with and without the blind view")


@pytest.mark.parametrize("n,w,h,deg,nviews", [(3000, 200, 120, 3, 1), (70_001, 640, 360, 2, 5), (5, 64, 48, 0, 2)])
def test_stage_1_counts_the_pairs_the_scan_will_report(n, w, h, deg, nviews):
    """The pair total that stage 1 accumulates (gsr_counters.pairs_wide, what the host sizes its buffers from) equals
    the total of the scan along the depth order and the sum of the tile counts, per view, in both entry points."""
    from gsplat_amd import rasterizer as R
    dev, params, views = _scene(n, w, h, deg, nviews)
    begun = R.forward_begin_views(*params, views) if nviews > 1 else [R.forward_begin(*params, None, views[0])]
    device_counters = [st._counters for st in begun]
    torch.cuda.synchronize()
    for k, st in enumerate(begun):
        words = device_counters[k].cpu().numpy().reshape(-1, 8)[k if nviews > 1 else 0].astype(np.int64) & 0xFFFFFFFF
        R.read_counters(st)
        total = int(st.ntiles.cpu().numpy().astype(np.int64).sum())
        assert (int(words[1]), int(words[4]), int(words[5])) == (total, total, 0), k
        assert st.num_pairs == total and st.num_visible == int(st.mask_u8.sum())


def test_pipelined_steps_back_to_back_equal_the_one_stream_step():
    """Several steps through the three-stream pipeline without a host synchronisation between them, with allocator
    churn in between: every gradient equals, bit for bit, the step rendered on one stream.  (The buffers that cross
    streams are ordered by stream-level waits, not by the allocator: a buffer handed out again too early, or a missing
    wait, shows here as a difference.)"""
    from gsplat_amd.minibatch import render_views
    dev, params, views = _scene(150_000, 640, 360, 3, 6)
    gen = torch.Generator().manual_seed(5)
    cots = [(torch.rand(640, 360, 3, generator=gen) / 6).to(dev) for _ in views]
    _, ref, ref_screens = render_views(params, views, lambda i, st: cots[i], overlap=False)
    ref = [g.clone() for g in list(ref) + list(ref_screens)]
    kept = []
    for step in range(8):
        churn = [torch.empty(1 << (16 + step % 5), device=dev).normal_() for _ in range(3)]
        del churn
        _, got, screens = render_views(params, views, lambda i, st: cots[i])
        kept.append([g.clone() for g in list(got) + list(screens)])
    torch.cuda.synchronize()
    for step, tensors in enumerate(kept):
        for a, b in zip(tensors, ref):
            _same(a, b, f"step {step}")


@pytest.mark.parametrize("n,w,h,deg,nviews", [(3000, 200, 120, 3, 3), (70_001, 640, 360, 2, 8), (5, 64, 48, 0, 2), (40_000, 320, 200, 1, 11)])
def test_deferred_slot_bases_equal_the_ones_of_the_scan(n, w, h, deg, nviews):
    """forward_begin_views(defer_slot_base=True) + forward_slot_bases (gsr_slot_bases) leaves the slot bases that the
    scan itself writes otherwise, and everything else unchanged; record_sums runs the deferred call if nobody has."""
    from gsplat_amd import rasterizer as R
    dev, params, views = _scene(n, w, h, deg, nviews)
    plain = R.forward_begin_views(*params, views)
    later = R.forward_begin_views(*params, views, defer_slot_base=True)
    assert all(st._slot_rest is not None for st in later) and later[1]._slot_rest is later[0]._slot_rest   # (one holder per group of views)
    R.forward_slot_bases(later)
    R.forward_slot_bases(later)          # (a second call does nothing)
    torch.cuda.synchronize()
    for k in range(nviews):
        for f in ("order", "slot_base", "slot_seq", "ntiles"):
            _same(getattr(later[k], f), getattr(plain[k], f), f"view {k}: {f}")
    lazy = R.forward_begin_views(*params, views, defer_slot_base=True)
    st = lazy[-1]
    R.forward_finish(st)
    if st.num_visible:
        cot = torch.rand(w, h, 3, generator=torch.Generator().manual_seed(4)).to(dev)
        ref = plain[-1]
        R.forward_finish(ref)
        _same(R.backward_records(st, cot), R.backward_records(ref, cot), "record sums behind the lazily run call")
    assert st._slot_rest is None or not st.num_visible


def test_forward_only_states_free_their_memory_when_dropped():
    """A forward-only render_views defers the slot bases and never runs them; the states it returns must not keep each
    other alive: dropping them gives the per-view arrays back at once, with Python's cyclic collector switched off."""
    import gc
    from gsplat_amd.minibatch import render_views
    dev, params, views = _scene(60_000, 320, 200, 1, 4)
    render_views(params, views, None)          # (warm-up: streams, pinned buffers, library)
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated(dev)
        states, _, _ = render_views(params, views, None)
        torch.cuda.synchronize()
        held = torch.cuda.memory_allocated(dev)
        assert held > base + 60_000 * 4 * 40          # (the N-indexed arrays of four views at least)
        del states
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(dev) <= base, "states dropped without a collection still hold device memory"
    finally:
        gc.enable()


def test_a_high_priority_stream_is_refused():
    """The render path hands work to side streams and waits for them; with a HIGH-priority stream current those waits can
    stall for good, so the call is refused before anything is enqueued - by ``render_views`` and by the one-view path."""
    from gsplat_amd import rasterizer as R
    from gsplat_amd.minibatch import render_views
    dev, params, views = _scene(3000, 200, 120, 1, 2)
    high = torch.cuda.Stream(device=dev, priority=-1)
    assert high.priority < 0
    with torch.cuda.stream(high):
        with pytest.raises(RuntimeError, match="high-priority stream"):
            render_views(params, views, None)
        with pytest.raises(RuntimeError, match="high-priority stream"):
            render_views(params, views, None, overlap=False)
        with pytest.raises(RuntimeError, match="high-priority stream"):
            R.forward_begin(*params, None, views[0])
    states, _, _ = render_views(params, views, None)          # (the default stream still works afterwards)
    torch.cuda.synchronize()
    assert all(st.num_visible for st in states)


@pytest.mark.parametrize("n,w,h,deg", [(3000, 200, 120, 3), (70_001, 640, 360, 2), (5, 64, 48, 0)])
def test_one_view_without_projection_outputs_equals_the_full_one(n, w, h, deg):
    """``forward_begin(..., projection=False)`` - what ``render_views`` uses for a rank with a single view - goes through the
    kernel of the view minibatches with one view and leaves everything but ``pos2d`` / ``cov2d``, bit for bit."""
    from gsplat_amd import rasterizer as R
    dev, params, views = _scene(n, w, h, deg, 1)
    full = R.forward_begin(*params, None, views[0])
    lean = R.forward_begin(*params, None, views[0], projection=False)
    R.forward_finish(full)
    R.forward_finish(lean)
    assert lean.pos2d is None and lean.cov2d is None
    assert (lean.num_visible, lean.num_pairs) == (full.num_visible, full.num_pairs)
    for f in FIELDS:
        if f not in ("pos2d", "cov2d"):
            _same(getattr(lean, f), getattr(full, f), f)
    if full.num_visible:
        for f in ("tile_offsets", "img", "final_T"):
            _same(getattr(lean, f), getattr(full, f), f)
