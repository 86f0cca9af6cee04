"""The render path as a ``torch.autograd.Function`` over the HIP kernels of ``libgsr.so``.

Replaces the body of ``GaussianRenderer.render`` (``GaussianRenderer.py:52-113``): forward
produces the image, the visibility mask and the radii; backward produces the gradients of the
five raw parameter tensors and of ``screen_coords``.  PyTorch only owns the memory and the
stream; all arithmetic runs in the library.
"""
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib


@dataclass
class ViewSpec:
    """Per-view constants (host side)."""
    sh_degree: int
    width: int
    height: int
    tile_length: int
    w2m: tuple        # 16 floats, row-major world->model
    proj: tuple       # 16 floats, row-major projection
    cam_pos: tuple    # 3 floats

    @staticmethod
    def from_camera(cam, sh_degree: int, tile_length: int) -> "ViewSpec":
        from .camera import camera_constants
        w2m, proj, pos = camera_constants(cam)
        return ViewSpec(int(sh_degree), int(cam.width), int(cam.height), int(tile_length),
                        tuple(float(x) for x in w2m), tuple(float(x) for x in proj),
                        tuple(float(x) for x in pos))


class SingularCovarianceError(RuntimeError):
    """A Gaussian that lands in a tile has a singular 2-D covariance.

    The reference raises ``torch.linalg.LinAlgError`` from ``torch.linalg.inv``
    (``Utils/Probability.py:15``) in this situation."""


class StageTimer:
    """Optional per-stage device timing: HIP events recorded on the stream the kernels run on.

    ``with StageTimer() as t: ...render...`` then ``t.summary()`` -> {stage: (calls, total_ms)}.
    ``only``: names of the entry points to time (default: all); an event pair costs a few
    microseconds of device time, which adds up over a hundred calls per step.
    """
    active = None

    def __init__(self, only=None):
        self.events = []
        self.only = None if only is None else set(only)

    def __enter__(self):
        StageTimer.active = self
        return self

    def __exit__(self, *exc):
        StageTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.events:
            n, ms = out.get(name, (0, 0.0))
            out[name] = (n + 1, ms + a.elapsed_time(b))
        return out


def _call(name, fn, *args):
    """Invoke one C-ABI entry point; raise on a non-zero code; time it when a StageTimer is active."""
    timer = StageTimer.active
    if timer is None or (timer.only is not None and name not in timer.only):
        _lib.check(fn(*args), name)
        return
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(fn(*args), name)
    b.record()
    timer.events.append((name, a, b))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _check_param(name, t, shape_tail, n=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a tensor on the GPU (there is no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if tuple(t.shape[1:]) != tuple(shape_tail) or (n is not None and t.shape[0] != n):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected [N,{','.join(map(str, shape_tail))}]")
    return t if t.is_contiguous() else t.contiguous()


class RenderState:
    """Everything one forward leaves behind for inspection and for the backward."""
    __slots__ = ("view", "cview", "n", "num_visible", "num_pairs", "mask_u8", "pos2d", "cov2d", "radius_full",
                 "bbox", "ntiles", "rec", "order", "slot_base", "slot_seq", "tile_offsets", "tile_order", "pair_slot", "sorted_gauss",
                 "stream_rec", "final_T", "img", "pairgrad", "blocks", "_counters", "_host", "_ready", "_slot_rest")

    def __init__(self):
        for s in self.__slots__:
            setattr(self, s, None)


_copy_streams = {}
_pinned_pool = []


def _copy_stream(dev):
    key = (dev.type, dev.index)
    if key not in _copy_streams:
        _copy_streams[key] = torch.cuda.Stream(device=dev)
    return _copy_streams[key]


def refuse_high_priority_stream(dev):
    """Raises when a HIGH-priority stream is current.  Every render hands work to side streams (the counter read-back;
    in a view minibatch the list building and the record sums) and waits for them.  With the caller's stream in the
    high-priority class those waits stalled for good: with side streams of normal priority in the one run that tried it
    (a stream of the high class waiting for queues of the lower class is a priority inversion the queue scheduler need not
    resolve), and with side streams created in the high class as well in one run out of three.  HIP gives no progress
    guarantee to lean on here, so the case is refused instead of left to hang."""
    if torch.cuda.current_stream(dev).priority < 0:
        raise RuntimeError("a high-priority stream is current: the render path synchronises with side streams of normal "
                           "priority and does not run on a high-priority stream (use a stream of default priority)")


def forward_begin(coords, scale_raw, rot_raw, opac_raw, sh, screen_coords, view: ViewSpec, projection: bool = True) -> RenderState:
    """Stage 1 and the depth order of one view, plus an asynchronous read-back of the three counters
    (visible Gaussians, pairs, error) on a side stream.  Nothing here waits for the GPU, so a caller
    with several views can enqueue this part for all of them before it calls ``forward_finish`` on
    the first: the counters of the later views are then long on the host when they are needed and the
    GPU never idles on a round trip.  ``projection=False`` (only without ``screen_coords``): ``pos2d`` and ``cov2d``,
    which nothing after stage 1 reads, are not written (28 bytes per Gaussian less, through the kernel that serves the
    view minibatches); everything else is the same bit for bit."""
    lib = _lib.load()
    dev = coords.device
    n = coords.shape[0]
    coords = _check_param("coords", coords, (3,))
    refuse_high_priority_stream(dev)
    scale_raw = _check_param("scale", scale_raw, (3,), n)
    rot_raw = _check_param("rotation", rot_raw, (4,), n)
    if opac_raw.dim() == 1:
        opac_raw = opac_raw.unsqueeze(1)
    opac_raw = _check_param("opacity", opac_raw, (1,), n)
    sh = _check_param("sh", sh, (3, 16), n)
    if screen_coords is not None:
        screen_coords = _check_param("screen_coords", screen_coords, (3,), n)
    st = RenderState()
    st.view, st.n = view, n
    cv = _lib.make_view(n, view.sh_degree, view.width, view.height, view.tile_length, view.w2m, view.proj, view.cam_pos)
    st.cview = cv
    stream = _stream(dev)
    i32 = dict(dtype=torch.int32, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    st.mask_u8 = torch.empty(n, dtype=torch.uint8, device=dev)
    depth_key = torch.empty(n, **i32)
    lean = not projection and screen_coords is None
    st.pos2d = None if lean else torch.empty(n, 3, **f32)
    st.cov2d = None if lean else torch.empty(n, 4, **f32)
    st.radius_full = torch.empty(n, **f32)
    st.bbox = torch.empty(n, 4, **i32)
    st.ntiles = torch.empty(n, **i32)
    st.rec = torch.empty(n, _lib.REC_FLOATS, **f32)
    counters = torch.zeros(8, **i32)
    if lean:
        _call("gsr_preprocess_fwd_views", lib.gsr_preprocess_fwd_views, (_lib.GsrView * 1)(cv), 1, _ptr(coords), _ptr(scale_raw),
              _ptr(rot_raw), _ptr(opac_raw), _ptr(sh), _ptr(st.mask_u8), _ptr(depth_key), None, None, _ptr(st.radius_full),
              _ptr(st.bbox), _ptr(st.ntiles), _ptr(st.rec), _ptr(counters), stream)
    else:
        _call("gsr_preprocess_fwd", lib.gsr_preprocess_fwd, cv, _ptr(coords), _ptr(scale_raw), _ptr(rot_raw), _ptr(opac_raw), _ptr(sh),
              _ptr(screen_coords), _ptr(st.mask_u8), _ptr(depth_key), _ptr(st.pos2d),
              _ptr(st.cov2d), _ptr(st.radius_full), _ptr(st.bbox), _ptr(st.ntiles),
              _ptr(st.rec), _ptr(counters), stream)
    # Read the counters back on a side stream, right behind stage 1: the number of visible Gaussians, the error word
    # and the number of pairs (stage 1 sums the tile counts) are final there, so the copy, the host's wait for it and
    # the sizing of the per-pair buffers run beside the depth sort that is enqueued next.
    main = torch.cuda.current_stream(dev)
    side = _copy_stream(dev)
    done = torch.cuda.Event()
    done.record(main)
    st._counters = counters
    st._host = _pinned_pool.pop() if _pinned_pool else torch.empty(8, dtype=torch.int32, pin_memory=True)
    st._ready = torch.cuda.Event()
    with torch.cuda.stream(side):
        side.wait_event(done)
        st._host.copy_(counters, non_blocking=True)
        st._ready.record(side)
    st.order = torch.empty(n, **i32)
    st.slot_base = torch.empty(n, **i32)
    st.slot_seq = torch.empty(n + 1, **i32)
    ws_bytes = int(lib.gsr_bin_offsets_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _call("gsr_bin_offsets", lib.gsr_bin_offsets, _ptr(depth_key), _ptr(st.ntiles), n, _ptr(st.order), _ptr(st.slot_base),
                                   _ptr(st.slot_seq), _ptr(counters), _ptr(ws), ws_bytes, stream)
    return st


def forward_begin_views(coords, scale_raw, rot_raw, opac_raw, sh, views, projection: bool = False,
                        defer_slot_base: bool = False) -> list:
    """``forward_begin`` for several views of the same Gaussians (at most ``_lib.MAX_VIEWS`` per call; more are
    taken in groups): ONE pass over the parameters for stage 1 of all of them, ONE set of launches for their depth
    orders and ONE read-back of all counters.  Returns one ``RenderState`` per view, the same - bit for bit - as
    ``forward_begin`` leaves (their arrays are slices of arrays that hold all views), except that ``pos2d`` and
    ``cov2d`` - which nothing after stage 1 reads - are only written, and kept, with ``projection=True``.
    ``defer_slot_base``: the Gaussian-indexed ``slot_base``, which only the backward's record sums read, is not written
    here but by ``forward_slot_bases(states)`` - called by a caller with several streams beside the first view's lists,
    or by the first ``record_sums`` that needs it."""
    lib = _lib.load()
    views = list(views)
    if len(views) > _lib.MAX_VIEWS:
        out = []
        for k in range(0, len(views), _lib.MAX_VIEWS):
            out += forward_begin_views(coords, scale_raw, rot_raw, opac_raw, sh, views[k:k + _lib.MAX_VIEWS], projection,
                                       defer_slot_base)
        return out
    if len(views) == 1:
        return [forward_begin(coords, scale_raw, rot_raw, opac_raw, sh, None, views[0])]
    dev = coords.device
    n = coords.shape[0]
    nv = len(views)
    coords = _check_param("coords", coords, (3,))
    refuse_high_priority_stream(dev)
    scale_raw = _check_param("scale", scale_raw, (3,), n)
    rot_raw = _check_param("rotation", rot_raw, (4,), n)
    if opac_raw.dim() == 1:
        opac_raw = opac_raw.unsqueeze(1)
    opac_raw = _check_param("opacity", opac_raw, (1,), n)
    sh = _check_param("sh", sh, (3, 16), n)
    stream = _stream(dev)
    i32 = dict(dtype=torch.int32, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    cviews = (_lib.GsrView * nv)(*[_lib.make_view(n, v.sh_degree, v.width, v.height, v.tile_length, v.w2m, v.proj, v.cam_pos)
                                   for v in views])
    mask = torch.empty(nv, n, dtype=torch.uint8, device=dev)
    depth_key = torch.empty(nv, n, **i32)
    pos2d = torch.empty(nv, n, 3, **f32) if projection else None
    cov2d = torch.empty(nv, n, 4, **f32) if projection else None
    radius = torch.empty(nv, n, **f32)
    bbox = torch.empty(nv, n, 4, **i32)
    ntiles = torch.empty(nv, n, **i32)
    rec = torch.empty(nv, n, _lib.REC_FLOATS, **f32)
    counters = torch.zeros(nv, 8, **i32)
    _call("gsr_preprocess_fwd_views", lib.gsr_preprocess_fwd_views, cviews, nv, _ptr(coords), _ptr(scale_raw), _ptr(rot_raw),
          _ptr(opac_raw), _ptr(sh), _ptr(mask), _ptr(depth_key), _ptr(pos2d), _ptr(cov2d), _ptr(radius), _ptr(bbox),
          _ptr(ntiles), _ptr(rec), _ptr(counters), stream)
    main = torch.cuda.current_stream(dev)
    side = _copy_stream(dev)
    done = torch.cuda.Event()
    done.record(main)
    host = torch.empty(nv, 8, dtype=torch.int32, pin_memory=True)
    ready = torch.cuda.Event()
    with torch.cuda.stream(side):        # (the counters are final after stage 1: see forward_begin)
        side.wait_event(done)
        host.copy_(counters, non_blocking=True)
        ready.record(side)
    order = torch.empty(nv, n, **i32)
    slot_base = torch.empty(nv, n, **i32)
    slot_seq = torch.empty(nv, n + 1, **i32)
    ws_bytes = int(lib.gsr_bin_offsets_views_workspace_bytes(n, nv))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _call("gsr_bin_offsets_views", lib.gsr_bin_offsets_views, _ptr(depth_key), _ptr(ntiles), n, nv, _ptr(order),
          None if defer_slot_base else _ptr(slot_base), _ptr(slot_seq), _ptr(counters), _ptr(ws), ws_bytes, stream)
    states = []
    for k in range(nv):
        st = RenderState()
        st.view, st.n, st.cview = views[k], n, cviews[k]
        st.mask_u8, st.radius_full = mask[k], radius[k]
        st.pos2d, st.cov2d = (pos2d[k], cov2d[k]) if projection else (None, None)
        st.bbox, st.ntiles, st.rec = bbox[k], ntiles[k], rec[k]
        st.order, st.slot_base, st.slot_seq = order[k], slot_base[k], slot_seq[k]
        st._counters, st._host, st._ready = counters, host[k], ready
        states.append(st)
    if defer_slot_base:
        rest = _SlotRest(order, slot_seq, slot_base, n, nv)
        for st in states:
            st._slot_rest = rest
    return states


class _SlotRest:
    """The deferred ``gsr_slot_bases`` call of one group of views: the whole-group arrays it takes.  Every state of the
    group refers to this holder and the holder to no state, so dropping the states frees the arrays at once (no
    reference cycle for the collector to find)."""
    __slots__ = ("args",)

    def __init__(self, *args):
        self.args = args


def forward_slot_bases(states):
    """Enqueues, on the current stream, the ``slot_base`` arrays that ``forward_begin_views(..., defer_slot_base=True)``
    left out (once per group of views; later calls do nothing).  Returns whether anything was enqueued."""
    enqueued = False
    for st in states:
        rest, st._slot_rest = st._slot_rest, None
        if rest is not None and rest.args is not None:
            (order, slot_seq, slot_base, n, nv), rest.args = rest.args, None
            _call("gsr_slot_bases", _lib.load().gsr_slot_bases, _ptr(order), _ptr(slot_seq), n, nv, _ptr(slot_base),
                  _stream(order.device))
            enqueued = True
    return enqueued


def forward_lists(st: RenderState) -> RenderState:
    """Waits for the view's counters (the one synchronisation of a render), sizes the per-pair
    buffers and builds the per-tile lists, the record stream and the tile order."""
    if st._ready is not None:
        read_counters(st)
    return _lists_of(st)


def _lists_of(st: RenderState) -> RenderState:
    lib = _lib.load()
    view, cv = st.view, st.cview
    dev = st.mask_u8.device
    stream = _stream(dev)
    i32 = dict(dtype=torch.int32, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    if st.num_visible == 0:
        return st
    T = ((view.width + view.tile_length - 1) // view.tile_length) * ((view.height + view.tile_length - 1) // view.tile_length)
    I = st.num_pairs
    st.tile_offsets = torch.empty(T + 1, **i32)
    st.sorted_gauss = torch.empty(max(I, 1), **i32)
    st.stream_rec = torch.empty(max(I, 1), _lib.STREAM_DWORDS, **f32)
    ws_bytes = int(lib.gsr_bin_lists_workspace_bytes(I))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _call("gsr_bin_lists", lib.gsr_bin_lists, cv, _ptr(st.order), _ptr(st.slot_base), _ptr(st.slot_seq), _ptr(st.bbox), _ptr(st.ntiles), _ptr(st.rec), I,
                                 _ptr(st.tile_offsets), None, _ptr(st.sorted_gauss),
                                 _ptr(st.stream_rec), _ptr(ws), ws_bytes, stream)
    st.tile_order = None
    if _block_lists_of(st).blocks is not None:
        return st                                                # (the compositing kernels take the blocks' order)
    if os.environ.get("GSR_TILE_ORDER", "length") != "id":      # "id": tile-id order (for comparisons)
        st.tile_order = torch.empty(T, **i32)
        _call("gsr_tile_order", lib.gsr_tile_order, cv, _ptr(st.tile_offsets), _ptr(st.tile_order), stream)
    return st


class BlockLists:
    """The per-block streams of a view whose tiles are 2x2, 3x3 or 4x4 blocks (``gsr_block_lists``): what the compositing
    kernels of one-block tiles walk instead of the tile lists."""
    __slots__ = ("offsets", "order", "stream", "combo", "cview16")

    def __init__(self, offsets, order, stream, combo, cview16):
        self.offsets, self.order, self.stream, self.combo, self.cview16 = offsets, order, stream, combo, cview16


def block_lists_enabled() -> bool:
    """``GSR_BLOCK_LISTS=0`` keeps tiles of several blocks on the tile-group kernels (read at every call: tests compare
    the two paths inside one process).  By default every tile edge the library supports (32, 48, 64) takes the per-block
    lists: 1.1, 1.6 and 1.9 times the rate of the tile groups at 1 M Gaussians, 1080p."""
    return os.environ.get("GSR_BLOCK_LISTS", "1") != "0"


def _block_lists_of(st: RenderState) -> RenderState:
    """Builds ``st.blocks`` when the view qualifies (tile edge 32, 48 or 64, few enough pairs); else leaves it None."""
    st.blocks = None
    if not st.num_visible or not st.num_pairs or not block_lists_enabled():
        return st
    lib = _lib.load()
    cv, I = st.cview, st.num_pairs
    if not lib.gsr_block_lists_supported(cv, I):
        return st
    dev = st.mask_u8.device
    stream = _stream(dev)
    S2 = (st.view.tile_length // 16) ** 2
    T16 = int(lib.gsr_block_count(cv))
    i32 = dict(dtype=torch.int32, device=dev)
    offsets = torch.empty(T16 + 1, **i32)
    block_stream = torch.empty(I * S2, _lib.STREAM_DWORDS, dtype=torch.float32, device=dev)   # (the upper bound)
    combo = torch.empty(I * S2, dtype=torch.int64 if int(lib.gsr_block_combo_bytes(cv, I)) == 8 else torch.int32, device=dev)
    ws_bytes = int(lib.gsr_block_lists_workspace_bytes(cv, I))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _call("gsr_block_lists", lib.gsr_block_lists, cv, _ptr(st.tile_offsets), _ptr(st.stream_rec), I, _ptr(offsets), _ptr(block_stream),
          _ptr(combo), _ptr(ws), ws_bytes, stream)
    cv16 = _lib.GsrView.from_buffer_copy(cv)
    cv16.tile_length = 16
    order = None
    if os.environ.get("GSR_TILE_ORDER", "length") != "id":
        order = torch.empty(T16, **i32)
        _call("gsr_tile_order", lib.gsr_tile_order, cv16, _ptr(offsets), _ptr(order), stream)
    st.blocks = BlockLists(offsets, order, block_stream, combo, cv16)
    return st


def read_counters(st: RenderState):
    """Waits for the view's counters and checks the error word (the one synchronisation of a render); does nothing
    when they were read before."""
    if st._ready is None:
        return
    st._ready.synchronize()
    host = st._host
    words = host.tolist()                  # (one conversion: indexing the pinned tensor word by word costs 3 us each)
    # gsr_counters: num_visible, num_pairs (written later, by the scan), error, -, the 64-bit pair total of stage 1
    st.num_visible, err = words[0], words[2]
    pairs = (words[4] & 0xFFFFFFFF) | ((words[5] & 0xFFFFFFFF) << 32)
    st.num_pairs = pairs & 0xFFFFFFFF
    if pairs >> 32:
        err |= 2
    if host._base is None:
        _pinned_pool.append(host)          # (a row of the buffer that several views share is not recycled)
    st._host = st._counters = st._ready = None
    if err & 1:
        raise SingularCovarianceError("singular 2-D covariance in a Gaussian that lands in a tile")
    if err & 2:
        raise RuntimeError("the view has more than 2^32 - 1 (tile, Gaussian) pairs: the per-tile lists cannot be indexed")


def forward_lists_views(states) -> list:
    """``forward_lists`` for the states ``forward_begin_views`` returned (views of one group of at most
    ``_lib.MAX_VIEWS``, consecutive slices of the same arrays): pair emission, tile sort, record stream and tile order
    of all of them with one set of launches.  Falls back to ``forward_lists`` view by view when the views do not
    qualify (different sizes, or more tiles, Gaussians and pairs than one 64-bit word per pair holds)."""
    import ctypes as C
    states = list(states)
    if not states:
        return states
    lib = _lib.load()
    for st in states:
        if st._ready is not None:
            read_counters(st)
    first = states[0]
    nv, n = len(states), first.n
    base_t = first.order._base
    batched = (nv > 1 and nv <= _lib.MAX_VIEWS and base_t is not None and os.environ.get("GSR_TILE_ORDER", "length") != "id" and
               all(st.order._base is base_t and st.order.storage_offset() == k * n for k, st in enumerate(states)))
    pairs = (C.c_uint32 * nv)(*[st.num_pairs if st.num_visible else 0 for st in states])
    cviews = (_lib.GsrView * nv)(*[st.cview for st in states])
    if batched:
        batched = bool(lib.gsr_bin_lists_views_supported(cviews, nv, pairs))
    if not batched:
        for st in states:
            if st.tile_offsets is None:
                _lists_of(st)
        return states
    dev = first.mask_u8.device
    stream = _stream(dev)
    i32 = dict(dtype=torch.int32, device=dev)
    view = first.view
    T = ((view.width + view.tile_length - 1) // view.tile_length) * ((view.height + view.tile_length - 1) // view.tile_length)
    bases = (C.c_uint32 * nv)()
    total = int(lib.gsr_bin_lists_views_layout(nv, pairs, bases))
    tile_offsets = torch.empty(nv, T + 1, **i32)
    tile_order = torch.empty(nv, T, **i32)
    sorted_gauss = torch.empty(total, **i32)
    stream_rec = torch.empty(total, _lib.STREAM_DWORDS, dtype=torch.float32, device=dev)
    ws_bytes = int(lib.gsr_bin_lists_views_workspace_bytes(nv, pairs))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

    def whole(name):          # the array that holds all views
        return getattr(first, name)._base

    _call("gsr_bin_lists_views", lib.gsr_bin_lists_views, cviews, nv, _ptr(whole("order")), _ptr(whole("slot_base")),
          _ptr(whole("slot_seq")), _ptr(whole("bbox")), _ptr(whole("ntiles")), _ptr(whole("rec")), pairs, _ptr(tile_offsets),
          _ptr(tile_order), _ptr(sorted_gauss), _ptr(stream_rec), _ptr(ws), ws_bytes, stream)
    for k, st in enumerate(states):
        if not st.num_visible:
            continue
        b, I = int(bases[k]), max(st.num_pairs, 1)
        st.tile_offsets, st.tile_order = tile_offsets[k], tile_order[k]
        st.sorted_gauss, st.stream_rec = sorted_gauss[b:b + I], stream_rec[b:b + I]
        _block_lists_of(st)
    return states


def forward_composite(st: RenderState) -> RenderState:
    """Compositing forward of a view whose lists are built: fills ``st.img`` and ``st.final_T``."""
    if st.num_visible == 0:
        return st
    lib = _lib.load()
    view = st.view
    dev = st.mask_u8.device
    f32 = dict(dtype=torch.float32, device=dev)
    st.img = torch.empty(view.width, view.height, 3, **f32)
    st.final_T = torch.empty(view.width, view.height, **f32)
    if st.blocks is not None:
        b = st.blocks
        _call("gsr_composite_blocks_fwd", lib.gsr_composite_blocks_fwd, st.cview, st.num_pairs, _ptr(b.offsets), _ptr(b.order),
              _ptr(b.stream), _ptr(st.img), _ptr(st.final_T), _stream(dev))
        return st
    _call("gsr_composite_fwd", lib.gsr_composite_fwd, st.cview, _ptr(st.tile_offsets), _ptr(st.tile_order), _ptr(st.stream_rec),
          _ptr(st.img), _ptr(st.final_T), _stream(dev))
    return st


def forward_finish(st: RenderState) -> RenderState:
    """``forward_lists`` + ``forward_composite`` on the current stream."""
    return forward_composite(forward_lists(st))


def forward_pipeline(coords, scale_raw, rot_raw, opac_raw, sh, screen_coords, view: ViewSpec) -> RenderState:
    """Stage 1 -> binning/sort -> compositing of one view.  One host synchronisation (three counters)
    in the middle, needed to size the per-pair buffers."""
    return forward_finish(forward_begin(coords, scale_raw, rot_raw, opac_raw, sh, screen_coords, view))


def _alloc_pairgrad(lib, st: RenderState, dev) -> torch.Tensor:
    """The gradient records of a view: one row per (tile, Gaussian) pair and per workgroup of the pair's tile."""
    rows = max(st.num_pairs, 1) * pairgrad_parts(st)
    return torch.empty(rows, _lib.PAIRGRAD_FLOATS, dtype=torch.float32, device=dev)


def pairgrad_parts(st: RenderState) -> int:
    """Rows of ``pairgrad`` per pair as ``_composite_bwd`` leaves them (1 with per-block lists: their rows are summed per
    pair before the per-Gaussian sums)."""
    return 1 if st.blocks is not None else int(_lib.load().gsr_pairgrad_parts(st.view.tile_length))


def _composite_bwd(lib, st: RenderState, dimg, dev) -> torch.Tensor:
    """Compositing backward on the current stream; returns the gradient records at the pairs' emission slots."""
    stream = _stream(dev)
    pairgrad = _alloc_pairgrad(lib, st, dev)
    if st.blocks is None:
        _call("gsr_composite_bwd", lib.gsr_composite_bwd, st.cview, _ptr(st.tile_offsets), _ptr(st.tile_order), _ptr(st.stream_rec),
              _ptr(st.final_T), _ptr(dimg), st.num_pairs, _ptr(pairgrad), stream)
        return pairgrad
    b, I = st.blocks, st.num_pairs
    S2 = (st.view.tile_length // 16) ** 2
    f32 = dict(dtype=torch.float32, device=dev)
    block_grad = torch.empty(max(I, 1) * S2, _lib.PAIRGRAD_FLOATS, **f32)     # one row per stream record (upper bound)
    _call("gsr_composite_blocks_bwd", lib.gsr_composite_blocks_bwd, st.cview, I, _ptr(b.offsets), _ptr(b.order), _ptr(b.stream),
          _ptr(st.final_T), _ptr(dimg), _ptr(block_grad), stream)
    _call("gsr_block_rows_reduce", lib.gsr_block_rows_reduce, st.cview, I, _ptr(st.stream_rec), _ptr(b.combo), _ptr(block_grad),
          _ptr(pairgrad), stream)
    return pairgrad


def backward_pipeline(st: RenderState, coords, scale_raw, rot_raw, opac_raw, sh, dimg, accumulate_into=None,
                      overwrite=False):
    """Compositing backward + stage-1 backward.  Returns the six dense gradients
    (coords, scale, rotation, opacity[N,1], sh, screen_coords).  ``accumulate_into`` may hold six
    existing buffers to add to (view minibatches); with ``overwrite`` they are written instead (the
    first view of a minibatch).  The screen_coords gradient is always overwritten."""
    lib = _lib.load()
    dev = coords.device
    n = st.n
    stream = _stream(dev)
    f32 = dict(dtype=torch.float32, device=dev)
    dimg = dimg.to(torch.float32).contiguous()
    I = st.num_pairs
    if st._slot_rest is not None:
        forward_slot_bases([st])
    pairgrad = _composite_bwd(lib, st, dimg, dev)
    st.pairgrad = pairgrad
    if st.blocks is not None:
        # (one row per pair: the one-view convenience call below would read gsr_pairgrad_parts(tile edge) rows)
        gsum = record_sums(st, pairgrad)
        five = None if accumulate_into is None else tuple(accumulate_into[:5])
        grads5, g_screens = backward_views([st], [gsum], coords, scale_raw, rot_raw, opac_raw, sh, grads=five,
                                           accumulate=accumulate_into is not None and not overwrite)
        if accumulate_into is None:
            return tuple(grads5) + (g_screens[0],)
        accumulate_into[5].copy_(g_screens[0])
        return accumulate_into
    if accumulate_into is None:
        grads = (torch.empty(n, 3, **f32), torch.empty(n, 3, **f32), torch.empty(n, 4, **f32),
                 torch.empty(n, 1, **f32), torch.empty(n, 3, 16, **f32), torch.empty(n, 3, **f32))
        acc = 0
    else:
        grads, acc = accumulate_into, (0 if overwrite else 1)
    if opac_raw.dim() == 1:
        opac_raw = opac_raw.unsqueeze(1)
    ws_bytes = int(lib.gsr_preprocess_bwd_workspace_bytes(n))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _call("gsr_preprocess_bwd", lib.gsr_preprocess_bwd, st.cview, _ptr(coords.contiguous()), _ptr(scale_raw.contiguous()),
                                      _ptr(rot_raw.contiguous()), _ptr(opac_raw.contiguous()), _ptr(sh.contiguous()),
                                      _ptr(st.mask_u8), _ptr(st.slot_base), _ptr(st.ntiles),
                                      _ptr(pairgrad), *[_ptr(g) for g in grads], acc, _ptr(ws), ws_bytes, stream)
    return grads


def backward_composite(st: RenderState, dimg) -> torch.Tensor:
    """Compositing backward of one view on the current stream.  Returns the gradient records of its pairs, for
    ``record_sums``."""
    lib = _lib.load()
    dev = st.mask_u8.device
    dimg = dimg.to(torch.float32).contiguous()
    return _composite_bwd(lib, st, dimg, dev)


def record_sums(st: RenderState, pairgrad: torch.Tensor, gsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The per-Gaussian sums of a view's gradient records, on the current stream: gsum [N,12]."""
    lib = _lib.load()
    if st._slot_rest is not None:
        forward_slot_bases([st])
    dev = st.mask_u8.device
    if gsum is None:
        gsum = torch.empty(st.n, 12, dtype=torch.float32, device=dev)
    _call("gsr_pair_reduce", lib.gsr_pair_reduce, st.n, pairgrad_parts(st), _ptr(st.mask_u8),
          _ptr(st.slot_base), _ptr(st.ntiles), _ptr(pairgrad), _ptr(gsum), _stream(dev))
    return gsum


def backward_records(st: RenderState, dimg) -> torch.Tensor:
    """First half of the backward of one view: compositing backward, then the per-Gaussian sums of
    its gradient records.  Returns gsum [N,12]; pass it with the state to ``backward_views``."""
    return record_sums(st, backward_composite(st, dimg))


def backward_views(states, gsums, coords, scale_raw, rot_raw, opac_raw, sh, grads=None, accumulate=False, g_colors=None,
                   skip_sh: bool = False):
    """Second half for a whole view minibatch: chains the per-Gaussian sums of all views to the raw
    parameters and writes (or, with ``accumulate``, adds to) the five dense gradients ONCE.
    ``states``/``gsums``: the views that had something visible.  Returns (grads, [g_screen per view]).

    ``g_colors`` (one ``[N,3]`` tensor per state): the sh gradient is NOT written; instead every view's
    colour gradient is stored there, for ranks that rebuild the sh gradient of the whole minibatch
    with ``sh_grad_views`` after gathering them (``minibatch.reduce_gradients``).  ``skip_sh``: the sh gradient is not
    written and nothing takes its place (the caller has the colour gradients already: they are columns 6..8 of the
    record sums)."""
    import ctypes as C
    lib = _lib.load()
    dev = coords.device
    n = coords.shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    if grads is None:
        grads = (torch.empty(n, 3, **f32), torch.empty(n, 3, **f32), torch.empty(n, 4, **f32),
                 torch.empty(n, 1, **f32), torch.empty(n, 3, 16, **f32))
    nv = len(states)
    if nv == 0:
        if not accumulate:
            for g in (grads[:4] if g_colors is not None or skip_sh else grads):
                g.zero_()
        return grads, []
    g_screens = [torch.empty(n, 3, **f32) for _ in range(nv)]
    views = (_lib.GsrView * nv)(*[st.cview for st in states])
    vp = C.c_void_p
    masks = (vp * nv)(*[_ptr(st.mask_u8) for st in states])
    sums = (vp * nv)(*[_ptr(g) for g in gsums])
    screens = (vp * nv)(*[_ptr(g) for g in g_screens])
    colors = None
    if g_colors is not None:
        if len(g_colors) != nv:
            raise ValueError(f"{len(g_colors)} colour-gradient buffers for {nv} views")
        for g in g_colors:
            if tuple(g.shape) != (n, 3) or g.dtype != torch.float32 or not g.is_contiguous() or g.device != dev:
                raise ValueError("a colour-gradient buffer must be a contiguous float32 [N,3] tensor on the parameters' device")
        colors = (vp * nv)(*[_ptr(g) for g in g_colors])
    if opac_raw.dim() == 1:
        opac_raw = opac_raw.unsqueeze(1)
    _call("gsr_preprocess_bwd_views", lib.gsr_preprocess_bwd_views, views, nv, _ptr(coords.contiguous()),
          _ptr(scale_raw.contiguous()), _ptr(rot_raw.contiguous()), _ptr(opac_raw.contiguous()), _ptr(sh.contiguous()),
          masks, sums, *[_ptr(g) for g in grads[:4]], None if g_colors is not None or skip_sh else _ptr(grads[4]), screens, colors,
          1 if accumulate else 0, _stream(dev))
    return grads, g_screens


def sh_grad_views(views, coords, sh, g_colors, g_sh, accumulate=False):
    """``g_sh[N,3,16]`` of a view minibatch from the colour gradients of ALL its views
    (``g_colors[v]``: ``[N,3]``, in the order of ``views``, a sequence of ``ViewSpec``)."""
    import ctypes as C
    lib = _lib.load()
    n = coords.shape[0]
    nv = len(views)
    if len(g_colors) != nv:
        raise ValueError(f"{len(g_colors)} colour gradients for {nv} views")
    cviews = (_lib.GsrView * nv)(*[_lib.make_view(n, v.sh_degree, v.width, v.height, v.tile_length, v.w2m, v.proj, v.cam_pos)
                                   for v in views])
    cols = (C.c_void_p * nv)(*[_ptr(g) for g in g_colors])
    _call("gsr_sh_grad_views", lib.gsr_sh_grad_views, cviews, nv, _ptr(coords.contiguous()), _ptr(sh.contiguous()), cols,
          _ptr(g_sh), 1 if accumulate else 0, _stream(coords.device))
    return g_sh


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, scale_raw, rot_raw, opac_raw, sh, screen_coords, view):
        st = forward_pipeline(coords.detach(), scale_raw.detach(), rot_raw.detach(), opac_raw.detach(), sh.detach(),
                              None if screen_coords is None else screen_coords.detach(), view)
        ctx.state = st
        mask = st.mask_u8.bool()
        if st.num_visible == 0:
            ctx.mark_non_differentiable(mask)
            return coords.new_zeros(0), mask, coords.new_zeros(0)
        ctx.save_for_backward(coords, scale_raw, rot_raw, opac_raw, sh)
        radius = st.radius_full[mask]          # length M, like get_radius on the culled set
        ctx.mark_non_differentiable(mask, radius)
        return st.img, mask, radius

    @staticmethod
    def backward(ctx, dimg, _dmask, _dradius):
        st = ctx.state
        if st.num_visible == 0:
            return (None,) * 7
        coords, scale_raw, rot_raw, opac_raw, sh = ctx.saved_tensors
        g = backward_pipeline(st, coords, scale_raw, rot_raw, opac_raw, sh, dimg)
        g_opac = g[3] if opac_raw.dim() == 2 else g[3].squeeze(1)
        need = ctx.needs_input_grad
        out = [g[0], g[1], g[2], g_opac, g[4], g[5]]
        return tuple(o if need[i] else None for i, o in enumerate(out)) + (None,)


def rasterize(coords, scale_raw, rot_raw, opac_raw, sh, screen_coords, view: ViewSpec):
    """-> (img [W,H,3], mask [N] bool, radius [M]) or (None, mask, None) when nothing is visible."""
    img, mask, radius = _Rasterize.apply(coords, scale_raw, rot_raw, opac_raw, sh, screen_coords, view)
    if img.numel() == 0:
        return None, mask, None
    return img, mask, radius
