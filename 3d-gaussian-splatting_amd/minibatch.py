"""View-minibatch data parallelism: the views of one optimisation step are independent renders, so
they are split over the ranks; every rank accumulates the parameter gradients of its views in one
flat fp32 bucket and the ranks exchange them (RCCL over xGMI on the GPUs, gloo in the CPU tests):
``reduce_gradients`` sums the 11 small floats per Gaussian and rebuilds the sh gradient from gathered
colour gradients, ``GradBucket.all_reduce`` sums the whole bucket, ``reduce_densify_stats`` sums the
densification statistics of the step's views.  The reference renders one view per step on one device
(``Trainer.py:54-57``); this module is what carries that step to N GPUs (``trainer.MinibatchTrainer``
is the training loop on top of it).
"""
from typing import Callable, List, Optional, Sequence, Tuple

import os

import torch

# floats per Gaussian in the bucket: coords 3, scale 3, rotation 4, opacity 1, sh 3*16
FIELD_SIZES = (3, 3, 4, 1, 48)
FLOATS_PER_GAUSSIAN = sum(FIELD_SIZES)


def shard_views(n_views: int, world_size: int, rank: int) -> List[int]:
    """Contiguous block of views for ``rank``; the first ``n_views % world_size`` ranks get one more."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside 0..{world_size - 1}")
    base, extra = divmod(n_views, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


class GradBucket:
    """One contiguous fp32 buffer holding the five dense parameter gradients of N Gaussians."""

    def __init__(self, n: int, device):
        self.n = n
        self.flat = torch.zeros(n * FLOATS_PER_GAUSSIAN, dtype=torch.float32, device=device)
        self.views: Tuple[torch.Tensor, ...] = self._make_views()

    def _make_views(self):
        out, off = [], 0
        for size in FIELD_SIZES:
            chunk = self.flat[off:off + self.n * size]
            out.append(chunk.view(self.n, 3, 16) if size == 48 else chunk.view(self.n, size))
            off += self.n * size
        return tuple(out)

    @property
    def coords(self): return self.views[0]

    @property
    def scale(self): return self.views[1]

    @property
    def rotation(self): return self.views[2]

    @property
    def opacity(self): return self.views[3]

    @property
    def sh(self): return self.views[4]

    def add_(self, grads: Sequence[torch.Tensor]):
        """Accumulate five gradient tensors (host-side path; the HIP backward adds in place)."""
        for dst, g in zip(self.views, grads):
            dst.add_(g.reshape(dst.shape))

    def all_reduce(self, group=None):
        """Sum the bucket over the ranks: one collective for all five tensors."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        return self


_side_streams = {}


def _side_stream(role: str, dev):
    """A side stream of the pipelined minibatch (``role``: "front" or "sums"), one per device."""
    key = (role, dev.type, dev.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


def _front_stream(dev):
    return _side_stream("front", dev)


def _sums_stream(dev):
    return _side_stream("sums", dev)


def _front_follows(front, cur):
    """Orders whatever is enqueued on the side stream from now on after everything enqueued on ``cur`` so far."""
    if front is not cur:
        done = torch.cuda.Event()
        done.record(cur)
        front.wait_event(done)


def reserve_streams(device=None):
    """Creates the three side streams of the pipelined minibatch (front part of the next views, record sums, counter read-back)
    and runs a trivial operation on each, which binds them to hardware queues of their own.

    Call this BEFORE ``torch.distributed.init_process_group("nccl", ...)``.  A process has few hardware queues
    (``GPU_MAX_HW_QUEUES``, 4 by default) and HIP hands them to streams in the order the streams are first used;
    RCCL takes its share when the process group is initialised.  Side streams first used after that end up on the
    queue of the stream they are meant to run beside, and everything ``render_views`` enqueues is then executed one
    kernel after the other: measured on the headline workload, 15.1 ms per step instead of 13.5 with nothing else
    changed."""
    from . import rasterizer as R
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    for stream in (_front_stream(dev), _sums_stream(dev), R._copy_stream(dev)):
        with torch.cuda.stream(stream):
            torch.zeros(1, device=dev).add_(1.0)
    torch.cuda.synchronize(dev)


def render_views(params: Sequence[torch.Tensor], views: Sequence, cotangent: Optional[Callable], grads=None,
                 accumulate: bool = False, overlap: bool = True, lookahead: int = 2, g_colors=None, batch_front: bool = True,
                 colors_ready: Optional[Callable] = None):
    """Forward + backward of the views of one rank against the same parameters.

    Stage 1 and the depth order of ALL views are one set of launches (``batch_front``: the parameters are read
    once, the sort kernels take the views as a batch).  The rest of a view's front part (pair emission, tile sort,
    record stream: bound by memory bandwidth) runs on a side stream beside the compositing of the view before it
    (forward, backward: bound by vector instruction issue); the record sums of a view run on a third stream beside the
    forward of the next one.  Without ``batch_front`` the whole front
    part of view i+1.. runs on the side stream, ``lookahead`` views begun before the first one's counters are
    waited for, so that the host never stalls the GPU.

    ``params``: coords, scale_raw, rot_raw, opac_raw, sh.  ``views``: ``ViewSpec``s.
    ``cotangent(i, state)`` returns dL/dimg of view ``i`` (called with the current stream current, after
    ``state.img`` is enqueued); ``None`` renders forward only.  The five dense gradients are written to
    (``accumulate``: added to) ``grads`` by ONE pass over all views.  ``g_colors`` (a ``[len(views),N,3]``
    tensor): the sh gradient is left out and the colour gradient of every view is stored there instead,
    for ``reduce_gradients`` - the colour gradient of a view is three columns of its record sums, copied as soon as
    the sums exist, and ``colors_ready()`` (if given) is called with all of them enqueued and ordered before the current
    stream's next work, BEFORE the pass that chains the views to the parameters: a caller that gathers them from other
    ranks starts its collective there, beside that pass.  ``batch_front``: stage 1 and the depth order of all views at once
    (``rasterizer.forward_begin_views``) instead of view by view; the results are bit-identical.
    Returns ``(states, grads, g_screens)`` - ``states`` holds every view (``num_visible == 0`` ones have
    no image), ``g_screens`` one tensor per view that had something visible."""
    from . import rasterizer as R
    dev = params[0].device
    R.refuse_high_priority_stream(dev)
    cur = torch.cuda.current_stream(dev)
    front = _front_stream(dev) if overlap and len(views) > 1 else cur
    if front is not cur:
        start = torch.cuda.Event()
        start.record(cur)
        front.wait_event(start)          # the parameters may have been written on the current stream
    # The buffers of a view are allocated while the side stream is current and read by kernels on the current stream.
    # They are NOT marked as in use there (Tensor.record_stream): the caching allocator would record an event per
    # buffer when the states are dropped - over a hundred markers queued behind the step's last kernel, 0.2 ms in
    # which the GPU ran nothing.  Instead the side stream is made to wait, below, for everything this call enqueues
    # on the current stream: memory that the allocator hands out again for the side stream is then only touched after
    # the kernels that read it here have finished.
    begun: List = []

    def begin(i):
        with torch.cuda.stream(front):
            begun.append(R.forward_begin(*params, None, views[i], projection=False))      # (nothing here reads pos2d / cov2d)

    lists_ready = None
    batch_lists = False
    slots_written = None
    if batch_front and len(views) > 1:
        # stage 1 and the depth orders of ALL views with one set of launches (the parameters are read once, the
        # small sort kernels fill the chip, one counter read-back)
        from . import _lib
        # The Gaussian-indexed slot bases (4-byte writes at random) are only read by the record sums of the backward:
        # with streams to spare they are written on the sums stream, beside the lists of view 0, instead of in front of
        # them; a forward-only call never writes them.
        sums_apart = front is not cur
        with torch.cuda.stream(front):
            begun.extend(R.forward_begin_views(*params, views, defer_slot_base=sums_apart))
            if sums_apart and cotangent is not None:
                ordered = torch.cuda.Event()
                ordered.record(front)
                with torch.cuda.stream(_sums_stream(dev)):
                    _sums_stream(dev).wait_event(ordered)
                    if R.forward_slot_bases(begun):
                        slots_written = torch.cuda.Event()
                        slots_written.record(_sums_stream(dev))
            # The lists of all views with one set of launches as well - when the record streams of all views together
            # stay inside the GPU's last-level cache.  Larger ones are better built view by view, right before the
            # view is composited: the forward then reads most of its stream from that cache (measured: one set of
            # launches is +2 % at 100 k Gaussians / 640x480 and -5 % at 1 M / 1080p).
            for st in begun:
                R.read_counters(st)
            stream_bytes = sum(st.num_pairs for st in begun if st.num_visible) * 4 * _lib.STREAM_DWORDS
            batch_lists = stream_bytes <= BATCHED_LISTS_MAX_STREAM_BYTES
            if batch_lists:
                for k in range(0, len(begun), _lib.MAX_VIEWS):
                    R.forward_lists_views(begun[k:k + _lib.MAX_VIEWS])
                if front is not cur:
                    lists_ready = torch.cuda.Event()
                    lists_ready.record(front)
        lookahead = len(views)
    # (beginning view 0 alone first, so that its lists are not queued behind the stage 1 of view 1, was
    # measured no faster: the host then waits for the counters before it has queued anything else)
    for i in range(len(begun), min(max(lookahead, 1), len(views))):
        begin(i)
    states, live, gsums = [], [], []
    sums = _sums_stream(dev) if front is not cur and cotangent is not None else cur
    pending = None
    readies = {}

    def lists_of(i):
        """Enqueues the lists of view i on the front stream (once) and returns the event behind them."""
        if i not in readies:
            ready = None
            with torch.cuda.stream(front):
                R.forward_lists(begun[i])
                if front is not cur:
                    ready = torch.cuda.Event()
                    ready.record(front)
            readies[i] = ready
        return readies[i]

    for i in range(len(views)):
        st = begun[i]
        if batch_lists:
            ready = lists_ready            # (the lists of all views were enqueued above)
        else:
            ready = lists_of(i)
        if i + lookahead < len(views):
            begin(i + lookahead)
        states.append(st)
        if st.num_visible == 0:
            continue
        if front is not cur:
            cur.wait_event(ready)
        R.forward_composite(st)
        if cotangent is None:
            continue
        live.append(st)
        if sums is cur:
            gsums.append(R.backward_records(st, cotangent(i, st)))
            if g_colors is not None:
                g_colors[i].copy_(gsums[-1][:, COLOR_COLUMNS])
            continue
        # The record sums of a view (bound by memory bandwidth) run on a stream of their own beside the forward of the
        # next view (bound by instruction issue).  The records stay referenced until the current stream has waited for
        # the sums: memory that the allocator hands out again is then only written after they have been read.
        dimg = cotangent(i, st)
        if pending is not None:
            cur.wait_event(pending[1])
            pending = None
        pairgrad = R.backward_composite(st, dimg)
        composited = torch.cuda.Event()
        composited.record(cur)
        gsum = torch.empty(st.n, 12, dtype=torch.float32, device=dev)
        with torch.cuda.stream(sums):
            sums.wait_event(composited)
            R.record_sums(st, pairgrad, gsum)
            if g_colors is not None:
                g_colors[i].copy_(gsum[:, COLOR_COLUMNS])
            summed = torch.cuda.Event()
            summed.record(sums)
        pending = (pairgrad, summed)
        gsums.append(gsum)
    if pending is not None:
        cur.wait_event(pending[1])
        pending = None
    elif slots_written is not None:
        # no view of the step had record sums for the current stream to wait for: it waits for the slot bases themselves,
        # so that their memory, once the states are dropped, is not handed out again under the kernel that writes them
        cur.wait_event(slots_written)
    if cotangent is None:
        _front_follows(front, cur)
        return states, None, []
    if g_colors is not None:
        for i, st in enumerate(states):
            if not st.num_visible:
                g_colors[i].zero_()
        if colors_ready is not None:
            colors_ready()
    grads, g_screens = R.backward_views(live, gsums, *params, grads=grads, accumulate=accumulate, skip_sh=g_colors is not None)
    _front_follows(front, cur)
    return states, grads, g_screens


# render_views builds the lists of all views of a step with one set of launches when their record streams together
# are at most this large (about two thirds of the 256 MB last-level cache of an MI355X)
BATCHED_LISTS_MAX_STREAM_BYTES = int(os.environ.get("GSR_BATCHED_LISTS_MAX_MB", "160")) << 20

# dL/d(colour) of a view as it reaches the clamp: columns 6..8 of its record sums (zero rows for culled Gaussians)
COLOR_COLUMNS = slice(6, 9)

SMALL_FLOATS = sum(FIELD_SIZES[:4])     # coords, scale, rotation, opacity: what is summed over the ranks


def can_factor(n_views: int, world_size: int) -> bool:
    """The factored exchange gathers equal blocks of views from the ranks."""
    return world_size >= 1 and n_views % world_size == 0


def factored_moves_less(n_views: int) -> bool:
    """Whether the factored exchange moves fewer bytes than summing the sh gradient itself: each rank receives
    3 floats per Gaussian and view of the step from the gather, against about 2 x 48 per Gaussian that a ring
    all-reduce of the sh gradient sends and receives - the gather wins below 32 views per step."""
    return 3 * n_views < 2 * 48


def gather_colors(g_colors_local: torch.Tensor, n_views: int, group=None, force_collectives: bool = False):
    """Starts the gather of the ranks' colour gradients (``[V/world, N, 3]`` each) into ``[V, N, 3]`` and returns
    ``(g_all, work)`` for ``reduce_gradients(..., gathered=...)``, or ``None`` where the gather is left to
    ``reduce_gradients`` (one rank without ``force_collectives``, or a backend other than RCCL).  Meant to be called
    from ``render_views``'s ``colors_ready``: the collective then runs beside the pass that chains the views to the
    parameters, whose results the sum over the ranks has to wait for anyway."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    world = dist.get_world_size(group)
    if (world == 1 and not force_collectives) or dist.get_backend(group) != "nccl":
        return None
    if not can_factor(n_views, world) or g_colors_local.dim() != 3 or g_colors_local.shape[0] != n_views // world:
        raise ValueError(f"g_colors_local has shape {tuple(g_colors_local.shape)}, expected [{n_views}/{world},N,3]")
    g_all = torch.empty((n_views,) + tuple(g_colors_local.shape[1:]), dtype=g_colors_local.dtype, device=g_colors_local.device)
    return g_all, dist.all_gather_into_tensor(g_all, g_colors_local, group=group, async_op=True)


def reduce_gradients(bucket: GradBucket, g_colors_local: torch.Tensor, n_views: int, rebuild_sh: Callable, group=None,
                     force_collectives: bool = False, gathered=None):
    """Gradient exchange of one step between ranks that hold the same parameters, without moving the
    sh gradient: the 11 floats per Gaussian of coords / scale / rotation / opacity are summed with one
    all-reduce, the colour gradients of the rank's views (``g_colors_local`` ``[V/world, N, 3]``, from
    ``render_views(..., g_colors=...)``) are gathered from all ranks into ``[V, N, 3]`` (rank r holds the
    views ``shard_views(V, world, r)``), and ``rebuild_sh(g_colors_all)`` writes ``bucket.sh`` from them
    (``rasterizer.sh_grad_views``, beside the all-reduce: the gather is issued first).  Per rank that is 11 N floats reduced and 3 N floats per view
    gathered in place of 59 N floats reduced; the sh gradient comes out identical on every rank and
    identical to a single-rank run.  ``force_collectives`` issues the two collectives for a single
    rank too (a rehearsal of the multi-rank path).  ``gathered``: what ``gather_colors`` returned for the same
    ``g_colors_local`` (``None``: the gather is issued here).  Returns the gathered colour gradients."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if not can_factor(n_views, world):
        raise ValueError(f"{n_views} views do not split evenly over {world} ranks")
    per_rank = n_views // world
    # the bucket may have more rows than there are live Gaussians (a model's capacity): its rows
    # beyond the colour gradients' N hold zeros and take part in the sum
    if (g_colors_local.dim() != 3 or g_colors_local.shape[0] != per_rank or g_colors_local.shape[1] > bucket.n
            or g_colors_local.shape[2] != 3 or not g_colors_local.is_contiguous()):
        raise ValueError(f"g_colors_local has shape {tuple(g_colors_local.shape)}, expected a contiguous "
                         f"[{per_rank},N<={bucket.n},3]")
    n = g_colors_local.shape[1]
    if world == 1 and not (force_collectives and dist.is_available() and dist.is_initialized()):
        rebuild_sh(g_colors_local)
        return g_colors_local
    small = bucket.flat[:bucket.n * SMALL_FLOATS]
    if dist.get_backend(group) == "nccl":
        # RCCL: the gather first (or earlier still: gather_colors), then the sum, both asynchronous: the sh rebuild needs
        # only the gathered colours and runs on the compute stream while the all-reduce is still on the wire (the two are
        # independent of each other)
        if gathered is None:
            gathered = gather_colors(g_colors_local, n_views, group, force_collectives=True)
        g_all, gather_work = gathered
        summed = dist.all_reduce(small, op=dist.ReduceOp.SUM, group=group, async_op=True)
        gather_work.wait()
        rebuild_sh(g_all)
        summed.wait()
    else:
        g_all = torch.empty((n_views, n, 3), dtype=g_colors_local.dtype, device=g_colors_local.device)
        # other backends (gloo in the tests, which stages device tensors through the host on streams of its own):
        # one collective after the other - two of them in flight beside a kernel stalled a two-rank run on one GPU
        dist.all_reduce(small, op=dist.ReduceOp.SUM, group=group)
        dist.all_gather_into_tensor(g_all, g_colors_local, group=group)
        rebuild_sh(g_all)
    return g_all


def reduce_densify_stats(model, delta_grads: torch.Tensor, delta_denom: torch.Tensor, group=None):
    """Densification statistics of a step whose views are spread over ranks: ``delta_grads`` ``[size,1]``
    f32 and ``delta_denom`` ``[size,1]`` i32 are what this rank's views add (``densify.stats_delta``); they
    are summed over the ranks (two all-reduces, 8 bytes per Gaussian) and added to the model's
    ``_coords_grads`` / ``_denom``, so every rank holds the statistics of ALL views of the step - what
    ``GaussianModel.update_densify_stats`` (``GaussianModel.py:230-232``) called once per view on one
    device leaves, up to the order of the float additions - and every rank then takes the same
    densification decisions."""
    import torch.distributed as dist
    size = int(model._size)
    if tuple(delta_grads.shape) != (size, 1) or tuple(delta_denom.shape) != (size, 1):
        raise ValueError(f"the deltas have shapes {tuple(delta_grads.shape)} / {tuple(delta_denom.shape)}, expected [{size},1]")
    if delta_grads.dtype != torch.float32 or delta_denom.dtype != torch.int32:
        raise TypeError("delta_grads must be float32 and delta_denom int32")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(delta_grads, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(delta_denom, op=dist.ReduceOp.SUM, group=group)
    with torch.no_grad():
        model._coords_grads[:size] += delta_grads
        model._denom[:size] += delta_denom
    return model
