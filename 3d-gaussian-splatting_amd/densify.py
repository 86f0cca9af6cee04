"""Densification bookkeeping on the GPU: host side of ``gsr_densify_*`` and ``gsr_remove_*``.

The functions take a model with the reference's attributes (``GaussianModel.py:17-27``):
``_coords _scale _rotation _opacity _sh`` (``nn.Parameter``s of capacity rows), ``_coords_grads``
``[capacity,1]`` f32, ``_denom`` ``[capacity,1]`` i32, ``_size``, ``_device``, ``curr_sh_degree``,
``max_sh_degree`` - the reference's own ``GaussianModel`` on the GPU or ``model.GaussianModel``
here - and do what its methods of the same names do (``GaussianModel.py:110-245``), including the
places where that differs from the published method (see ``csrc/densify.hip``).  One host
synchronisation per ``densify`` / ``remove`` reads the row counts back, as the reference's
``mask.sum()`` does.  There is no CPU path.
"""
import torch
import torch.nn as nn

from . import _lib
from .rasterizer import _call, _stream

GRAD_THRESHOLD = 0.0002     # GaussianModel.py:240,244
SCALE_FRACTION = 0.01       # of the scene radius, GaussianModel.py:240,244
PARAM_NAMES = ("_coords", "_opacity", "_sh", "_scale", "_rotation")   # order of GaussianModel.py:122-136


def _rows(model):
    ts = (model._coords, model._scale, model._rotation, model._opacity, model._sh)
    for t in ts:
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("the model's parameters must be contiguous float32 tensors on the GPU (there is no CPU path)")
    return ts


def _mask_u8(mask: torch.Tensor, n: int) -> torch.Tensor:
    if mask.dim() != 1 or mask.shape[0] != n:
        raise ValueError(f"mask has shape {tuple(mask.shape)}, expected [{n}]")
    return mask.to(torch.uint8).contiguous() if mask.dtype != torch.uint8 else mask.contiguous()


@torch.no_grad()
def update_densify_stats(model, pos2d: torch.Tensor, mask: torch.Tensor):
    """``GaussianModel.update_densify_stats`` (``GaussianModel.py:230-232``); ``pos2d`` is the
    ``screen_coords`` leaf a render returned, read through its ``.grad``."""
    n = int(model._size)
    g = pos2d.grad if pos2d.grad is not None else None
    if g is None:
        raise RuntimeError("screen_coords has no gradient: call backward() first")
    if tuple(g.shape) != (n, 3) or not g.is_cuda:
        raise ValueError(f"screen_coords.grad has shape {tuple(g.shape)}, expected [{n},3] on the GPU")
    accumulate_stats(model, g, mask)


@torch.no_grad()
def accumulate_stats(model, g_screen: torch.Tensor, mask: torch.Tensor):
    """The same from the gradient tensor itself (``[size,3]``), e.g. a view's ``g_screen`` as
    ``rasterizer.backward_views`` returns it."""
    n = int(model._size)
    m8 = _mask_u8(mask, n)
    lib = _lib.load()
    _call("gsr_densify_stats", lib.gsr_densify_stats, n, m8.data_ptr(), g_screen.contiguous().data_ptr(),
          model._coords_grads.data_ptr(), model._denom.data_ptr(), _stream(g_screen.device))


def _stats_views(size: int, g_screens, masks, coords_grads, denom):
    """``gsr_densify_stats_views``: the contributions of several views with one launch (per 8 views), added in the
    order of the views - bit for bit what one ``gsr_densify_stats`` call per view leaves."""
    import ctypes as C
    if len(g_screens) != len(masks):
        raise ValueError(f"{len(g_screens)} gradients for {len(masks)} masks")
    if not g_screens:
        return
    dev = g_screens[0].device
    keep = []
    for g, mask in zip(g_screens, masks):
        if tuple(g.shape) != (size, 3) or not g.is_cuda or g.dtype != torch.float32:
            raise ValueError(f"g_screen has shape {tuple(g.shape)}, expected float32 [{size},3] on the GPU")
        keep.append((g.contiguous(), _mask_u8(mask, size)))
    nv = len(keep)
    grads_p = (C.c_void_p * nv)(*[g.data_ptr() for g, _ in keep])
    masks_p = (C.c_void_p * nv)(*[m.data_ptr() for _, m in keep])
    _call("gsr_densify_stats_views", _lib.load().gsr_densify_stats_views, size, nv, masks_p, grads_p, coords_grads.data_ptr(),
          denom.data_ptr(), _stream(dev))


@torch.no_grad()
def accumulate_stats_views(model, g_screens, masks):
    """``accumulate_stats`` for the views of one step (``g_screens`` and ``masks`` in the order of the views)."""
    _stats_views(int(model._size), g_screens, masks, model._coords_grads, model._denom)


@torch.no_grad()
def stats_delta(size: int, g_screens, masks):
    """What ``update_densify_stats`` would add for several views, as tensors of their own:
    ``(grads [size,1] f32, denom [size,1] i32)`` starting from zero.  For ranks that each render a part
    of a step's views: the deltas are summed over the ranks (``minibatch.reduce_densify_stats``) and
    added to every rank's statistics."""
    if len(g_screens) != len(masks):
        raise ValueError(f"{len(g_screens)} gradients for {len(masks)} masks")
    if not g_screens:
        raise ValueError("stats_delta needs at least one view (the device is taken from it)")
    dev = g_screens[0].device
    grads = torch.zeros((size, 1), dtype=torch.float32, device=dev)
    denom = torch.zeros((size, 1), dtype=torch.int32, device=dev)
    _stats_views(size, g_screens, masks, grads, denom)
    return grads, denom


@torch.no_grad()
def expand_capacity(model, new_capacity: int):
    """``GaussianModel.expand_capacity`` (``GaussianModel.py:116-140``): zero rows appended, the
    parameters re-created as new ``nn.Parameter`` objects."""
    cap = model._coords.shape[0]
    if cap >= new_capacity:
        return
    for name in PARAM_NAMES:
        old = getattr(model, name)
        pad = torch.zeros((new_capacity - cap,) + tuple(old.shape[1:]), dtype=old.dtype, device=old.device)
        setattr(model, name, nn.Parameter(torch.cat((old.detach(), pad), dim=0)).requires_grad_(True))
    for name in ("_coords_grads", "_denom"):
        old = getattr(model, name)
        pad = torch.zeros((new_capacity - cap,) + tuple(old.shape[1:]), dtype=old.dtype, device=old.device)
        setattr(model, name, torch.cat((old, pad), dim=0))


def _zero_stats(model):
    cap, dev = model._coords.shape[0], model._coords.device
    model._coords_grads = torch.zeros((cap, 1), dtype=torch.float32, device=dev)
    model._denom = torch.zeros((cap, 1), dtype=torch.int32, device=dev)


@torch.no_grad()
def densify(model, scene_radius, noise: torch.Tensor = None, generator: torch.Generator = None):
    """``GaussianModel.densify`` (``GaussianModel.py:234-245``) = selection, ``clone``
    (``:179-197``), ``split`` (``:199-228``).  ``noise`` ``[>= 2 * num_split, 3]``: the
    standard-normal draws behind the split's ``torch.normal`` (drawn with ``torch.randn`` on the
    device, from ``generator`` when given, if omitted).  Returns ``(num_clone, num_split)``."""
    lib = _lib.load()
    size = int(model._size)
    if size == 0:
        return 0, 0
    dev = model._coords.device
    stream = _stream(dev)
    _rows(model)
    cls = torch.empty(size, dtype=torch.uint8, device=dev)
    ranks = torch.empty((2, size), dtype=torch.int32, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    ws_bytes = lib.gsr_densify_select_workspace_bytes(size)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    thr = float(torch.as_tensor(SCALE_FRACTION, dtype=torch.float32) * torch.as_tensor(scene_radius, dtype=torch.float32).cpu())
    _call("gsr_densify_select", lib.gsr_densify_select, size, model._coords_grads.data_ptr(), model._denom.data_ptr(),
          model._scale.data_ptr(), GRAD_THRESHOLD, thr, cls.data_ptr(), ranks[0].data_ptr(), ranks[1].data_ptr(),
          counts.data_ptr(), ws.data_ptr(), ws_bytes, stream)
    num_clone, num_split = (int(c) for c in counts.cpu())
    if num_clone == 0 and num_split == 0:
        return 0, 0            # the statistics persist (GaussianModel.py:181-182,201-202)
    # capacity as clone, then split, grow it: doubled once each when the new size reaches it
    if size + num_clone >= model._coords.shape[0] and num_clone:
        expand_capacity(model, 2 * model._coords.shape[0])
    if num_split and size + num_clone + num_split >= model._coords.shape[0]:
        expand_capacity(model, 2 * model._coords.shape[0])
    if size + num_clone + num_split > model._coords.shape[0]:
        raise RuntimeError("densify: the doubled capacity cannot hold the new rows (the reference fails here too)")
    if num_split:
        if noise is None:
            noise = torch.randn((2 * num_split, 3), dtype=torch.float32, device=dev, generator=generator)
        if noise.dim() != 2 or noise.shape[1] != 3 or noise.shape[0] < 2 * num_split or noise.dtype != torch.float32:
            raise ValueError(f"noise has shape {tuple(noise.shape)}, expected float32 [>= {2 * num_split}, 3]")
        noise = noise.to(dev).contiguous()
    coords, scale, rot, opac, sh = _rows(model)
    _call("gsr_densify_apply", lib.gsr_densify_apply, size, num_clone, num_split, cls.data_ptr(), ranks[0].data_ptr(),
          ranks[1].data_ptr(), None if noise is None else noise.data_ptr(), coords.data_ptr(), scale.data_ptr(),
          rot.data_ptr(), opac.data_ptr(), sh.data_ptr(), stream)
    _zero_stats(model)
    model._size = size + num_clone + num_split
    return num_clone, num_split


def _mask_ranks(mask: torch.Tensor, size: int):
    """(flags u8, rank i32 = number of set rows below each row, count) of a ``[size]`` mask, the ranks by
    ``gsr_exclusive_scan``; one read-back for the count, as the reference's ``mask.sum()``."""
    lib = _lib.load()
    flags = _mask_u8(mask, size)
    dev = flags.device
    if not flags.is_cuda:
        raise RuntimeError("the mask must be on the GPU (there is no CPU path)")
    wide = flags.to(torch.int32)
    rank = torch.empty(size, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.gsr_scan_workspace_bytes(size)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _call("gsr_exclusive_scan", lib.gsr_exclusive_scan, wide.data_ptr(), rank.data_ptr(), size, total.data_ptr(), ws.data_ptr(),
          ws_bytes, _stream(dev))
    return flags, rank, int(total.cpu())


def _grow_for(model, new_size: int, what: str):
    """The capacity rule of ``clone`` and ``split`` (``GaussianModel.py:184-187,204-207``): doubled once when
    the new size reaches it."""
    if new_size >= model._coords.shape[0]:
        expand_capacity(model, 2 * model._coords.shape[0])
    if new_size > model._coords.shape[0]:
        raise RuntimeError(f"{what}: the doubled capacity cannot hold the new rows (the reference fails here too)")


@torch.no_grad()
def clone(model, mask: torch.Tensor, expand_factor: int = 2):
    """``GaussianModel.clone(mask)`` (``GaussianModel.py:179-197``) on its own: the masked rows are appended
    after the live rows, the statistics zeroed.  ``expand_factor`` is accepted and, as in the reference,
    not used (the capacity doubles).  Returns the number of cloned rows."""
    lib = _lib.load()
    size = int(model._size)
    if size == 0:
        return 0
    flags, rank, num = _mask_ranks(mask, size)
    if num == 0:
        return 0                      # the statistics persist (GaussianModel.py:181-182)
    _grow_for(model, size + num, "clone")
    coords, scale, rot, opac, sh = _rows(model)
    _call("gsr_densify_apply", lib.gsr_densify_apply, size, num, 0, flags.data_ptr(), rank.data_ptr(), rank.data_ptr(), None,
          coords.data_ptr(), scale.data_ptr(), rot.data_ptr(), opac.data_ptr(), sh.data_ptr(), _stream(coords.device))
    _zero_stats(model)
    model._size = size + num
    return num


@torch.no_grad()
def split(model, mask: torch.Tensor, expand_factor: int = 2, split_factor: int = 2, noise: torch.Tensor = None,
          generator: torch.Generator = None):
    """``GaussianModel.split(mask)`` (``GaussianModel.py:199-228``) on its own, for the reference's
    ``split_factor`` of 2: every masked row is replaced by a sample around the origin of its own frame and a
    second sample is appended, both with the scale divided by 1.6.  ``noise`` as in ``densify``.  Returns
    the number of split rows."""
    if split_factor != 2:
        raise NotImplementedError("split_factor 2 only (the value every call of the reference uses)")
    lib = _lib.load()
    size = int(model._size)
    if size == 0:
        return 0
    flags, rank, num = _mask_ranks(mask, size)
    if num == 0:
        return 0
    _grow_for(model, size + num, "split")
    dev = model._coords.device
    if noise is None:
        noise = torch.randn((2 * num, 3), dtype=torch.float32, device=dev, generator=generator)
    if noise.dim() != 2 or noise.shape[1] != 3 or noise.shape[0] < 2 * num or noise.dtype != torch.float32:
        raise ValueError(f"noise has shape {tuple(noise.shape)}, expected float32 [>= {2 * num}, 3]")
    noise = noise.to(dev).contiguous()
    if noise.shape[0] != 2 * num:
        # the kernel reads the second sample of row k at noise[num_split + k]
        noise = noise[:2 * num].contiguous()
    cls = flags * 2                   # 2 = split (gsr_densify_select's encoding)
    coords, scale, rot, opac, sh = _rows(model)
    _call("gsr_densify_apply", lib.gsr_densify_apply, size, 0, num, cls.data_ptr(), rank.data_ptr(), rank.data_ptr(),
          noise.data_ptr(), coords.data_ptr(), scale.data_ptr(), rot.data_ptr(), opac.data_ptr(), sh.data_ptr(), _stream(dev))
    _zero_stats(model)
    model._size = size + num
    return num


@torch.no_grad()
def remove(model, mask: torch.Tensor, inplace: bool = True):
    """``GaussianModel.remove(mask, inplace=True)`` (``GaussianModel.py:148-164``) for a 1-D mask,
    literally as the reference does it (see ``gsr_remove_apply`` in ``include/gsr.h``)."""
    if not inplace:
        raise NotImplementedError("only the in-place form (the reference trainer's call, Trainer.py:69)")
    lib = _lib.load()
    size = int(model._size)
    if size == 0:
        return 0
    dev = model._coords.device
    if mask.dim() != 1:
        # Trainer.py:68 hands over a mask broadcast to [N,M]; the reference returns at once when nothing
        # is set in it (GaussianModel.py:149-151) and fails on the indexing otherwise
        if not bool(mask.any()):
            return 0
        raise IndexError(f"remove: the mask has shape {tuple(mask.shape)}; a set entry of a mask that is not [size] cannot be "
                         "applied (the reference fails on its boolean indexing here)")
    stream = _stream(dev)
    m8 = _mask_u8(mask, size)
    rank = torch.empty(size, dtype=torch.int32, device=dev)
    kept = torch.empty(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.gsr_remove_select_workspace_bytes(size)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _call("gsr_remove_select", lib.gsr_remove_select, size, m8.data_ptr(), rank.data_ptr(), kept.data_ptr(), ws.data_ptr(),
          ws_bytes, stream)
    num_removed = size - int(kept.cpu())
    if num_removed == 0:
        return 0
    coords, scale, rot, opac, sh = _rows(model)
    st_bytes = lib.gsr_remove_apply_workspace_bytes(num_removed)
    staging = torch.empty(st_bytes, dtype=torch.uint8, device=dev)
    _call("gsr_remove_apply", lib.gsr_remove_apply, size, num_removed, m8.data_ptr(), rank.data_ptr(), coords.data_ptr(),
          scale.data_ptr(), rot.data_ptr(), opac.data_ptr(), sh.data_ptr(), model._coords_grads.data_ptr(),
          model._denom.data_ptr(), staging.data_ptr(), st_bytes, stream)
    model._size = size - num_removed
    return num_removed


@torch.no_grad()
def reset_opacity(model):
    """``GaussianModel.reset_opacity`` (``GaussianModel.py:110-111``): every raw opacity, over the capacity."""
    model._opacity.fill_(0.01)


def add_sh_degree(model):
    """``GaussianModel.add_sh_degree`` (``GaussianModel.py:113-114``)."""
    model.curr_sh_degree = min(model.max_sh_degree, 1 + model.curr_sh_degree)
