"""ctypes front-end of the CPU oracle (``libgso_oracle.so``).  TEST INFRASTRUCTURE ONLY.

Imported by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``; the shipped renderer never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgso_oracle.so")


class GsoView(C.Structure):
    _fields_ = [("n", C.c_int32), ("sh_degree", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("tile_length", C.c_int32),
                ("w2m", C.c_float * 16), ("proj", C.c_float * 16), ("cam_pos", C.c_float * 3)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("gso_oracle.c", "gso_loss.c", "gso_composite.inc")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgso_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gso_preprocess_fwd.restype = C.c_int64
        _lib.gso_bin_count.restype = C.c_int64
        _lib.gso_num_tiles.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def make_view(n, sh_degree, width, height, tile_length, w2m, proj, cam_pos) -> GsoView:
    v = GsoView()
    v.n, v.sh_degree, v.width, v.height, v.tile_length = int(n), int(sh_degree), int(width), int(height), int(tile_length)
    v.w2m[:] = [float(x) for x in np.asarray(w2m, dtype=np.float32).reshape(16)]
    v.proj[:] = [float(x) for x in np.asarray(proj, dtype=np.float32).reshape(16)]
    v.cam_pos[:] = [float(x) for x in np.asarray(cam_pos, dtype=np.float32).reshape(3)]
    return v


class SingularCovariance(RuntimeError):
    pass


def render(coords, scale, rotation, opacity, sh, sh_degree, width, height, tile_length,
           w2m, proj, cam_pos, screen_coords=None, dimg=None, want_lists=True, threads=None, precise=False):
    """Forward (and backward when ``dimg`` is given) of the whole path on the CPU.

    Inputs are the raw parameters ([N,3],[N,3],[N,4],[N,1] or [N],[N,3,16]).  Returns a dict
    with every stage's output (N-indexed unless noted) or ``{"img": None}`` when nothing is
    visible (GaussianRenderer.py:67-68).
    """
    L = lib()
    if threads is not None:
        L.gso_set_threads(int(threads))
    coords, scale, rotation, sh = _f32(coords), _f32(scale), _f32(rotation), _f32(sh)
    opacity = _f32(opacity).reshape(-1)
    n = coords.shape[0]
    assert sh.shape == (n, 3, 16)
    sc = None if screen_coords is None else _f32(screen_coords)
    v = make_view(n, sh_degree, width, height, tile_length, w2m, proj, cam_pos)
    mask = np.zeros(n, np.uint8)
    pos2d = np.zeros((n, 3), np.float32)
    cov2d = np.zeros((n, 4), np.float32)
    conic = np.zeros((n, 4), np.float32)
    opac = np.zeros(n, np.float32)
    color = np.zeros((n, 3), np.float32)
    radius = np.zeros(n, np.float32)
    bbox = np.zeros((n, 4), np.int32)
    vis = L.gso_preprocess_fwd(C.byref(v), _p(coords), _p(scale), _p(rotation), _p(opacity), _p(sh), _p(sc),
                               _p(mask), _p(pos2d), _p(cov2d), _p(conic), _p(opac), _p(color), _p(radius), _p(bbox))
    if vis == -2:
        raise SingularCovariance("singular 2-D covariance (the reference raises torch.linalg.LinAlgError)")
    if vis < 0:
        raise ValueError("bad argument")
    out = {"mask": mask.astype(bool), "num_visible": int(vis)}
    if vis == 0:
        out["img"] = None
        return out
    out.update(pos2d=pos2d, cov2d=cov2d, conic=conic, opacity_act=opac, color=color,
               radius_full=radius, radius=radius[mask.astype(bool)], bbox=bbox)
    T = int(L.gso_num_tiles(C.byref(v)))
    counts = np.zeros(T, np.int64)
    total = int(L.gso_bin_count(C.byref(v), _p(mask), _p(bbox), _p(counts)))
    offsets = np.zeros(T + 1, np.int64)
    lists = np.zeros(max(total, 1), np.int32)
    L.gso_bin_sort(C.byref(v), _p(mask), _p(bbox), _p(pos2d), _p(offsets), _p(lists))
    lists = lists[:total]
    out.update(tile_offsets=offsets, tile_lists=lists, num_pairs=total)
    img = np.zeros((width, height, 3), np.float32)
    L.gso_composite_fwd(C.byref(v), _p(offsets), _p(lists), _p(pos2d), _p(conic), _p(opac), _p(color), _p(img), int(precise))
    out["img"] = img
    if dimg is not None:
        dimg = _f32(dimg)
        assert dimg.shape == img.shape
        d_pos2d = np.zeros((n, 2), np.float64)
        d_conic = np.zeros((n, 4), np.float64)
        d_opac = np.zeros(n, np.float64)
        d_color = np.zeros((n, 3), np.float64)
        L.gso_composite_bwd(C.byref(v), _p(offsets), _p(lists), _p(pos2d), _p(conic), _p(opac), _p(color),
                            _p(dimg), _p(d_pos2d), _p(d_conic), _p(d_opac), _p(d_color), int(precise))
        g_coords = np.zeros((n, 3), np.float32)
        g_scale = np.zeros((n, 3), np.float32)
        g_rot = np.zeros((n, 4), np.float32)
        g_opac = np.zeros(n, np.float32)
        g_sh = np.zeros((n, 3, 16), np.float32)
        g_screen = np.zeros((n, 3), np.float32)
        L.gso_preprocess_bwd(C.byref(v), _p(coords), _p(scale), _p(rotation), _p(opacity), _p(sh), _p(sc),
                             _p(mask), _p(d_pos2d), _p(d_conic), _p(d_opac), _p(d_color),
                             _p(g_coords), _p(g_scale), _p(g_rot), _p(g_opac), _p(g_sh), _p(g_screen))
        out.update(d_pos2d=d_pos2d, d_conic=d_conic, d_opacity_act=d_opac, d_color=d_color,
                   g_coords=g_coords, g_scale=g_scale, g_rotation=g_rot, g_opacity=g_opac.reshape(n, 1),
                   g_sh=g_sh, g_screen=g_screen)
    return out


def render_scene(scene, tile_length, dimg=None, threads=None, screen_coords=None, precise=False):
    """``render`` on a ``gsplat_amd.scenes.Scene`` (camera built on the host)."""
    cam = scene.camera(device="cpu")
    return render(scene.coords.numpy(), scene.scale.numpy(), scene.rotation.numpy(),
                  scene.opacity.numpy(), scene.sh.numpy(), scene.sh_degree, scene.width, scene.height,
                  tile_length, cam.w2m_host, cam.proj_host, np.asarray(scene.cam_position, np.float32),
                  screen_coords=screen_coords, dimg=dimg, threads=threads, precise=precise)


def loss(pred, gt, w_l1=0.8, w_ssim=-0.2, want_grad=True):
    """l1 (Utils/Loss.py:4-5), mean SSIM over [W,H,3] read as W planes of [H,3] (Utils/Loss.py:69-97),
    and ``w_l1 * d l1/d pred + w_ssim * d ssim/d pred``.  The reference's training loss
    (Trainer.py:35-38) is ``(1 - lam) * l1 + lam * (1 - ssim)``, i.e. weights ``(1 - lam, -lam)``."""
    L = lib()
    pred, gt = _f32(pred), _f32(gt)
    assert pred.shape == gt.shape and pred.ndim == 3 and pred.shape[2] == 3
    w, h = pred.shape[:2]
    out = np.zeros(2, np.float64)
    dpred = np.zeros_like(pred) if want_grad else None
    rc = L.gso_loss(_p(pred), _p(gt), C.c_int(w), C.c_int(h), C.c_double(w_l1), C.c_double(w_ssim), _p(out), _p(dpred))
    if rc != 0:
        raise ValueError("bad argument")
    return {"l1": float(out[0]), "ssim": float(out[1]), "dpred": dpred}


def max_threads() -> int:
    return int(lib().gso_max_threads())
