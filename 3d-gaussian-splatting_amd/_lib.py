"""ctypes binding of ``libgsr.so`` (the C ABI declared in ``include/gsr.h``).

The library is built in-tree by ``make -C csrc`` (hipcc, gfx950).  There is no CPU fallback:
``load()`` raises when the shared object is missing, and every wrapper raises ``RuntimeError``
with the library's message on a non-zero return code.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GSR_LIB_PATH") or os.path.join(CSRC, "libgsr.so")

REC_FLOATS = 12
STREAM_DWORDS = 16
PAIRGRAD_FLOATS = 12
MAX_VIEWS = 8
ADAM_MAX_TENSORS = 8
VERSION = 2
# what gsr_version() of a library with these row widths returns (GSR_VERSION_WORD in include/gsr.h)
VERSION_WORD = VERSION | REC_FLOATS << 8 | STREAM_DWORDS << 16 | PAIRGRAD_FLOATS << 24

# Every symbol include/gsr.h declares.
SYMBOLS = (
    "gsr_last_error", "gsr_version", "gsr_preprocess_fwd", "gsr_preprocess_fwd_views", "gsr_preprocess_bwd",
    "gsr_preprocess_bwd_workspace_bytes",
    "gsr_pair_reduce", "gsr_preprocess_bwd_views", "gsr_sh_grad_views",
    "gsr_sort_workspace_bytes", "gsr_sort_pairs", "gsr_sort_pairs64", "gsr_scan_workspace_bytes", "gsr_exclusive_scan",
    "gsr_exclusive_scan_static",
    "gsr_bin_offsets_workspace_bytes", "gsr_bin_offsets", "gsr_bin_offsets_views_workspace_bytes", "gsr_bin_offsets_views", "gsr_slot_bases",
    "gsr_bin_lists_workspace_bytes", "gsr_bin_lists", "gsr_tile_order", "gsr_bin_lists_views_supported",
    "gsr_bin_lists_views_layout", "gsr_bin_lists_views_workspace_bytes", "gsr_bin_lists_views",
    "gsr_composite_fwd", "gsr_composite_bwd", "gsr_pairgrad_parts",
    "gsr_block_lists_supported", "gsr_block_combo_bytes", "gsr_block_count", "gsr_block_lists_workspace_bytes", "gsr_block_lists",
    "gsr_composite_blocks_fwd", "gsr_composite_blocks_bwd", "gsr_block_rows_reduce",
    "gsr_loss_workspace_bytes", "gsr_loss_fwd_bwd",
    "gsr_adam_step", "gsr_knn_dist2",
    "gsr_densify_stats", "gsr_densify_stats_views", "gsr_densify_select_workspace_bytes", "gsr_densify_select", "gsr_densify_apply",
    "gsr_remove_select_workspace_bytes", "gsr_remove_select", "gsr_remove_apply_workspace_bytes", "gsr_remove_apply",
)


class GsrView(C.Structure):
    _fields_ = [("n", C.c_int32), ("sh_degree", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("tile_length", C.c_int32), ("w2m", C.c_float * 16), ("proj", C.c_float * 16),
                ("cam_pos", C.c_float * 3)]


def make_view(n, sh_degree, width, height, tile_length, w2m, proj, cam_pos) -> GsrView:
    v = GsrView()
    v.n, v.sh_degree, v.width, v.height, v.tile_length = int(n), int(sh_degree), int(width), int(height), int(tile_length)
    v.w2m[:] = [float(x) for x in w2m]
    v.proj[:] = [float(x) for x in proj]
    v.cam_pos[:] = [float(x) for x in cam_pos]
    return v


def build(force: bool = False) -> str:
    """Compile libgsr.so for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built (run `make -C {CSRC}` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # libgsr.so needs libamdhip64.so.7.  PyTorch ships its own copy of that runtime; the device
    # pointers and streams handed to the library come from torch, so both must share ONE runtime.
    # Importing torch first makes the dynamic loader resolve the soname to torch's copy (a second
    # runtime loaded from /opt/rocm ahead of torch's reports "no ROCm-capable device").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name in SYMBOLS:
        if not hasattr(lib, name):
            raise RuntimeError(f"{LIB_PATH} does not export {name}")
    lib.gsr_last_error.restype = C.c_char_p
    # The buffers are allocated here by the row widths above: a library built with other widths (a stale build, a
    # variant of an experiment) would write rows of its own width into them - out of bounds.  Refuse it.
    if lib.gsr_version() != VERSION_WORD:
        raise RuntimeError(f"{LIB_PATH} reports version word {lib.gsr_version():#x}, this host code expects "
                           f"{VERSION_WORD:#x} (version | record floats << 8 | stream dwords << 16 | gradient-record "
                           "floats << 24): rebuild the library from this tree")
    for name in ("gsr_sort_workspace_bytes", "gsr_scan_workspace_bytes", "gsr_bin_offsets_workspace_bytes",
                 "gsr_bin_lists_workspace_bytes", "gsr_preprocess_bwd_workspace_bytes",
                 "gsr_densify_select_workspace_bytes", "gsr_remove_select_workspace_bytes",
                 "gsr_remove_apply_workspace_bytes"):
        getattr(lib, name).restype = C.c_size_t
        getattr(lib, name).argtypes = [C.c_uint32]
    vp = C.c_void_p
    lib.gsr_preprocess_fwd.argtypes = [C.POINTER(GsrView)] + [vp] * 16
    lib.gsr_preprocess_fwd_views.argtypes = [C.POINTER(GsrView), C.c_uint32] + [vp] * 15
    lib.gsr_preprocess_bwd.argtypes = [C.POINTER(GsrView)] + [vp] * 15 + [C.c_int, vp, C.c_size_t, vp]
    lib.gsr_pair_reduce.argtypes = [C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, vp]
    lib.gsr_pairgrad_parts.restype = C.c_uint32
    lib.gsr_pairgrad_parts.argtypes = [C.c_int]
    lib.gsr_preprocess_bwd_views.argtypes = ([C.POINTER(GsrView), C.c_uint32] + [vp] * 5 + [C.POINTER(vp)] * 2 + [vp] * 5 +
                                             [C.POINTER(vp), C.POINTER(vp), C.c_int, vp])
    lib.gsr_sh_grad_views.argtypes = [C.POINTER(GsrView), C.c_uint32, vp, vp, C.POINTER(vp), vp, C.c_int, vp]
    lib.gsr_sort_pairs.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_int, C.c_int, vp, C.c_size_t, vp]
    lib.gsr_sort_pairs64.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_int, C.c_int, vp, C.c_size_t, vp]
    lib.gsr_exclusive_scan.argtypes = [vp, vp, C.c_uint32, vp, vp, C.c_size_t, vp]
    lib.gsr_exclusive_scan_static.argtypes = [vp, vp, C.c_uint32, vp, vp]
    lib.gsr_bin_offsets.argtypes = [vp, vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.gsr_bin_offsets_views_workspace_bytes.restype = C.c_size_t
    lib.gsr_bin_offsets_views_workspace_bytes.argtypes = [C.c_uint32, C.c_uint32]
    lib.gsr_bin_offsets_views.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.gsr_slot_bases.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp]
    lib.gsr_bin_lists.argtypes = [C.POINTER(GsrView), vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.gsr_tile_order.argtypes = [C.POINTER(GsrView), vp, vp, vp]
    u32p = C.POINTER(C.c_uint32)
    lib.gsr_bin_lists_views_supported.argtypes = [C.POINTER(GsrView), C.c_uint32, u32p]
    lib.gsr_bin_lists_views_layout.restype = C.c_uint64
    lib.gsr_bin_lists_views_layout.argtypes = [C.c_uint32, u32p, u32p]
    lib.gsr_bin_lists_views_workspace_bytes.restype = C.c_size_t
    lib.gsr_bin_lists_views_workspace_bytes.argtypes = [C.c_uint32, u32p]
    lib.gsr_bin_lists_views.argtypes = [C.POINTER(GsrView), C.c_uint32] + [vp] * 6 + [u32p] + [vp] * 5 + [C.c_size_t, vp]
    lib.gsr_composite_fwd.argtypes = [C.POINTER(GsrView), vp, vp, vp, vp, vp, vp]
    lib.gsr_block_lists_supported.argtypes = [C.POINTER(GsrView), C.c_uint32]
    lib.gsr_block_count.restype = C.c_uint32
    lib.gsr_block_combo_bytes.restype = C.c_uint32
    lib.gsr_block_combo_bytes.argtypes = [C.POINTER(GsrView), C.c_uint32]
    lib.gsr_block_count.argtypes = [C.POINTER(GsrView)]
    lib.gsr_block_lists_workspace_bytes.restype = C.c_size_t
    lib.gsr_block_lists_workspace_bytes.argtypes = [C.POINTER(GsrView), C.c_uint32]
    lib.gsr_block_lists.argtypes = [C.POINTER(GsrView), vp, vp, C.c_uint32, vp, vp, vp, vp, C.c_size_t, vp]
    lib.gsr_composite_blocks_fwd.argtypes = [C.POINTER(GsrView), C.c_uint32, vp, vp, vp, vp, vp, vp]
    lib.gsr_composite_blocks_bwd.argtypes = [C.POINTER(GsrView), C.c_uint32, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_block_rows_reduce.argtypes = [C.POINTER(GsrView), C.c_uint32, vp, vp, vp, vp, vp]
    lib.gsr_composite_bwd.argtypes = [C.POINTER(GsrView), vp, vp, vp, vp, vp, C.c_uint32, vp, vp]
    lib.gsr_loss_workspace_bytes.restype = C.c_size_t
    lib.gsr_loss_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.gsr_loss_fwd_bwd.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp, vp, C.c_size_t, vp]
    lib.gsr_adam_step.argtypes = ([C.c_uint32] + [C.POINTER(vp)] * 4 + [C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)] +
                                  [C.c_double] * 4 + [vp])
    lib.gsr_knn_dist2.argtypes = [C.c_uint32, vp, vp, vp]
    lib.gsr_densify_stats.argtypes = [C.c_uint32, vp, vp, vp, vp, vp]
    lib.gsr_densify_stats_views.argtypes = [C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]
    lib.gsr_densify_select.argtypes = [C.c_uint32, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.gsr_densify_apply.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32] + [vp] * 9 + [vp]
    lib.gsr_remove_select.argtypes = [C.c_uint32, vp, vp, vp, vp, C.c_size_t, vp]
    lib.gsr_remove_apply.argtypes = [C.c_uint32, C.c_uint32] + [vp] * 9 + [vp, C.c_size_t, vp]
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().gsr_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed ({code}): {msg}")
