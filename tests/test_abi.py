"""CPU-only checks of the C-ABI boundary: the shared library loads and exports every function that
include/gsr.h declares (no kernel is launched here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "gsr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for n in ("gsr_preprocess_fwd", "gsr_preprocess_bwd", "gsr_sort_pairs", "gsr_bin_offsets", "gsr_bin_lists",
              "gsr_composite_fwd", "gsr_composite_bwd", "gsr_exclusive_scan", "gsr_last_error"):
        assert n in names


def test_library_exports_every_declared_symbol():
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"libgsr.so does not export {name}"
    assert set(declared_functions()) == set(_lib.SYMBOLS)
    assert _lib.load().gsr_version() == _lib.VERSION_WORD          # (load() itself refuses any other)


def test_struct_layout_matches_header():
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib
    # 5 x int32 + 16 + 16 + 3 floats, no padding
    assert ctypes.sizeof(_lib.GsrView) == 4 * (5 + 16 + 16 + 3)


def test_workspace_queries_do_not_need_a_gpu():
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib
    lib = _lib.load()
    assert lib.gsr_sort_workspace_bytes(0) > 0
    assert lib.gsr_sort_workspace_bytes(1 << 20) >= (1 << 20) // 4096 * 256 * 4
    assert lib.gsr_bin_lists_workspace_bytes(1000) >= 5 * 1000 * 4
    assert lib.gsr_bin_offsets_workspace_bytes(1000) >= 3 * 1000 * 4


def test_renderer_refuses_cpu_tensors():
    """The product path has no CPU fallback: it fails loudly instead."""
    import torch
    import gsplat_amd  # noqa: F401
    from gsplat_amd.model import GaussianParams
    from gsplat_amd.renderer import GaussianRenderer
    from gsplat_amd.scenes import synthetic_scene
    sc = synthetic_scene(16, 32, 32, 0, 0)
    r = GaussianRenderer(GaussianParams.from_scene(sc, "cpu"))
    with pytest.raises(RuntimeError):
        r.render(sc.camera(device="cpu"), 16)


def test_bin_lists_refuses_a_record_stream_of_4_gib():
    """An argument check that runs before anything is launched: 2^26 pairs x 64 bytes reach the 32-bit
    stream offsets the compositing kernels use."""
    import ctypes as C
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib
    lib = _lib.load()
    view = _lib.make_view(10, 0, 3840, 2160, 16, [0.0] * 16, [0.0] * 16, [0.0] * 3)
    dummy = (C.c_uint32 * 4)()
    addr = C.addressof(dummy)
    rc = lib.gsr_bin_lists(C.byref(view), addr, addr, addr, addr, addr, addr, 1 << 26, addr, None, addr, addr, addr, 1 << 40, None)
    assert rc == -1          # GSR_ERR_ARG
    with pytest.raises(RuntimeError, match="4 GiB"):
        _lib.check(rc, "bin_lists")


def test_block_list_queries_do_not_need_a_gpu():
    """Which views take the per-block lists, how many blocks they have and how much scratch the builder wants."""
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib
    lib = _lib.load()
    eye = [1.0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]

    def view(tile):
        return _lib.make_view(1000, 3, 1920, 1080, tile, eye, eye, [0.0, 0.0, 0.0])

    for tile, want in ((16, 0), (32, 1), (40, 0), (48, 1), (64, 1), (80, 0), (128, 0)):
        assert lib.gsr_block_lists_supported(view(tile), 100_000) == want
    assert lib.gsr_block_lists_supported(view(64), (1 << 24) - 1) == 1       # pairs x 16 blocks must stay below 2^28
    assert lib.gsr_block_lists_supported(view(64), 1 << 24) == 0
    assert lib.gsr_block_lists_supported(view(32), (1 << 26) - 1) == 1
    # 32-bit combination words below 2^26 - 1 pair-block combinations, 64-bit ones from there on
    assert lib.gsr_block_combo_bytes(view(64), (1 << 22) - 1) == 4 and lib.gsr_block_combo_bytes(view(64), 1 << 22) == 8
    assert lib.gsr_block_count(view(64)) == 120 * 68
    tiles = 30 * 17
    assert lib.gsr_block_lists_workspace_bytes(view(64), 1_000_000) >= 4 * (16 * (1_000_000 // 64 + tiles) + tiles + 120 * 68)
