"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Floating-point bar of the path: fp32 rtol 1e-4.  Gradients are sums with cancellation, so an
# element much smaller than its tensor's largest entry cannot carry a relative bound; those are
# held to an absolute floor of ATOL_FRAC * max|reference| instead.
RTOL = 1e-4
ATOL_FRAC = 2e-6


def _names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def golden_names():
    """Fixtures of the render path (tests/golden/make_golden.py)."""
    other = ("loss_", "adam_", "stats_", "densify_", "clone_", "split_", "remove_", "camera_", "config1_")   # fixtures of the loss, of the training-state updates and of the reference's own CPU case at its size
    return [n for n in _names() if not n.startswith(other) and not n.startswith(("nonfinite_", "nancolour_"))]


def config1_golden_names():
    """The reference's own CPU case at its size (tests/golden/make_config1_golden.py)."""
    return [n for n in _names() if n.startswith("config1_") and not n.endswith("_inputs")]


def config1_inputs(g):
    """The scene of a config-1 fixture from the stored inputs (uint16 upper halves of the fp32 patterns), checked against
    the fixture's digest of what the reference rendered.  -> (scene, target [W,H,3] float32 0..255)"""
    import hashlib

    import torch

    import gsplat_amd  # noqa: F401
    from gsplat_amd.scenes import Scene
    d = load_golden(f"config1_n{int(g['n'])}_inputs")

    def f32(bits):
        return torch.from_numpy((bits.astype(np.uint32) << 16).view(np.float32).copy())

    sh = torch.full((int(g["n"]), 3, 16), float(d["sh_rest"]), dtype=torch.float32)
    sh[:, :, 0] = f32(d["sh0"])
    scene = Scene(f32(d["coords"]), f32(d["scale"]), f32(d["rotation"]), f32(d["opacity"]), sh, int(g["sh_degree"]),
                  int(g["width"]), int(g["height"]), d["cam_position"], d["cam_rotation"], float(d["fx"]), float(d["fy"]))
    digest = hashlib.sha256(b"".join(t.numpy().tobytes() for t in scene.params())).hexdigest()
    assert digest == str(g["inputs_sha256"]), "the stored inputs are not the ones the reference rendered"
    return scene, g["target"].astype(np.float32)


def check_config1(g, out, loss3, grads):
    """A render of a config-1 fixture's scene against what the reference produced: ``out`` holds mask, tile_offsets,
    tile_lists, img, radius (length M); ``loss3`` = (loss, l1, ssim); ``grads``: the six gradients by name."""
    import hashlib
    n, stride = int(g["n"]), int(g["row_stride"])
    mask = np.unpackbits(g["ref_mask_bits"])[:n].astype(bool)
    assert_bit_equal(np.asarray(out["mask"], dtype=bool), mask, "mask")
    assert_bit_equal(np.asarray(out["tile_offsets"], dtype=np.int64), g["ref_tile_offsets"], "tile_offsets (list lengths)")
    digest = hashlib.sha256(np.ascontiguousarray(out["tile_lists"], dtype=np.int32).tobytes()).hexdigest()
    assert digest == str(g["ref_tile_lists_sha256"]), "tile lists differ from the reference's"
    assert_close(out["img"], g["ref_img"], "img")
    assert_close(np.asarray(out["radius"])[::stride], g["ref_radius_rows"], "radius")
    assert_close(loss3[0], g["ref_loss"], "loss", rtol=1e-5, atol_frac=0)
    assert_close(loss3[1], g["ref_l1"], "l1", rtol=1e-5, atol_frac=0)
    assert_close(loss3[2], g["ref_ssim"], "ssim", rtol=1e-4, atol_frac=0)
    for k, v in grads.items():
        rows, ref = np.asarray(v)[::stride], g["ref_" + k + "_rows"]
        # the absolute floor refers to the largest entry of the WHOLE reference tensor, of which the fixture keeps every 16th row
        scale = float(g["ref_" + k + "_absmax"])
        err = np.abs(rows.astype(np.float64) - ref.astype(np.float64))
        tol = RTOL * np.abs(ref.astype(np.float64)) + ATOL_FRAC * scale
        assert rows.shape == ref.shape and np.array_equal(np.isnan(rows), np.isnan(ref)), k
        bad = err > tol
        assert not bad.any(), f"{k}: {int(bad.sum())}/{bad.size} elements outside rtol {RTOL} (+{ATOL_FRAC}*{scale:.3e}); worst {err.max():.3e}"


def nan_colour_golden_names():
    """Render fixtures in which a NaN colour or opacity spreads NaN over whole tiles of the image."""
    return [n for n in _names() if n.startswith("nancolour_")]


def nonfinite_golden_names():
    """Render fixtures whose inputs hold NaN / infinite parameters (compared NaN for NaN)."""
    return [n for n in _names() if n.startswith("nonfinite_")]


def assert_close_nan(got, ref, name="", skip_rows=None, **kw):
    """assert_close for arrays that may hold NaN: the NaN positions must coincide and the rest must be
    close.  ``skip_rows``: rows (first axis) left out of the comparison."""
    got = np.array(got, dtype=np.float64, copy=True)
    ref = np.array(ref, dtype=np.float64, copy=True)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    if skip_rows is not None:
        keep = np.ones(got.shape[0], bool)
        keep[skip_rows] = False
        got, ref = got[keep], ref[keep]
    gn, rn = np.isnan(got), np.isnan(ref)
    assert np.array_equal(gn, rn), f"{name}: NaN at {np.argwhere(gn != rn)[:5].tolist()} on one side only"
    got[gn] = 0.0
    ref[rn] = 0.0
    assert_close(got, ref, name, **kw)


def loss_golden_names():
    """Fixtures of the training loss (tests/golden/make_loss_golden.py)."""
    return [n for n in _names() if n.startswith("loss_")]


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


def assert_close(got, ref, name="", rtol=RTOL, atol_frac=ATOL_FRAC, outlier_rows=None, outlier_rtol=0.0,
                 outlier_atol_frac=0.0):
    """|got - ref| <= rtol*|ref| + atol_frac*max|ref| for every element.

    ``outlier_rows`` = (k, f) lets the elements of at most max(k, f * rows) ROWS (first axis: Gaussians) miss the
    bound, provided they still meet the looser ``outlier_rtol`` / ``outlier_atol_frac`` bound.  It exists for one
    documented reason, and test_every_gradient_outlier_sits_on_a_clamp_threshold measures it (none of 200 k
    Gaussians at 800x800, 3 of a million at 1080p, each of them a Gaussian the oracle itself moves).
    alpha = clamp(opacity*G, 0.01, 0.99) passes gradient only inside the interval, so a pixel whose
    opacity*G lies within one ulp of 0.01 is a DISCRETE decision, and two correct fp32 evaluations
    of G (the GPU's exp2 of a pre-scaled exponent, the CPU's expf) can decide it differently.  A
    render at 1 M Gaussians / 1080p evaluates 1.4e9 pixel-Gaussian pairs and a handful of them sit
    that close to the threshold; for a sharp splat that passes ~20 pixels one such pixel is worth ~1 % of
    its position gradient.  Rounding noise proper is ~1e-7 of max|ref| and needs no allowance.
    """
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    # non-finite values must sit at the same positions (NaN for NaN, infinities with their sign); the
    # tolerance applies to the finite remainder
    gn, rn = np.isnan(got), np.isnan(ref)
    assert np.array_equal(gn, rn), f"{name}: NaN at {np.argwhere(gn != rn)[:5].tolist()} on one side only"
    gi, ri = np.isinf(got), np.isinf(ref)
    assert np.array_equal(gi, ri) and np.array_equal(got[gi], ref[ri]), f"{name}: infinities differ"
    finite = ~(gn | gi)
    if not finite.all():
        got, ref = np.where(finite, got, 0.0), np.where(finite, ref, 0.0)
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    tol = rtol * np.abs(ref) + atol_frac * scale
    err = np.abs(got - ref)
    bad = err > tol
    rows = bad.shape[0] if bad.ndim else 1
    bad_rows = int(bad.reshape(rows, -1).any(axis=1).sum()) if bad.size else 0
    allowed = max(int(outlier_rows[0]), int(np.ceil(outlier_rows[1] * rows))) if outlier_rows else 0
    worse = err > np.maximum(tol, outlier_rtol * np.abs(ref) + outlier_atol_frac * scale)
    if bad_rows > allowed or worse.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.size} elements in {bad_rows} rows outside rtol={rtol} (+{atol_frac}*max), {allowed} rows allowed, "
            f"{int(worse.sum())} beyond the outlier bound; worst at {i}: "
            f"got {got[i]!r} ref {ref[i]!r} err {err[i]:.3e} tol {tol[i]:.3e} (max|ref|={scale:.3e})")


# Gradients compared across devices: the gradients of at most 2 Gaussians, or 4 in a million, may differ by a
# clamp-threshold pixel (see assert_close), and then by no more than a few per cent of the element.
GRAD_OUTLIERS = dict(outlier_rows=(2, 4e-6), outlier_rtol=0.05, outlier_atol_frac=2e-3)
# ... and only where the number of pixel-Gaussian pairs makes such a pixel likely: the fixtures and the small and
# mid-size scenes (below 100 000 Gaussians) are held to the plain bound, every row.
GRAD_OUTLIERS_FROM_ROWS = 100_000


def grad_tolerance(rows: int) -> dict:
    """Keyword arguments of assert_close for a gradient tensor of ``rows`` Gaussians."""
    return dict(GRAD_OUTLIERS) if rows >= GRAD_OUTLIERS_FROM_ROWS else {}


def assert_bit_equal(got, ref, name=""):
    got = np.ascontiguousarray(got)
    ref = np.ascontiguousarray(ref)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    assert got.dtype == ref.dtype, f"{name}: dtype {got.dtype} vs {ref.dtype}"
    if got.dtype.kind == "f":
        same = got.view(np.uint32) == ref.view(np.uint32)
    else:
        same = got == ref
    assert same.all(), f"{name}: {int((~same).sum())}/{same.size} elements differ"
