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


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


def assert_close(got, ref, name="", rtol=RTOL, atol_frac=ATOL_FRAC, outlier_frac=0.0, outlier_factor=1.0):
    """|got - ref| <= rtol*|ref| + atol_frac*max|ref| for every element.

    ``outlier_frac`` > 0 would let that fraction of the elements miss the bound by up to
    ``outlier_factor``; no test uses it (the HIP path sits at the reference's own rounding-noise
    level, about 1e-7 of max|ref|, once the backward divides the transmittance with a correctly
    rounded quotient).
    """
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    assert np.isfinite(got).all() == np.isfinite(ref).all(), f"{name}: non-finite mismatch"
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    tol = rtol * np.abs(ref) + atol_frac * scale
    err = np.abs(got - ref)
    bad = err > tol
    allowed = int(np.ceil(outlier_frac * bad.size)) if outlier_frac > 0 else 0
    worse = err > outlier_factor * tol
    if int(bad.sum()) > allowed or worse.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.size} elements outside rtol={rtol} (+{atol_frac}*max), {allowed} allowed, "
            f"{int(worse.sum())} beyond x{outlier_factor}; worst at {i}: "
            f"got {got[i]!r} ref {ref[i]!r} err {err[i]:.3e} tol {tol[i]:.3e} (max|ref|={scale:.3e})")


# Extra keyword arguments for gradient comparisons across devices (none needed).
GRAD_OUTLIERS = dict()


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
