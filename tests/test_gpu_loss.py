"""GPU parity tests of the training loss: the HIP kernel (through the C ABI) against the fixtures
made by the reference's own Utils/Loss.py and against the CPU oracle.  Floats within fp32 rtol 1e-4
(util.assert_close); the scalar terms, which are means over the whole image, within 1e-5."""
import numpy as np
import pytest
import torch

from util import assert_close, load_golden, loss_golden_names

pytestmark = pytest.mark.gpu

# dloss/dpred is a window sum of terms of both signs; an element is held to rtol 1e-4 of itself plus
# 1e-5 of the largest gradient in the image.
GRAD_TOL = dict(rtol=1e-4, atol_frac=1e-5)


def _gpu():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def run_hip(pred, gt, lam=0.2, grad_scale=1.0, want_grad=True):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.loss import loss_and_grad
    dev = _gpu()
    p = torch.as_tensor(np.ascontiguousarray(pred), dtype=torch.float32).to(dev)
    g = torch.as_tensor(np.ascontiguousarray(gt), dtype=torch.float32).to(dev)
    out, dpred = loss_and_grad(p, g, lam, grad_scale, want_grad)
    out = out.cpu().numpy()
    return {"loss": float(out[0]), "l1": float(out[1]), "ssim": float(out[2]),
            "dpred": None if dpred is None else dpred.cpu().numpy()}


def check_against_oracle(oracle_mod, pred, gt, lam=0.2):
    h = run_hip(pred, gt, lam)
    o = oracle_mod.loss(pred, gt, 1.0 - lam, -lam)
    assert_close(h["l1"], o["l1"], "l1", rtol=1e-5, atol_frac=0)
    assert_close(h["ssim"], o["ssim"], "ssim", rtol=1e-5, atol_frac=0)
    assert_close(h["loss"], (1.0 - lam) * o["l1"] + lam * (1.0 - o["ssim"]), "loss", rtol=1e-5, atol_frac=0)
    assert_close(h["dpred"], o["dpred"], "dloss/dpred", **GRAD_TOL)
    return h, o


@pytest.mark.parametrize("name", loss_golden_names())
def test_loss_matches_reference_fixture(name):
    g = load_golden(name)
    h = run_hip(g["pred"], g["gt"], float(g["loss_lambda"]))
    assert_close(h["l1"], g["ref_l1"], "l1", rtol=1e-5, atol_frac=0)
    assert_close(h["ssim"], g["ref_ssim"], "ssim", rtol=1e-5, atol_frac=0)
    assert_close(h["loss"], g["ref_loss"], "loss", rtol=1e-5, atol_frac=0)
    assert_close(h["dpred"], g["ref_dpred"], "dloss/dpred", **GRAD_TOL)


@pytest.mark.parametrize("w,h", [(3, 246), (5, 247), (2, 256), (7, 493), (4, 1), (1, 11), (33, 739)])
def test_loss_matches_oracle_across_row_chunks(oracle_mod, w, h):
    """A workgroup owns 246 rows of one image column; heights at and around multiples of that, and
    heights below the window size."""
    rng = np.random.default_rng(w * 1000 + h)
    pred = rng.random((w, h, 3), dtype=np.float32)
    gt = (0.6 * pred + 0.4 * rng.random((w, h, 3), dtype=np.float32)).astype(np.float32)
    check_against_oracle(oracle_mod, pred, gt)


@pytest.mark.parametrize("lam", [0.0, 0.2, 1.0])
def test_loss_lambda_and_scale(oracle_mod, lam):
    rng = np.random.default_rng(5)
    pred = rng.random((20, 300, 3), dtype=np.float32)
    gt = rng.random((20, 300, 3), dtype=np.float32)
    h, o = check_against_oracle(oracle_mod, pred, gt, lam)
    scaled = run_hip(pred, gt, lam, grad_scale=-0.125)
    assert_close(scaled["dpred"], -0.125 * h["dpred"].astype(np.float64), "scaled gradient", rtol=1e-6, atol_frac=1e-7)
    assert scaled["loss"] == h["loss"]
    none = run_hip(pred, gt, lam, want_grad=False)
    assert none["dpred"] is None and none["loss"] == h["loss"] and none["ssim"] == h["ssim"]


def test_loss_of_identical_images():
    """pred == gt: l1 = 0, ssim = 1, loss = 0 and the l1 term of the gradient is exactly zero."""
    rng = np.random.default_rng(6)
    img = rng.random((16, 280, 3), dtype=np.float32)
    h = run_hip(img, img.copy(), 0.0)
    assert h["l1"] == 0.0 and h["loss"] == 0.0
    assert not h["dpred"].any()
    h = run_hip(img, img.copy(), 0.2)
    assert abs(h["ssim"] - 1.0) < 1e-6
    assert abs(h["dpred"]).max() < 1e-8   # SSIM is at its maximum


def test_loss_full_size_against_oracle(oracle_mod):
    """The bench size, 1920x1080, with a 0..255 target as the data loader makes them."""
    gen = torch.Generator().manual_seed(9)
    pred = torch.rand(1920, 1080, 3, generator=gen).numpy()
    gt = torch.floor(torch.rand(1920, 1080, 3, generator=gen) * 256).numpy()
    check_against_oracle(oracle_mod, pred, gt)
    gt01 = (0.5 * pred + 0.5 * torch.rand(1920, 1080, 3, generator=gen).numpy()).astype(np.float32)
    check_against_oracle(oracle_mod, pred, gt01)


def test_autograd_wrappers_mirror_the_reference_functions(oracle_mod):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.loss import compute_loss, l1_loss, ssim_loss
    dev = _gpu()
    rng = np.random.default_rng(7)
    pred_np = rng.random((24, 260, 3), dtype=np.float32)
    gt_np = rng.random((24, 260, 3), dtype=np.float32)
    gt = torch.as_tensor(gt_np).to(dev)

    def grad_of(fn, factor):
        p = torch.as_tensor(pred_np).to(dev).requires_grad_(True)
        v = fn(p, gt)
        (v * factor).backward()
        return float(v.detach()), p.grad.cpu().numpy()

    o = oracle_mod.loss(pred_np, gt_np, 0.8, -0.2)
    v, g = grad_of(lambda p, t: compute_loss(p, t, 0.2), 3.0)
    assert_close(v, 0.8 * o["l1"] + 0.2 * (1.0 - o["ssim"]), "compute_loss", rtol=1e-5, atol_frac=0)
    assert_close(g, 3.0 * o["dpred"].astype(np.float64), "d compute_loss", **GRAD_TOL)
    o1 = oracle_mod.loss(pred_np, gt_np, 1.0, 0.0)
    v, g = grad_of(l1_loss, 1.0)
    assert_close(v, o1["l1"], "l1_loss", rtol=1e-5, atol_frac=0)
    assert_close(g, o1["dpred"], "d l1_loss", **GRAD_TOL)
    o2 = oracle_mod.loss(pred_np, gt_np, 0.0, 1.0)
    v, g = grad_of(ssim_loss, 1.0)
    assert_close(v, o2["ssim"], "ssim_loss", rtol=1e-5, atol_frac=0)
    assert_close(g, o2["dpred"], "d ssim_loss", **GRAD_TOL)
    # no gradient requested: nothing is kept
    assert not compute_loss(torch.as_tensor(pred_np).to(dev), gt).requires_grad


def test_loss_rejects_bad_arguments():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.loss import loss_and_grad
    dev = _gpu()
    a = torch.rand(8, 8, 3, device=dev)
    with pytest.raises(RuntimeError):
        loss_and_grad(a.cpu(), a.cpu())
    with pytest.raises(ValueError):
        loss_and_grad(a, torch.rand(8, 9, 3, device=dev))
    with pytest.raises(ValueError):
        loss_and_grad(torch.rand(8, 8, 4, device=dev), torch.rand(8, 8, 4, device=dev))
    with pytest.raises(RuntimeError):
        loss_and_grad(a, a, loss_lambda=1.5)
