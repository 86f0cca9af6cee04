"""The training loss of one view on the GPU: host side of ``gsr_loss_fwd_bwd``.

Mirrors the reference's loss code: ``l1_loss`` (``Utils/Loss.py:4-5``), ``ssim_loss``
(``Utils/Loss.py:69-77``) and ``Trainer.compute_loss`` (``Trainer.py:35-38``), for ``pred`` and
``gt`` of shape ``[W,H,3]`` as ``GaussianRenderer.render`` and ``DataLoader.sample`` return them.
As in the reference, SSIM reads such a tensor as W channels of ``[H,3]`` planes
(``Utils/Loss.py:70``).  One kernel launch produces the loss and dloss/dpred; the autograd wrappers
below keep that gradient for their backward.  There is no CPU path.
"""
import torch

from . import _lib
from .rasterizer import _call, _stream


def _check_pair(pred: torch.Tensor, gt: torch.Tensor):
    for name, t in (("pred", pred), ("gt", gt)):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError(f"{name} must be a tensor on the GPU (there is no CPU path)")
        if t.dtype != torch.float32:
            raise TypeError(f"{name} must be float32, got {t.dtype}")
    if pred.dim() != 3 or pred.shape[2] != 3:
        raise ValueError(f"pred has shape {tuple(pred.shape)}, expected [W,H,3]")
    if gt.shape != pred.shape:
        raise ValueError(f"gt has shape {tuple(gt.shape)}, pred {tuple(pred.shape)}")
    return pred.contiguous(), gt.contiguous()


def loss_and_grad(pred, gt, loss_lambda: float = 0.2, grad_scale: float = 1.0, want_grad: bool = True):
    """``(out, dpred)``: ``out`` is a device tensor ``[loss, l1, ssim]`` with
    ``loss = (1 - loss_lambda) * l1 + loss_lambda * (1 - ssim)`` (``Trainer.py:38``) and ``dpred`` is
    ``grad_scale * dloss/dpred`` (``None`` unless ``want_grad``).  Nothing is read back to the host."""
    lib = _lib.load()
    pred, gt = _check_pair(pred.detach(), gt.detach())
    w, h = int(pred.shape[0]), int(pred.shape[1])
    dev = pred.device
    out = torch.empty(3, dtype=torch.float32, device=dev)
    dpred = torch.empty_like(pred) if want_grad else None
    ws_bytes = lib.gsr_loss_workspace_bytes(w, h)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _call("gsr_loss_fwd_bwd", lib.gsr_loss_fwd_bwd, pred.data_ptr(), gt.data_ptr(), w, h, float(loss_lambda),
          float(grad_scale), out.data_ptr(), None if dpred is None else dpred.data_ptr(), ws.data_ptr(),
          ws_bytes, _stream(dev))
    return out, dpred


class _Loss(torch.autograd.Function):
    """``out[pick]`` of ``loss_and_grad`` with the matching gradient saved for the backward."""

    @staticmethod
    def forward(ctx, pred, gt, loss_lambda, grad_scale, pick):
        out, dpred = loss_and_grad(pred, gt, loss_lambda, grad_scale, want_grad=ctx.needs_input_grad[0])
        ctx.dpred = dpred
        return out[pick].clone()

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.dpred is None:
            return None, None, None, None, None
        return ctx.dpred * grad_out, None, None, None, None


def compute_loss(pred, gt, loss_lambda: float = 0.2):
    """``Trainer.compute_loss`` (``Trainer.py:35-38``)."""
    return _Loss.apply(pred, gt, float(loss_lambda), 1.0, 0)


def l1_loss(pred, gt):
    """``Utils/Loss.py:4-5``."""
    return _Loss.apply(pred, gt, 0.0, 1.0, 1)


def ssim_loss(img1, img2, window_size: int = 11, size_average: bool = True):
    """``Utils/Loss.py:69-77`` for ``[W,H,3]`` inputs, the default window and the mean."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("only window_size=11 with size_average=True (the reference trainer's call)")
    return _Loss.apply(img1, img2, 1.0, -1.0, 2)
