"""Generates tests/golden/loss_*.npz by running the REFERENCE loss (Utils/Loss.py, combined as in
Trainer.py:35-38 with loss_lambda = 0.2) on seeded [W,H,3] image pairs.

Run in the build container only (needs /root/reference):

    python tests/golden/make_loss_golden.py

Each fixture holds pred, gt, the reference's l1, ssim and loss (fp32 scalars) and d loss / d pred
from the reference's autograd.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
LAMBDA = 0.2


def smooth_image(w, h, gen, hi):
    """Low-frequency picture plus a little noise, values in [0, hi]: flat and textured areas both occur."""
    xs = torch.linspace(0, 1, w)[:, None, None]
    ys = torch.linspace(0, 1, h)[None, :, None]
    ph = torch.rand(3, generator=gen)[None, None, :] * 6.28
    img = 0.5 + 0.35 * torch.sin(5 * xs + 3 * ys + ph) * torch.cos(4 * ys - 2 * xs + ph)
    img = img + 0.05 * torch.rand(w, h, 3, generator=gen)
    return (img.clamp(0, 1) * hi).float()


# name -> (W, H, kind of pred, kind of gt)
CASES = {
    "loss_rand_40x30": (40, 30, "rand", "rand"),              # white noise against white noise
    "loss_smooth_64x48": (64, 48, "smooth", "smooth"),        # smooth pictures, 0..1 targets
    "loss_target255_48x64": (48, 64, "smooth", "rand255"),    # targets as the data loader gives them (0..255)
    "loss_photo255_48x40": (48, 40, "smooth", "smooth255"),   # smooth 0..255 target, rounded like an 8-bit picture
    "loss_short_17x7": (17, 7, "rand", "smooth"),             # fewer rows than the window
    "loss_column_1x33": (1, 33, "rand", "rand"),              # one image column
    "loss_equal_24x20": (24, 20, "quant", "quant_shift"),     # many exact ties pred == gt (sign(0) = 0)
}


def make(kind, w, h, gen):
    if kind == "rand":
        return torch.rand(w, h, 3, generator=gen)
    if kind == "rand255":
        return torch.floor(torch.rand(w, h, 3, generator=gen) * 256).clamp(0, 255)
    if kind == "smooth":
        return smooth_image(w, h, gen, 1.0)
    if kind == "smooth255":
        return torch.round(smooth_image(w, h, gen, 255.0))
    if kind == "quant":
        return torch.floor(torch.rand(w, h, 3, generator=gen) * 4) / 4
    if kind == "quant_shift":
        return torch.floor(torch.rand(w, h, 3, generator=gen) * 4) / 4
    raise KeyError(kind)


def main():
    assert os.path.isdir(REFERENCE), "needs /root/reference"
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE)
    from Utils.Loss import l1_loss, ssim_loss  # the reference's own functions

    for seed, (name, (w, h, kp, kg)) in enumerate(CASES.items()):
        gen = torch.Generator().manual_seed(500 + seed)
        pred = make(kp, w, h, gen).requires_grad_(True)
        gt = make(kg, w, h, gen)
        ssim = ssim_loss(pred, gt)
        l1 = l1_loss(pred, gt)
        loss = (1 - LAMBDA) * l1 + LAMBDA * (1 - ssim)   # Trainer.py:38
        loss.backward()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, pred=pred.detach().numpy(), gt=gt.numpy(), loss_lambda=np.float32(LAMBDA),
                            ref_l1=np.float32(l1.item()), ref_ssim=np.float32(ssim.item()),
                            ref_loss=np.float32(loss.item()), ref_dpred=pred.grad.numpy())
        print(f"{name}: l1 {l1.item():.6f} ssim {ssim.item():.6f} loss {loss.item():.6f} "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
