"""Device time of the loss kernel at the bench size (1920x1080), with and without the gradient."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsplat_amd  # noqa: E402,F401
from gsplat_amd.loss import loss_and_grad  # noqa: E402

dev = torch.device("cuda:0")
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
pred = torch.rand(w, h, 3, device=dev)
gt = torch.floor(torch.rand(w, h, 3, device=dev) * 256)
for want in (True, False):
    for _ in range(3):
        loss_and_grad(pred, gt, 0.2, 1.0, want)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    a.record()
    for _ in range(reps):
        loss_and_grad(pred, gt, 0.2, 1.0, want)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    byt = w * h * 3 * 4 * (3 if want else 2)
    print(f"{w}x{h} grad={want}: {ms * 1e3:.1f} us per call, {byt / ms / 1e6:.1f} GB/s of {byt / 1e6:.1f} MB")
