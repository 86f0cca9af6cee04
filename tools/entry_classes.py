"""Fractions of list entries by the path the compositing backward takes for them (bench scene)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsplat_amd  # noqa
from gsplat_amd.rasterizer import ViewSpec, forward_pipeline
from gsplat_amd.scenes import synthetic_scene
dev = torch.device("cuda:0")
W, H = 1920, 1080
sc = synthetic_scene(1000000, W, H, 3, 0)
p = [t.to(dev) for t in sc.params()]
st = forward_pipeline(*p, None, ViewSpec.from_camera(sc.camera(device=dev), 3, 16))
off = st.tile_offsets.cpu().long()
nbh = (H + 15) // 16
g = torch.Generator().manual_seed(0)
tiles = torch.randint(0, off.numel() - 1, (200,), generator=g).tolist()
tot = flagged = geom = nongeom = 0
npass = []; half_tb = []; half_lr = []; bands = []
runs = []
for t in tiles:
    b, e = int(off[t]), int(off[t + 1])
    if e <= b: continue
    r = st.stream_rec[b:e]
    xi, yi = t // nbh, t % nbh
    xs = torch.arange(xi * 16, min(xi * 16 + 16, W), device=dev).float()
    ys = torch.arange(yi * 16, min(yi * 16 + 16, H), device=dev).float()
    # stream layout: 0 y, 1 x, 2 D', 3 A', 4 opacity, 5 B', 11 flags
    dx = r[:, 1, None, None] - xs[None, :, None]
    dy = r[:, 0, None, None] - ys[None, None, :]
    q = r[:, 3, None, None] * dx * dx + r[:, 5, None, None] * dx * dy + r[:, 2, None, None] * dy * dy
    raw = r[:, 4][:, None, None] * torch.exp2(q)
    ps = ((raw >= 0.01) & (raw <= 0.99)).flatten(1)
    anyp = ps.any(1)
    fl = r[:, 11].view(torch.int32) == 3
    tot += e - b; flagged += int(fl.sum()); geom += int((~fl & anyp).sum()); nongeom += int((~fl & ~anyp).sum())
    npass.append(ps[~fl & anyp].float().sum(1))
    low = (raw < 0.01)[~fl]                      # [n, x, y]
    if low.shape[1] == 16 and low.shape[2] == 16:
        top, bot = low[:, :, :8].flatten(1).all(1), low[:, :, 8:].flatten(1).all(1)
        lef, rig = low[:, :8, :].flatten(1).all(1), low[:, 8:, :].flatten(1).all(1)
        half_tb.append((top.float() + bot.float())); half_lr.append((lef.float() + rig.float()))
        q = [low[:, :, 4 * b:4 * b + 4].flatten(1).all(1).float() for b in range(4)]
        bands.append(sum(q))
    f = fl.cpu().tolist(); k = 0
    while k < len(f):
        if f[k]:
            j = k
            while j < len(f) and f[j]: j += 1
            runs.append(j - k); k = j
        else: k += 1
npass = torch.cat(npass)
print("entries %d: flagged %.3f  geometric %.3f  clamped everywhere but not flagged %.3f" % (tot, flagged / tot, geom / tot, nongeom / tot))
print("geometric entries: passing pixels of 256: mean %.1f median %.0f; <=16 px: %.3f, <=64 px: %.3f" % (
    npass.mean(), npass.median(), (npass <= 16).float().mean(), (npass <= 64).float().mean()))
runs = torch.tensor(runs).float()
print("flagged runs: mean length %.2f, entries in runs >= 2: %.3f, >= 4: %.3f" % (runs.mean(), runs[runs >= 2].sum() / runs.sum(), runs[runs >= 4].sum() / runs.sum()))

tb = torch.cat(half_tb); lr = torch.cat(half_lr); bd = torch.cat(bands)
print("non-flagged entries: fully-clamped-low halves (top/bottom split): mean %.3f of 2 ; P(exactly one) %.3f" % (tb.mean(), (tb == 1).float().mean()))
print("non-flagged entries: left/right split: mean %.3f of 2" % lr.mean())
print("non-flagged entries: four 4-row bands clamped-low: mean %.3f of 4" % bd.mean())
