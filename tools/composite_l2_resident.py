"""Experiment: compositing kernels on a stream that stays L2-resident (every tile walks one of four
short lists), to separate instruction-issue cost from memory effects.  Needs a variant library whose
kernels take pixel coordinates relative to the tile (GSR_LIB_PATH), so that one real list, shifted to
the origin, is meaningful for every tile; the product library would see the splats far away."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsplat_amd  # noqa
from gsplat_amd import _lib
from gsplat_amd.rasterizer import ViewSpec, forward_pipeline
from gsplat_amd.scenes import synthetic_scene

dev = torch.device("cuda:0")
n_list = int(os.environ.get("LIST", 640)); W = H = 2048
# real records: take them from a real render so that the flagged fraction and the value ranges are realistic
lib = _lib.load()
sc = synthetic_scene(1000000, 1920, 1080, 3, 0)
p = [t.to(dev) for t in sc.params()]
st = forward_pipeline(*p, None, ViewSpec.from_camera(sc.camera(device=dev), 3, 16))
nbh_src = (1080 + 15) // 16
off = st.tile_offsets.cpu().numpy().astype(np.int64)
cand = np.nonzero((off[1:] - off[:-1] >= n_list) & (np.arange(len(off) - 1) % nbh_src < nbh_src - 1))[0]
t0 = int(cand[len(cand) // 2])
rec = st.stream_rec[off[t0]:off[t0] + n_list].clone()
rec[:, 1] -= 16.0 * (t0 // nbh_src)   # x
rec[:, 0] -= 16.0 * (t0 % nbh_src)    # y
rec[:, 9] = torch.arange(n_list, device=dev, dtype=torch.int32).view(torch.float32)   # slots inside the small pairgrad
fl = rec[:, 11].view(torch.int32)
flag_frac = float((fl == 3).float().mean())
one_frac = float(((fl == 1) | (fl == 2)).float().mean())
# move the list to tile (0,0) coordinates: every tile sees the same offsets relative to its corner
stream = rec.repeat(4, 1).contiguous()
T = (W // 16) * (H // 16)
o = np.zeros(T + 1, np.int64)
pat = np.array([0, 1, 2, 3, 4]) * n_list
o[:] = np.resize(pat, T + 1)
tile_offsets = torch.tensor(o, dtype=torch.int32, device=dev)
active = int(((o[1:] - o[:-1]) > 0).sum())
view = _lib.make_view(1, 3, W, H, 16, [0.0] * 16, [0.0] * 16, [0.0] * 3)
img = torch.empty(W, H, 3, device=dev); fT = torch.empty(W, H, device=dev)
dimg = torch.rand(W, H, 3, device=dev)
I = 4 * n_list
pairgrad = torch.empty(I, 12, device=dev)
s = torch.cuda.current_stream().cuda_stream
def fwd(): _lib.check(lib.gsr_composite_fwd(view, tile_offsets.data_ptr(), None, stream.data_ptr(), img.data_ptr(), fT.data_ptr(), s), "fwd")
def bwd(): _lib.check(lib.gsr_composite_bwd(view, tile_offsets.data_ptr(), None, stream.data_ptr(), fT.data_ptr(), dimg.data_ptr(), I, pairgrad.data_ptr(), s), "bwd")
for name, f in (("fwd", fwd), ("bwd", bwd)):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    we = active * n_list
    print("%s %.4f ms  active waves %d  wave-entries %d  -> %.1f cycles/wave-entry/SIMD at 2.4 GHz (both halves low %.2f, one %.2f)"
          % (name, ms, active, we, ms * 1e-3 * 2.4e9 / (we / 1024.0), flag_frac, one_frac))
