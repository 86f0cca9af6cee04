"""Per-stage device times (HIP events) of the bench workload, one view, forward + backward."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsplat_amd  # noqa
from gsplat_amd.rasterizer import StageTimer, ViewSpec, backward_pipeline, forward_pipeline
from gsplat_amd.scenes import synthetic_scene
n = int(os.environ.get("N", 1000000)); W = int(os.environ.get("W", 1920)); H = int(os.environ.get("H", 1080))
tile = int(os.environ.get("TILE", 16)); reps = int(os.environ.get("REPS", 10))
dev = torch.device("cuda:0")
sc = synthetic_scene(n, W, H, 3, 0)
p = [t.to(dev) for t in sc.params()]
view = ViewSpec.from_camera(sc.camera(device=dev), 3, tile)
cot = torch.rand(W, H, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(3):
    st = forward_pipeline(*p, None, view)
    g = backward_pipeline(st, *p, cot)
torch.cuda.synchronize()
with StageTimer() as t:
    for _ in range(reps):
        st = forward_pipeline(*p, None, view)
        g = backward_pipeline(st, *p, cot)
s = t.summary()
tot = 0.0
for k, (c, ms) in s.items():
    print("%-22s %8.4f ms" % (k, ms / c)); tot += ms / c
print("%-22s %8.4f ms   pairs %d" % ("total", tot, st.num_pairs))
print("checksums img %.6e  g_coords %.6e  g_sh %.6e" % (st.img.double().sum().item(), g[0].double().abs().sum().item(),
                                                       g[4].double().abs().sum().item()))
