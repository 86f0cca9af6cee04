"""How many 16x16 tiles of the bench scene have a pixel whose final transmittance is below the bounds of the three backward sweep cases."""
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
T = st.final_T
pad = torch.ones(W, 1088 - H, device=dev)
Tm = torch.cat([T, pad], 1).reshape(W // 16, 16, 1088 // 16, 16).amin(dim=(1, 3))
for thr in (1e-20, 1e-26, 1.17549435e-38, 0.0):
    print("tiles with min T <= %g: %.4f   pixels: %.5f" % (thr, float((Tm <= thr).float().mean()), float((T <= thr).float().mean())))
print("T quantiles", torch.quantile(T.flatten()[:4000000].float(), torch.tensor([0.001, 0.01, 0.1, 0.5], device=dev)).tolist())
rec = st.stream_rec[:st.num_pairs]
fl = rec[:, 11].view(torch.int32); top = (fl & 1) != 0; bot = (fl & 2) != 0
print("flags: both %.4f  top only %.4f  bottom only %.4f  none %.4f" % (
    float((top & bot).float().mean()), float((top & ~bot).float().mean()), float((~top & bot).float().mean()), float((~top & ~bot).float().mean())))
