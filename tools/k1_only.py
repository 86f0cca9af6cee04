"""Times gsr_preprocess_fwd alone (bench scene); safe for experimental variants of that kernel."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsplat_amd  # noqa
from gsplat_amd import _lib
from gsplat_amd.rasterizer import ViewSpec
from gsplat_amd.scenes import synthetic_scene
dev = torch.device("cuda:0")
n, W, H = 1000000, 1920, 1080
sc = synthetic_scene(n, W, H, 3, 0)
p = [t.to(dev) for t in sc.params()]
view = ViewSpec.from_camera(sc.camera(device=dev), 3, 16)
lib = _lib.load()
cv = _lib.make_view(n, 3, W, H, 16, view.w2m, view.proj, view.cam_pos)
i32 = dict(dtype=torch.int32, device=dev); f32 = dict(dtype=torch.float32, device=dev)
mask = torch.empty(n, dtype=torch.uint8, device=dev); key = torch.empty(n, **i32)
pos = torch.empty(n, 3, **f32); cov = torch.empty(n, 4, **f32); rad = torch.empty(n, **f32)
bbox = torch.empty(n, 4, **i32); nt = torch.empty(n, **i32); rec = torch.empty(n, 12, **f32)
cnt = torch.zeros(8, **i32)
s = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(lib.gsr_preprocess_fwd(cv, p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(), p[3].data_ptr(), p[4].data_ptr(), None,
               mask.data_ptr(), key.data_ptr(), pos.data_ptr(), cov.data_ptr(), rad.data_ptr(), bbox.data_ptr(), nt.data_ptr(),
               rec.data_ptr(), cnt.data_ptr(), s), "k1")
for _ in range(3): run()
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): run()
b.record(); torch.cuda.synchronize()
print("preprocess_fwd alone: %.4f ms" % (a.elapsed_time(b) / 20))
