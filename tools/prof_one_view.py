"""Profiling driver: a few forward+backward renders of the bench workload (one view), nothing else."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsplat_amd  # noqa
from gsplat_amd.rasterizer import ViewSpec, backward_pipeline, forward_pipeline
from gsplat_amd.scenes import synthetic_scene
n = int(os.environ.get("N", 1000000)); W = int(os.environ.get("W", 1920)); H = int(os.environ.get("H", 1080))
tile = int(os.environ.get("TILE", 16)); reps = int(os.environ.get("REPS", 3))
dev = torch.device("cuda:0")
sc = synthetic_scene(n, W, H, 3, 0)
p = [t.to(dev) for t in sc.params()]
if os.environ.get("CAM", "identity") == "ring0":
    # view 0 of bench.py's minibatch (a ring of radius 0.15 around the scene's base pose)
    import numpy as np
    from gsplat_amd.camera import Camera
    cam = Camera(np.array([0.15, 0.0, 0.0]), np.array([1.0, 0.0, 0.0, 0.0]), sc.fx, sc.fy, W / 2, H / 2, W, H, device=dev)
else:
    cam = sc.camera(device=dev)
view = ViewSpec.from_camera(cam, 3, tile)
cot = torch.rand(W, H, 3, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(reps):
    st = forward_pipeline(*p, None, view)
    backward_pipeline(st, *p, cot)
torch.cuda.synchronize()
print("pairs", st.num_pairs)
