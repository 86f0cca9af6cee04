"""Renders the bench's minibatch repeatedly through the pipelined path (three streams) and compares every gradient,
bit for bit, with the same minibatch rendered on one stream: a buffer handed out again too early, or a missing wait
between the streams, shows as a difference.  Usage: python tools/stress_overlap.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsplat_amd  # noqa: F401
from gsplat_amd.minibatch import render_views
from gsplat_amd.rasterizer import ViewSpec
from gsplat_amd.scenes import synthetic_scene
import bench

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
n, w, h = 1_000_000, 1920, 1080
scene = synthetic_scene(n, w, h, 3, seed=0)
params = [t.to(dev) for t in scene.params()]
views = [ViewSpec.from_camera(c, 3, 16) for c in bench.minibatch_cameras(scene, 8, dev)]
gen = torch.Generator().manual_seed(1)
cots = [(torch.rand(w, h, 3, generator=gen) / 8).to(dev) for _ in range(8)]
_, ref, ref_screens = render_views(params, views, lambda i, st: cots[i], overlap=False)
ref = [g.clone() for g in ref]
ref_screens = [g.clone() for g in ref_screens]
torch.cuda.synchronize()
# first without a host synchronisation between the steps (the host runs ahead as in training), compared at the end
kept = []
for it in range(min(iters, 16)):
    _, got, screens = render_views(params, views, lambda i, st: cots[i])
    kept.append([g.clone() for g in list(got) + list(screens)])
torch.cuda.synchronize()
ahead = sum(not all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(k, ref + ref_screens)) for k in kept)
print(f"{len(kept)} steps back to back, {ahead} differ", flush=True)
del kept
bad = ahead
for it in range(iters):
    junk = [torch.empty(1 << 20, device=dev).normal_() for _ in range(4)]     # churn in the allocator between steps
    del junk
    _, got, screens = render_views(params, views, lambda i, st: cots[i])
    same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(list(got) + list(screens), ref + ref_screens))
    bad += not same
    if it % 10 == 9:
        print(f"{it + 1} iterations, {bad} differ", flush=True)
print("ok" if bad == 0 else f"{bad} of {iters} iterations differ from the one-stream result")
sys.exit(1 if bad else 0)
