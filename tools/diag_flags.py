"""Diagnostic: check the 'alpha pinned at 0.01 on the whole tile' flags of the record stream against
an exact per-pixel evaluation, for the Gaussians whose gradients disagree with the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gsplat_amd, oracle
from gsplat_amd.scenes import synthetic_scene
from gsplat_amd.rasterizer import ViewSpec, forward_pipeline, backward_pipeline
from test_gpu_parity import _scene_arrays
n, w, h, deg, tile = 1_000_000, 1920, 1080, 3, 16
sc = synthetic_scene(n, w, h, deg, 0)
a = _scene_arrays(sc)
cot = torch.rand(w, h, 3, generator=torch.Generator().manual_seed(7)).numpy()
dev = torch.device("cuda:0")
t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(dev)
p = [t(a["coords"]), t(a["scale"]), t(a["rotation"]), t(a["opacity"].reshape(-1, 1)), t(a["sh"])]
view = ViewSpec(deg, w, h, tile, tuple(a["w2m"].reshape(16).tolist()), tuple(a["proj"].reshape(16).tolist()), tuple(a["cam_pos"].tolist()))
st = forward_pipeline(*p, None, view)
g = backward_pipeline(st, *p, t(cot))
torch.cuda.synchronize()
ora = oracle.render(a["coords"], a["scale"], a["rotation"], a["opacity"], a["sh"], deg, w, h, tile, a["w2m"], a["proj"], a["cam_pos"], dimg=cot)
gc = g[0].cpu().numpy(); rc = ora["g_coords"]
err = np.abs(gc - rc).max(1); tol = 1e-4 * np.abs(rc).max(1) + 2e-6 * np.abs(rc).max()
bad = np.argsort(-(err - tol))[:6]
stream = st.stream_rec.cpu().numpy()[:st.num_pairs]
owner = st.sorted_gauss.cpu().numpy()[:st.num_pairs]
off = st.tile_offsets.cpu().numpy().astype(np.int64)
nbh = (h + tile - 1) // tile
con = ora["conic"]; pos = ora["pos2d"]; op = ora["opacity_act"]
print("worst Gaussians", bad, "err", err[bad])
for gi in bad:
    ps = np.nonzero(owner == gi)[0]
    print(f"g={gi} pos=({pos[gi,0]:.2f},{pos[gi,1]:.2f}) opac={op[gi]:.4f} conic={con[gi]} radius={ora['radius_full'][gi]:.2f} pairs={len(ps)} hip={gc[gi]} ref={rc[gi]}")
    for pp in ps:
        tid = np.searchsorted(off, pp, side="right") - 1
        xi, yi = divmod(tid, nbh)
        xs = np.arange(xi * tile, min((xi + 1) * tile, w)); ys = np.arange(yi * tile, min((yi + 1) * tile, h))
        dx = pos[gi, 0] - xs[:, None].astype(np.float64); dy = pos[gi, 1] - ys[None, :].astype(np.float64)
        ex = (dx * con[gi, 0] + dy * con[gi, 2]) * dx + (dx * con[gi, 1] + dy * con[gi, 3]) * dy
        raw = op[gi] * np.exp(-0.5 * ex)
        flagged = stream[pp, 5] < 0 or (stream[pp, 5] == 0 and np.signbit(stream[pp, 5]))
        print(f"   tile ({xi},{yi}) flagged={flagged} max raw={raw.max():.6f} n_pass={(raw >= 0.01).sum()}")
