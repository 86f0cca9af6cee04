"""Diagnostic: per-Gaussian stage-3 gradients, HIP vs oracle (fp32 and fp64 pair math)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gsplat_amd, oracle
from gsplat_amd.scenes import synthetic_scene
from gsplat_amd.rasterizer import ViewSpec, forward_pipeline, backward_pipeline
from test_gpu_parity import _scene_arrays
n, w, h, deg, tile = 20000, 320, 200, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 64
sc = synthetic_scene(n, w, h, deg, 3)
a = _scene_arrays(sc)
cot = torch.rand(w, h, 3, generator=torch.Generator().manual_seed(7)).numpy()
dev = torch.device("cuda:0")
t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(dev)
p = [t(a["coords"]), t(a["scale"]), t(a["rotation"]), t(a["opacity"].reshape(-1, 1)), t(a["sh"])]
view = ViewSpec(deg, w, h, tile, tuple(a["w2m"].reshape(16).tolist()), tuple(a["proj"].reshape(16).tolist()), tuple(a["cam_pos"].tolist()))
st = forward_pipeline(*p, None, view)
g = backward_pipeline(st, *p, t(cot))
torch.cuda.synchronize()
pg = st.pairgrad.cpu().numpy().astype(np.float64)[:st.num_pairs]
owner = np.zeros(st.num_pairs, np.int64)     # pairgrad rows are in emission-slot order
owner[st.pair_slot.cpu().numpy()[:st.num_pairs]] = st.sorted_gauss.cpu().numpy()[:st.num_pairs]
S = np.zeros((n, 12)); np.add.at(S, owner, pg)
o32 = oracle.render(a["coords"], a["scale"], a["rotation"], a["opacity"], a["sh"], deg, w, h, tile, a["w2m"], a["proj"], a["cam_pos"], dimg=cot)
o64 = oracle.render(a["coords"], a["scale"], a["rotation"], a["opacity"], a["sh"], deg, w, h, tile, a["w2m"], a["proj"], a["cam_pos"], dimg=cot, precise=True)
def D_from_conicgrad(o):
    C = o["conic"].astype(np.float64).reshape(-1, 2, 2); G = o["d_conic"].reshape(-1, 2, 2)
    return -np.einsum('nji,njk,nlk->nil', C, G, C)   # -C^T G C^T
hip = dict(d_pos=np.stack([-S[:, 0], -S[:, 1]], 1), D=0.5 * np.stack([S[:, 2], S[:, 3], S[:, 3], S[:, 4]], 1).reshape(-1, 2, 2),
           d_o=S[:, 5], d_col=S[:, 6:9])
def pack(o): return dict(d_pos=o["d_pos2d"], D=D_from_conicgrad(o), d_o=o["d_opacity_act"], d_col=o["d_color"])
r32, r64 = pack(o32), pack(o64)
for k in ("d_pos", "D", "d_o", "d_col"):
    mx = np.abs(r64[k]).max()
    e_h = np.abs(hip[k] - r64[k]); e_o = np.abs(r32[k] - r64[k])
    print(f"{k}: max|ref|={mx:.3e}  hip-vs-f64 max={e_h.max()/mx:.2e} mean={e_h.mean()/mx:.2e} | o32-vs-f64 max={e_o.max()/mx:.2e} mean={e_o.mean()/mx:.2e}")
e = np.abs(hip["d_pos"] - r64["d_pos"]).max(1)
worst = np.argsort(-e)[:8]
rad = st.radius_full.cpu().numpy(); nt = st.ntiles.cpu().numpy(); pos = st.pos2d.cpu().numpy(); cov = st.cov2d.cpu().numpy()
for i in worst:
    ev = np.linalg.eigvalsh(0.5 * (cov[i].reshape(2, 2) + cov[i].reshape(2, 2).T))
    print(f"g={i} err={e[i]:.3e} d_pos hip={hip['d_pos'][i]} f64={r64['d_pos'][i]} o32={r32['d_pos'][i]} radius={rad[i]:.1f} nt={nt[i]} pos=({pos[i,0]:.1f},{pos[i,1]:.1f}) sig=({np.sqrt(ev[0]):.2f},{np.sqrt(ev[1]):.2f}) opac={o32['opacity_act'][i]:.3f}")
# final T consistency: forward image vs oracle
print("img hip-vs-f64 max rel", (np.abs(st.img.cpu().numpy() - o64["img"]) / (np.abs(o64["img"]) + 1e-3)).max(), " o32-vs-f64", (np.abs(o32["img"] - o64["img"]) / (np.abs(o64["img"]) + 1e-3)).max())
