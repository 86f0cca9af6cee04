"""Diagnostic: how far are the HIP gradients from the oracle, in units of the tolerance."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gsplat_amd, oracle
from gsplat_amd.scenes import synthetic_scene, adversarial_scene, orbit_scene
from test_gpu_parity import run_hip, _scene_arrays, GRADS
cases = [("syn", 20000, 320, 200, 3, 16), ("syn", 20000, 320, 200, 2, 64), ("adv", 30000, 250, 130, 3, 16),
         ("syn", 100000, 640, 480, 3, 16), ("orbit", 3000, 160, 120, 3, 16)]
for kind, n, w, h, deg, tile in cases:
    sc = {"syn": lambda: synthetic_scene(n, w, h, deg, 3), "adv": lambda: adversarial_scene(n, w, h, deg, 4),
          "orbit": lambda: orbit_scene(n, w, h, deg, 3, 8, 5)}[kind]()
    a = _scene_arrays(sc)
    cot = torch.rand(w, h, 3, generator=torch.Generator().manual_seed(7)).numpy()
    hip = run_hip(cot=cot, tile=tile, **a)
    ora = oracle.render(a["coords"], a["scale"], a["rotation"], a["opacity"], a["sh"], deg, w, h, tile, a["w2m"], a["proj"], a["cam_pos"], dimg=cot)
    line = f"{kind} n={n} {w}x{h} L={tile} I={hip['num_pairs']}:"
    for k in ("img",) + GRADS:
        g, r = hip[k].astype(np.float64), ora[k].astype(np.float64)
        mx = np.abs(r).max()
        need = ((np.abs(g - r) - 1e-4 * np.abs(r)) / mx).max()   # atol fraction needed on top of rtol 1e-4
        line += f" {k}={need:.1e}"
    print(line, flush=True)
