"""Generates tests/golden/*.npz by running the REFERENCE renderer on seeded scenes.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Each fixture holds the raw inputs, the camera constants, the cotangent used for the backward
(loss = sum(img * cotangent)), and everything the reference produced: visibility mask, per-stage
intermediates (compacted to the visible set, as the reference holds them), per-tile depth-sorted
lists (stable order, original Gaussian indices), the image and the six gradients.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import gsplat_amd  # noqa: E402,F401
from gsplat_amd.scenes import (adversarial_scene, all_culled_scene, orbit_scene,  # noqa: E402
                               synthetic_scene)
import ref_harness  # noqa: E402

def nonfinite_scene():
    """A synthetic scene in which a few Gaussians carry non-finite or degenerate raw parameters: NaN and
    infinite coordinates (culled: every comparison with NaN is false, Utils/Camera.py:106-108), a zero and
    a NaN quaternion (the latter is visible but gets a NaN radius and an empty box, Utils/Screen.py:33-36),
    raw opacities of plus and minus infinity (sigmoid 1 and 0), raw scales of minus infinity and -60 (zero
    extent: visible, in no list, and the reference never inverts their covariance)."""
    sc = synthetic_scene(512, 64, 48, 1, seed=21)
    nan, inf = float("nan"), float("inf")
    sc.coords[3] = nan
    sc.coords[4, 0] = inf
    sc.rotation[5] = 0.0
    sc.rotation[6] = nan
    sc.opacity[7] = inf
    sc.opacity[8] = -inf
    sc.scale[9] = -inf          # exp -> 0: a singular covariance, but the empty box keeps it out of every list
    sc.scale[11] = -60.0        # the same with a covariance that underflows
    return sc


def nan_colour_scene():
    """A NaN sh coefficient and a NaN raw opacity: the reference's image is NaN in every pixel of the tiles
    those two Gaussians are listed in (each entry is blended into each pixel of its tile,
    GaussianRenderer.py:104-112)."""
    sc = synthetic_scene(256, 64, 48, 0, seed=22)
    sc.sh[10, 0, 0] = float("nan")
    sc.opacity[12] = float("nan")
    return sc


# name -> (scene factory, tile_length)
CASES = {
    "syn_n256_64x48_deg0_t16": (lambda: synthetic_scene(256, 64, 48, 0, seed=10), 16),
    "syn_n256_64x48_deg1_t64": (lambda: synthetic_scene(256, 64, 48, 1, seed=11), 64),
    "syn_n1024_128x96_deg2_t16": (lambda: synthetic_scene(1024, 128, 96, 2, seed=12), 16),
    "syn_n2048_128x96_deg3_t64": (lambda: synthetic_scene(2048, 128, 96, 3, seed=13), 64),
    "adv_n2048_128x96_deg2_t16": (lambda: adversarial_scene(2048, 128, 96, 2, seed=14), 16),
    "adv_n1024_130x70_deg3_t64": (lambda: adversarial_scene(1024, 130, 70, 3, seed=15), 64),
    "orbit_n512_100x100_deg3_t16": (lambda: orbit_scene(512, 100, 100, 3, view=1, n_views=8, seed=16), 16),
    "syn_n512_64x48_deg3_t10": (lambda: synthetic_scene(512, 64, 48, 3, seed=17), 10),
    "culled_n256_64x48_deg0_t16": (lambda: all_culled_scene(256, 64, 48, 0, seed=18), 16),
    "nonfinite_n512_64x48_deg1_t16": (nonfinite_scene, 16),
    "nancolour_n256_64x48_deg0_t16": (nan_colour_scene, 16),
}


def main():
    assert ref_harness.reference_available(), "needs /root/reference"
    only = sys.argv[1:]
    for name, (factory, tile) in CASES.items():
        if only and name not in only:
            continue
        sc = factory()
        cot = torch.rand(sc.width, sc.height, 3, generator=torch.Generator().manual_seed(1234),
                         dtype=torch.float32).numpy()
        out = ref_harness.run_reference(sc, tile, cot)
        cam = sc.camera(device="cpu")
        data = dict(
            coords=sc.coords.numpy(), scale=sc.scale.numpy(), rotation=sc.rotation.numpy(),
            opacity=sc.opacity.numpy(), sh=sc.sh.numpy(), sh_degree=np.int32(sc.sh_degree),
            width=np.int32(sc.width), height=np.int32(sc.height), tile_length=np.int32(tile),
            w2m=cam.w2m_host, proj=cam.proj_host,
            cam_pos=np.asarray(sc.cam_position, dtype=np.float32), cotangent=cot,
            visible=np.bool_(out["img"] is not None),
        )
        if out["img"] is not None:
            for k, v in out.items():
                data["ref_" + k] = v
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB"
              + ("" if out["img"] is None else f", visible {int(out['mask'].sum())}/{sc.n}, "
                 f"pairs {len(out['tile_lists'])}"))


if __name__ == "__main__":
    main()
