"""Pins the path at the SIZE of the reference's own CPU case (about 50 k Gaussians, SH degree 0, a head photograph
downscaled by four as the target) with the reference itself, and compares the oracle with the reference at 200 k.

Run in the build container only (needs /root/reference):

    python tests/golden/make_config1_golden.py fixtures     # writes config1_n50000_180x180_deg0_t{64,16}.npz
    python tests/golden/make_config1_golden.py compare200k  # prints the oracle-vs-reference comparison, writes nothing

``fixtures``: the imported reference renders ``orbit_scene(50_000, 180, 180, 0, view=3, seed=31)`` - its parameters
cut to the upper 16 bits of their fp32 patterns, so that they are stored exactly in half the space (the generator's
``pow``, ``log`` and normal draws are not reproducible bit for bit from one CPU model to the next) - at tile edges 64
(what ``Trainer.py:55`` passes by default) and 16, the loss is the reference's ``0.8 * l1 + 0.2 * (1 - ssim)``
(``Trainer.py:35-38`` over ``Utils/Loss.py``) against ``HumanHeadImages/head000.jpg`` read with PIL, reduced from 720 to
180 pixels and kept as 0..255 floats (``Utils/DataLoader.py:81-83``), and ``loss.backward()`` gives the gradients.  The
inputs are stored once (``config1_n50000_inputs.npz``: uint16 upper halves; degree 0 reads one sh coefficient per channel,
the other fifteen are the constant 0.25); a fixture stores the target, the image, the mask bits, the tile offsets (list
lengths), a SHA-256 of the concatenated tile lists, the loss and every 16th row of the six gradients.

``compare200k``: 200 k Gaussians, degree 3, 800 x 800, tile 16 - mask, positions with the depth key, boxes, offsets and
lists of the oracle against the reference bit for bit, image and gradients within rtol 1e-4.  Takes about a minute and
16 GB; its output is quoted where the oracle is described.
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import gsplat_amd  # noqa: E402,F401
from gsplat_amd.scenes import orbit_scene, synthetic_scene  # noqa: E402
import ref_harness  # noqa: E402

N, W, H, DEG, VIEW, SEED = 50_000, 180, 180, 0, 3, 31
ROW_STRIDE = 16
LOSS_LAMBDA = 0.2
HEAD = os.path.join(ref_harness.REF, "HumanHeadImages", "head000.jpg")
GRADS = ("g_coords", "g_scale", "g_rotation", "g_opacity", "g_sh", "g_screen")


def _upper16(t):
    """fp32 values cut to the upper 16 bits of their patterns (sign, exponent, 7 fraction bits): exact in a uint16."""
    return (t.contiguous().view(torch.int32) & -65536).view(torch.float32)


def config1_scene():
    sc = orbit_scene(N, W, H, DEG, view=VIEW, n_views=8, seed=SEED)
    sh = torch.full_like(sc.sh, 0.25)          # degree 0 reads coefficient 0 only; the rest must not matter
    sh[:, :, 0] = _upper16(sc.sh[:, :, 0])
    sc.coords, sc.scale, sc.rotation, sc.opacity, sc.sh = (_upper16(sc.coords), _upper16(sc.scale), _upper16(sc.rotation),
                                                           _upper16(sc.opacity), sh)
    return sc


def _bits16(t):
    return (t.contiguous().view(torch.int32) >> 16).to(torch.int16).numpy().view(np.uint16)


def save_inputs(sc):
    path = os.path.join(HERE, f"config1_n{N}_inputs.npz")
    np.savez_compressed(path, coords=_bits16(sc.coords), scale=_bits16(sc.scale), rotation=_bits16(sc.rotation),
                        opacity=_bits16(sc.opacity), sh0=_bits16(sc.sh[:, :, 0]), sh_rest=np.float32(0.25),
                        cam_position=np.asarray(sc.cam_position, dtype=np.float64),
                        cam_rotation=np.asarray(sc.cam_rotation, dtype=np.float64), fx=np.float64(sc.fx), fy=np.float64(sc.fy))
    print(f"{os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


def head_target():
    """The photograph as the reference's loader hands it to the loss: RGB, float, 0..255, [rows, cols, 3]."""
    from PIL import Image
    im = Image.open(HEAD).convert("RGB")
    assert im.size == (720, 720)
    im = im.resize((H, W), Image.BILINEAR)          # (PIL takes (columns, rows); the scene is square)
    return np.asarray(im, dtype=np.uint8)


def lists_digest(tile_lists):
    return hashlib.sha256(np.ascontiguousarray(tile_lists, dtype=np.int32).tobytes()).hexdigest()


def reference_training_pass(scene, tile, target_u8):
    """render -> compute_loss -> backward, with the reference's own functions."""
    ref_harness._import_reference()
    from Utils.Loss import l1_loss, ssim_loss
    model, cam, renderer = ref_harness.build_reference(scene)
    gt = torch.from_numpy(target_u8).float()
    t0 = time.time()
    with ref_harness._StableArgsort():
        img, mask, screen_coords, radius = renderer.render(cam, tile)
        t1 = time.time()
        ssim = ssim_loss(img, gt)
        l1 = l1_loss(img, gt)
        loss = (1 - LOSS_LAMBDA) * l1 + LOSS_LAMBDA * (1 - ssim)
        loss.backward()
    t2 = time.time()
    out = dict(img=img.detach().numpy().copy(), mask=mask.numpy().copy(), radius=radius.detach().numpy().copy(),
               loss=np.float32(loss.item()), l1=np.float32(l1.item()), ssim=np.float32(ssim.item()),
               g_coords=model._coords.grad.numpy().copy(), g_scale=model._scale.grad.numpy().copy(),
               g_rotation=model._rotation.grad.numpy().copy(), g_opacity=model._opacity.grad.numpy().copy(),
               g_sh=model._sh.grad.numpy().copy(), g_screen=screen_coords.grad.numpy().copy())
    print(f"  reference: forward {t1 - t0:.1f} s, loss + backward {t2 - t1:.1f} s", flush=True)
    return out


def make_fixtures():
    target = head_target()
    save_inputs(config1_scene())
    for tile in (64, 16):
        scene = config1_scene()
        print(f"config 1 at tile {tile}:", flush=True)
        out = reference_training_pass(scene, tile, target)
        lists = ref_harness.run_reference(scene, tile, None, want_intermediates=True)
        assert np.array_equal(lists["mask"], out["mask"]) and np.array_equal(lists["img"], out["img"])
        cam = scene.camera(device="cpu")
        data = dict(
            n=np.int32(N), width=np.int32(W), height=np.int32(H), sh_degree=np.int32(DEG), view=np.int32(VIEW),
            seed=np.int32(SEED), tile_length=np.int32(tile), loss_lambda=np.float32(LOSS_LAMBDA),
            row_stride=np.int32(ROW_STRIDE), w2m=cam.w2m_host, proj=cam.proj_host,
            cam_pos=np.asarray(scene.cam_position, dtype=np.float32),
            inputs_sha256=np.str_(hashlib.sha256(b"".join(t.numpy().tobytes() for t in scene.params())).hexdigest()),
            target=target, ref_img=out["img"], ref_mask_bits=np.packbits(out["mask"]),
            ref_radius_rows=out["radius"][::ROW_STRIDE].copy(), ref_loss=out["loss"], ref_l1=out["l1"], ref_ssim=out["ssim"],
            ref_tile_offsets=lists["tile_offsets"], ref_tile_lists_sha256=np.str_(lists_digest(lists["tile_lists"])),
        )
        for k in GRADS:
            data["ref_" + k + "_rows"] = out[k][::ROW_STRIDE].copy()
            data["ref_" + k + "_absmax"] = np.float32(np.abs(out[k]).max())
        path = os.path.join(HERE, f"config1_n{N}_{W}x{H}_deg{DEG}_t{tile}.npz")
        np.savez_compressed(path, **data)
        print(f"  {os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB, visible {int(out['mask'].sum())}/{N}, "
              f"pairs {len(lists['tile_lists'])}, loss {float(out['loss']):.6f}", flush=True)


def compare_200k():
    import oracle
    from util import GRAD_OUTLIERS, assert_bit_equal, assert_close
    oracle.build()
    scene = synthetic_scene(200_000, 800, 800, 3, seed=0)
    cot = torch.rand(800, 800, 3, generator=torch.Generator().manual_seed(1234), dtype=torch.float32).numpy()
    t0 = time.time()
    ref = ref_harness.run_reference(scene, 16, cot)
    print(f"reference: {time.time() - t0:.1f} s (fwd + bwd + lists)", flush=True)
    cam = scene.camera(device="cpu")
    t0 = time.time()
    o = oracle.render(scene.coords.numpy(), scene.scale.numpy(), scene.rotation.numpy(), scene.opacity.numpy(), scene.sh.numpy(),
                      3, 800, 800, 16, cam.w2m_host, cam.proj_host, np.asarray(scene.cam_position, dtype=np.float32), dimg=cot)
    print(f"oracle: {time.time() - t0:.1f} s", flush=True)
    m = ref["mask"]
    assert_bit_equal(o["mask"], m, "mask")
    assert_bit_equal(o["pos2d"][m], ref["pos2d"], "pos2d")
    assert_bit_equal(o["bbox"][m], ref["bbox"], "bbox")
    assert_bit_equal(o["tile_offsets"], ref["tile_offsets"], "tile_offsets")
    assert_bit_equal(o["tile_lists"], ref["tile_lists"], "tile_lists")
    print(f"bit-equal: mask ({int(m.sum())} visible), pos2d incl. depth key, bbox, tile_offsets, tile_lists "
          f"({len(ref['tile_lists'])} pairs)", flush=True)
    assert_close(o["img"], ref["img"], "img")
    worst = {"img": float(np.abs(o["img"] - ref["img"]).max() / np.abs(ref["img"]).max())}
    for k in GRADS:
        assert_close(o[k], ref[k], k, **GRAD_OUTLIERS)
        worst[k] = float(np.abs(o[k] - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-30))
    print("within rtol 1e-4: img and six gradients; max |diff| / max |ref|: " +
          ", ".join(f"{k} {v:.1e}" for k, v in worst.items()), flush=True)


if __name__ == "__main__":
    assert ref_harness.reference_available(), "needs /root/reference"
    mode = sys.argv[1] if len(sys.argv) > 1 else "fixtures"
    {"fixtures": make_fixtures, "compare200k": compare_200k}[mode]()
