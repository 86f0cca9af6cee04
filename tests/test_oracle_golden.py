"""The CPU oracle against the fixtures produced by the reference itself (CPU only)."""
import numpy as np
import pytest

from util import (assert_bit_equal, assert_close, assert_close_nan, check_config1, config1_golden_names, config1_inputs,
                  golden_names, load_golden, loss_golden_names, nan_colour_golden_names, nonfinite_golden_names)

GRADS = ("g_coords", "g_scale", "g_rotation", "g_opacity", "g_sh", "g_screen")


def run_oracle(oracle_mod, g, with_grad=True):
    return oracle_mod.render(g["coords"], g["scale"], g["rotation"], g["opacity"], g["sh"], int(g["sh_degree"]),
                             int(g["width"]), int(g["height"]), int(g["tile_length"]), g["w2m"], g["proj"],
                             g["cam_pos"], dimg=g["cotangent"] if with_grad else None)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference(oracle_mod, name):
    g = load_golden(name)
    o = run_oracle(oracle_mod, g)
    if not bool(g["visible"]):
        assert o["img"] is None            # GaussianRenderer.py:67-68 returns four Nones
        assert not o["mask"].any()
        return
    m = g["ref_mask"]
    # discrete results: bit-exact
    assert_bit_equal(o["mask"], m, "mask")
    assert_bit_equal(o["pos2d"][m], g["ref_pos2d"], "pos2d (pixel x, pixel y, NDC depth)")
    assert_bit_equal(o["bbox"][m], g["ref_bbox"], "bbox")
    assert_bit_equal(o["tile_offsets"], g["ref_tile_offsets"], "tile_offsets")
    assert_bit_equal(o["tile_lists"], g["ref_tile_lists"], "tile_lists")
    # floats: rtol 1e-4
    assert_close(o["cov2d"][m], g["ref_cov2d"].reshape(-1, 4), "cov2d")
    assert_close(o["radius"], g["ref_radius"], "radius")
    assert_close(o["color"][m], g["ref_color"], "color")
    assert_close(o["opacity_act"][m], g["ref_opacity_act"][:, 0], "opacity")
    assert_close(o["img"], g["ref_img"], "img")
    for k in GRADS:
        assert_close(o[k], g["ref_" + k], k)
    # structure of the gradients (reference behaviour)
    K = (int(g["sh_degree"]) + 1) ** 2
    assert not o["g_sh"][:, :, K:].any()
    assert not o["g_screen"][:, 2].any()
    assert not o["g_screen"][~m].any()


def test_fixture_coverage():
    """The fixture set exercises the cases that decide visibility and tile membership."""
    names = golden_names()
    assert len(names) >= 9
    seen_culled = seen_empty_bbox = seen_ties = seen_behind = seen_none = False
    for n in names:
        g = load_golden(n)
        if not bool(g["visible"]):
            seen_none = True
            continue
        m = g["ref_mask"]
        seen_culled |= bool((~m).any())
        bb = g["ref_bbox"]
        seen_empty_bbox |= bool(((bb[:, 2] <= bb[:, 0]) | (bb[:, 3] <= bb[:, 1])).any())
        d = g["ref_pos2d"][:, 2]
        seen_ties |= len(np.unique(d)) < len(d)
        seen_behind |= bool((d > 1.0).any())       # NDC z > 1: behind the camera, still visible
    assert seen_culled and seen_empty_bbox and seen_ties and seen_behind and seen_none


def test_oracle_thread_count_invariance(oracle_mod):
    g = load_golden("adv_n2048_128x96_deg2_t16")
    a = run_oracle(oracle_mod, g)
    oracle_mod.lib().gso_set_threads(1)
    b = run_oracle(oracle_mod, g)
    oracle_mod.lib().gso_set_threads(oracle_mod.max_threads())
    assert_bit_equal(a["img"], b["img"], "img")
    assert_bit_equal(a["tile_lists"], b["tile_lists"], "lists")
    for k in GRADS:
        assert_close(a[k], b[k], k, rtol=1e-6, atol_frac=1e-9)


@pytest.mark.parametrize("name", loss_golden_names())
def test_loss_oracle_matches_reference(oracle_mod, name):
    """The loss oracle against the reference's own l1_loss / ssim_loss and autograd (Utils/Loss.py,
    combined as in Trainer.py:35-38)."""
    g = load_golden(name)
    lam = float(g["loss_lambda"])
    o = oracle_mod.loss(g["pred"], g["gt"], 1.0 - lam, -lam)
    assert_close(o["l1"], g["ref_l1"], "l1", rtol=1e-6, atol_frac=0)
    assert_close(o["ssim"], g["ref_ssim"], "ssim", rtol=1e-5, atol_frac=0)
    loss = (1.0 - lam) * o["l1"] + lam * (1.0 - o["ssim"])
    assert_close(loss, g["ref_loss"], "loss", rtol=1e-6, atol_frac=0)
    assert_close(o["dpred"], g["ref_dpred"], "dloss/dpred")


def test_loss_fixture_coverage():
    names = loss_golden_names()
    assert len(names) >= 7
    shapes = [load_golden(n)["pred"].shape for n in names]
    assert any(s[1] < 11 for s in shapes)            # fewer rows than the window
    assert any(s[0] == 1 for s in shapes)            # a single image column
    assert any(load_golden(n)["gt"].max() > 200 for n in names)   # 0..255 targets (Utils/DataLoader.py:83)
    assert any((load_golden(n)["pred"] == load_golden(n)["gt"]).mean() > 0.1 for n in names)   # ties


def check_nonfinite(o, g):
    """What holds for a fixture with NaN / infinite parameters (used by the GPU test as well): the discrete
    results are those of the reference, floats agree NaN for NaN - including the NaN coordinate gradient
    the reference gives a CULLED Gaussian whose coordinates are not finite (autograd divides its zero
    cotangent by the clip-space w and multiplies it by the clip coordinates, Utils/Camera.py:103)."""
    m = g["ref_mask"]
    assert_bit_equal(o["mask"], m, "mask")
    # a NaN radius: the reference's int() of the NaN box edges gives INT_MIN four times (no tile contains
    # it, Utils/Screen.py:38-62), here the box is stored as empty; either way the Gaussian is in no list
    nan_box = np.isnan(g["ref_radius"])
    assert nan_box.any() and (g["ref_bbox"][nan_box] == np.iinfo(np.int32).min).all()
    assert_bit_equal(o["bbox"][m][~nan_box], g["ref_bbox"][~nan_box], "bbox")
    assert not np.isin(np.nonzero(m)[0][nan_box], o["tile_lists"]).any()
    assert_bit_equal(o["tile_offsets"], g["ref_tile_offsets"], "tile_offsets")
    assert_bit_equal(o["tile_lists"], g["ref_tile_lists"], "tile_lists")
    assert_close_nan(o["pos2d"][m], g["ref_pos2d"], "pos2d", rtol=1e-6)
    assert_close_nan(o["cov2d"][m], g["ref_cov2d"].reshape(-1, 4), "cov2d")
    assert_close_nan(o["radius"], g["ref_radius"], "radius")
    assert np.isfinite(g["ref_img"]).all()
    assert_close(o["img"], g["ref_img"], "img")
    dead = np.nonzero(~np.isfinite(g["coords"]).all(axis=1) & ~m)[0]
    assert len(dead) >= 2 and np.isnan(g["ref_g_coords"][dead]).all() and np.isnan(o["g_coords"][dead]).all()
    for k in GRADS:
        assert_close_nan(o[k], g["ref_" + k], k)


@pytest.mark.parametrize("name", nonfinite_golden_names())
def test_oracle_matches_reference_on_nonfinite_parameters(oracle_mod, name):
    g = load_golden(name)
    assert (~np.isfinite(g["coords"])).any() and np.isnan(g["rotation"]).any() and np.isinf(g["opacity"]).any()
    check_nonfinite(run_oracle(oracle_mod, g), g)


def check_nan_colour(o, g):
    """A NaN colour or opacity: lists as the reference's, the image NaN in exactly the reference's pixels
    (whole tiles) and close elsewhere, and the gradients carry NaN too, so the loss shows the divergence.
    (Which gradient elements are NaN depends on where a 0 * NaN happens to be skipped; not compared.)"""
    m = g["ref_mask"]
    assert_bit_equal(o["mask"], m, "mask")
    assert_bit_equal(o["tile_offsets"], g["ref_tile_offsets"], "tile_offsets")
    assert_bit_equal(o["tile_lists"], g["ref_tile_lists"], "tile_lists")
    bad = np.isnan(g["ref_img"])
    assert bad.any() and not bad.all() and bad.all(axis=2).any() and (bad.any(axis=2) & ~bad.all(axis=2)).any()
    assert_close_nan(o["img"], g["ref_img"], "img")
    assert np.isnan(o["g_opacity"]).any() and np.isnan(o["g_coords"]).any()


@pytest.mark.parametrize("name", nan_colour_golden_names())
def test_oracle_spreads_nan_like_the_reference(oracle_mod, name):
    g = load_golden(name)
    assert np.isnan(g["sh"]).any() and np.isnan(g["opacity"]).any()
    check_nan_colour(run_oracle(oracle_mod, g), g)


@pytest.mark.parametrize("name", config1_golden_names())
def test_oracle_matches_reference_at_config1_size(oracle_mod, name):
    """The reference's own CPU case at its size: 50 k Gaussians, degree 0, a 180 x 180 view, tile edges 64 (the default
    of GaussianRenderer.py:49) and 16, the loss of Trainer.py:35-38 against a head photograph reduced by four, backward.
    The oracle runs the same pass: render, loss oracle, backward with the loss's gradient as cotangent."""
    g = load_golden(name)
    scene, target = config1_inputs(g)
    tile, lam = int(g["tile_length"]), float(g["loss_lambda"])

    def render(dimg=None):          # (the camera matrices are the stored ones the reference was given)
        return oracle_mod.render(*(t.numpy() for t in scene.params()), int(g["sh_degree"]), int(g["width"]), int(g["height"]), tile,
                                 g["w2m"], g["proj"], g["cam_pos"], dimg=dimg)

    ol = oracle_mod.loss(render()["img"], target, 1.0 - lam, -lam)
    o = render(ol["dpred"])
    loss = (1.0 - lam) * ol["l1"] + lam * (1.0 - ol["ssim"])
    check_config1(g, o, (loss, ol["l1"], ol["ssim"]), {k: o[k] for k in GRADS})


def test_config1_fixture_coverage():
    names = config1_golden_names()
    assert sorted(int(load_golden(n)["tile_length"]) for n in names) == [16, 64]
    for n in names:
        g = load_golden(n)
        assert int(g["n"]) == 50_000 and int(g["sh_degree"]) == 0 and g["target"].shape == (180, 180, 3)
        assert g["target"].dtype == np.uint8 and g["target"].max() > 200        # 0..255 targets (Utils/DataLoader.py:83)
        assert int(g["ref_tile_offsets"][-1]) > 100_000
