"""CPU: the host-side scene ingest helpers.  Camera.from_sfm against matrices the reference's own
Camera.from_sfm (Utils/Camera.py:30-46) produced for the same structure-from-motion poses
(tests/golden/camera_from_sfm.npz, written by tests/golden/make_train_golden.py);
compute_scene_radius (Utils/DataLoader.py:91-106) against its definition - the reference's DataLoader
needs pycolmap and cv2 and cannot be imported, so that one has no fixture."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_from_sfm_matches_the_reference():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.camera import Camera
    d = np.load(os.path.join(ROOT, "tests", "golden", "camera_from_sfm.npz"))
    for k in range(int(d["cases"])):
        q, center = d[f"quat_xyzw_{k}"], d[f"center_{k}"]
        fx, fy, cx, cy, w, h = d[f"intrinsics_{k}"]
        image = types.SimpleNamespace(cam_from_world=types.SimpleNamespace(rotation=types.SimpleNamespace(quat=q.copy())),
                                      projection_center=lambda c=center: c.copy())
        model = types.SimpleNamespace(focal_length_x=fx, focal_length_y=fy, principal_point_x=cx, principal_point_y=cy)
        cam = Camera.from_sfm(image, model, int(w), int(h), device="cpu")
        assert np.array_equal(q, d[f"quat_xyzw_{k}"])                     # the pose object is not modified
        np.testing.assert_array_equal(cam.world2model_matrix.numpy(), d[f"w2m_{k}"])
        np.testing.assert_array_equal(cam.projection_matrix.numpy(), d[f"proj_{k}"])
        np.testing.assert_allclose(np.asarray(cam.transform.position, np.float64), d[f"position_{k}"])
        assert (cam.width, cam.height) == (int(w), int(h))


def test_scene_radius():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.camera import compute_scene_radius
    rng = np.random.default_rng(2)
    t = rng.normal(size=(17, 3)).astype(np.float32) * np.array([5, 1, 0.2], np.float32)
    r = compute_scene_radius(t)
    want = np.abs(t - t.mean(axis=0)).max(axis=0)
    assert r.dtype == np.float32 and r.shape == (3,)
    np.testing.assert_allclose(r, want, rtol=1e-5)
    np.testing.assert_array_equal(compute_scene_radius(t[:1]), np.zeros(3, np.float32))


def test_project_gaussian_matches_the_reference_fixtures():
    """Camera.project_gaussian and GaussianModel.covariance (Utils/Camera.py:74-117, GaussianModel.py:99-108,
    the pair Trainer.test calls) against what the reference's render produced on the golden scenes."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    from util import golden_names, load_golden
    import gsplat_amd  # noqa: F401
    from gsplat_amd.camera import Camera
    from gsplat_amd.model import GaussianModel
    checked = 0
    for name in golden_names():
        g = load_golden(name)
        t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32))  # noqa: E731
        model = GaussianModel(t("coords"), t("scale"), t("rotation"), t("opacity").reshape(-1, 1), t("sh"),
                              sh_degree=int(g["sh_degree"]))
        cam = Camera(np.zeros(3), np.array([1.0, 0, 0, 0]), 1.0, 1.0, 0.0, 0.0, int(g["width"]), int(g["height"]), device="cpu")
        cam.world2model_matrix = torch.from_numpy(np.asarray(g["w2m"], np.float32).reshape(4, 4))
        cam.projection_matrix = torch.from_numpy(np.asarray(g["proj"], np.float32).reshape(4, 4))
        with torch.no_grad():        # Trainer.test runs under no_grad (Trainer.py:106)
            pos2d, cov2d, mask = cam.project_gaussian(torch.zeros_like(model.coords), model.coords, model.covariance)
        if not bool(g["visible"]):
            assert pos2d is None and cov2d is None and not mask.any()
            continue
        assert np.array_equal(mask.numpy(), g["ref_mask"])
        np.testing.assert_allclose(pos2d.numpy(), g["ref_pos2d"], rtol=1e-5, atol=1e-4)
        ref = g["ref_cov2d"].reshape(-1, 2, 2)
        np.testing.assert_allclose(cov2d.numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
        checked += 1
    assert checked >= 8
