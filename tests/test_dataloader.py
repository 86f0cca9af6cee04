"""CPU: the scene-ingest mirror of the reference's DataLoader (Utils/DataLoader.py:52-106) on a stand-in
reconstruction.  Parity unpinned for get_pcd / split / sample: the reference's DataLoader imports pycolmap and cv2,
which are absent, so no fixture of it exists; what is checked is the behaviour its code states (orders, dtypes, the
draws of Python's random module, the bilinear rescale, the camera of Camera.from_sfm, which IS pinned by
tests/test_camera_ingest.py)."""
import os
import random
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def reconstruction(n_images=10, n_points=7, seed=3):
    rng = np.random.default_rng(seed)
    points = {100 + 3 * k: types.SimpleNamespace(xyz=rng.normal(size=3), color=rng.integers(0, 256, 3).astype(np.uint8))
              for k in range(n_points)}
    images = {}
    for k in range(n_images):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        centre = rng.normal(size=3)
        images[k + 1] = types.SimpleNamespace(
            name=f"im{k}.npy", camera_id=1,
            cam_from_world=types.SimpleNamespace(rotation=types.SimpleNamespace(quat=q), translation=rng.normal(size=3)),
            projection_center=lambda c=centre: c.copy())
    cameras = {1: types.SimpleNamespace(focal_length_x=50.0, focal_length_y=48.0, principal_point_x=12.0, principal_point_y=8.0)}
    return types.SimpleNamespace(points3D=points, point3D_ids=lambda: list(points.keys()), num_points3D=lambda: len(points),
                                 reg_image_ids=lambda: list(images.keys()), images=images, cameras=cameras)


def test_point_cloud_of_the_reconstruction():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.dataloader import DataLoader
    rec = reconstruction()
    pcd = DataLoader("unused", reconstruction=rec).get_pcd()
    assert pcd.size == 7 and pcd.coords.dtype == np.float32 and pcd.colors.dtype == np.float32
    for row, pid in enumerate(rec.point3D_ids()):
        np.testing.assert_array_equal(pcd.coords[row], rec.points3D[pid].xyz.astype(np.float32))
        np.testing.assert_array_equal(pcd.colors[row], rec.points3D[pid].color.astype(np.float32))
    with pytest.raises(ValueError):
        DataLoader("unused").get_pcd()


def test_split_and_sampling_follow_pythons_random_module():
    import gsplat_amd  # noqa: F401
    from gsplat_amd.dataloader import DataLoader
    rec = reconstruction()
    loader = DataLoader("unused", seed=11, reconstruction=rec, read_image=lambda path: np.zeros((4, 6, 3), np.uint8), device="cpu")
    loader.split_train_validate_data(0.8)
    random.seed(11)
    ids = list(rec.reg_image_ids()); random.shuffle(ids)
    assert loader.train_image_ids == ids[:8] and loader.validate_image_ids == ids[8:]
    want = [random.choice(ids[:8]), random.choice(ids[8:])]
    random.seed(11); ids2 = list(rec.reg_image_ids()); random.shuffle(ids2)     # rewind to just after the split
    got = []
    for is_train in (True, False):
        loader.read_image = lambda path, got=got: got.append(path) or np.zeros((4, 6, 3), np.uint8)
        loader.sample(is_train)
    assert got == [f"unused/{rec.images[i].name}" for i in want]


def test_sample_returns_the_sfm_camera_and_the_rescaled_image(tmp_path):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.camera import Camera
    from gsplat_amd.dataloader import DataLoader
    rec = reconstruction(n_images=1)
    pixels = np.random.default_rng(0).integers(0, 256, (12, 20, 3)).astype(np.uint8)
    np.save(tmp_path / "im0.npy", pixels)
    loader = DataLoader(str(tmp_path), seed=0, reconstruction=rec, device="cpu")
    loader.split_train_validate_data(1.0)
    cam, image = loader.sample(True)
    assert image.dtype == torch.float32 and tuple(image.shape) == (12, 20, 3)
    np.testing.assert_array_equal(image.numpy(), pixels.astype(np.float32))
    ref = Camera.from_sfm(rec.images[1], rec.cameras[1], 12, 20, device="cpu")   # first image axis = the camera's width
    assert (cam.width, cam.height) == (12, 20)
    np.testing.assert_array_equal(cam.world2model_matrix.numpy(), ref.world2model_matrix.numpy())
    np.testing.assert_array_equal(cam.projection_matrix.numpy(), ref.projection_matrix.numpy())
    cam_half, half = loader.sample(True, image_scale=0.5)
    assert tuple(half.shape) == (6, 10, 3) and (cam_half.width, cam_half.height) == (6, 10)
    want = torch.nn.functional.interpolate(torch.from_numpy(pixels).float().permute(2, 0, 1)[None], scale_factor=0.5,
                                           mode="bilinear")[0].permute(1, 2, 0)
    np.testing.assert_array_equal(half.numpy(), want.numpy())


def test_ppm_reader_and_scene_radius(tmp_path):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.camera import compute_scene_radius
    from gsplat_amd.dataloader import DataLoader, read_image_file
    pixels = np.random.default_rng(1).integers(0, 256, (5, 7, 3)).astype(np.uint8)
    with open(tmp_path / "a.ppm", "wb") as f:
        f.write(b"P6\n# a comment\n7 5\n255\n" + pixels.tobytes())
    np.testing.assert_array_equal(read_image_file(str(tmp_path / "a.ppm")), pixels)
    rec = reconstruction()
    loader = DataLoader("unused", reconstruction=rec)
    radius = loader.compute_scene_radius()
    np.testing.assert_array_equal(radius, compute_scene_radius([rec.images[i].cam_from_world.translation for i in rec.reg_image_ids()]))
    assert loader.scene_radius is radius and radius.shape == (3,)
