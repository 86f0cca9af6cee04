"""GPU: the statements of the reference's main.py (main.py:25-37) on this package's modules, from a scene loader to
training steps: dataloader.DataLoader over a stand-in reconstruction whose images were rendered from a known model,
GaussianModel.from_pcd on its point cloud, Trainer on both.  (The loader's own behaviour is checked on the CPU in
tests/test_dataloader.py; this is the hand-over between the pieces.)"""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def test_loader_point_cloud_model_and_trainer_fit_together(tmp_path):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.camera import Camera
    from gsplat_amd.dataloader import DataLoader
    from gsplat_amd.model import GaussianModel
    from gsplat_amd.renderer import GaussianRenderer
    from gsplat_amd.scenes import synthetic_scene
    from gsplat_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    W, H = 96, 64
    sc = synthetic_scene(1500, W, H, 0, seed=4)
    truth = GaussianRenderer(GaussianModel(*[t.to(dev) for t in sc.params()], sh_degree=0))
    # six poses looking down -z (identity rotation: x, y, z, w = 0, 0, 0, 1) from a small ring
    images = {}
    for k in range(6):
        centre = np.array([0.08 * np.cos(k), 0.08 * np.sin(k), 0.0])
        cam = Camera(centre, np.array([1.0, 0.0, 0.0, 0.0]), sc.fx, sc.fy, W / 2, H / 2, W, H, device=dev)
        with torch.no_grad():
            img = truth(cam, 16)[0]
        np.save(tmp_path / f"view{k}.npy", (img * 255).clamp(0, 255).round().to(torch.uint8).cpu().numpy())   # [W,H,3]: first axis = width
        images[k + 1] = types.SimpleNamespace(
            name=f"view{k}.npy", camera_id=1, projection_center=lambda c=centre: c.copy(),
            cam_from_world=types.SimpleNamespace(rotation=types.SimpleNamespace(quat=np.array([0.0, 0.0, 0.0, 1.0])), translation=-centre))
    coords = sc.coords.numpy()
    points = {i: types.SimpleNamespace(xyz=coords[i], color=np.array([200, 120, 40], np.uint8)) for i in range(0, 1500, 3)}
    rec = types.SimpleNamespace(points3D=points, point3D_ids=lambda: list(points.keys()), num_points3D=lambda: len(points),
                                reg_image_ids=lambda: list(images.keys()), images=images,
                                cameras={1: types.SimpleNamespace(focal_length_x=sc.fx, focal_length_y=sc.fy,
                                                                  principal_point_x=W / 2, principal_point_y=H / 2)})
    loader = DataLoader(str(tmp_path), "unused", seed=0, reconstruction=rec, device="cuda:0")
    loader.split_train_validate_data(0.8)
    loader.compute_scene_radius()
    assert len(loader.train_image_ids) == 4 and len(loader.validate_image_ids) == 2
    pcd = loader.get_pcd()
    model = GaussianModel.from_pcd(pcd, device="cuda:0")
    assert model._size == 500 and model.curr_sh_degree == 0
    cam, target = loader.sample(True)
    assert (cam.width, cam.height) == (W, H) and tuple(target.shape) == (W, H, 3) and target.is_cuda
    trainer = Trainer(model, loader, lr=1e-3, train_steps=6, tile_length=16)
    assert float(trainer.scene_radius) == pytest.approx(float(np.max(loader.scene_radius)))
    before = [p.detach().clone() for p in model.parameters()]
    losses = [float(trainer.train_step(i + 1)) for i in range(6)]
    assert all(np.isfinite(losses))
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))
    assert np.isfinite(float(trainer.evaluate()))
