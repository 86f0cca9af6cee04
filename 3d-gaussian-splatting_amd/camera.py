"""Host-side camera: the two 4x4 fp32 matrices the rasterizer consumes.

Mirrors the attribute surface of the reference camera
(``Utils/Camera.py:10-28``: ``width``, ``height``, ``fx``, ``fy``, ``cx``,
``cy``, ``transform.position``, ``world2model_matrix``, ``projection_matrix``)
so that the renderer accepts either this class or the reference's own
``Camera`` object.  The matrices are built on the host exactly the way the
reference builds them:

* world->model = inv(T @ R @ S) with T, R, S assembled in float32 numpy
  (``Utils/Transform.py:28-38,73-84,109-138``),
* OpenGL-style projection computed in float64 and cast to float32
  (``Utils/Camera.py:48-64``), near 0.1, far 100.

They are 32 floats per view and stay on the host; the kernels receive them
as launch constants.
"""
import math

import numpy as np
import torch


class _Pose:
    """Position + unit quaternion (w, x, y, z) + per-axis scale, the data of ``Utils/Transform.py:6-25``."""

    def __init__(self, position, rotation=None, scale=None):
        self.position = np.asarray(position)
        self.rotation = np.array([1, 0, 0, 0]) if rotation is None else np.asarray(rotation)
        self.scale = np.array([1, 1, 1]) if scale is None else np.asarray(scale)
        if self.rotation.shape != (4,):
            raise ValueError(f"a pose takes a quaternion of 4 numbers (w, x, y, z), got shape {self.rotation.shape}")
        if self.scale.shape != (3,):
            raise ValueError(f"a pose takes 3 scale factors, got shape {self.scale.shape}")

    def model2world(self) -> np.ndarray:
        """translation @ rotation @ scale as three float32 4x4 products (``Utils/Transform.py:28-38,73-84,
        109-138``).  The entries are evaluated in the inputs' precision with the reference's association -
        ``2 (w w + x x) - 1`` on the diagonal, ``2 (x y -+ w z)`` off it - and only then stored as float32,
        which is what makes the inverse below agree with the reference's to the last bit."""
        w, x, y, z = (self.rotation[i] for i in range(4))
        rot = np.zeros((4, 4), dtype=np.float32)
        rot[3, 3] = 1
        for a, (p, q, r) in enumerate(((x, y, z), (y, z, x), (z, x, y))):     # axis a, then the two after it, cyclically
            b, c = (a + 1) % 3, (a + 2) % 3
            rot[a, a] = 2 * (w * w + p * p) - 1
            rot[a, b] = 2 * (p * q - w * r)      # one axis on
            rot[a, c] = 2 * (p * r + w * q)      # two axes on
        place = np.eye(4, dtype=np.float32)
        place[:3, 3] = [self.position[i] for i in range(3)]
        stretch = np.diag(np.array([self.scale[0], self.scale[1], self.scale[2], 1], dtype=np.float32))
        return place @ rot @ stretch

    def world2model(self) -> np.ndarray:
        return np.linalg.inv(self.model2world())


class Camera:
    """Pinhole camera handed to ``GaussianRenderer.render``.

    ``width`` is the slow image axis and ``height`` the fast one: the renderer
    writes ``[width, height, 3]`` (``GaussianRenderer.py:84``).
    """

    def __init__(self, position, rotation, focal_length_x, focal_length_y,
                 principal_point_x, principal_point_y, width, height, device=None):
        self.transform = _Pose(position, rotation=rotation)
        self.fx = focal_length_x
        self.fy = focal_length_y
        self.cx = principal_point_x
        self.cy = principal_point_y
        self.height = height
        self.width = width
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = device
        self.w2m_host = np.ascontiguousarray(self.transform.world2model(), dtype=np.float32)
        self.proj_host = np.ascontiguousarray(self.get_projection_matrix(), dtype=np.float32)
        self.world2model_matrix = torch.from_numpy(self.w2m_host).to(device=device, dtype=torch.float32)
        self.projection_matrix = torch.from_numpy(self.proj_host).to(device=device, dtype=torch.float32)

    @staticmethod
    def from_sfm(sfm_image, sfm_camera, width, height, device=None) -> "Camera":
        """``Camera.from_sfm`` (``Utils/Camera.py:30-46``): a camera from a structure-from-motion pose.
        ``sfm_image`` and ``sfm_camera`` are read through the attributes the reference reads of
        ``pycolmap.Image`` / ``pycolmap.Camera`` (``cam_from_world.rotation.quat`` in x, y, z, w order,
        ``projection_center()``, ``focal_length_x/y``, ``principal_point_x/y``); no SfM package is
        imported here.  The quaternion is rolled to w, x, y, z and conjugated, as the reference does."""
        rotation = np.roll(np.asarray(sfm_image.cam_from_world.rotation.quat, dtype=np.float64), 1)
        rotation[1:] = -rotation[1:]
        return Camera(sfm_image.projection_center(), rotation, sfm_camera.focal_length_x, sfm_camera.focal_length_y,
                      sfm_camera.principal_point_x, sfm_camera.principal_point_y, width, height, device=device)

    def get_projection_matrix(self, near: float = 0.1, far: float = 100) -> np.ndarray:
        """The symmetric OpenGL frustum of the intrinsics in fp64 (``Utils/Camera.py:48-72``), bit for bit:
        the reference goes through the field of view, ``near * tan(fov / 2)`` with ``fov = 2 atan(size / 2f)``;
        doubling and halving are exact, and so is ``right - left = 2 right`` of the symmetric frustum, hence
        ``2 near / (right - left) = near / right`` to the last bit.  The two off-centre terms are ``-0.0``."""
        half_w = near * math.tan(math.atan(self.width / (2 * self.fx)))
        half_h = near * math.tan(math.atan(self.height / (2 * self.fy)))
        m = np.zeros((4, 4), dtype=float)
        m[0, 0] = near / half_w
        m[1, 1] = near / half_h
        m[0, 3] = m[1, 3] = -0.0
        m[2, 2] = -(far + near) / (far - near)
        m[2, 3] = 2 * far * near / (near - far)
        m[3, 2] = -1.0
        return m

    def get_covariance2d(self, pos, cov3d: torch.Tensor) -> torch.Tensor:
        """``Camera.get_covariance2d`` (``Utils/Camera.py:74-87``): ``J W cov3d W^T J^T`` with the Jacobian
        taken at the WORLD-space point, as the reference does.  ``pos``: ``[M,>=3]``; returns ``[M,2,2]``.
        (The renderer computes the same inside its first kernel; this is for callers such as
        ``Trainer.test``.)"""
        x, y, z = pos[:, 0], pos[:, 1], pos[:, 2]
        zero = torch.zeros_like(z)
        jac = torch.stack((1 / z, zero, -x / z, zero, 1 / z, -y / z), dim=1).view(-1, 2, 3)
        m = jac @ self.world2model_matrix.to(pos.device)[:3, :3].T
        return m @ cov3d @ m.transpose(1, 2)

    def project_gaussian(self, screen_coords: torch.Tensor, pos3d: torch.Tensor, cov3d: torch.Tensor, relax_factor: float = 1.3):
        """``Camera.project_gaussian`` (``Utils/Camera.py:89-117``) in tensor operations: ``(pos2d [M,3] in
        pixels with the NDC depth third, cov2d [M,2,2], mask [N])``, or ``(None, None, mask)`` when the mask
        is empty."""
        dev = pos3d.device
        homo = torch.nn.functional.pad(pos3d, (0, 1), value=1.0)
        clip = homo @ self.world2model_matrix.to(dev).T @ self.projection_matrix.to(dev).T
        ndc = clip[:, :3] / clip[:, 3:4]
        mask = (ndc[:, 2] >= 0.1) & (-relax_factor <= ndc[:, 0]) & (ndc[:, 0] <= relax_factor) \
            & (-relax_factor <= ndc[:, 1]) & (ndc[:, 1] <= relax_factor)
        if not bool(mask.any()):
            return None, None, mask
        shifted = screen_coords[mask] + ndc[mask]
        pos2d = torch.stack((0.5 * (shifted[:, 0] + 1) * self.width, (1 - 0.5 * (shifted[:, 1] + 1)) * self.height,
                             shifted[:, 2]), dim=1)
        return pos2d, self.get_covariance2d(homo[mask], cov3d[mask]), mask


def camera_constants(cam):
    """Return ``(w2m[16], proj[16], position[3])`` as contiguous float32 numpy
    arrays from either this module's ``Camera`` or the reference's."""
    if hasattr(cam, "w2m_host"):
        w2m, proj = cam.w2m_host, cam.proj_host
    else:
        w2m = cam.world2model_matrix.detach().cpu().numpy()
        proj = cam.projection_matrix.detach().cpu().numpy()
    w2m = np.ascontiguousarray(w2m, dtype=np.float32).reshape(16)
    proj = np.ascontiguousarray(proj, dtype=np.float32).reshape(16)
    # GaussianRenderer.py:73 casts the camera position to float32 before the subtraction.
    pos = np.ascontiguousarray(np.asarray(cam.transform.position), dtype=np.float32).reshape(3)
    return w2m, proj, pos


def compute_scene_radius(translations) -> np.ndarray:
    """``DataLoader.compute_scene_radius`` (``Utils/DataLoader.py:91-106``): per axis, the largest distance of
    a registered image's ``cam_from_world.translation`` from the mean of those translations.  Returns the
    three extents (the trainer uses their maximum, ``Trainer.py:33``)."""
    t = np.asarray(translations, dtype=np.float32).reshape(-1, 3)
    mean = np.zeros(3, dtype=np.float32)
    for row in t:                     # accumulated in fp32 in image order, as the reference does
        mean += row
    mean /= len(t)
    radius = np.zeros(3, dtype=np.float32)
    for row in t:
        radius = np.maximum(radius, np.abs(row - mean))
    return radius
