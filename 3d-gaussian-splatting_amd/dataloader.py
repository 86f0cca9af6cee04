"""Scene ingest on the host: the reference's ``DataLoader`` (``Utils/DataLoader.py:12-106``) over a
structure-from-motion reconstruction that already exists.

The reference builds the reconstruction itself with ``pycolmap`` (``extract_keypoint``,
``Utils/DataLoader.py:26-50``) and reads images with ``cv2``; neither package is part of this build, so this
class takes the reconstruction as an object (anything with the attributes the reference reads:
``points3D``, ``point3D_ids()``, ``num_points3D()``, ``reg_image_ids()``, ``images``, ``cameras``) and the
image reader as a callable.  Everything downstream of those two is the reference's behaviour: the point cloud
of ``get_pcd``, the shuffled train / validation split, ``sample`` with its bilinear rescale and the camera of
``Camera.from_sfm``, and the per-axis scene radius.  The draws come from Python's ``random`` module in the
reference's order, so a seed gives the reference's split and sampling sequence.
"""
import random
from typing import Callable, Optional

import numpy as np
import torch

from .camera import Camera, compute_scene_radius


class PointCloud:
    """The part of the reference's ``PointCloud`` (``Utils/PointCloud.py``) that ``get_pcd`` and
    ``GaussianModel.from_pcd`` use: float32 coordinates and colours (0..255) of ``size`` points."""

    def __init__(self, coords=None, colors=None):
        self._coords = np.empty((0, 3), dtype=np.float32)
        self._colors = np.empty((0, 3), dtype=np.float32)
        if coords is not None:
            self.set_coords(coords)
        if colors is not None:
            self.set_colors(colors)

    def set_coords(self, coords):
        self._coords = np.ascontiguousarray(coords, dtype=np.float32).reshape(-1, 3)

    def set_colors(self, colors):
        self._colors = np.ascontiguousarray(colors, dtype=np.float32).reshape(-1, 3)

    @property
    def coords(self) -> np.ndarray:
        return self._coords

    @property
    def colors(self) -> np.ndarray:
        return self._colors

    @property
    def size(self) -> int:
        return self._coords.shape[0]

    def __repr__(self):
        return f"coords: {self._coords.shape[0]} \t colors: {self._colors.shape[0]}"


def read_image_file(path: str) -> np.ndarray:
    """``[rows, columns, 3]`` uint8 RGB from a ``.npy`` array or a binary PPM (``P6``, maxval 255) file; other
    formats need ``cv2`` as in the reference (``Utils/DataLoader.py:81-82``) and raise when it is absent."""
    if path.endswith(".npy"):
        img = np.load(path)
    elif path.endswith(".ppm"):
        with open(path, "rb") as f:
            data = f.read()
        fields, pos = [], 0
        while len(fields) < 4:                      # magic, columns, rows, maxval, separated by whitespace
            while data[pos:pos + 1].isspace():
                pos += 1
            if data[pos:pos + 1] == b"#":
                pos = data.index(b"\n", pos) + 1
                continue
            end = pos
            while not data[end:end + 1].isspace():
                end += 1
            fields.append(data[pos:end])
            pos = end
        if fields[0] != b"P6" or int(fields[3]) != 255:
            raise ValueError(f"{path}: not a binary PPM with maxval 255")
        cols, rows = int(fields[1]), int(fields[2])
        img = np.frombuffer(data, dtype=np.uint8, count=rows * cols * 3, offset=pos + 1).reshape(rows, cols, 3)
    else:
        try:
            import cv2
        except ImportError as e:
            raise RuntimeError(f"reading {path} needs cv2 (absent): pass read_image= to DataLoader") from e
        img = cv2.cvtColor(cv2.imread(path), cv2.COLOR_BGR2RGB)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError(f"{path}: expected [rows, columns, 3], got {img.shape}")
    return img


class DataLoader:
    """``DataLoader(image_dir, sfm_dir, seed)`` of the reference with the reconstruction handed in
    (``reconstruction=``; the reference's ``extract_keypoint`` fills ``sfm_reconstruction`` instead)."""

    def __init__(self, image_dir: str, sfm_dir: str = "", seed: Optional[int] = None, reconstruction=None,
                 read_image: Callable[[str], np.ndarray] = read_image_file, device: Optional[str] = None):
        if seed is not None:
            random.seed(seed)
        self.image_dir = image_dir
        self.sfm_dir = sfm_dir
        self.database_file = f"{self.sfm_dir}/database.db"
        self.sparse_dir = f"{self.sfm_dir}/sparse"
        self.sfm_reconstruction = reconstruction
        self.train_image_ids = None
        self.validate_image_ids = None
        self.scene_radius = None
        self.read_image = read_image
        self.device = device

    def extract_keypoint(self):
        """The reference runs COLMAP here; this build has no ``pycolmap``."""
        if self.sfm_reconstruction is None:
            raise RuntimeError("no structure-from-motion package in this build: pass reconstruction= to DataLoader")

    def _reconstruction(self):
        if self.sfm_reconstruction is None:
            raise ValueError("SfM Reconstruction is None.")
        return self.sfm_reconstruction

    def get_pcd(self) -> PointCloud:
        """One point per 3-D point of the reconstruction, in the order of ``point3D_ids()``
        (``Utils/DataLoader.py:52-63``)."""
        rec = self._reconstruction()
        ids = list(rec.point3D_ids())
        points = np.zeros((len(ids), 3), dtype=np.float32)
        colors = np.zeros((len(ids), 3), dtype=np.float32)
        for row, point_id in enumerate(ids):
            point = rec.points3D[point_id]
            points[row] = point.xyz
            colors[row] = point.color
        return PointCloud(points, colors)

    def split_train_validate_data(self, ratio: float = 0.8):
        """Shuffles the registered image ids and cuts them at ``int(len * ratio)`` (``Utils/DataLoader.py:65-70``)."""
        image_ids = list(self._reconstruction().reg_image_ids())
        random.shuffle(image_ids)
        cut = int(len(image_ids) * ratio)
        self.train_image_ids = image_ids[:cut]
        self.validate_image_ids = image_ids[cut:]

    def sample(self, is_train: bool = True, image_scale: float = 1):
        """A random image of the split as ``(Camera, image)``: the image is the file's ``[rows, columns, 3]``
        RGB array as float32 0..255, rescaled bilinearly by ``image_scale``; the camera takes the image's first
        axis as its width and the second as its height, as the reference does (``Utils/DataLoader.py:72-89``)."""
        rec = self._reconstruction()
        image_ids = self.train_image_ids if is_train else self.validate_image_ids
        if not image_ids:
            raise ValueError("split_train_validate_data() has not been called, or the split is empty")
        sfm_image = rec.images[random.choice(image_ids)]
        sfm_camera = rec.cameras[sfm_image.camera_id]
        device = self.device or ("cuda" if torch.cuda.is_available() else "cpu")
        pixels = np.ascontiguousarray(self.read_image(self.image_dir + "/" + sfm_image.name))
        image = torch.from_numpy(pixels).float().to(device)
        if image_scale != 1:
            planes = image.permute(2, 0, 1)[None]
            planes = torch.nn.functional.interpolate(planes, scale_factor=image_scale, mode="bilinear")
            image = planes[0].permute(1, 2, 0)
        return Camera.from_sfm(sfm_image, sfm_camera, image.shape[0], image.shape[1], device=device), image

    def compute_scene_radius(self):
        """Per-axis extent of the camera translations about their mean (``Utils/DataLoader.py:91-106``)."""
        rec = self._reconstruction()
        translations = [rec.images[i].cam_from_world.translation for i in rec.reg_image_ids()]
        self.scene_radius = compute_scene_radius(translations)
        return self.scene_radius
