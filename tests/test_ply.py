"""CPU: the reference's model file format (GaussianModel.save_ply / load_ply, GaussianModel.py:256-318)
written and read by ply.py.  The reference goes through the plyfile package, which is absent here and
on the GPU box, so there is no file written by the reference to compare with (parity unpinned): the
tests pin the documented layout - header text, property order, float32 little-endian rows - and the
round trip."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_model(cap=7, size=5, seed=0):
    import gsplat_amd  # noqa: F401
    from gsplat_amd.model import GaussianModel
    g = torch.Generator().manual_seed(seed)
    return GaussianModel(torch.randn(cap, 3, generator=g), torch.randn(cap, 3, generator=g), torch.randn(cap, 4, generator=g),
                         torch.randn(cap, 1, generator=g), torch.randn(cap, 3, 16, generator=g), sh_degree=2, size=size)


def test_header_and_row_layout(tmp_path):
    m = make_model()
    path = str(tmp_path / "model.ply")
    m.save_ply(path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode("ascii").split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 7"]     # the whole capacity is saved
    props = [l.split()[2] for l in lines[3:] if l]
    assert all(l.startswith("property float ") for l in lines[3:] if l)
    assert props == (["x", "y", "z", "nx", "ny", "nz"] + [f"sh_{i}" for i in range(48)] + ["opacity", "scale_0", "scale_1",
                     "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    rows = np.frombuffer(body, dtype="<f4").reshape(7, 62)
    np.testing.assert_array_equal(rows[:, 0:3], m._coords.detach().numpy())
    assert not rows[:, 3:6].any()
    np.testing.assert_array_equal(rows[:, 6:54], m._sh.detach().numpy().reshape(7, 48))      # channel-major [3,16]
    np.testing.assert_array_equal(rows[:, 54], m._opacity.detach().numpy()[:, 0])
    np.testing.assert_array_equal(rows[:, 55:58], m._scale.detach().numpy())
    np.testing.assert_array_equal(rows[:, 58:62], m._rotation.detach().numpy())


def test_round_trip(tmp_path):
    m = make_model()
    path = str(tmp_path / "model.ply")
    m.save_ply(path)
    other = make_model(cap=3, size=3, seed=9)
    other._coords_grads += 1.0
    other.load_ply(path)
    assert other._size == 7 and other.capacity == 7                  # every row of the file is live, GaussianModel.py:289
    for name in ("_coords", "_scale", "_rotation", "_opacity", "_sh"):
        got, want = getattr(other, name), getattr(m, name)
        assert isinstance(got, torch.nn.Parameter) and got.requires_grad and got.shape == want.shape
        assert torch.equal(got.detach(), want.detach())
    assert other._coords_grads.shape == (7, 1) and not other._coords_grads.any() and other._denom.dtype == torch.int32


def test_reads_ascii_and_big_endian_and_shuffled_properties(tmp_path):
    import gsplat_amd  # noqa: F401
    from gsplat_amd import ply
    names = ply.attribute_names()
    rng = np.random.default_rng(3)
    table = rng.normal(size=(4, len(names))).astype(np.float32)
    order = list(reversed(range(len(names))))                       # properties in another order
    head = "ply\nformat {fmt} 1.0\ncomment made by a test\nelement vertex 4\n" + "".join(
        f"property float {names[i]}\n" for i in order) + "element face 0\nproperty list uchar int vertex_indices\nend_header\n"
    p_ascii, p_be = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    with open(p_ascii, "w") as f:
        f.write(head.format(fmt="ascii"))
        for r in table:
            f.write(" ".join(repr(float(r[i])) for i in order) + "\n")
    with open(p_be, "wb") as f:
        f.write(head.format(fmt="binary_big_endian").encode("ascii"))
        f.write(table[:, order].astype(">f4").tobytes())
    for path in (p_ascii, p_be):
        cols = ply.read_vertex_table(path)
        for i, n in enumerate(names):
            np.testing.assert_array_equal(cols[n].astype(np.float32), table[:, i])
        m = make_model()
        m.load_ply(path)
        np.testing.assert_array_equal(m._sh.detach().numpy().reshape(4, 48), table[:, 6:54])
        np.testing.assert_array_equal(m._rotation.detach().numpy(), table[:, 58:62])


def test_rejects_what_the_reference_rejects(tmp_path):
    import gsplat_amd  # noqa: F401
    from gsplat_amd import ply
    path = str(tmp_path / "short.ply")
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 0\nproperty float x\nproperty float y\nproperty float z\n"
                b"property float opacity\nproperty float sh_0\nend_header\n")
    with pytest.raises(ValueError):
        make_model().load_ply(path)                                  # 1 sh property instead of 48 (GaussianModel.py:294)
    with open(path, "wb") as f:
        f.write(b"not a ply\n")
    with pytest.raises(ValueError):
        ply.read_vertex_table(path)
