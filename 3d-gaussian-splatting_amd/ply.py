"""The reference's model file format: ``GaussianModel.save_ply`` / ``load_ply``
(``GaussianModel.py:256-318``), written and read with numpy alone.

The reference goes through the ``plyfile`` package, which is not part of this tree; the format is the
standard PLY container, one ``vertex`` element whose float32 properties are, in this order
(``GaussianModel.py:256-266,278``): ``x y z nx ny nz sh_0 .. sh_47 opacity scale_0 scale_1 scale_2
rot_0 rot_1 rot_2 rot_3`` - the raw parameter tensors over the whole CAPACITY (the reference saves
``_coords`` etc. unsliced and makes every loaded row live), normals zero, ``sh`` flattened
channel-major ``[3,16]``.  ``plyfile`` writes ``binary_little_endian`` on a little-endian host; the
reader here also accepts ``ascii`` and ``binary_big_endian`` files and properties in any order.
Host-side I/O only: tensors go through the CPU.
"""
import numpy as np
import torch
import torch.nn as nn

_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
          "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4",
          "float64": "f8"}


def attribute_names(sh_floats: int = 48, scale_floats: int = 3, rot_floats: int = 4):
    """``GaussianModel.construct_list_of_attributes`` (``GaussianModel.py:256-266``)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"sh_{i}" for i in range(sh_floats)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(scale_floats)]
    names += [f"rot_{i}" for i in range(rot_floats)]
    return names


def save_ply(model, path: str):
    """``GaussianModel.save_ply`` (``GaussianModel.py:268-281``)."""
    coords = model._coords.detach().cpu().numpy().astype(np.float32)
    rows = coords.shape[0]
    sh = model._sh.detach().flatten(start_dim=1).contiguous().cpu().numpy().astype(np.float32)
    opacity = model._opacity.detach().cpu().numpy().astype(np.float32).reshape(rows, 1)
    scale = model._scale.detach().cpu().numpy().astype(np.float32)
    rotation = model._rotation.detach().cpu().numpy().astype(np.float32)
    table = np.concatenate((coords, np.zeros_like(coords), sh, opacity, scale, rotation), axis=1).astype("<f4")
    names = attribute_names(sh.shape[1], scale.shape[1], rotation.shape[1])
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % rows
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table).tobytes())


def read_vertex_table(path: str):
    """-> dict name -> 1-D numpy array of the ``vertex`` element of a PLY file."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex, seen_vertex = None, 0, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header without end_header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    if props:
                        raise ValueError(f"{path}: two vertex elements")
                    count, seen_vertex = int(tok[2]), True
                elif not seen_vertex:
                    raise ValueError(f"{path}: an element precedes 'vertex' (only vertex-first files are supported)")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property in the vertex element")
                if tok[1] not in _TYPES:
                    raise ValueError(f"{path}: unknown property type {tok[1]}")
                props.append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if not seen_vertex:
            raise ValueError(f"{path}: no vertex element")
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(count)]
            cols = {n: np.array([r[i] for r in rows], dtype=np.float64).astype(t) for i, (n, t) in enumerate(props)}
            return cols
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported format {fmt}")
        order = "<" if fmt == "binary_little_endian" else ">"
        dtype = np.dtype([(n, order + t) for n, t in props])
        data = np.frombuffer(f.read(count * dtype.itemsize), dtype=dtype, count=count)
        return {n: np.asarray(data[n]) for n, _ in props}


def _numbered(cols, prefix):
    names = [n for n in cols if n.startswith(prefix)]
    return sorted(names, key=lambda n: int(n.split("_")[-1]))


def load_ply(model, path: str, device=None):
    """``GaussianModel.load_ply`` (``GaussianModel.py:283-318``): every row of the file becomes a live
    Gaussian, the densification statistics start at zero."""
    cols = read_vertex_table(path)
    device = device if device is not None else getattr(model, "_device", "cpu")
    coords = np.stack((cols["x"], cols["y"], cols["z"]), axis=1)
    opacity = np.asarray(cols["opacity"])[..., np.newaxis]
    max_deg = getattr(model, "max_sh_degree", 3)
    sh_names = _numbered(cols, "sh_")
    if len(sh_names) != 3 * (max_deg + 1) ** 2:                                       # the reference asserts this
        raise ValueError(f"{path}: {len(sh_names)} sh properties, expected {3 * (max_deg + 1) ** 2}")
    sh = np.stack([cols[n] for n in sh_names], axis=1).reshape(coords.shape[0], 3, (max_deg + 1) ** 2)
    scale = np.stack([cols[n] for n in _numbered(cols, "scale_")], axis=1)
    rot = np.stack([cols[n] for n in _numbered(cols, "rot")], axis=1)

    def param(a):
        native = np.ascontiguousarray(a, dtype=np.float32)      # also converts a big-endian file to the host byte order
        return nn.Parameter(torch.tensor(native, dtype=torch.float, device=device).requires_grad_(True))

    model._size = coords.shape[0]
    model._device = device
    model._coords, model._sh, model._opacity = param(coords), param(sh), param(opacity)
    model._scale, model._rotation = param(scale), param(rot)
    model._coords_grads = torch.zeros((coords.shape[0], 1), dtype=torch.float, device=device)
    model._denom = torch.zeros((coords.shape[0], 1), dtype=torch.int, device=device)
    return model
