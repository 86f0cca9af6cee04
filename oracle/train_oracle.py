"""CPU restatement (numpy, fp32) of the reference's per-iteration training-state updates.
TEST INFRASTRUCTURE ONLY: imported by ``tests/`` alone; the shipped package never imports it.

Covers what ``Trainer.train`` does to the model around a render (``Trainer.py:57-80``):

* ``adam_step``      - ``optimizer.step()`` (``Trainer.py:80``) of the ``torch.optim.Adam`` built at
                       ``Trainer.py:24`` (lr, betas given; eps 1e-8, no weight decay, no amsgrad).  The
                       algorithm lives in PyTorch (requirements.txt pins torch~=2.2.0), not in the
                       reference tree; this restates its single-tensor path and is pinned by fixtures
                       generated with ``torch.optim.Adam`` itself.
* ``densify_stats``  - ``GaussianModel.update_densify_stats`` (``GaussianModel.py:230-232``).
* ``densify``        - ``GaussianModel.densify`` with ``clone`` and ``split``
                       (``GaussianModel.py:179-245``), incl. ``expand_capacity`` (``:116-140``).
* ``remove``         - ``GaussianModel.remove(mask, inplace=True)`` (``GaussianModel.py:142-164``).

The model is a dict of arrays of CAPACITY rows: coords[C,3] scale[C,3] rotation[C,4] opacity[C,1]
sh[C,3,16] coords_grads[C,1] (f32) denom[C,1] (i32) plus the int "size".  Functions return new dicts.
"""
import numpy as np

F32 = np.float32
PARAM_KEYS = ("coords", "scale", "rotation", "opacity", "sh")


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps=1e-8):
    """One Adam update of one tensor; ``step`` is the 1-based count of this update.
    Scalars are formed in double precision on the host and applied in fp32, as PyTorch does."""
    p, g = np.asarray(param, F32), np.asarray(grad, F32)
    m, v = np.asarray(exp_avg, F32), np.asarray(exp_avg_sq, F32)
    w = F32(1.0 - beta1)
    diff = g - m
    # Tensor.lerp_: start + w * diff for w < 0.5, end - diff * (1 - w) otherwise
    m = (m + w * diff) if abs(float(w)) < 0.5 else (g - diff * (F32(1.0) - w))
    v = v * F32(beta2) + (F32(1.0 - beta2) * g) * g
    bias1 = 1.0 - beta1 ** step
    bias2 = 1.0 - beta2 ** step
    step_size = lr / bias1
    denom = np.sqrt(v) / F32(bias2 ** 0.5) + F32(eps)
    p = p + F32(-step_size) * (m / denom)
    return p.astype(F32), m.astype(F32), v.astype(F32)


def densify_stats(model, g_screen, mask):
    """GaussianModel.py:230-232: coords_grads[mask] += |g_screen[mask, :2]|, denom[mask] += 1."""
    out = {k: np.array(v, copy=True) for k, v in model.items()}
    n = int(model["size"])
    mask = np.asarray(mask, bool)
    g = np.asarray(g_screen, F32)
    norm = np.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]).astype(F32)
    out["coords_grads"][:n, 0][mask] += norm[mask]
    out["denom"][:n, 0][mask] += 1
    return out


def _expand(model, new_capacity):
    """GaussianModel.py:116-140: zero rows appended to every tensor."""
    cap = model["coords"].shape[0]
    if cap >= new_capacity:
        return model
    out = dict(model)
    for k in PARAM_KEYS + ("coords_grads", "denom"):
        a = model[k]
        pad = np.zeros((new_capacity - cap,) + a.shape[1:], a.dtype)
        out[k] = np.concatenate([a, pad], axis=0)
    return out


def rotation_matrices(q):
    """Utils/Transform.py:55-63 for unit quaternions [w, x, y, z] -> [n,3,3]."""
    w, x, y, z = (q[:, i] for i in range(4))
    two = F32(2.0)
    one = F32(1.0)
    R = np.zeros((q.shape[0], 3, 3), F32)
    R[:, 0, 0] = two * (w * w + x * x) - one
    R[:, 0, 1] = two * (x * y - w * z)
    R[:, 0, 2] = two * (x * z + w * y)
    R[:, 1, 0] = two * (x * y + w * z)
    R[:, 1, 1] = two * (w * w + y * y) - one
    R[:, 1, 2] = two * (y * z - w * x)
    R[:, 2, 0] = two * (x * z - w * y)
    R[:, 2, 1] = two * (y * z + w * x)
    R[:, 2, 2] = two * (w * w + z * z) - one
    return R


def normalize(q):
    """torch.nn.functional.normalize (eps 1e-12), GaussianModel.py:30,56."""
    nrm = np.sqrt((q * q).sum(axis=1, keepdims=True)).astype(F32)
    return (q / np.maximum(nrm, F32(1e-12))).astype(F32)


def clone(model, mask):
    """GaussianModel.clone (GaussianModel.py:179-197): the masked rows appended after the live rows, the
    capacity doubled when the new size reaches it, the statistics zeroed.  Returns (model, num_clone)."""
    m = {k: np.array(v, copy=True) for k, v in model.items()}
    size = int(m["size"])
    mask = np.asarray(mask, bool)
    num_clone = int(mask.sum())
    if num_clone:
        new_size = size + num_clone
        if new_size >= m["coords"].shape[0]:
            m = _expand(m, 2 * m["coords"].shape[0])
        for k in PARAM_KEYS:
            m[k][size:new_size] = m[k][:size][mask]
        m["coords_grads"] = np.zeros((m["coords"].shape[0], 1), F32)
        m["denom"] = np.zeros((m["coords"].shape[0], 1), np.int32)
        size = new_size
    m["size"] = size
    return m, num_clone


def split(model, mask, noise, split_factor=2):
    """GaussianModel.split (GaussianModel.py:199-228) with split_factor 2.

    ``noise``: the standard-normal draws behind ``torch.normal(0, std)``, shape [2 * num_split, 3] (or more
    rows; the first 2 * num_split are used): sample = noise * std.  Returns (model, num_split)."""
    assert split_factor == 2
    m = {k: np.array(v, copy=True) for k, v in model.items()}
    size = int(m["size"])
    split_mask = np.asarray(mask, bool)
    num_split = int(split_mask.sum())
    if num_split:
        new_size = size + num_split
        if new_size >= m["coords"].shape[0]:
            m = _expand(m, 2 * m["coords"].shape[0])
        std = np.exp(m["scale"][:size][split_mask]).astype(F32)                     # activated scale
        stds = np.concatenate([std, std], axis=0)
        samples = (np.asarray(noise, F32)[:2 * num_split] * stds).astype(F32)       # torch.normal(0, std)
        R = rotation_matrices(normalize(m["rotation"][:size][split_mask]))
        R2 = np.concatenate([R, R], axis=0)
        # row vector times R^T (:211-212); the 4x4 rotation has no translation, so the offset added at
        # :212 is zero: the new centres are NOT displaced by the old centre
        means = np.zeros_like(samples)
        for j in range(3):
            acc = np.zeros(samples.shape[0], F32)
            for kk in range(3):
                acc = acc + samples[:, kk] * R2[:, j, kk]
            means[:, j] = acc
        means = means + F32(0.0)
        scales = np.log(stds / F32(0.8 * split_factor)).astype(F32)                 # :213
        rows = np.nonzero(split_mask)[0]
        m["coords"][rows] = means[:num_split]
        m["scale"][rows] = scales[:num_split]
        m["coords"][size:new_size] = means[num_split:]
        m["scale"][size:new_size] = scales[num_split:]
        for key in ("rotation", "opacity", "sh"):
            m[key][size:new_size] = m[key][rows]
        m["coords_grads"] = np.zeros((m["coords"].shape[0], 1), F32)
        m["denom"] = np.zeros((m["coords"].shape[0], 1), np.int32)
        size = new_size
    m["size"] = size
    return m, num_split


def densify(model, scene_radius, noise, grad_threshold=0.0002, split_factor=2):
    """GaussianModel.py:234-245: the two selections around ``clone`` and ``split`` above.
    Returns (model, num_clone, num_split)."""
    assert split_factor == 2
    m = {k: np.array(v, copy=True) for k, v in model.items()}
    size = int(m["size"])
    with np.errstate(divide="ignore", invalid="ignore"):
        grads = (m["coords_grads"] / m["denom"].astype(F32)).astype(F32)   # :236, over the capacity
    grads[np.isnan(grads)] = 0.0                                           # :237
    thr = F32(0.01) * F32(scene_radius)
    smax = m["scale"][:size].max(axis=1)                                   # raw (log) scale, as the reference compares it
    clone_mask = (np.abs(grads[:size, 0]) >= F32(grad_threshold)) & (smax <= thr)   # :240
    m, num_clone = clone(m, clone_mask)
    size = int(m["size"])
    # split mask (:242-244): the averaged statistics of the rows below the OLD capacity
    pad = np.zeros((size, 1), F32)
    k = min(size, grads.shape[0])
    pad[:k] = grads[:size]
    smax = m["scale"][:size].max(axis=1)
    split_mask = (np.abs(pad[:, 0]) >= F32(grad_threshold)) & (smax > thr)
    m, num_split = split(m, split_mask, noise, split_factor)
    return m, num_clone, num_split


def remove(model, mask):
    """GaussianModel.py:148-164, literally: with K the kept rows in order and r rows to remove, the last
    r kept rows are written to rows [size - 2r, size - r) - opacity, rotation and scale in their
    ACTIVATED form (sigmoid, normalised, exp), the rest raw - and the size drops by r.  Rows below
    size - 2r stay as they are, removed or not."""
    m = {k: np.array(v, copy=True) for k, v in model.items()}
    size = int(m["size"])
    mask = np.asarray(mask, bool)
    r = int(mask.sum())
    if r == 0:
        return m
    keep = ~mask
    start = size - 2 * r
    if start < 0:
        raise RuntimeError("remove: more than half of the rows are to be removed (the reference fails on the slice shapes)")
    src = np.nonzero(keep)[0][-r:]
    act_opacity = (F32(1.0) / (F32(1.0) + np.exp(-m["opacity"][src]))).astype(F32)
    act_rot = normalize(m["rotation"][src])
    act_scale = np.exp(m["scale"][src]).astype(F32)
    vals = {"coords": m["coords"][src], "opacity": act_opacity, "rotation": act_rot, "sh": m["sh"][src],
            "scale": act_scale, "coords_grads": m["coords_grads"][src], "denom": m["denom"][src]}
    for k, v in vals.items():
        m[k][start:start + r] = v
    m["size"] = size - r
    return m


def knn_dist2(points):
    """distCUDA2 of the simple_knn extension as GaussianModel.from_pcd uses it (GaussianModel.py:82):
    per point the mean of the squared distances to its three nearest other points (its own index
    skipped, coincident points included with distance zero).  The extension is Inria's simple-knn
    (un-vendored and unpinned in the reference, absent here); this restates its published result with
    an exact k-d tree search in double precision.  PARITY UNPINNED: the reference holds no output of it."""
    from scipy.spatial import cKDTree
    p = np.asarray(points, np.float64)
    n = p.shape[0]
    assert n >= 4
    # coincident points make "the point itself" ambiguous among the zero distances: ask for enough
    # neighbours to drop the query's own index explicitly
    k = 4
    while True:
        dist, idx = cKDTree(p).query(p, k=min(k, n))
        own = idx == np.arange(n)[:, None]
        if own.any(axis=1).all() or k >= n:
            break
        k *= 2
    out = np.empty(n, np.float64)
    for i in range(n):
        d = dist[i][~own[i]][:3] if own[i].any() else dist[i][:3]
        out[i] = (d[0] ** 2 + d[1] ** 2 + d[2] ** 2) / 3.0
    return out.astype(F32)


def from_pcd(coords, colors, max_sh_degree=3):
    """GaussianModel.from_pcd (GaussianModel.py:71-97) given the neighbour distances of knn_dist2."""
    coords = np.asarray(coords, F32)
    n = coords.shape[0]
    sh = np.zeros((n, 3, (max_sh_degree + 1) ** 2), F32)
    sh[:, :, 0] = (np.asarray(colors)[:, :3].astype(F32) / F32(255) - F32(0.5)) / F32(0.28209479177387814)
    dist2 = np.maximum(knn_dist2(coords), F32(0.0000001))
    scale = np.repeat(np.log(np.sqrt(dist2))[:, None], 3, axis=1).astype(F32)
    rotation = np.zeros((n, 4), F32)
    rotation[:, 0] = 1
    return {"coords": coords, "scale": scale, "rotation": rotation, "opacity": np.ones((n, 1), F32), "sh": sh, "size": n}
