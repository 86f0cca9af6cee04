"""Generates tests/golden/adam_*.npz, stats_*.npz, densify_*.npz, clone_*.npz, split_*.npz and remove_*.npz by running the
REFERENCE's training-state code on the CPU:

* ``torch.optim.Adam(params, lr=1e-3, betas=(0.5, 0.999))`` exactly as ``Trainer.py:24`` builds it,
  stepped as ``Trainer.py:53,80`` do;
* ``GaussianModel.update_densify_stats / densify / clone / split / remove`` imported from
  /root/reference (``GaussianModel.py:142-245``), called under ``torch.no_grad()`` as
  ``Trainer.py:60-69`` does.

Run in the build container only (needs /root/reference):

    python tests/golden/make_train_golden.py

The split draws ``torch.normal(0, std)``; the fixture stores the standard-normal draws behind it
(``torch.randn`` under the same seed - the generator asserts that noise * std reproduces the
reference's samples), so the draws are an INPUT of the parity tests.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

KEYS = ("coords", "scale", "rotation", "opacity", "sh")


def new_model(GM, cap, size, gen, scale_mean=-1.0):
    m = GM.GaussianModel()
    m._size = size
    m._device = "cpu"
    m._coords = torch.nn.Parameter(torch.randn(cap, 3, generator=gen))
    m._scale = torch.nn.Parameter(torch.randn(cap, 3, generator=gen) * 0.5 + scale_mean)
    m._rotation = torch.nn.Parameter(torch.randn(cap, 4, generator=gen))
    m._opacity = torch.nn.Parameter(torch.randn(cap, 1, generator=gen) * 2)
    m._sh = torch.nn.Parameter(torch.randn(cap, 3, 16, generator=gen))
    m._coords_grads = torch.zeros(cap, 1)
    m._denom = torch.zeros(cap, 1, dtype=torch.int32)
    return m


def snapshot(m, prefix):
    out = {prefix + "size": np.int64(int(m._size))}
    for k in KEYS:
        out[prefix + k] = getattr(m, "_" + k).detach().numpy().copy()
    out[prefix + "coords_grads"] = m._coords_grads.numpy().copy()
    out[prefix + "denom"] = m._denom.numpy().copy()
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def make_adam():
    # shapes of the five parameter tensors in the order model.parameters() yields them
    # (GaussianModel.py:87-91), two row counts, 7 steps; step 3 leaves two tensors without a gradient
    # (zero_grad(set_to_none=True), Trainer.py:53: Adam skips them and their step count)
    for name, n, lr, betas in (("adam_n37", 37, 1e-3, (0.5, 0.999)), ("adam_n128_b09", 128, 5e-2, (0.9, 0.99))):
        gen = torch.Generator().manual_seed(700 + n)
        shapes = [(n, 3), (n, 1), (n, 3, 16), (n, 3), (n, 4)]
        params = [torch.nn.Parameter(torch.randn(s, generator=gen)) for s in shapes]
        opt = torch.optim.Adam(params, lr=lr, betas=betas)
        out = {"lr": np.float64(lr), "beta1": np.float64(betas[0]), "beta2": np.float64(betas[1]), "steps": np.int64(7)}
        for i, p in enumerate(params):
            out[f"p{i}_init"] = p.detach().numpy().copy()
        for step in range(7):
            opt.zero_grad(set_to_none=True)
            for i, p in enumerate(params):
                if step == 3 and i in (1, 4):
                    out[f"g{i}_s{step}"] = np.zeros(0, np.float32)   # no gradient this step
                    continue
                scale = 10.0 ** float(torch.randint(-6, 3, (1,), generator=gen))
                g = torch.randn(p.shape, generator=gen) * scale
                if step == 5:
                    g[::3] = 0.0      # rows that received no gradient (culled Gaussians)
                p.grad = g
                out[f"g{i}_s{step}"] = g.numpy().copy()
            opt.step()
            for i, p in enumerate(params):
                out[f"p{i}_s{step}"] = p.detach().numpy().copy()
        st = opt.state_dict()["state"]
        for i in range(len(params)):
            out[f"m{i}_final"] = st[i]["exp_avg"].numpy().copy()
            out[f"v{i}_final"] = st[i]["exp_avg_sq"].numpy().copy()
            out[f"step{i}_final"] = np.int64(int(st[i]["step"]))
        save(name, **out)


def make_stats(GM):
    gen = torch.Generator().manual_seed(811)
    cap, size = 40, 33
    m = new_model(GM, cap, size, gen)
    out = snapshot(m, "in_")
    for it in range(3):
        mask = torch.rand(size, generator=gen) < 0.6
        sc = torch.zeros(size, 3, requires_grad=True)
        sc.grad = torch.randn(size, 3, generator=gen) * 10.0 ** float(torch.randint(-5, 0, (1,), generator=gen))
        with torch.no_grad():
            m.update_densify_stats(sc, mask)
        out[f"mask_{it}"] = mask.numpy().copy()
        out[f"g_screen_{it}"] = sc.grad.numpy().copy()
        out.update(snapshot(m, f"out{it}_"))
    out["iters"] = np.int64(3)
    save("stats_n33", **out)


def set_stats(m, gen, frac_hot, stale=True, min_denom=0):
    size, cap = int(m._size), m.capacity
    hot = torch.rand(size, generator=gen) < frac_hot
    denom = torch.randint(min_denom, 4, (size,), generator=gen).to(torch.int32)
    g = torch.where(hot, torch.rand(size, generator=gen) * 0.01 + 0.0003, torch.rand(size, generator=gen) * 0.00019)
    m._coords_grads[:size, 0] = g * denom          # a zero count leaves 0/0 = nan -> 0
    m._denom[:size, 0] = denom
    if stale and cap > size:
        # statistics left behind by an earlier remove() in rows at and above the size
        m._coords_grads[size:, 0] = torch.rand(cap - size, generator=gen) * 0.02
        m._denom[size:, 0] = 1


def make_densify(GM):
    # name: capacity, size, scene radius, fraction of rows with a large gradient, mean log-scale
    cases = {
        "densify_room": (64, 20, 60.0, 0.6, 0.3),           # clone and split, capacity suffices
        "densify_grow_clone": (24, 20, 200.0, 0.9, 0.3),    # clones only; the clone doubles the capacity
        "densify_grow_split": (30, 20, 2.0, 0.9, 0.3),      # splits only; the split doubles the capacity
        "densify_grow_twice": (20, 20, 60.0, 1.0, 0.3),     # every row selected: both double it
        "densify_none": (40, 30, 60.0, 0.0, 0.3),           # nothing selected: the statistics persist
        "densify_full": (300, 257, 60.0, 0.5, 0.3),         # several waves of rows
    }
    for seed, (name, (cap, size, radius, hot, smean)) in enumerate(cases.items()):
        gen = torch.Generator().manual_seed(900 + seed)
        m = new_model(GM, cap, size, gen, scale_mean=smean)
        set_stats(m, gen, hot, min_denom=1 if name == "densify_grow_twice" else 0)
        out = snapshot(m, "in_")
        out["scene_radius"] = np.float32(radius)
        pre_scale = m._scale.detach().clone()
        torch.manual_seed(4000 + seed)
        noise = torch.randn(4 * size, 3)          # more than any split can use; a prefix is what the split draws
        torch.manual_seed(4000 + seed)
        with torch.no_grad():
            m.densify(torch.tensor(radius))
        post = snapshot(m, "out_")
        out.update(post)
        # recover the counts and check the noise equivalence on the new split rows
        new_size = int(m._size)
        thr = np.float32(0.01) * np.float32(radius)
        smax = pre_scale[:size].max(dim=1).values.numpy()
        with np.errstate(divide="ignore", invalid="ignore"):
            g = out["in_coords_grads"][:size, 0] / out["in_denom"][:size, 0].astype(np.float32)
        g[np.isnan(g)] = 0
        n_clone = int(((g >= 0.0002) & (smax <= thr)).sum())
        n_split = new_size - size - n_clone
        out["num_clone"], out["num_split"] = np.int64(n_clone), np.int64(n_split)
        if n_split:
            # torch.normal draws exactly 2 * n_split rows: regenerate them at that length
            torch.manual_seed(4000 + seed)
            noise = torch.randn(2 * n_split, 3)
            rows = np.nonzero((g >= 0.0002) & (smax > thr))[0]
            assert len(rows) == n_split
            std = torch.exp(pre_scale[rows])
            want = torch.log(std / 1.6).numpy()
            np.testing.assert_allclose(post["out_scale"][rows], want, rtol=1e-6)
        out["noise"] = noise.numpy().copy()
        print(f"  {name}: size {size} -> {new_size} (clone {n_clone}, split {n_split}), capacity {cap} -> {m.capacity}")
        save(name, **out)


def make_clone_split(GM):
    """GaussianModel.clone(mask) and GaussianModel.split(mask) (GaussianModel.py:179-228) called on their own
    with arbitrary masks, as a caller other than densify would."""
    # name: method, capacity, size, fraction of rows in the mask
    cases = {
        "clone_room": ("clone", 64, 40, 0.3),
        "clone_grow": ("clone", 44, 40, 0.5),         # the new size reaches the capacity: doubled
        "clone_none": ("clone", 64, 40, 0.0),         # empty mask: nothing changes, the statistics persist
        "split_room": ("split", 300, 130, 0.4),
        "split_grow": ("split", 40, 33, 0.6),
        "split_none": ("split", 64, 40, 0.0),
    }
    for seed, (name, (method, cap, size, frac)) in enumerate(cases.items()):
        gen = torch.Generator().manual_seed(980 + seed)
        m = new_model(GM, cap, size, gen, scale_mean=-0.5)
        set_stats(m, gen, 0.5)
        mask = torch.rand(size, generator=gen) < frac
        out = snapshot(m, "in_")
        out["mask"] = mask.numpy().copy()
        n = int(mask.sum())
        torch.manual_seed(5000 + seed)
        noise = torch.randn(max(2 * n, 1), 3)        # what torch.normal draws below, standardised
        torch.manual_seed(5000 + seed)
        pre_scale = m._scale.detach().clone()
        with torch.no_grad():
            getattr(m, method)(mask)
        out.update(snapshot(m, "out_"))
        if method == "split" and n:
            rows = np.nonzero(mask.numpy())[0]
            want = torch.log(torch.exp(pre_scale[rows]) / 1.6).numpy()
            np.testing.assert_allclose(out["out_scale"][rows], want, rtol=1e-6)
        out["noise"] = noise.numpy().copy()
        print(f"  {name}: size {size} -> {int(m._size)}, capacity {cap} -> {m.capacity}")
        save(name, **out)


def make_remove(GM):
    for seed, (name, cap, size, frac) in enumerate((("remove_few", 40, 33, 0.1), ("remove_many", 300, 280, 0.45),
                                                     ("remove_none", 20, 16, 0.0))):
        gen = torch.Generator().manual_seed(950 + seed)
        m = new_model(GM, cap, size, gen)
        set_stats(m, gen, 0.5)
        mask = torch.rand(size, generator=gen) < frac
        out = snapshot(m, "in_")
        out["mask"] = mask.numpy().copy()
        with torch.no_grad():
            m.remove(mask)
        out.update(snapshot(m, "out_"))
        print(f"  {name}: size {size} -> {int(m._size)}")
        save(name, **out)


def make_camera(RefCamera):
    """Camera.from_sfm (Utils/Camera.py:30-46) on stand-ins for pycolmap's Image and Camera that carry the
    attributes the reference reads; the reference's own matrices are the expected output."""
    import types
    rng = np.random.default_rng(77)
    out = {"cases": np.int64(5)}
    for k in range(5):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)                                   # x, y, z, w
        center = rng.normal(size=3) * 3
        fx, fy = 500 + 300 * rng.random(), 480 + 300 * rng.random()
        w, h = int(rng.integers(100, 900)), int(rng.integers(100, 900))
        cx, cy = w / 2 + rng.normal(), h / 2 + rng.normal()
        image = types.SimpleNamespace(cam_from_world=types.SimpleNamespace(rotation=types.SimpleNamespace(quat=q.copy())),
                                      projection_center=lambda c=center: c.copy())
        cam_model = types.SimpleNamespace(focal_length_x=fx, focal_length_y=fy, principal_point_x=cx, principal_point_y=cy)
        cam = RefCamera.from_sfm(image, cam_model, w, h)
        out[f"quat_xyzw_{k}"], out[f"center_{k}"] = q, center
        out[f"intrinsics_{k}"] = np.array([fx, fy, cx, cy, w, h], np.float64)
        out[f"w2m_{k}"] = cam.world2model_matrix.numpy().copy()
        out[f"proj_{k}"] = cam.projection_matrix.numpy().copy()
        out[f"position_{k}"] = np.asarray(cam.transform.position, np.float64)
    save("camera_from_sfm", **out)


def main():
    assert ref_harness.reference_available(), "needs /root/reference"
    GM, _, RefCamera, _ = ref_harness._import_reference()
    make_camera(RefCamera)
    make_adam()
    make_stats(GM)
    make_densify(GM)
    make_clone_split(GM)
    make_remove(GM)


if __name__ == "__main__":
    main()
