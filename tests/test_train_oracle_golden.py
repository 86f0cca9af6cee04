"""CPU: the numpy restatement of the training-state updates (oracle/train_oracle.py) against the
fixtures generated from torch.optim.Adam as Trainer.py:24 builds it and from the reference's
GaussianModel methods (tests/golden/make_train_golden.py)."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import train_oracle as to  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
RTOL = 1e-4   # floats; sizes, capacities, counts and integer statistics are exact


def fixtures(prefix):
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, prefix + "_*.npz")))


def load_model(d, prefix):
    m = {k: d[prefix + k] for k in to.PARAM_KEYS + ("coords_grads", "denom")}
    m["size"] = int(d[prefix + "size"])
    return m


def assert_model(got, d, prefix, exact_rows=False):
    assert int(got["size"]) == int(d[prefix + "size"])
    for k in to.PARAM_KEYS + ("coords_grads", "denom"):
        want = d[prefix + k]
        assert got[k].shape == want.shape, k
        if k == "denom":
            np.testing.assert_array_equal(got[k], want)
        else:
            np.testing.assert_allclose(got[k], want, rtol=RTOL, atol=1e-7, err_msg=k)


def test_fixture_coverage():
    assert len(fixtures("adam")) >= 2 and len(fixtures("densify")) >= 6 and len(fixtures("remove")) >= 3 and fixtures("stats")
    assert len(fixtures("clone")) >= 3 and len(fixtures("split")) >= 3
    grow = [np.load(os.path.join(GOLDEN, n + ".npz")) for n in fixtures("densify")]
    assert any(d["out_coords"].shape[0] == 4 * d["in_coords"].shape[0] for d in grow)      # capacity doubled twice
    assert any(int(d["num_clone"]) and int(d["num_split"]) for d in grow)
    assert any(int(d["num_clone"]) == 0 and int(d["num_split"]) == 0 for d in grow)


@pytest.mark.parametrize("name", fixtures("adam"))
def test_adam_oracle_matches_torch(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    lr, b1, b2 = float(d["lr"]), float(d["beta1"]), float(d["beta2"])
    for i in range(5):
        p = d[f"p{i}_init"]
        m, v, step = np.zeros_like(p), np.zeros_like(p), 0
        for s in range(int(d["steps"])):
            g = d[f"g{i}_s{s}"]
            if g.size:
                step += 1
                p, m, v = to.adam_step(p, g, m, v, step, lr, b1, b2)
            np.testing.assert_allclose(p, d[f"p{i}_s{s}"], rtol=1e-5, atol=1e-6)
        assert step == int(d[f"step{i}_final"])
        np.testing.assert_allclose(m, d[f"m{i}_final"], rtol=RTOL, atol=1e-6 * np.abs(d[f"m{i}_final"]).max())
        np.testing.assert_allclose(v, d[f"v{i}_final"], rtol=RTOL, atol=1e-6 * np.abs(d[f"v{i}_final"]).max())


@pytest.mark.parametrize("name", fixtures("stats"))
def test_stats_oracle_matches_reference(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = load_model(d, "in_")
    for it in range(int(d["iters"])):
        m = to.densify_stats(m, d[f"g_screen_{it}"], d[f"mask_{it}"])
        assert_model(m, d, f"out{it}_")


@pytest.mark.parametrize("name", fixtures("densify"))
def test_densify_oracle_matches_reference(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    m, n_clone, n_split = to.densify(load_model(d, "in_"), float(d["scene_radius"]), d["noise"])
    assert (n_clone, n_split) == (int(d["num_clone"]), int(d["num_split"]))
    assert_model(m, d, "out_")


@pytest.mark.parametrize("name", fixtures("clone") + fixtures("split"))
def test_clone_and_split_oracle_match_reference(name):
    """GaussianModel.clone / GaussianModel.split called on their own with an arbitrary mask."""
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    if name.startswith("clone"):
        m, n = to.clone(load_model(d, "in_"), d["mask"])
    else:
        m, n = to.split(load_model(d, "in_"), d["mask"], d["noise"])
    assert n == int(d["mask"].sum())
    assert_model(m, d, "out_")


@pytest.mark.parametrize("name", fixtures("remove"))
def test_remove_oracle_matches_reference(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = to.remove(load_model(d, "in_"), d["mask"])
    assert_model(m, d, "out_")


def test_remove_of_more_than_half_raises():
    d = np.load(os.path.join(GOLDEN, "remove_few.npz"))
    m = load_model(d, "in_")
    with pytest.raises(RuntimeError):
        to.remove(m, np.ones(m["size"], bool))


def test_knn_oracle_against_exhaustive_search():
    """The k-d tree restatement of simple_knn's distCUDA2 against a plain O(N^2) search, with duplicates."""
    rng = np.random.default_rng(5)
    p = rng.normal(size=(300, 3)).astype(np.float32)
    p[10] = p[11] = p[12] = p[13]                 # four coincident points: three neighbours at distance zero
    p[20] = p[21]
    got = to.knn_dist2(p)
    d2 = ((p[:, None, :].astype(np.float64) - p[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)                   # the point's own index is skipped, nothing else
    want = np.sort(d2, axis=1)[:, :3].mean(axis=1)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-12)
    assert got[10] == 0.0 and got[20] > 0.0
