"""GPU tests of the scan and radix-sort entry points (bit-exact against numpy)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib
    return _lib, _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("n", [0, 1, 63, 2048, 2049, 100_000, 3_000_001])
def test_exclusive_scan(n):
    L, lib = _lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n)
    x = rng.integers(0, 1000, size=n, dtype=np.int64).astype(np.int32)
    tin = torch.from_numpy(x).to(dev)
    tout = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    total = torch.full((1,), -1, dtype=torch.int32, device=dev)
    wsb = int(lib.gsr_scan_workspace_bytes(n))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    L.check(lib.gsr_exclusive_scan(tin.data_ptr() if n else None, tout.data_ptr(), n, total.data_ptr(), ws.data_ptr(),
                                   wsb, _stream()), "scan")
    ref = np.concatenate([[0], np.cumsum(x.astype(np.int64))])
    assert int(total.cpu()[0]) == int(ref[-1])
    if n:
        assert np.array_equal(tout.cpu().numpy()[:n].astype(np.int64), ref[:-1])


@pytest.mark.parametrize("n,bits,distinct", [(0, 32, 10), (1, 32, 10), (4095, 32, 1 << 30), (4096, 13, 8160),
                                             (4097, 32, 50), (1_000_003, 32, 1 << 31), (2_500_000, 15, 32400),
                                             (300_000, 0, 5), (70_000, 9, 400), (3_200_001, 32, 1 << 31)])
def test_sort_pairs_is_a_stable_radix_sort(n, bits, distinct):
    L, lib = _lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, distinct, size=n, dtype=np.int64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    tk = torch.from_numpy(keys.view(np.int32).copy()).to(dev)
    tv = torch.from_numpy(vals.view(np.int32).copy()).to(dev)
    ko = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    vo = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    wsb = int(lib.gsr_sort_workspace_bytes(n))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    L.check(lib.gsr_sort_pairs(tk.data_ptr(), tv.data_ptr(), ko.data_ptr(), vo.data_ptr(), n, 0, bits, ws.data_ptr(),
                               wsb, _stream()), "sort")
    masked = keys & np.uint32((1 << bits) - 1 if bits < 32 else 0xFFFFFFFF)
    order = np.argsort(masked, kind="stable")
    assert np.array_equal(vo.cpu().numpy()[:n].view(np.uint32), vals[order])
    assert np.array_equal(ko.cpu().numpy()[:n].view(np.uint32), keys[order])


@pytest.mark.parametrize("w,h,tile,kind", [(16, 16, 16, "rand"), (1920, 1080, 16, "rand"), (3840, 2160, 16, "rand"),
                                           (300, 200, 64, "equal"), (640, 480, 16, "empty"), (2000, 1000, 16, "spike")])
def test_tile_order_is_a_permutation_by_descending_length(w, h, tile, kind):
    L, lib = _lib()
    dev = torch.device("cuda:0")
    T = ((w + tile - 1) // tile) * ((h + tile - 1) // tile)
    rng = np.random.default_rng(T)
    lens = {"rand": lambda: rng.integers(0, 2000, size=T), "equal": lambda: np.full(T, 77), "empty": lambda: np.zeros(T, np.int64),
            "spike": lambda: np.where(rng.random(T) < 0.01, 100000, rng.integers(0, 50, size=T))}[kind]().astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    toff = torch.from_numpy(off.view(np.int32).copy()).to(dev)
    order = torch.full((T,), -1, dtype=torch.int32, device=dev)
    ident = np.eye(4, dtype=np.float32).reshape(16)
    view = L.make_view(0, 0, w, h, tile, ident, ident, (0.0, 0.0, 0.0))
    L.check(lib.gsr_tile_order(view, toff.data_ptr(), order.data_ptr(), _stream()), "tile_order")
    o = order.cpu().numpy().astype(np.int64)
    assert np.array_equal(np.sort(o), np.arange(T)), "not a permutation of the tile ids"
    got = lens[o]
    # longest first up to the width of a length class (1/1023 of the longest list, rounded up)
    slack = int(lens.max()) // 1023 + 1
    assert (got[:-1] + slack >= got[1:]).all()
    assert got[0] + slack >= lens.max()


@pytest.mark.parametrize("n,per,expect", [(70_000, 65_536, True),      # total 4.6e9: over 32 bits across blocks
                                          (3_000, 2_000_000, True),    # 6e9 inside one block of the scan
                                          (65_535, 65_536, False),     # 4 294 901 760: just fits
                                          (1_000_000, 5, False)])
def test_bin_offsets_flags_a_pair_count_beyond_32_bits(n, per, expect):
    """gsr_bin_offsets sums the per-Gaussian tile counts; a total that does not fit 32 bits raises bit 1 of
    the error word (the host then refuses to build lists), a total that fits leaves it alone."""
    L, lib = _lib()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(n)
    depth = torch.randint(0, 2**31 - 1, (n,), generator=gen, dtype=torch.int32).to(dev)
    ntiles = torch.full((n,), per, dtype=torch.int32, device=dev)
    order = torch.empty(n, dtype=torch.int32, device=dev)
    slot_base = torch.empty(n, dtype=torch.int32, device=dev)
    slot_seq = torch.empty(n + 1, dtype=torch.int32, device=dev)
    counters = torch.zeros(8, dtype=torch.int32, device=dev)
    wsb = int(lib.gsr_bin_offsets_workspace_bytes(n))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    L.check(lib.gsr_bin_offsets(depth.data_ptr(), ntiles.data_ptr(), n, order.data_ptr(), slot_base.data_ptr(),
                                slot_seq.data_ptr(), counters.data_ptr(), ws.data_ptr(), wsb, _stream()), "bin_offsets")
    c = counters.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert bool(c[2] & 2) == expect
    assert c[2] & ~2 == 0
    if not expect:
        assert c[1] == n * per
