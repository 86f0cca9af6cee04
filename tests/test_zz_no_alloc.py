"""GPU test of the allocation-free scan entry point: gsr_exclusive_scan_static (block sums in a device
array inside the library) against gsr_exclusive_scan (block sums in a caller workspace), bit for bit, on
two calls in a row - the second on a smaller input, so that what the first call left in the shared array
must not leak into it."""
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _alloc_events(dev):
    stats = torch.cuda.memory_stats(dev)
    return (stats.get("allocation.all.allocated", 0), stats.get("segment.all.allocated", 0))


def _scan_both(L, lib, x):
    dev = torch.device("cuda:0")
    n = x.size
    stream = torch.cuda.current_stream().cuda_stream
    tin = torch.from_numpy(x).to(dev)
    out_ws = torch.full((max(n, 1),), -1, dtype=torch.int32, device=dev)
    out_st = torch.full((max(n, 1),), -2, dtype=torch.int32, device=dev)
    tot_ws = torch.full((1,), -1, dtype=torch.int32, device=dev)
    tot_st = torch.full((1,), -2, dtype=torch.int32, device=dev)
    wsb = int(lib.gsr_scan_workspace_bytes(n))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    L.check(lib.gsr_exclusive_scan(tin.data_ptr(), out_ws.data_ptr(), n, tot_ws.data_ptr(), ws.data_ptr(), wsb, stream), "scan")
    torch.cuda.synchronize()
    # What must not happen inside the window is an ALLOCATION: torch's allocator counts its allocation and segment
    # events (monotonic, so tensors of earlier tests that Python's collector frees meanwhile do not show), and the
    # device's free memory must not shrink (a raw hipMalloc would take some; it may grow: child processes of earlier
    # tests can still be giving their memory back).  The collector is run first and held off for the window.
    gc.collect()
    torch.cuda.synchronize()
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        before = _alloc_events(dev)
        free_before = torch.cuda.mem_get_info(dev)[0]
        L.check(lib.gsr_exclusive_scan_static(tin.data_ptr(), out_st.data_ptr(), n, tot_st.data_ptr(), stream), "scan_static")
        torch.cuda.synchronize()
        after = _alloc_events(dev)
        free_after = torch.cuda.mem_get_info(dev)[0]
    finally:
        if gc_was_on:
            gc.enable()
    assert after == before, f"torch allocated inside gsr_exclusive_scan_static: {before} -> {after}"
    assert free_after >= free_before
    return out_ws.cpu().numpy()[:n], int(tot_ws.cpu()[0]), out_st.cpu().numpy()[:n], int(tot_st.cpu()[0])


def test_static_scan_matches_the_workspace_scan_on_two_calls_in_a_row():
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(5)
    for n in (3_000_001, 70_003):          # 1465 block sums, then 35: the second call reuses part of the array
        x = rng.integers(0, 1000, size=n, dtype=np.int64).astype(np.int32)
        a, ta, b, tb = _scan_both(L, lib, x)
        ref = np.concatenate([[0], np.cumsum(x.astype(np.int64))])
        assert np.array_equal(a.astype(np.int64), ref[:-1]) and ta == int(ref[-1])
        assert np.array_equal(a, b) and ta == tb


def test_static_scan_refuses_more_than_its_scratch_holds():
    import gsplat_amd  # noqa: F401
    from gsplat_amd import _lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    x = torch.zeros(8, dtype=torch.int32, device=dev)
    rc = lib.gsr_exclusive_scan_static(x.data_ptr(), x.data_ptr(), (1 << 24) + 1, None, torch.cuda.current_stream().cuda_stream)
    assert rc == -1
    with pytest.raises(RuntimeError, match="exceed the built-in scratch"):
        L.check(rc, "scan_static")
