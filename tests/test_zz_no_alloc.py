"""cugraph_amd_edgelist_source_weight_sums_noalloc (static device scratch, allocates nothing) against
cugraph_amd_edgelist_source_weight_sums, through the C ABI: two calls in a row, the second with a smaller
input (the scratch array keeps the first call's contents), skipped ids, the size limit, and a call
captured into a HIP graph and replayed (nothing in it may allocate, copy to the host or synchronise)."""
import ctypes

import numpy as np
import pytest
import torch

from cugraph_amd import _capi, pylibcugraph as plc
from cugraph_amd.pylibcugraph.utils import create_view_from_py_obj, free_view

pytestmark = pytest.mark.gpu


def _call(name, handle, *tensors):
    lib = _capi.load()
    views = [create_view_from_py_obj(t) for t in tensors]
    err = ctypes.c_void_p()
    try:
        code = getattr(lib, name)(handle.c_resource_handle_ptr, *views, ctypes.byref(err))
        _capi.check(code, err, name)
    finally:
        for v in views:
            free_view(v)


def _both(handle, src, wgt, nv):
    want = torch.full((nv,), -1.0, dtype=torch.float64, device="cuda")
    got = torch.full((nv,), -2.0, dtype=torch.float64, device="cuda")
    skipped = torch.full((1,), -7, dtype=torch.int64, device="cuda")
    _call("cugraph_amd_edgelist_source_weight_sums", handle, src, wgt, want)
    _call("cugraph_amd_edgelist_source_weight_sums_noalloc", handle, src, wgt, got, skipped)
    torch.cuda.synchronize()
    return want.cpu().numpy(), got.cpu().numpy(), int(skipped.item())


@pytest.mark.parametrize("id_dtype", [torch.int32, torch.int64])
def test_no_alloc_source_sums_two_calls_in_a_row(id_dtype):
    h = plc.ResourceHandle()
    rng = np.random.default_rng(11)
    nv = 50_000
    for ne, weighted in ((3_000_000, False), (1_234_567, True), (1000, False), (0, False)):
        src = torch.from_numpy(rng.integers(0, nv, ne)).to(id_dtype).cuda()
        wgt = torch.from_numpy(rng.random(ne)).cuda() if weighted else None
        want, got, skipped = _both(h, src, wgt, nv)
        assert skipped == 0
        if weighted:  # fp64 atomic sums in arrival order: equal to rounding
            assert np.max(np.abs(got - want)) <= 1e-12 * max(1.0, float(np.max(want)))
            ref = np.bincount(src.cpu().numpy(), weights=wgt.cpu().numpy(), minlength=nv)
            assert np.max(np.abs(got - ref)) <= 1e-10
        else:  # counts: exact
            assert np.array_equal(got, want)
            assert np.array_equal(got, np.bincount(src.cpu().numpy(), minlength=nv).astype(np.float64))


def test_no_alloc_source_sums_skips_and_counts_foreign_ids_and_has_a_limit():
    h = plc.ResourceHandle()
    nv = 1000
    src = torch.tensor([5, 7, 7, -3, 1000, 999, 2**31 + 7, 0], dtype=torch.int64, device="cuda")
    got = torch.zeros(nv, dtype=torch.float64, device="cuda")
    skipped = torch.zeros(1, dtype=torch.int64, device="cuda")
    _call("cugraph_amd_edgelist_source_weight_sums_noalloc", h, src, None, got, skipped)
    torch.cuda.synchronize()
    want = np.zeros(nv)
    want[[5, 999, 0]] = 1
    want[7] = 2
    assert np.array_equal(got.cpu().numpy(), want) and int(skipped.item()) == 3
    # without the counter argument
    _call("cugraph_amd_edgelist_source_weight_sums_noalloc", h, src, None, got, None)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    big = torch.zeros((1 << 22) + 1, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError, match="at most 4194304 edges per call"):
        _call("cugraph_amd_edgelist_source_weight_sums_noalloc", h, big, None, got, skipped)


def test_no_alloc_source_sums_under_graph_capture():
    """The property the entry point exists for: nothing in it allocates, copies to the host or
    synchronises.  A call is captured into a HIP graph on the handle's stream -- under capture a
    hipMalloc, a blocking copy or a synchronisation fails the capture (relaxed mode off) -- and
    the graph is replayed twice, on the first input and on a second one written into the same
    buffers, against the allocating entry point (which is not capturable: it allocates its staging
    buffer and reads the id range back; trying it here would leave the stream in a broken capture)."""
    rng = np.random.default_rng(5)
    nv, ne = 20_000, 500_000
    stream = torch.cuda.Stream()
    h = plc.ResourceHandle(stream=stream.cuda_stream)
    src = torch.from_numpy(rng.integers(0, nv, ne)).to(torch.int32).cuda()
    wgt = torch.from_numpy(rng.random(ne)).cuda()
    got = torch.full((nv,), -2.0, dtype=torch.float64, device="cuda")
    skipped = torch.full((1,), -7, dtype=torch.int64, device="cuda")
    # warm-up outside capture (first call on a device: symbol address, module load)
    with torch.cuda.stream(stream):
        _call("cugraph_amd_edgelist_source_weight_sums_noalloc", h, src, wgt, got, skipped)
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream, capture_error_mode="global"):
        _call("cugraph_amd_edgelist_source_weight_sums_noalloc", h, src, wgt, got, skipped)
    for round_ in range(2):
        if round_ == 1:  # new contents, same buffers
            src.copy_(torch.from_numpy(rng.integers(0, nv, ne)).to(torch.int32))
            wgt.copy_(torch.from_numpy(rng.random(ne)))
        got.fill_(-2.0)
        skipped.fill_(-7)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        ref = np.bincount(src.cpu().numpy(), weights=wgt.cpu().numpy(), minlength=nv)
        assert np.max(np.abs(got.cpu().numpy() - ref)) <= 1e-10
        assert int(skipped.item()) == 0
        want = torch.full((nv,), -1.0, dtype=torch.float64, device="cuda")
        h0 = plc.ResourceHandle()
        _call("cugraph_amd_edgelist_source_weight_sums", h0, src, wgt, want)
        torch.cuda.synchronize()
        assert np.max(np.abs(got.cpu().numpy() - want.cpu().numpy())) <= 1e-12 * max(1.0, float(ref.max()))
