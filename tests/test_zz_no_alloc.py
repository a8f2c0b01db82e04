"""cugraph_amd_edgelist_source_weight_sums_noalloc (static device scratch, allocates nothing) against
cugraph_amd_edgelist_source_weight_sums, through the C ABI: two calls in a row, the second with a smaller
input (the scratch array keeps the first call's contents), skipped ids, and the size limit."""
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
