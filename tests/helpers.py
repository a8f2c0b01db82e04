"""Shared helpers for the parity tests (GPU side goes through the C ABI via cugraph_amd)."""
import ctypes
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_csv(name):
    a = np.loadtxt(os.path.join(GOLDEN_DIR, name))
    return a[:, 0].astype(np.int32), a[:, 1].astype(np.int32), a[:, 2].astype(np.float32)


def case_arrays(case):
    if "file" in case:
        return load_csv(case["file"])
    return (np.array(case["src"], np.int32), np.array(case["dst"], np.int32),
            np.array(case["wgt"], np.float32))


def to_dev(a):
    import torch

    if a is None:
        return None
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_sg_graph(src, dst, wgt=None, *, store_transposed=True, renumber=False, vertices=None,
                  multigraph=False, handle=None):
    from cugraph_amd import pylibcugraph as plc

    h = handle or plc.ResourceHandle()
    props = plc.GraphProperties(is_symmetric=False, is_multigraph=multigraph)
    g = plc.SGGraph(h, props, to_dev(src), to_dev(dst), weight_array=to_dev(wgt),
                    store_transposed=store_transposed, renumber=renumber, do_expensive_check=False,
                    vertices_array=to_dev(vertices))
    return g, h


def view_to_numpy(handle, view_ptr):
    """Copy a type-erased device view to host through the C ABI and free the view."""
    from cugraph_amd import _capi
    from cugraph_amd.pylibcugraph.utils import get_numpy_type_from_c_type

    lib = _capi.load()
    if not view_ptr:
        return None
    n = lib.cugraph_type_erased_device_array_view_size(view_ptr)
    t = lib.cugraph_type_erased_device_array_view_type(view_ptr)
    out = np.empty(n, dtype=get_numpy_type_from_c_type(t))
    err = ctypes.c_void_p()
    code = lib.cugraph_type_erased_device_array_view_copy_to_host(
        handle.c_resource_handle_ptr, out.ctypes.data_as(ctypes.c_void_p), view_ptr,
        ctypes.byref(err))
    _capi.check(code, err, "copy_to_host")
    lib.cugraph_type_erased_device_array_view_free(view_ptr)
    return out


def compressed_arrays(graph, handle):
    """(offsets, indices, weights|None, number_map|None) of the graph as numpy arrays."""
    from cugraph_amd import _capi

    lib = _capi.load()
    off, idx, w, nm = (ctypes.c_void_p() for _ in range(4))
    err = ctypes.c_void_p()
    code = lib.cugraph_amd_graph_get_compressed(graph.c_graph_ptr, ctypes.byref(off),
                                                ctypes.byref(idx), ctypes.byref(w),
                                                ctypes.byref(nm), ctypes.byref(err))
    _capi.check(code, err, "cugraph_amd_graph_get_compressed")
    return tuple(view_to_numpy(handle, v) for v in (off, idx, w, nm))


def compressed_tensors(graph):
    """(offsets, indices, weights|None, number_map|None) as torch tensors ALIASING the graph's
    device arrays (no copy; for the full-size property tests)."""
    from cugraph_amd import _capi
    from cugraph_amd.pylibcugraph.utils import get_numpy_type_from_c_type
    from cugraph_amd.session import _alias_device_memory

    lib = _capi.load()
    views = [ctypes.c_void_p() for _ in range(4)]
    err = ctypes.c_void_p()
    code = lib.cugraph_amd_graph_get_compressed(graph.c_graph_ptr, *[ctypes.byref(v) for v in views],
                                                ctypes.byref(err))
    _capi.check(code, err, "cugraph_amd_graph_get_compressed")
    out = []
    for v in views:
        if not v:
            out.append(None)
            continue
        n = lib.cugraph_type_erased_device_array_view_size(v)
        t = lib.cugraph_type_erased_device_array_view_type(v)
        p = lib.cugraph_type_erased_device_array_view_pointer(v)
        out.append(_alias_device_memory(p, n, get_numpy_type_from_c_type(t)))
        lib.cugraph_type_erased_device_array_view_free(v)
    return tuple(out)
