"""R-MAT edge lists on the device (benchmark inputs).

Mirrors the role of cugraph.generators.rmat / cugraph::generate_rmat_edgelist
(cpp/src/generators/generate_rmat_edgelist.cu:34-113) with this project's own counter-based
random stream, so that any slice [first, first+n) of the edge stream can be produced on any rank
and the CPU oracle emits the identical list.
"""
import ctypes

from . import _capi
from .pylibcugraph.utils import create_view_from_py_obj, free_view


def rmat(resource_handle, scale, num_edges, *, seed=42, a=0.57, b=0.19, c=0.19,
         scramble_vertex_ids=False, first_edge=0):
    """Returns (src, dst) int32 torch tensors with edges [first_edge, first_edge+num_edges)."""
    import torch

    lib = _capi.load()
    src = torch.empty(num_edges, dtype=torch.int32, device="cuda")
    dst = torch.empty(num_edges, dtype=torch.int32, device="cuda")
    sv, dv = create_view_from_py_obj(src), create_view_from_py_obj(dst)
    err = ctypes.c_void_p()
    try:
        code = lib.cugraph_amd_generate_rmat_edgelist(
            resource_handle.c_resource_handle_ptr, int(scale), int(first_edge), int(seed), float(a),
            float(b), float(c), int(bool(scramble_vertex_ids)), sv, dv, ctypes.byref(err))
        _capi.check(code, err, "cugraph_amd_generate_rmat_edgelist")
    finally:
        free_view(sv)
        free_view(dv)
    return src, dst


def rmat_weights(resource_handle, num_edges, *, seed=43, first_edge=0, dtype=None):
    """Edge weights in (0, 1] for edges [first_edge, first_edge+num_edges)."""
    import torch

    lib = _capi.load()
    w = torch.empty(num_edges, dtype=dtype or torch.float64, device="cuda")
    wv = create_view_from_py_obj(w)
    err = ctypes.c_void_p()
    try:
        code = lib.cugraph_amd_generate_rmat_weights(resource_handle.c_resource_handle_ptr,
                                                     int(first_edge), int(seed), wv,
                                                     ctypes.byref(err))
        _capi.check(code, err, "cugraph_amd_generate_rmat_weights")
    finally:
        free_view(wv)
    return w
