"""Step-wise PageRank session (extension; include/cugraph_amd_c.h part 2).

`pylibcugraph.pagerank` is a loop over this inside the library; the multi-GPU driver and bench.py
drive it from Python because they interleave collectives / timers with the iterations.
"""
import ctypes

import numpy as np

from . import _capi
from .pylibcugraph.utils import (copy_to_device_array, create_view_from_py_obj, device_pointer,
                                 free_view, get_numpy_type_from_c_type)


class PagerankSession:
    def __init__(self, resource_handle, graph, *, alpha=0.85, epsilon=1e-6,
                 precomputed_vertex_out_weight_vertices=None,
                 precomputed_vertex_out_weight_sums=None, initial_guess_vertices=None,
                 initial_guess_values=None, personalization_vertices=None,
                 personalization_values=None, do_expensive_check=False):
        self._lib = _capi.load()
        self._handle = resource_handle
        self._graph = graph
        views = [create_view_from_py_obj(a) for a in (
            precomputed_vertex_out_weight_vertices, precomputed_vertex_out_weight_sums,
            initial_guess_vertices, initial_guess_values, personalization_vertices,
            personalization_values)]
        ptr = ctypes.c_void_p()
        err = ctypes.c_void_p()
        try:
            code = self._lib.cugraph_amd_pagerank_session_create(
                resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, *views, float(alpha),
                float(epsilon), int(bool(do_expensive_check)), ctypes.byref(ptr), ctypes.byref(err))
            _capi.check(code, err, "cugraph_amd_pagerank_session_create")
        finally:
            for v in views:
                free_view(v)
        self._ptr = ptr

    def iterate(self, n=1):
        err = ctypes.c_void_p()
        code = self._lib.cugraph_amd_pagerank_session_iterate(self._ptr, int(n), ctypes.byref(err))
        _capi.check(code, err, "cugraph_amd_pagerank_session_iterate")

    def iterate_phase(self, phase):
        """One iteration in two steps on a graph whose gathered vector has two regions: phase 0
        needs region 0 only, phase 1 the whole vector (cugraph_amd_pagerank_session_iterate_phase)."""
        err = ctypes.c_void_p()
        code = self._lib.cugraph_amd_pagerank_session_iterate_phase(self._ptr, int(phase), ctypes.byref(err))
        _capi.check(code, err, "cugraph_amd_pagerank_session_iterate_phase")

    def timed_iterate(self, n):
        """Returns (iteration_ms[n], spmv_kernel_ms[n]) measured with HIP events; synchronises."""
        it = (ctypes.c_float * n)()
        km = (ctypes.c_float * n)()
        err = ctypes.c_void_p()
        code = self._lib.cugraph_amd_pagerank_session_timed_iterate(self._ptr, int(n), it, km,
                                                                    ctypes.byref(err))
        _capi.check(code, err, "cugraph_amd_pagerank_session_timed_iterate")
        return np.array(it[:], dtype=np.float64), np.array(km[:], dtype=np.float64)

    def status(self):
        """(iterations, stopped, l1_delta, dangling_mass); synchronises the stream."""
        it = ctypes.c_size_t()
        st = ctypes.c_int()
        delta = ctypes.c_double()
        dang = ctypes.c_double()
        err = ctypes.c_void_p()
        code = self._lib.cugraph_amd_pagerank_session_status(
            self._ptr, ctypes.byref(it), ctypes.byref(st), ctypes.byref(delta), ctypes.byref(dang),
            ctypes.byref(err))
        _capi.check(code, err, "cugraph_amd_pagerank_session_status")
        return int(it.value), bool(st.value), float(delta.value), float(dang.value)

    def info(self):
        arr = (ctypes.c_int64 * 8)()
        err = ctypes.c_void_p()
        code = self._lib.cugraph_amd_pagerank_session_info(self._ptr, arr, ctypes.byref(err))
        _capi.check(code, err, "cugraph_amd_pagerank_session_info")
        keys = ["bytes_per_iteration", "edges", "low_degree_vertices", "mid_degree_vertices",
                "hub_vertices", "hub_work_items", "spmv_workgroups", "dominant_kernel_bytes"]
        return dict(zip(keys, [int(x) for x in arr]))

    def exchange_buffers(self):
        """(x_full, scalars) as torch tensors ALIASING the session's device buffers: the vector
        the next iteration gathers from (this rank's slice freshly written) and the two local
        scalars {dangling mass, L1 delta}.  Valid until the next iterate()."""
        import torch

        xv, sv = ctypes.c_void_p(), ctypes.c_void_p()
        err = ctypes.c_void_p()
        code = self._lib.cugraph_amd_pagerank_session_exchange_buffers(
            self._ptr, ctypes.byref(xv), ctypes.byref(sv), ctypes.byref(err))
        _capi.check(code, err, "cugraph_amd_pagerank_session_exchange_buffers")
        out = []
        cache = self.__dict__.setdefault("_aliases", {})  # the two x buffers and the scalars never move
        for v in (xv, sv):
            p = self._lib.cugraph_type_erased_device_array_view_pointer(v)
            t = cache.get(p)
            if t is None:
                n = self._lib.cugraph_type_erased_device_array_view_size(v)
                ty = self._lib.cugraph_type_erased_device_array_view_type(v)
                t = cache[p] = _alias_device_memory(p, n, get_numpy_type_from_c_type(ty))
            out.append(t)
            self._lib.cugraph_type_erased_device_array_view_free(v)
        return tuple(out)

    def result(self, max_iterations):
        """(vertices, pageranks, iterations, converged) as torch tensors / ints."""
        res = ctypes.c_void_p()
        err = ctypes.c_void_p()
        code = self._lib.cugraph_amd_pagerank_session_result(self._ptr, int(max_iterations),
                                                             ctypes.byref(res), ctypes.byref(err))
        _capi.check(code, err, "cugraph_amd_pagerank_session_result")
        try:
            lib = self._lib
            verts = copy_to_device_array(self._handle.c_resource_handle_ptr,
                                         lib.cugraph_centrality_result_get_vertices(res))
            vals = copy_to_device_array(self._handle.c_resource_handle_ptr,
                                        lib.cugraph_centrality_result_get_values(res))
            iters = int(lib.cugraph_centrality_result_get_num_iterations(res))
            conv = bool(lib.cugraph_centrality_result_converged(res))
        finally:
            self._lib.cugraph_centrality_result_free(res)
        return verts, vals, iters, conv

    def close(self):
        if getattr(self, "_ptr", None):
            self._lib.cugraph_amd_pagerank_session_free(self._ptr)
            self._ptr = None

    __del__ = close


class _DeviceAlias:
    """Minimal __cuda_array_interface__ carrier so torch can wrap library-owned device memory."""

    def __init__(self, ptr, n, np_dtype):
        self.__cuda_array_interface__ = {
            "shape": (int(n),),
            "typestr": np.dtype(np_dtype).str,
            "data": (int(ptr), False),
            "version": 2,
            "strides": None,
        }


def _alias_device_memory(ptr, n, np_dtype):
    import torch

    if n == 0 or not ptr:
        return torch.empty(0, dtype=getattr(torch, np.dtype(np_dtype).name), device="cuda")
    return torch.as_tensor(_DeviceAlias(ptr, n, np_dtype), device="cuda")
