"""pylibcugraph.katz_centrality / eigenvector_centrality / hits (reference:
python/pylibcugraph/pylibcugraph/katz_centrality.pyx:55-153, eigenvector_centrality.pyx:57-136,
hits.pyx:58-193): same arguments and return tuples; failure to converge surfaces as the
RuntimeError the reference's error mapping produces (utils.pyx:37-83)."""
import ctypes

from .. import _capi
from .utils import copy_to_device_array, create_view_from_py_obj, free_view


def _centrality(api_name, call, resource_handle, views):
    lib = _capi.load()
    result = ctypes.c_void_p()
    err = ctypes.c_void_p()
    try:
        code = call(ctypes.byref(result), ctypes.byref(err))
        _capi.check(code, err, api_name)
        h = resource_handle.c_resource_handle_ptr
        vertices = copy_to_device_array(h, lib.cugraph_centrality_result_get_vertices(result))
        values = copy_to_device_array(h, lib.cugraph_centrality_result_get_values(result))
    finally:
        if result:
            lib.cugraph_centrality_result_free(result)
        for v in views:
            free_view(v)
    return (vertices, values)


def katz_centrality(resource_handle, graph, betas, alpha, beta, epsilon, max_iterations,
                    do_expensive_check):
    """(vertices, values); betas: optional device array, one value per vertex id, else `beta`."""
    lib = _capi.load()
    views = [create_view_from_py_obj(betas)]

    def call(result_ref, err_ref):
        return lib.cugraph_katz_centrality(
            resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, views[0], float(alpha),
            float(beta), float(epsilon), int(max_iterations), int(bool(do_expensive_check)),
            result_ref, err_ref)

    return _centrality("cugraph_katz_centrality", call, resource_handle, views)


def eigenvector_centrality(resource_handle, graph, epsilon, max_iterations, do_expensive_check):
    """(vertices, values)"""
    lib = _capi.load()

    def call(result_ref, err_ref):
        return lib.cugraph_eigenvector_centrality(
            resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, float(epsilon),
            int(max_iterations), int(bool(do_expensive_check)), result_ref, err_ref)

    return _centrality("cugraph_eigenvector_centrality", call, resource_handle, [])


def hits(resource_handle, graph, tol, max_iter, initial_hubs_guess_vertices,
         initial_hubs_guess_values, normalized, do_expensive_check):
    """(vertices, hubs, authorities)"""
    lib = _capi.load()
    views = [create_view_from_py_obj(a) for a in (initial_hubs_guess_vertices,
                                                  initial_hubs_guess_values)]
    result = ctypes.c_void_p()
    err = ctypes.c_void_p()
    try:
        code = lib.cugraph_hits(resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, float(tol),
                                int(max_iter), views[0], views[1], int(bool(normalized)),
                                int(bool(do_expensive_check)), ctypes.byref(result),
                                ctypes.byref(err))
        _capi.check(code, err, "cugraph_hits")
        h = resource_handle.c_resource_handle_ptr
        vertices = copy_to_device_array(h, lib.cugraph_hits_result_get_vertices(result))
        hubs = copy_to_device_array(h, lib.cugraph_hits_result_get_hubs(result))
        authorities = copy_to_device_array(h, lib.cugraph_hits_result_get_authorities(result))
    finally:
        if result:
            lib.cugraph_hits_result_free(result)
        for v in views:
            free_view(v)
    return (vertices, hubs, authorities)
