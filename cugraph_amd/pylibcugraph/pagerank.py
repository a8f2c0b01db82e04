"""pylibcugraph.pagerank (reference: python/pylibcugraph/pylibcugraph/pagerank.pyx:54-258)."""
import ctypes

from .. import _capi
from .exceptions import FailedToConvergeError
from .utils import copy_to_device_array, create_view_from_py_obj, free_view


def _run(api_name, call, resource_handle, fail_on_nonconvergence, views):
    lib = _capi.load()
    result = ctypes.c_void_p()
    err = ctypes.c_void_p()
    try:
        code = call(ctypes.byref(result), ctypes.byref(err))
        _capi.check(code, err, api_name)
        converged = bool(lib.cugraph_centrality_result_converged(result))
        vertices = pageranks = None
        # Only extract results if necessary (pagerank.pyx:232-239)
        if (fail_on_nonconvergence is False) or converged:
            vertices = copy_to_device_array(resource_handle.c_resource_handle_ptr,
                                            lib.cugraph_centrality_result_get_vertices(result))
            pageranks = copy_to_device_array(resource_handle.c_resource_handle_ptr,
                                             lib.cugraph_centrality_result_get_values(result))
    finally:
        if result:
            lib.cugraph_centrality_result_free(result)
        for v in views:
            free_view(v)
    if fail_on_nonconvergence is False:
        return (vertices, pageranks, converged)
    if converged:
        return (vertices, pageranks)
    raise FailedToConvergeError


def pagerank(resource_handle,
             graph,
             precomputed_vertex_out_weight_vertices,
             precomputed_vertex_out_weight_sums,
             initial_guess_vertices,
             initial_guess_values,
             alpha,
             epsilon,
             max_iterations,
             do_expensive_check,
             fail_on_nonconvergence=True):
    """Same contract as pylibcugraph.pagerank: returns (vertices, pageranks) device arrays, or
    (vertices, pageranks, converged) when fail_on_nonconvergence is False; raises
    FailedToConvergeError otherwise.  ``epsilon`` is compared with the L1 change of the rank
    vector as is (cpp/src/link_analysis/pagerank_impl.cuh:330)."""
    lib = _capi.load()
    views = [create_view_from_py_obj(a) for a in (precomputed_vertex_out_weight_vertices,
                                                  precomputed_vertex_out_weight_sums,
                                                  initial_guess_vertices, initial_guess_values)]

    def call(result_ref, err_ref):
        return lib.cugraph_pagerank_allow_nonconvergence(
            resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, views[0], views[1], views[2],
            views[3], float(alpha), float(epsilon), int(max_iterations),
            int(bool(do_expensive_check)), result_ref, err_ref)

    return _run("cugraph_pagerank_allow_nonconvergence", call, resource_handle,
                fail_on_nonconvergence, views)
