"""pylibcugraph.personalized_pagerank
(reference: python/pylibcugraph/pylibcugraph/personalized_pagerank.pyx:54-285)."""
from .. import _capi
from .pagerank import _run
from .utils import create_view_from_py_obj


def personalized_pagerank(resource_handle,
                          graph,
                          precomputed_vertex_out_weight_vertices,
                          precomputed_vertex_out_weight_sums,
                          initial_guess_vertices,
                          initial_guess_values,
                          personalization_vertices,
                          personalization_values,
                          alpha,
                          epsilon,
                          max_iterations,
                          do_expensive_check,
                          fail_on_nonconvergence=True):
    """Same contract as pylibcugraph.personalized_pagerank; personalization values need not sum
    to one (they are divided by their sum, pagerank_impl.cuh:316-317)."""
    lib = _capi.load()
    views = [create_view_from_py_obj(a) for a in (precomputed_vertex_out_weight_vertices,
                                                  precomputed_vertex_out_weight_sums,
                                                  initial_guess_vertices, initial_guess_values,
                                                  personalization_vertices,
                                                  personalization_values)]

    def call(result_ref, err_ref):
        return lib.cugraph_personalized_pagerank_allow_nonconvergence(
            resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, views[0], views[1], views[2],
            views[3], views[4], views[5], float(alpha), float(epsilon), int(max_iterations),
            int(bool(do_expensive_check)), result_ref, err_ref)

    return _run("cugraph_personalized_pagerank_allow_nonconvergence", call, resource_handle,
                fail_on_nonconvergence, views)
