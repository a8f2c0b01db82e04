"""ms per iteration of Katz / eigenvector centrality / HITS on the bench graph (R-MAT scale S,
ef 16, scrambled, renumber=True, unweighted fp32), through the C ABI."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cugraph_amd import _capi, generators, pylibcugraph as plc  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
nv, ne = 1 << scale, 16 << scale
h = plc.ResourceHandle()
src, dst = generators.rmat(h, scale, ne, seed=42, scramble_vertex_ids=True)
g = plc.SGGraph(h, plc.GraphProperties(is_multigraph=True), src, dst, store_transposed=True, renumber=True,
                vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
del src, dst
lib = _capi.load()
deg_max = 1 << 21  # safe bound on the largest in-degree for the Katz attenuation factor


def timed(name, call, iters_of, free):
    res, err = ctypes.c_void_p(), ctypes.c_void_p()
    for rep in range(2):  # the first call builds the plan(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        code = call(ctypes.byref(res), ctypes.byref(err))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        _capi.check(code, err, name)
        it = iters_of(res)
        free(res)
        print(f"{name} run {rep}: {it} iterations in {dt * 1e3:.1f} ms -> {dt * 1e3 / it:.2f} ms/iteration, "
              f"{ne * it / dt / 1e9:.0f} GTEPS", flush=True)


hp, gp = h.c_resource_handle_ptr, g.c_graph_ptr
timed("katz", lambda r, e: lib.cugraph_katz_centrality(hp, gp, None, 0.5 / deg_max, 1.0, 1e-7, 1000, 0, r, e),
      lib.cugraph_centrality_result_get_num_iterations, lib.cugraph_centrality_result_free)
timed("eigenvector", lambda r, e: lib.cugraph_eigenvector_centrality(hp, gp, 1e-9, 1000, 0, r, e),
      lib.cugraph_centrality_result_get_num_iterations, lib.cugraph_centrality_result_free)
timed("hits", lambda r, e: lib.cugraph_hits(hp, gp, 1e-9, 1000, None, None, 1, 0, r, e),
      lib.cugraph_hits_result_get_number_of_iterations, lib.cugraph_hits_result_free)
