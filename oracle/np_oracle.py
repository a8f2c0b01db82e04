"""NumPy / SciPy / NetworkX side of the oracle (TEST INFRASTRUCTURE ONLY).

- csc_lexsort: the reference CSC by definition -- (dst, src)-lexicographic order,
  cpp/src/structure/detail/structure_utils.cuh:198-209,273-279 -- as numpy.lexsort + bincount.
- nx_pagerank: networkx.pagerank, the CPU backend nx-cugraph dispatches from
  (python/nx-cugraph/nx_cugraph/algorithms/link_analysis/pagerank_alg.py:29-96); the oracle
  BASELINE.json's north_star names.  NetworkX 3.4.2 is a third-party package of this image,
  not part of /root/reference.
- scipy_pagerank: restatement of networkx's _pagerank_scipy loop
  (networkx/algorithms/link_analysis/pagerank_alg.py:456-500) on raw edge arrays, for graphs
  too big for an nx.Graph of Python dicts.
"""
import numpy as np


def csc_lexsort(num_vertices, src, dst, weights=None):
    """Returns (offsets[int64], indices[int32], weights or None) in (dst, src) order, ties stable."""
    src = np.asarray(src)
    dst = np.asarray(dst)
    order = np.lexsort((src, dst))  # last key is primary; lexsort is stable
    offsets = np.zeros(num_vertices + 1, dtype=np.int64)
    np.cumsum(np.bincount(dst, minlength=num_vertices), out=offsets[1:])
    w = None if weights is None else np.asarray(weights)[order]
    return offsets, src[order].astype(np.int32), w


def nx_pagerank(num_vertices, src, dst, weights=None, *, alpha=0.85, tol=1e-6, max_iter=100,
                nstart=None, personalization=None):
    """networkx.pagerank on a MultiDiGraph with nodes 0..num_vertices-1 (parallel edges add up,
    as every edge is a separate term in cuGraph's sum).  Returns a float64 vector."""
    import networkx as nx

    G = nx.MultiDiGraph()
    G.add_nodes_from(range(num_vertices))
    if weights is None:
        G.add_edges_from(zip(map(int, src), map(int, dst)))
        weight = None
    else:
        G.add_weighted_edges_from(zip(map(int, src), map(int, dst), map(float, weights)))
        weight = "weight"
    pr = nx.pagerank(G, alpha=alpha, tol=tol, max_iter=max_iter, nstart=nstart,
                     personalization=personalization, weight=weight)
    return np.array([pr[i] for i in range(num_vertices)], dtype=np.float64)


def scipy_pagerank(num_vertices, src, dst, weights=None, *, alpha=0.85, tol=1e-6, max_iter=100,
                   nstart=None, personalization=None, return_info=False, timings=None):
    """_pagerank_scipy's loop on raw arrays.  nstart / personalization are dense vectors or None.

    Raises RuntimeError after max_iter iterations without err < N*tol (NetworkX raises
    PowerIterationFailedConvergence there)."""
    import time

    import scipy.sparse as sp

    t_start = time.perf_counter()
    N = int(num_vertices)
    src = np.asarray(src)
    dst = np.asarray(dst)
    data = np.ones(src.shape[0], dtype=np.float64) if weights is None else np.asarray(weights, dtype=np.float64)
    A = sp.csr_array(sp.coo_array((data, (src, dst)), shape=(N, N)))  # duplicates are summed
    S = np.asarray(A.sum(axis=1)).reshape(-1)
    nz = S != 0
    inv = np.zeros(N, dtype=np.float64)
    inv[nz] = 1.0 / S[nz]
    A = sp.csr_array(sp.diags_array(inv) @ A)
    x = np.repeat(1.0 / N, N) if nstart is None else np.asarray(nstart, dtype=np.float64) / np.sum(nstart)
    if personalization is None:
        p = np.repeat(1.0 / N, N)
    else:
        p = np.asarray(personalization, dtype=np.float64)
        if p.sum() == 0:
            raise ZeroDivisionError
        p = p / p.sum()
    is_dangling = np.where(S == 0)[0]
    AT = sp.csr_array(A.T)  # x @ A == A.T @ x; csr mat-vec is the fast path
    t_loop = time.perf_counter()
    if timings is not None:
        timings["build_s"] = t_loop - t_start
    for it in range(max_iter):
        xlast = x
        x = alpha * (AT @ x + x[is_dangling].sum() * p) + (1 - alpha) * p
        err = np.absolute(x - xlast).sum()
        if timings is not None:
            timings["loop_s"] = time.perf_counter() - t_loop
            timings["iterations"] = it + 1
        if err < N * tol:
            return (x, it + 1) if return_info else x
    raise RuntimeError(f"power iteration failed to converge in {max_iter} iterations")
