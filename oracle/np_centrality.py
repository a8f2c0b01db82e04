"""TEST INFRASTRUCTURE ONLY -- CPU restatements (numpy, fp64) of the reference's sibling power
iterations; only tests/ may import this.  Each function follows the device code line by line:

  katz_centrality         cpp/src/centrality/katz_centrality_impl.cuh:95-196 (C API: no initial
                          guess, always normalised, cpp/src/c_api/katz.cpp:120-133)
  eigenvector_centrality  cpp/src/centrality/eigenvector_centrality_impl.cuh:76-165
  hits                    cpp/src/link_analysis/hits_impl.cuh:57-206

Pinned by the reference's own golden vectors (cpp/tests/c_api/{katz,eigenvector_centrality,hits}_test.c,
copied as data into tests/golden/reference_vectors.json) and against NetworkX
(tests/test_oracle.py)."""
import numpy as np


def _pull(num_vertices, src, dst, x, weights=None):
    """y[v] = sum over edges (u -> v) of x[u] * w"""
    contrib = x[src] if weights is None else x[src] * weights
    return np.bincount(dst, weights=contrib, minlength=num_vertices)


def katz_centrality(num_vertices, src, dst, weights=None, *, alpha=0.1, beta=1.0, betas=None,
                    epsilon=1e-6, max_iterations=100, normalize=True):
    """returns (values, iterations); raises RuntimeError when it does not converge"""
    n = int(num_vertices)
    x = np.zeros(n)
    w = None if weights is None else np.asarray(weights, dtype=np.float64)
    it = 0
    while True:
        old = x
        x = alpha * _pull(n, src, dst, old, w) + (beta if betas is None else np.asarray(betas, np.float64))
        diff = np.abs(x - old).sum()
        it += 1
        if diff < epsilon:
            break
        if it >= max_iterations:
            raise RuntimeError("Katz Centrality failed to converge.")
    if normalize:
        x = x / np.sqrt((x * x).sum())
    return x, it


def eigenvector_centrality(num_vertices, src, dst, weights=None, *, epsilon=1e-6, max_iterations=100):
    n = int(num_vertices)
    x = np.full(n, 1.0 / n)
    w = None if weights is None else np.asarray(weights, dtype=np.float64)
    it = 0
    while True:
        old = x
        x = _pull(n, src, dst, old, w) + old
        x = x / np.sqrt((x * x).sum())
        diff = np.abs(x - old).sum()
        it += 1
        if diff < n * epsilon:
            break
        if it >= max_iterations:
            raise RuntimeError("Eigenvector Centrality failed to converge.")
    return x, it


def hits(num_vertices, src, dst, *, epsilon=1e-5, max_iterations=100, initial_hubs=None,
         normalize=True):
    """returns (hubs, authorities, iterations, hub_score_difference); weights are ignored, as in
    the reference"""
    n = int(num_vertices)
    if initial_hubs is None:
        hubs = np.full(n, 1.0 / n)
    else:
        hubs = np.asarray(initial_hubs, dtype=np.float64)
        hubs = hubs / hubs.sum()
    it = 0
    while True:
        auth = np.bincount(dst, weights=hubs[src], minlength=n)
        new_hubs = np.bincount(src, weights=auth[dst], minlength=n)
        new_hubs = new_hubs / new_hubs.max()
        auth = auth / auth.max()
        diff = np.abs(new_hubs - hubs).sum()
        hubs = new_hubs
        it += 1
        if diff < n * epsilon:
            break
        if it >= max_iterations:
            raise RuntimeError("HITS failed to converge.")
    if normalize:
        hubs = hubs / hubs.sum()
        auth = auth / auth.sum()
    return hubs, auth, it, diff
