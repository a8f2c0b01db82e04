"""Katz / eigenvector centrality / HITS on the GPU, through the C ABI:
the golden vectors of cpp/tests/c_api/{katz,eigenvector_centrality,hits}_test.c and R-MAT graphs
against the numpy oracle (oracle/np_centrality.py), on both SpMV paths (degree buckets, XS tiles)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, make_sg_graph, to_dev

pytestmark = pytest.mark.gpu


def _cases(kind):
    with open(os.path.join(GOLDEN_DIR, "reference_vectors.json")) as f:
        return json.load(f)["c_api_centrality"][kind]


def _by_vertex(v, x, n):
    out = np.zeros(n)
    out[v.cpu().numpy()] = x.cpu().numpy()
    return out


@pytest.mark.parametrize("case", _cases("katz"), ids=lambda c: c["name"])
@pytest.mark.parametrize("renumber", [False, True])
def test_katz_goldens(case, renumber):
    from cugraph_amd import pylibcugraph as plc

    g, h = make_sg_graph(np.array(case["src"], np.int32), np.array(case["dst"], np.int32),
                         np.array(case["wgt"], np.float32), store_transposed=case["store_transposed"],
                         renumber=renumber)
    v, x = plc.katz_centrality(h, g, None, case["alpha"], case["beta"], case["epsilon"],
                               case["max_iterations"], False)
    assert np.max(np.abs(_by_vertex(v, x, case["num_vertices"]) - np.array(case["result"]))) < 1e-3


@pytest.mark.parametrize("case", _cases("eigenvector"), ids=lambda c: c["name"])
@pytest.mark.parametrize("transposed", [True, False])
def test_eigenvector_goldens(case, transposed):
    from cugraph_amd import pylibcugraph as plc

    g, h = make_sg_graph(np.array(case["src"], np.int32), np.array(case["dst"], np.int32),
                         np.array(case["wgt"], np.float32), store_transposed=transposed)
    v, x = plc.eigenvector_centrality(h, g, case["epsilon"], case["max_iterations"], False)
    assert np.max(np.abs(_by_vertex(v, x, case["num_vertices"]) - np.array(case["result"]))) < 1e-3


@pytest.mark.parametrize("case", _cases("hits"), ids=lambda c: c["name"])
@pytest.mark.parametrize("renumber", [False, True])
def test_hits_goldens(case, renumber):
    from cugraph_amd import pylibcugraph as plc

    n = case["num_vertices"]
    g, h = make_sg_graph(np.array(case["src"], np.int32), np.array(case["dst"], np.int32),
                         np.array(case["wgt"], np.float32), store_transposed=case["store_transposed"],
                         renumber=renumber, vertices=np.arange(n, dtype=np.int32))
    iv = ix = None
    if "initial_hubs" in case:
        iv = to_dev(np.array(case["initial_vertices"], np.int32))
        ix = to_dev(np.array(case["initial_hubs"], np.float32))
    v, hubs, auth = plc.hits(h, g, case["epsilon"], case["max_iterations"], iv, ix, case["normalize"], False)
    assert np.max(np.abs(_by_vertex(v, hubs, n) - np.array(case["hubs"]))) < 1e-3
    assert np.max(np.abs(_by_vertex(v, auth, n) - np.array(case["authorities"]))) < 1e-3


def test_non_convergence_is_an_error():
    """CUGRAPH_FAIL in the reference -> CUGRAPH_UNKNOWN_ERROR -> RuntimeError (utils.pyx:37-83)"""
    from cugraph_amd import pylibcugraph as plc

    c = _cases("katz")[0]
    g, h = make_sg_graph(np.array(c["src"], np.int32), np.array(c["dst"], np.int32),
                         np.array(c["wgt"], np.float32))
    with pytest.raises(RuntimeError, match="Katz Centrality failed to converge"):
        plc.katz_centrality(h, g, None, 0.01, 1.0, 1e-12, 2, False)
    with pytest.raises(RuntimeError, match="Eigenvector Centrality failed to converge"):
        plc.eigenvector_centrality(h, g, 1e-12, 1, False)
    with pytest.raises(RuntimeError, match="HITS failed to converge"):
        plc.hits(h, g, 1e-12, 1, None, None, True, False)
    # arguments the algorithm rejects come back as CUGRAPH_UNKNOWN_ERROR too (utils.hpp:53-55)
    with pytest.raises(RuntimeError, match=r"alpha should be in \[0.0, 1.0\]"):
        plc.katz_centrality(h, g, None, 1.5, 1.0, 1e-6, 100, False)
    with pytest.raises(RuntimeError, match="epsilon should be non-negative"):
        plc.katz_centrality(h, g, None, 0.01, 1.0, -1e-6, 100, False)
    with pytest.raises(RuntimeError, match="epsilon should be non-negative"):
        plc.eigenvector_centrality(h, g, -1.0, 10, False)
    with pytest.raises(RuntimeError, match="epsilon should be non-negative"):
        plc.hits(h, g, -1.0, 10, None, None, True, False)


@pytest.mark.parametrize("scale", [12, 16])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("xs_tiles", [False, True])
@pytest.mark.parametrize("renumber", [False, True])
def test_rmat_centralities_vs_oracle(scale, dtype, xs_tiles, renumber, monkeypatch):
    import oracle
    from oracle import np_centrality as npc
    from cugraph_amd import pylibcugraph as plc

    monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", "0" if xs_tiles else str(1 << 40))
    # the XS runs also take the compact source space big graphs get (the products then scatter
    # their input vector to the slots)
    monkeypatch.setenv("CUGRAPH_AMD_COMPACT", "1" if xs_tiles else "0")
    nv, ne = 1 << scale, 16 << scale
    src, dst = oracle.rmat_edges(scale, ne, seed=5, scramble=True)
    wgt = oracle.rmat_weights(ne, seed=6).astype(dtype)
    tol = 2e-5 if dtype == np.float32 else 1e-10
    g, h = make_sg_graph(src, dst, wgt, renumber=renumber, vertices=np.arange(nv, dtype=np.int32),
                         multigraph=True)
    # Katz: alpha well inside the convergence radius of this weighted graph
    alpha = 0.5 / np.bincount(dst, weights=wgt.astype(np.float64), minlength=nv).max()
    betas = np.linspace(0.5, 1.5, nv).astype(dtype)
    for b in (None, betas):
        v, x = plc.katz_centrality(h, g, None if b is None else to_dev(b), alpha, 1.0, 1e-9, 1000, False)
        want, _ = npc.katz_centrality(nv, src, dst, wgt, alpha=alpha, beta=1.0, betas=b, epsilon=1e-9,
                                      max_iterations=1000)
        assert np.max(np.abs(_by_vertex(v, x, nv) - want)) < tol
    v, x = plc.eigenvector_centrality(h, g, 1e-9, 1000, False)
    want, _ = npc.eigenvector_centrality(nv, src, dst, wgt, epsilon=1e-9, max_iterations=1000)
    assert np.max(np.abs(_by_vertex(v, x, nv) - want)) < tol
    for normalize in (False, True):
        v, hubs, auth = plc.hits(h, g, 1e-9, 1000, None, None, normalize, False)
        wh, wa, _, _ = npc.hits(nv, src, dst, epsilon=1e-9, max_iterations=1000, normalize=normalize)
        assert np.max(np.abs(_by_vertex(v, hubs, nv) - wh)) < tol * (1 if normalize else 10)
        assert np.max(np.abs(_by_vertex(v, auth, nv) - wa)) < tol * (1 if normalize else 10)
