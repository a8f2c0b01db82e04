"""CPU: pins the oracle (C restatement + NetworkX/SciPy) to every golden vector the reference's
own tests hold for the PageRank path, and the oracle's parts to each other."""
import os

import numpy as np
import pytest

import oracle
from helpers import case_arrays, load_csv


def _rel_close(actual, expected, rel):
    actual, expected = np.asarray(actual, float), np.asarray(expected, float)
    return np.all(np.abs(actual - expected) <= rel * np.maximum(np.abs(actual), np.abs(expected)))


@pytest.mark.parametrize("name", ["karate.csv", "dolphins.csv", "Simple_1", "Simple_2"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_c_oracle_vs_pylibcugraph_goldens(golden, name, dtype):
    block = golden["pylibcugraph"]
    case = block["cases"][name]
    src, dst, wgt = case_arrays(case)
    nv = len(case["expected"])
    off, idx, perm = oracle.compress(nv, dst, src)
    pr, it, conv = oracle.pagerank(nv, off, idx, wgt[perm], dtype=dtype, alpha=block["alpha"],
                                   epsilon=block["epsilon"], max_iterations=block["max_iterations"])
    assert conv and _rel_close(pr, case["expected"], block["rel_tol"])


def test_c_oracle_vs_c_api_goldens(golden):
    block = golden["c_api"]
    for name, case in block["cases"].items():
        src, dst, wgt = case_arrays(case)
        nv = case["num_vertices"]
        off, idx, perm = oracle.compress(nv, dst, src)
        pr, it, conv = oracle.pagerank(nv, off, idx, wgt[perm], dtype=np.float32, alpha=case["alpha"],
                                       epsilon=case["epsilon"], max_iterations=case["max_iterations"],
                                       pers_vertices=case.get("pers_vertices"),
                                       pers_values=case.get("pers_values"))
        assert _rel_close(pr, case["expected"], block["rel_tol"]), name
        assert conv == (not case["allow_nonconvergence"]), name


def test_docstring_example(golden):
    # personalized_pagerank.pyx:153-175
    off, idx, perm = oracle.compress(4, np.array([1, 2, 3]), np.array([0, 1, 2]))
    pr, _, _ = oracle.pagerank(4, off, idx, np.ones(3), dtype=np.float32, alpha=0.85, epsilon=1e-6,
                               max_iterations=500, pers_vertices=[0, 2], pers_values=[0.008309, 0.991691])
    assert _rel_close(pr, [0.00446455, 0.00379487, 0.53607565, 0.45566472], 1e-4)


@pytest.mark.parametrize("name", ["karate.csv", "dolphins.csv", "netscience.csv", "polbooks.csv"])
def test_networkx_scipy_and_c_oracle_agree(name):
    """The oracle north_star names (NetworkX) against the C restatement: NetworkX tol maps to
    epsilon = N*tol (nx-cugraph, pagerank_alg.py:81)."""
    src, dst, wgt = load_csv(name)
    nv = int(max(src.max(), dst.max())) + 1
    tol = 1e-10
    x = oracle.nx_pagerank(nv, src, dst, wgt, alpha=0.85, tol=tol, max_iter=1000)
    y, it_s = oracle.scipy_pagerank(nv, src, dst, wgt, alpha=0.85, tol=tol, max_iter=1000,
                                    return_info=True)
    off, idx, perm = oracle.compress(nv, dst, src)
    z, it_c, conv = oracle.pagerank(nv, off, idx, wgt[perm], dtype=np.float64, alpha=0.85,
                                    epsilon=nv * tol, max_iterations=1000)
    assert conv and it_c == it_s
    assert np.max(np.abs(x - y)) < 1e-14 and np.max(np.abs(y - z)) < 1e-13


def test_networkx_karate_reproduces_pylibcugraph_golden(golden):
    src, dst, _ = load_csv("karate.csv")
    x = oracle.nx_pagerank(34, src, dst, alpha=0.85, tol=1e-6 / 34, max_iter=500)
    assert _rel_close(x, golden["pylibcugraph"]["cases"]["karate.csv"]["expected"], 1e-4)


def test_personalization_and_nstart_vs_networkx():
    src, dst, wgt = load_csv("dolphins.csv")
    nv = 62
    rng = np.random.default_rng(1)
    pers = rng.random(nv)
    pers[rng.random(nv) < 0.5] = 0
    nstart = rng.random(nv)
    tol = 1e-11
    x = oracle.nx_pagerank(nv, src, dst, wgt, alpha=0.85, tol=tol, max_iter=1000,
                           nstart=dict(enumerate(nstart)), personalization=dict(enumerate(pers)))
    off, idx, perm = oracle.compress(nv, dst, src)
    nz = np.nonzero(pers)[0]
    z, _, conv = oracle.pagerank(nv, off, idx, wgt[perm], dtype=np.float64, alpha=0.85,
                                 epsilon=nv * tol, max_iterations=1000, nstart=nstart / nstart.sum(),
                                 pers_vertices=nz, pers_values=pers[nz])
    assert conv and np.max(np.abs(x - z)) < 1e-12


@pytest.mark.parametrize("scale", [6, 11])
def test_compress_equals_lexsort_definition(scale):
    nv, ne = 1 << scale, 16 << scale
    src, dst = oracle.rmat_edges(scale, ne, seed=5)
    w = oracle.rmat_weights(ne, seed=6)
    off, idx, perm = oracle.compress(nv, dst, src)
    o2, i2, w2 = oracle.csc_lexsort(nv, src, dst, w)
    assert np.array_equal(off, o2) and np.array_equal(idx, i2) and np.array_equal(w[perm], w2)


def test_rmat_generator_properties():
    # Philox4x32-10 known answers (Random123 kat_vectors: counter/key all zero, and all ones)
    import ctypes

    out = (ctypes.c_uint32 * 4)()
    oracle.lib().oracle_philox(0, 0, 0, 0, 0, 0, out)
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    oracle.lib().oracle_philox(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, out)
    assert [hex(v) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    # slices of the stream are position-addressable; quadrant probabilities follow (a, b, c)
    s, d = oracle.rmat_edges(12, 1 << 16, seed=3)
    s2, d2 = oracle.rmat_edges(12, 1000, seed=3, first=5000)
    assert np.array_equal(s[5000:6000], s2) and np.array_equal(d[5000:6000], d2)
    top_s, top_d = s >> 11, d >> 11
    frac = [np.mean((top_s == i) & (top_d == j)) for i in (0, 1) for j in (0, 1)]
    assert np.allclose(frac, [0.57, 0.19, 0.19, 0.05], atol=0.01)
    # scramble is a bijection
    ids = np.arange(1 << 10, dtype=np.int32)
    # every id appears as some scrambled id: generate many edges and compare id sets
    a, _ = oracle.rmat_edges(10, 1 << 18, seed=1, scramble=False)
    b, _ = oracle.rmat_edges(10, 1 << 18, seed=1, scramble=True)
    assert len(np.unique(a)) == len(np.unique(b))
    w = oracle.rmat_weights(10000)
    assert w.min() > 0 and w.max() <= 1 and abs(w.mean() - 0.5) < 0.02
    assert ids.size == 1024


# ---------------------------------------------------------------- sibling power iterations
def _centrality_cases(kind):
    import json

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")) as f:
        return json.load(f)["c_api_centrality"][kind]


def test_centrality_oracle_matches_reference_goldens():
    """the numpy restatements reproduce every golden of cpp/tests/c_api/{katz,eigenvector_centrality,
    hits}_test.c to the 0.001 those tests use"""
    from oracle import np_centrality as npc

    for c in _centrality_cases("katz"):
        x, _ = npc.katz_centrality(c["num_vertices"], np.array(c["src"]), np.array(c["dst"]),
                                   np.array(c["wgt"], np.float32), alpha=c["alpha"], beta=c["beta"],
                                   epsilon=c["epsilon"], max_iterations=c["max_iterations"])
        assert np.max(np.abs(x - np.array(c["result"]))) < 1e-3, c["name"]
    for c in _centrality_cases("eigenvector"):
        x, _ = npc.eigenvector_centrality(c["num_vertices"], np.array(c["src"]), np.array(c["dst"]),
                                          np.array(c["wgt"], np.float32), epsilon=c["epsilon"],
                                          max_iterations=c["max_iterations"])
        assert np.max(np.abs(x - np.array(c["result"]))) < 1e-3, c["name"]
    for c in _centrality_cases("hits"):
        init = None
        if "initial_hubs" in c:
            init = np.zeros(c["num_vertices"])
            init[c["initial_vertices"]] = c["initial_hubs"]
        h, a, _, _ = npc.hits(c["num_vertices"], np.array(c["src"]), np.array(c["dst"]),
                              epsilon=c["epsilon"], max_iterations=c["max_iterations"],
                              initial_hubs=init, normalize=c["normalize"])
        assert np.max(np.abs(h - np.array(c["hubs"]))) < 1e-3, c["name"]
        assert np.max(np.abs(a - np.array(c["authorities"]))) < 1e-3, c["name"]


def test_centrality_oracle_matches_networkx():
    import networkx as nx
    from oracle import np_centrality as npc

    src, dst, wgt = load_csv("karate.csv")
    G = nx.DiGraph()
    G.add_weighted_edges_from(zip(src.tolist(), dst.tolist(), wgt.tolist()))
    n = 34
    want = nx.katz_centrality(G, alpha=0.05, beta=1.0, max_iter=1000, tol=1e-12, weight="weight")
    x, _ = npc.katz_centrality(n, src, dst, wgt, alpha=0.05, beta=1.0, epsilon=1e-12, max_iterations=1000)
    assert np.max(np.abs(x - np.array([want[i] for i in range(n)]))) < 1e-9
    want = nx.eigenvector_centrality(G, max_iter=2000, tol=1e-13, weight="weight")
    x, _ = npc.eigenvector_centrality(n, src, dst, wgt, epsilon=1e-15, max_iterations=5000)
    assert np.max(np.abs(x - np.array([want[i] for i in range(n)]))) < 1e-7
    hw, aw = nx.hits(nx.DiGraph(list(zip(src.tolist(), dst.tolist()))), max_iter=1000, tol=1e-14)
    h, a, _, _ = npc.hits(n, src, dst, epsilon=1e-15, max_iterations=5000, normalize=True)
    assert np.max(np.abs(h - np.array([hw[i] for i in range(n)]))) < 1e-8
    assert np.max(np.abs(a - np.array([aw[i] for i in range(n)]))) < 1e-8
