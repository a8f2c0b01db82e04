"""cugraph-level API (cugraph_amd.cugraph): the reference's own wrapper tests restated
(python/cugraph/cugraph/tests/link_analysis/test_pagerank.py:205-212, 367-469), NetworkX as the
oracle, plus drop_self_loops / drop_multi_edges of the graph build."""
import networkx as nx
import numpy as np
import pandas as pd
import pytest

from helpers import compressed_arrays, load_csv, make_sg_graph

pytestmark = pytest.mark.gpu


def _karate_frame():
    src, dst, wgt = load_csv("karate.csv")
    return pd.DataFrame({"src": src, "dst": dst, "weights": wgt})


def _directed_graph(store_transposed=True, weighted=True, renumber=True):
    from cugraph_amd import cugraph

    G = cugraph.Graph(directed=True)
    G.from_cudf_edgelist(_karate_frame(), source="src", destination="dst",
                         edge_attr="weights" if weighted else None,
                         store_transposed=store_transposed, renumber=renumber)
    return G


def _nx_digraph():
    src, dst, wgt = load_csv("karate.csv")
    G = nx.DiGraph()
    G.add_weighted_edges_from(zip(src.tolist(), dst.tolist(), wgt.tolist()))
    return G


@pytest.mark.parametrize("renumber", [True, False])
def test_pagerank_frame_matches_networkx(renumber):
    from cugraph_amd import cugraph

    G = _directed_graph(renumber=renumber)
    assert G.number_of_vertices() == 34 and G.number_of_edges() == 156
    df = cugraph.pagerank(G, alpha=0.85, max_iter=500, tol=1.0e-7)
    assert list(df.columns) == ["vertex", "pagerank"] and len(df) == 34
    want = nx.pagerank(_nx_digraph(), alpha=0.85, tol=1e-9, max_iter=500)
    got = dict(zip(df["vertex"].tolist(), df["pagerank"].tolist()))
    assert set(got) == set(want)
    # the reference's own criterion is |diff| < 1.1 * tol for 99 % of the vertices; all here
    assert max(abs(got[v] - want[v]) for v in want) < 1e-6


def test_pagerank_personalization_nstart_frames():
    from cugraph_amd import cugraph

    G = _directed_graph()
    pers = pd.DataFrame({"vertex": np.array([17, 26], np.int32),
                         "values": np.array([0.5, 0.75], np.float32)})
    nstart = pd.DataFrame({"vertex": np.arange(34, dtype=np.int32),
                           "values": np.full(34, 1 / 34, np.float32)})
    df = cugraph.pagerank(G, personalization=pers, nstart=nstart, max_iter=500, tol=1e-7)
    want = nx.pagerank(_nx_digraph(), alpha=0.85, tol=1e-9, max_iter=500,
                       personalization={17: 0.5, 26: 0.75})
    got = dict(zip(df["vertex"].tolist(), df["pagerank"].tolist()))
    assert max(abs(got[v] - want[v]) for v in want) < 1e-6
    # precomputed out-weight sums: the true sums change nothing
    src, _, wgt = load_csv("karate.csv")
    sums = np.bincount(src, weights=wgt.astype(np.float64), minlength=34).astype(np.float32)
    pre = pd.DataFrame({"vertex": np.arange(34, dtype=np.int32), "sums": sums})
    df2 = cugraph.pagerank(G, personalization=pers, precomputed_vertex_out_weight=pre,
                           max_iter=500, tol=1e-7)
    got2 = dict(zip(df2["vertex"].tolist(), df2["pagerank"].tolist()))
    assert max(abs(got2[v] - want[v]) for v in want) < 1e-6


def test_pagerank_invalid_personalization_dtype():
    """test_pagerank.py:367-410, same message texts"""
    from cugraph_amd import cugraph

    G = _directed_graph()
    personalization = pd.DataFrame({"vertex": [17, 26], "values": [0.5, 0.75]})  # int64 / float64
    warning_msg = (
        "PageRank requires 'personalization' values to match the "
        "graph's 'edge_attr' type. edge_attr type is: "
        "float32 and got 'personalization' values "
        "of type: float64."
    )
    with pytest.warns(UserWarning, match=warning_msg):
        cugraph.pagerank(G, personalization=personalization)
    warning_msg = (
        "PageRank requires 'personalization' vertex to match the "
        "graph's 'vertex' type. input graph's vertex type is: "
        "int32 and got 'personalization' vertex "
        "of type: int64."
    )
    with pytest.warns(UserWarning, match=warning_msg):
        cugraph.pagerank(G, personalization=personalization)
    with pytest.raises(NotImplementedError):
        cugraph.pagerank(G, personalization={17: 0.5})


def test_pagerank_transposed_false():
    """test_pagerank.py:413-424; the result is the same as with store_transposed=True"""
    from cugraph_amd import cugraph

    G = _directed_graph(store_transposed=False)
    warning_msg = (
        "Pagerank expects the 'store_transposed' "
        "flag to be set to 'True' for optimal performance during "
        "the graph creation"
    )
    with pytest.warns(UserWarning, match=warning_msg):
        df = cugraph.pagerank(G, tol=1e-7, max_iter=500)
    ref = cugraph.pagerank(_directed_graph(), tol=1e-7, max_iter=500)
    a = df.sort_values("vertex")["pagerank"].to_numpy()
    b = ref.sort_values("vertex")["pagerank"].to_numpy()
    assert np.max(np.abs(a - b)) < 1e-7


def test_pagerank_non_convergence():
    """test_pagerank.py:427-469"""
    from cugraph_amd import cugraph

    G = _directed_graph()
    with pytest.raises(cugraph.exceptions.FailedToConvergeError):
        cugraph.pagerank(G, max_iter=1, fail_on_nonconvergence=True)
    (df, converged) = cugraph.pagerank(G, max_iter=1, fail_on_nonconvergence=False)
    assert type(df) is pd.DataFrame and type(converged) is bool and converged is False
    (df, converged) = cugraph.pagerank(G, fail_on_nonconvergence=False)
    assert type(df) is pd.DataFrame and type(converged) is bool and converged is True
    personalization = pd.DataFrame({"vertex": np.array([17, 26], np.int32),
                                    "values": np.array([0.5, 0.75], np.float32)})
    with pytest.raises(cugraph.exceptions.FailedToConvergeError):
        cugraph.pagerank(G, max_iter=1, personalization=personalization, fail_on_nonconvergence=True)
    (df, converged) = cugraph.pagerank(G, max_iter=1, personalization=personalization,
                                       fail_on_nonconvergence=False)
    assert converged is False
    (df, converged) = cugraph.pagerank(G, personalization=personalization,
                                       fail_on_nonconvergence=False)
    assert converged is True


def test_undirected_graph_and_networkx_input():
    """cugraph.Graph() is undirected: edges are symmetrised; a NetworkX graph in, a dict out"""
    from cugraph_amd import cugraph

    kc = nx.karate_club_graph()
    want = nx.pagerank(kc, alpha=0.85, tol=1e-10, max_iter=500, weight=None)
    edges = np.array(list(kc.edges()), dtype=np.int32)
    G = cugraph.Graph()
    G.from_pandas_edgelist(pd.DataFrame({"source": edges[:, 0], "destination": edges[:, 1]}),
                           store_transposed=True)
    assert not G.is_directed() and G.number_of_edges() == kc.number_of_edges()
    df = cugraph.pagerank(G, tol=1e-7, max_iter=500)
    got = dict(zip(df["vertex"].tolist(), df["pagerank"].tolist()))
    assert max(abs(got[v] - want[v]) for v in want) < 1e-6
    d = cugraph.pagerank(kc, tol=1e-7, max_iter=500)  # NetworkX graph in
    assert isinstance(d, dict) and set(d) == set(want)
    assert max(abs(d[v] - want[v]) for v in want) < 1e-6


@pytest.mark.parametrize("directed", [False, True])
def test_nx_backend_entry(directed):
    """nx-cugraph's pagerank (pagerank_alg.py:35-96) against NetworkX itself, NetworkX arguments"""
    from cugraph_amd import cugraph

    G = nx.les_miserables_graph()  # string node names, weighted
    if directed:
        G = nx.DiGraph([(u, v, d) for u, v, d in G.edges(data=True) if hash(u) % 3])
    pers = {n: float(i % 5 + 1) for i, n in enumerate(G) if i % 7 == 0}
    nstart = {n: float(i % 3 + 1) for i, n in enumerate(G)}
    for kw in ({}, {"personalization": pers}, {"nstart": nstart}, {"weight": None}):
        want = nx.pagerank(G, alpha=0.9, tol=1e-10, max_iter=1000, **kw)
        got = cugraph.nx_pagerank(G, alpha=0.9, tol=1e-9, max_iter=1000, dtype=np.float64, **kw)
        assert set(got) == set(want)
        assert max(abs(got[n] - want[n]) for n in want) < 1e-7
    with pytest.raises(nx.PowerIterationFailedConvergence):
        cugraph.nx_pagerank(G, max_iter=2, tol=1e-12)
    with pytest.raises(ZeroDivisionError):
        cugraph.nx_pagerank(G, personalization={n: 0 for n in G})
    assert cugraph.nx_pagerank(nx.DiGraph()) == {}


@pytest.mark.parametrize("name", ["karate_multi_edge.csv", "karate_s_loop.csv"])
@pytest.mark.parametrize("renumber", [False, True])
@pytest.mark.parametrize("symmetric", [False, True])
def test_drop_self_loops_and_multi_edges(name, renumber, symmetric):
    """graph_sg.cpp:185-208: self-loops dropped, one copy of every parallel edge kept (the minimum
    weight when the graph is flagged symmetric, the first in input order otherwise)"""
    import torch
    from cugraph_amd import pylibcugraph as plc

    src, dst, wgt = load_csv(name)
    rng = np.random.default_rng(1)
    wgt = (wgt * rng.integers(1, 5, len(wgt))).astype(np.float32)  # make the copies differ
    # add explicit copies and loops so that both flags have work in both files
    src = np.concatenate([src, src[:20], np.array([3, 3, 9], np.int32)])
    dst = np.concatenate([dst, dst[:20], np.array([3, 3, 9], np.int32)])
    wgt = np.concatenate([wgt, (wgt[:20] * 0.5).astype(np.float32), np.array([1, 2, 3], np.float32)])
    nv = int(max(src.max(), dst.max())) + 1
    h = plc.ResourceHandle()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    g = plc.SGGraph(h, plc.GraphProperties(is_symmetric=symmetric, is_multigraph=False), dev(src),
                    dev(dst), weight_array=dev(wgt), store_transposed=True, renumber=renumber,
                    vertices_array=dev(np.arange(nv, dtype=np.int32)), drop_self_loops=True,
                    drop_multi_edges=True)
    off, idx, w, nmap = compressed_arrays(g, h)
    # expected edge set on the host
    keep = src != dst
    s, d, x = src[keep], dst[keep], wgt[keep]
    best = {}
    for i in range(len(s)):
        k = (int(s[i]), int(d[i]))
        if k not in best or (symmetric and x[i] < best[k]):
            best[k] = x[i]
    assert len(idx) == len(best) and int(off[-1]) == len(best)
    ext = nmap if nmap is not None else np.arange(nv)
    rows = np.repeat(np.arange(len(off) - 1), np.diff(off))
    got = {(int(ext[idx[k]]), int(ext[rows[k]])): w[k] for k in range(len(idx))}
    assert got == best
    # PageRank on it equals NetworkX on the simple weighted digraph
    G = nx.DiGraph()
    G.add_nodes_from(range(nv))
    G.add_weighted_edges_from((a, b, float(v)) for (a, b), v in best.items())
    want = nx.pagerank(G, alpha=0.85, tol=1e-10, max_iter=500)
    v, r = plc.pagerank(h, g, None, None, None, None, 0.85, 1e-7, 500, False)
    got_pr = dict(zip(v.cpu().numpy().tolist(), r.cpu().numpy().tolist()))
    assert max(abs(got_pr[n] - want[n]) for n in want) < 1e-6


@pytest.mark.parametrize("kind", ["strings", "two_columns"])
@pytest.mark.parametrize("renumber", [True, False])
def test_string_and_multi_column_vertices(kind, renumber):
    """Vertices named by strings or by several columns (number_map.py; the reference's
    test_pagerank_multi_column, test_pagerank.py:300-366): ranks equal the integer-id run's, the
    result names its vertices the way unrenumber does (`vertex`, or `0_vertex`, `1_vertex`), and
    personalization / nstart frames use the same naming with the value column last."""
    from cugraph_amd import cugraph

    pdf = _karate_frame()
    base = cugraph.pagerank(_directed_graph(), max_iter=500, tol=1e-7)
    want = dict(zip(base["vertex"].tolist(), base["pagerank"].tolist()))
    pers_int = pd.DataFrame({"vertex": np.array([17, 26], np.int32), "values": np.array([0.5, 0.75], np.float32)})
    base_p = cugraph.pagerank(_directed_graph(), personalization=pers_int, max_iter=500, tol=1e-7)
    want_p = dict(zip(base_p["vertex"].tolist(), base_p["pagerank"].tolist()))
    G = cugraph.Graph(directed=True)
    if kind == "strings":
        name = lambda v: f"member-{v:02d}"  # noqa: E731
        frame = pd.DataFrame({"a": [name(v) for v in pdf["src"]], "b": [name(v) for v in pdf["dst"]],
                              "w": pdf["weights"]})
        G.from_cudf_edgelist(frame, source="a", destination="b", edge_attr="w", store_transposed=True,
                             renumber=renumber)
        pers = pd.DataFrame({"vertex": [name(17), name(26)], "values": np.array([0.5, 0.75], np.float32)})
        nstart = pd.DataFrame({"vertex": [name(v) for v in range(34)], "values": np.full(34, 1 / 34, np.float32)})
        key = lambda df: [int(v[-2:]) for v in df["vertex"]]  # noqa: E731
        cols = ["vertex", "pagerank"]
        bad = pd.DataFrame({"vertex": ["nobody"], "values": np.array([1.0], np.float32)})
    else:
        split = lambda s: (s // 10 + 1000, s % 10)  # noqa: E731
        frame = pd.DataFrame({"s0": split(pdf["src"])[0], "s1": split(pdf["src"])[1],
                              "d0": split(pdf["dst"])[0], "d1": split(pdf["dst"])[1], "w": pdf["weights"]})
        G.from_cudf_edgelist(frame, source=["s0", "s1"], destination=["d0", "d1"], edge_attr="w",
                             store_transposed=True, renumber=renumber)
        pers = pd.DataFrame({"0_vertex": [1001, 1002], "1_vertex": [7, 6], "values": np.array([0.5, 0.75], np.float32)})
        ids = np.arange(34)
        nstart = pd.DataFrame({"0_vertex": ids // 10 + 1000, "1_vertex": ids % 10,
                               "values": np.full(34, 1 / 34, np.float32)})
        key = lambda df: ((df["0_vertex"] - 1000) * 10 + df["1_vertex"]).tolist()  # noqa: E731
        cols = ["0_vertex", "1_vertex", "pagerank"]
        bad = pd.DataFrame({"0_vertex": [1001], "1_vertex": [99], "values": np.array([1.0], np.float32)})
    assert G.number_of_vertices() == 34
    df = cugraph.pagerank(G, max_iter=500, tol=1e-7)
    assert list(df.columns) == cols and len(df) == 34
    got = dict(zip(key(df), df["pagerank"].tolist()))
    assert set(got) == set(want) and max(abs(got[v] - want[v]) for v in want) < 1e-7
    df = cugraph.pagerank(G, personalization=pers, nstart=nstart, max_iter=500, tol=1e-7)
    got = dict(zip(key(df), df["pagerank"].tolist()))
    assert max(abs(got[v] - want_p[v]) for v in want_p) < 1e-6
    with pytest.raises(ValueError, match="vertex not found"):
        cugraph.pagerank(G, personalization=bad)


def test_adjacency_list_and_matrix_inputs():
    """graph_classes.py:206-271 from_cudf_adjlist (CSR arrays), :443-500 from_pandas_adjacency /
    from_numpy_array / from_numpy_matrix: the same ranks as the edge-list graph of the same edges."""
    import scipy.sparse
    from cugraph_amd import cugraph

    pdf = _karate_frame()
    base = cugraph.pagerank(_directed_graph(), max_iter=500, tol=1e-7)
    want = dict(zip(base["vertex"].tolist(), base["pagerank"].tolist()))
    M = scipy.sparse.coo_matrix((pdf["weights"], (pdf["src"], pdf["dst"])), shape=(34, 34)).tocsr()
    G = cugraph.Graph(directed=True)
    G.from_cudf_adjlist(pd.Series(M.indptr), pd.Series(M.indices), pd.Series(M.data), store_transposed=True)
    assert G.number_of_vertices() == 34 and G.number_of_edges() == 156
    df = cugraph.pagerank(G, max_iter=500, tol=1e-7)
    got = dict(zip(df["vertex"].tolist(), df["pagerank"].tolist()))
    assert max(abs(got[v] - want[v]) for v in want) < 1e-7
    with pytest.raises(RuntimeError, match="already has values"):
        G.from_cudf_adjlist(pd.Series(M.indptr), pd.Series(M.indices))
    with pytest.raises(ValueError):
        cugraph.Graph(directed=True).from_cudf_adjlist(pd.Series([0, 2, 1]), pd.Series([0, 1]))
    # an undirected graph stores both directions itself; a vertex without edges (33 rows + 1) still counts
    U = cugraph.Graph()
    sym = (M + M.T).tocsr()
    sym.resize((35, 35))
    U.from_cudf_adjlist(pd.Series(sym.indptr), pd.Series(sym.indices), store_transposed=True)
    assert U.number_of_vertices() == 35
    udf = cugraph.pagerank(U, max_iter=500, tol=1e-7)
    nxU = nx.Graph()
    nxU.add_nodes_from(range(35))
    nxU.add_edges_from(zip(pdf["src"].tolist(), pdf["dst"].tolist()))
    uw = nx.pagerank(nxU, alpha=0.85, tol=1e-9, max_iter=500)
    ug = dict(zip(udf["vertex"].tolist(), udf["pagerank"].tolist()))
    assert set(ug) == set(uw) and max(abs(ug[v] - uw[v]) for v in uw) < 1e-6
    # dense adjacency with labelled nodes
    A = np.zeros((34, 34))
    A[pdf["src"], pdf["dst"]] = pdf["weights"]
    labels = [f"n{v:02d}" for v in range(34)]
    D = cugraph.Graph(directed=True)
    D.from_pandas_adjacency(pd.DataFrame(A, columns=labels))
    with pytest.warns(UserWarning, match="store_transposed"):
        ddf = cugraph.pagerank(D, max_iter=500, tol=1e-7)
    dg = {int(k[1:]): v for k, v in zip(ddf["vertex"].tolist(), ddf["pagerank"].tolist())}
    assert max(abs(dg[v] - want[v]) for v in want) < 1e-6
    N = cugraph.Graph(directed=True)
    N.from_numpy_matrix(np.matrix(A))
    assert N.number_of_edges() == 156
    with pytest.raises(TypeError):
        cugraph.Graph().from_numpy_matrix(A)
    with pytest.raises(ValueError):
        cugraph.Graph().from_numpy_array(np.zeros(3))


def test_graph_queries():
    """view_edge_list / nodes / degrees / has_isolated_vertices in the caller's vertex names"""
    from cugraph_amd import cugraph

    pdf = _karate_frame()
    G = _directed_graph()
    el = G.view_edge_list()
    assert sorted(zip(el["src"], el["dst"])) == sorted(zip(pdf["src"], pdf["dst"])) and "weights" in el
    assert sorted(G.nodes().tolist()) == list(range(34)) and not G.has_isolated_vertices()
    ind = dict(zip(G.in_degree()["vertex"], G.in_degree()["degree"]))
    assert all(ind[v] == int((pdf["dst"] == v).sum()) for v in range(34))
    deg = dict(zip(G.degree()["vertex"], G.degree()["degree"]))
    assert all(deg[v] == int((pdf["dst"] == v).sum() + (pdf["src"] == v).sum()) for v in range(34))
    U = cugraph.Graph()
    U.from_pandas_edgelist(pd.DataFrame({"a": ["x", "y", "z"], "b": ["y", "z", "x"]}), "a", "b",
                           vertices=["x", "y", "z", "lonely"])
    assert sorted(U.nodes().tolist()) == ["lonely", "x", "y", "z"] and U.has_isolated_vertices()
    assert len(U.view_edge_list()) == 3 and U.number_of_edges() == 3
    d = dict(zip(U.degree()["vertex"], U.degree()["degree"]))
    assert d == {"x": 2, "y": 2, "z": 2, "lonely": 0}
    df = cugraph.pagerank(U, max_iter=500, tol=1e-9)
    pr = dict(zip(df["vertex"], df["pagerank"]))
    assert abs(pr["x"] - pr["y"]) < 1e-7 and pr["lonely"] < pr["x"] and abs(sum(pr.values()) - 1) < 1e-6


def test_sibling_centralities_at_the_frame_level():
    """cugraph.katz_centrality / eigenvector_centrality / hits (katz_centrality.py:30-205,
    eigenvector_centrality.py:27-140, hits.py:28-140) against NetworkX on karate, with string
    vertex names, NetworkX input, and the reference's argument checks."""
    from cugraph_amd import cugraph

    pdf = _karate_frame()
    name = lambda v: f"m{v:02d}"  # noqa: E731
    frame = pd.DataFrame({"a": [name(v) for v in pdf["src"]], "b": [name(v) for v in pdf["dst"]]})
    G = cugraph.Graph(directed=True)
    G.from_cudf_edgelist(frame, source="a", destination="b", store_transposed=True)
    nxG = nx.DiGraph()
    nxG.add_edges_from(zip(frame["a"], frame["b"]))
    alpha = 1 / max(d for _, d in nxG.degree())
    kz = cugraph.katz_centrality(G, max_iter=1000, tol=1e-9)
    assert list(kz.columns) == ["vertex", "katz_centrality"]
    want = nx.katz_centrality(nxG, alpha=alpha, beta=1.0, max_iter=1000, tol=1e-9, normalized=False)
    got = dict(zip(kz["vertex"], kz["katz_centrality"]))
    scale = np.linalg.norm(list(want.values())) if abs(np.linalg.norm(list(got.values())) - 1) < 1e-3 else 1.0
    assert max(abs(got[v] - want[v] / scale) for v in want) < 1e-5
    ev = cugraph.eigenvector_centrality(G, max_iter=1000, tol=1e-7)  # (an unweighted graph computes in fp32)
    want = nx.eigenvector_centrality(nxG, max_iter=2000, tol=1e-12)
    got = dict(zip(ev["vertex"], ev["eigenvector_centrality"]))
    assert max(abs(got[v] - want[v]) for v in want) < 1e-4
    ht = cugraph.hits(G, max_iter=1000, tol=1e-7)
    wh, wa = nx.hits(nxG, max_iter=2000, tol=1e-12, normalized=True)
    assert list(ht.columns) == ["vertex", "hubs", "authorities"]
    gh, ga = dict(zip(ht["vertex"], ht["hubs"])), dict(zip(ht["vertex"], ht["authorities"]))
    assert max(abs(gh[v] - wh[v]) for v in wh) < 1e-5 and max(abs(ga[v] - wa[v]) for v in wa) < 1e-5
    nstart = pd.DataFrame({"vertex": [name(v) for v in range(34)], "values": np.full(34, 1 / 34, np.float32)})
    ht2 = cugraph.hits(G, max_iter=1000, tol=1e-7, nstart=nstart)
    assert max(abs(a - b) for a, b in zip(sorted(ht2["hubs"]), sorted(ht["hubs"]))) < 1e-6
    # NetworkX graph in, dicts out
    h2, a2 = cugraph.hits(nxG, max_iter=1000, tol=1e-7)
    assert max(abs(h2[v] - wh[v]) for v in wh) < 1e-5 and set(a2) == set(wa)
    assert set(cugraph.eigenvector_centrality(nxG)) == set(nxG.nodes())
    with pytest.raises(ValueError, match="'alpha' must be a positive float"):
        cugraph.katz_centrality(G, alpha=-1.0)
    with pytest.raises(ValueError, match="'max_iter' must be a positive integer"):
        cugraph.eigenvector_centrality(G, max_iter=0)
    with pytest.raises(ValueError, match="'tol' must be a positive float"):
        cugraph.katz_centrality(G, tol=0)
    T = cugraph.Graph(directed=True)
    T.from_cudf_edgelist(frame, source="a", destination="b", store_transposed=False)
    with pytest.warns(UserWarning, match="HITS expects the 'store_transposed' flag"):
        cugraph.hits(T)
