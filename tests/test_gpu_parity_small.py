"""GPU parity, small cases: golden vectors of the reference's own tests + oracle comparisons,
all through the C ABI (pylibcugraph mirror).  Bit-exact for the CSC, reference tolerances for
the floating-point ranks."""
import numpy as np
import pytest

from helpers import case_arrays, compressed_arrays, load_csv, make_sg_graph, to_dev

pytestmark = pytest.mark.gpu


def _expect_close(actual, expected, rel):
    actual = np.asarray(actual, dtype=np.float64)
    expected = np.asarray(expected, dtype=np.float64)
    assert actual.shape == expected.shape
    assert np.all(np.abs(actual - expected) <= rel * np.maximum(np.abs(actual), np.abs(expected)))


@pytest.mark.parametrize("name", ["karate.csv", "dolphins.csv", "Simple_1", "Simple_2"])
def test_pylibcugraph_goldens(golden, name):
    # python/pylibcugraph/pylibcugraph/tests/test_pagerank.py:175-218
    from cugraph_amd import pylibcugraph as plc
    import torch

    block = golden["pylibcugraph"]
    case = block["cases"][name]
    src, dst, wgt = case_arrays(case)
    g, h = make_sg_graph(src, dst, wgt, store_transposed=True)
    verts, ranks = plc.pagerank(h, g, None, None, None, None, block["alpha"], block["epsilon"],
                                block["max_iterations"], False)
    assert verts.dtype == torch.int32 and ranks.dtype == torch.float32
    verts = verts.cpu().numpy()
    ranks = ranks.cpu().numpy()
    expected = np.array(case["expected"])
    assert sorted(verts.tolist()) == list(range(len(expected)))
    _expect_close(ranks, expected[verts], block["rel_tol"])


@pytest.mark.parametrize("name", [
    "test_pagerank", "test_pagerank_with_transpose", "test_pagerank_4",
    "test_pagerank_4_with_transpose", "test_pagerank_non_convergence",
    "test_personalized_pagerank", "test_personalized_pagerank_non_convergence"])
def test_c_api_goldens(golden, name):
    # cpp/tests/c_api/pagerank_test.c:396-555
    from cugraph_amd import pylibcugraph as plc

    block = golden["c_api"]
    case = block["cases"][name]
    src, dst, wgt = case_arrays(case)
    g, h = make_sg_graph(src, dst, wgt, store_transposed=case["store_transposed"])
    allow = case["allow_nonconvergence"]
    if "pers_vertices" in case:
        out = plc.personalized_pagerank(
            h, g, None, None, None, None, to_dev(np.array(case["pers_vertices"], np.int32)),
            to_dev(np.array(case["pers_values"], np.float32)), case["alpha"], case["epsilon"],
            case["max_iterations"], False, fail_on_nonconvergence=not allow)
    else:
        out = plc.pagerank(h, g, None, None, None, None, case["alpha"], case["epsilon"],
                           case["max_iterations"], False, fail_on_nonconvergence=not allow)
    verts, ranks = out[0].cpu().numpy(), out[1].cpu().numpy()
    if allow:
        assert out[2] is False  # both non-convergence cases stop at max_iterations
    _expect_close(ranks, np.array(case["expected"])[verts], block["rel_tol"])
    assert g.info()["store_transposed"] == 1  # transposed in place if it was not (pagerank.cpp:105-110)


def test_nonconvergence_raises(golden):
    # python/cugraph/cugraph/tests/link_analysis/test_pagerank.py:427-469
    from cugraph_amd import pylibcugraph as plc

    src, dst, wgt = load_csv("karate.csv")
    g, h = make_sg_graph(src, dst, wgt)
    with pytest.raises(plc.FailedToConvergeError):
        plc.pagerank(h, g, None, None, None, None, 0.85, 1e-12, 1, False)
    v, r, conv = plc.pagerank(h, g, None, None, None, None, 0.85, 1e-12, 1, False,
                              fail_on_nonconvergence=False)
    assert conv is False and len(r) == 34


@pytest.mark.parametrize("name", ["karate.csv", "dolphins.csv", "netscience.csv", "polbooks.csv",
                                  "karate_multi_edge.csv", "karate_s_loop.csv"])
@pytest.mark.parametrize("transposed", [True, False])
@pytest.mark.parametrize("weighted", [True, False])
def test_compressed_layout_bit_exact(name, transposed, weighted):
    """COO -> CSC/CSR equals the (major, minor)-lexicographic definition, bit for bit."""
    import oracle

    src, dst, wgt = load_csv(name)
    if weighted:  # make weights distinct so that a wrong permutation cannot hide
        wgt = (np.arange(len(src)) * 0.37 + 1.0).astype(np.float32)
    nv = int(max(src.max(), dst.max())) + 1
    g, h = make_sg_graph(src, dst, wgt if weighted else None, store_transposed=transposed,
                         multigraph=True)
    off, idx, w, nm = compressed_arrays(g, h)
    major, minor = (dst, src) if transposed else (src, dst)
    o_off, o_idx, o_w = oracle.csc_lexsort(nv, minor, major, wgt if weighted else None)
    assert np.array_equal(off.astype(np.int64), o_off)
    assert np.array_equal(idx, o_idx)
    if weighted:
        assert np.array_equal(w, o_w)
    else:
        assert w is None
    assert nm is None


@pytest.mark.parametrize("scale,ef", [(8, 4), (12, 16), (16, 16)])
@pytest.mark.parametrize("scramble", [False, True])
def test_rmat_generator_matches_oracle(scale, ef, scramble):
    import ctypes
    import torch
    import oracle
    from cugraph_amd import _capi, pylibcugraph as plc
    from cugraph_amd.pylibcugraph.utils import create_view_from_py_obj, free_view

    lib = _capi.load()
    h = plc.ResourceHandle()
    ne = ef << scale
    src = torch.empty(ne, dtype=torch.int32, device="cuda")
    dst = torch.empty(ne, dtype=torch.int32, device="cuda")
    sv, dv = create_view_from_py_obj(src), create_view_from_py_obj(dst)
    err = ctypes.c_void_p()
    code = lib.cugraph_amd_generate_rmat_edgelist(h.c_resource_handle_ptr, scale, 0, 42, 0.57, 0.19,
                                                  0.19, int(scramble), sv, dv, ctypes.byref(err))
    _capi.check(code, err, "rmat")
    free_view(sv), free_view(dv)
    o_src, o_dst = oracle.rmat_edges(scale, ne, seed=42, scramble=scramble)
    assert np.array_equal(src.cpu().numpy(), o_src)
    assert np.array_equal(dst.cpu().numpy(), o_dst)
    for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
        w = torch.empty(1000, dtype=tdt, device="cuda")
        wv = create_view_from_py_obj(w)
        code = lib.cugraph_amd_generate_rmat_weights(h.c_resource_handle_ptr, 5, 43, wv,
                                                     ctypes.byref(err))
        _capi.check(code, err, "rmat_weights")
        free_view(wv)
        assert np.array_equal(w.cpu().numpy(), oracle.rmat_weights(1000, seed=43, first=5).astype(dt))


@pytest.mark.parametrize("scale", [10, 14, 17])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("xs_tiles", [False, True, "xs2"])
def test_rmat_pagerank_vs_oracle(scale, dtype, weighted, xs_tiles, monkeypatch):
    """R-MAT (hubs, multi-edges, self-loops, dangling and isolated vertices): CSC bit-exact, ranks
    within 1e-6 (fp32) / 1e-10 (fp64) max-abs of the C oracle, same iteration count."""
    import oracle
    from cugraph_amd import pylibcugraph as plc

    # the XS tile path (mid-degree rows) is only chosen for big graphs unless forced
    monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", "0" if xs_tiles else str(1 << 40))
    if xs_tiles == "xs2":  # xs2_kernel (two accumulator banks) is the default only from 40 M entries on
        monkeypatch.setenv("CUGRAPH_AMD_XS_ASYNC", "1")
    nv, ne = 1 << scale, 16 << scale
    src, dst = oracle.rmat_edges(scale, ne, seed=42)
    wgt = oracle.rmat_weights(ne, seed=43).astype(dtype) if weighted else None
    verts = np.arange(nv, dtype=np.int32)
    g, h = make_sg_graph(src, dst, wgt, vertices=verts, multigraph=True)
    off, idx, w, _ = compressed_arrays(g, h)
    o_off, o_idx, perm = oracle.compress(nv, dst, src)
    assert np.array_equal(off.astype(np.int64), o_off)
    assert np.array_equal(idx, o_idx)
    if weighted:
        assert np.array_equal(w, wgt[perm])
    elif dtype == np.float64:
        pytest.skip("an unweighted graph always computes in fp32 (graph_sg.cpp:247)")
    eps = nv * 1e-8  # NetworkX tol=1e-8 through nx-cugraph's epsilon = N*tol (pagerank_alg.py:81)
    v, r, conv = plc.pagerank(h, g, None, None, None, None, 0.85, eps, 500, False,
                              fail_on_nonconvergence=False)
    o_pr, o_it, o_conv = oracle.pagerank(nv, o_off, o_idx, None if wgt is None else wgt[perm],
                                         dtype=np.float64, alpha=0.85, epsilon=eps,
                                         max_iterations=500)
    assert conv and o_conv
    r = r.cpu().numpy().astype(np.float64)
    assert np.array_equal(v.cpu().numpy(), verts)
    tol = 1e-6 if dtype == np.float32 else 1e-10
    assert np.max(np.abs(r - o_pr)) <= tol
    assert abs(r.sum() - 1.0) < (1e-4 if dtype == np.float32 else 1e-10)


@pytest.mark.parametrize("tile_edges,piece", [(65536, 1024), (1 << 30, 300), (4096, 1 << 20), (1 << 20, 1 << 14)])
@pytest.mark.parametrize("scale,dtype,weighted", [(16, np.float32, False), (14, np.float64, True)])
@pytest.mark.parametrize("compact,row_bits,sub_bits", [("0", "12", "0"), ("1", "13", "8"), ("1", "14", None)])
def test_xs_work_items_cut_into_pieces(tile_edges, piece, scale, dtype, weighted, compact, row_bits, sub_bits,
                                       monkeypatch):
    """The XS work-item machinery at sizes a test can hold: tiles cut into up to 15 pieces (each
    with partials of its own), one tile holding every row, tiles of a few rows -- all must give the
    ranks of the C oracle (1e-6 fp32 / 1e-10 fp64 max-abs) in the same number of iterations."""
    import oracle
    from cugraph_amd import pylibcugraph as plc

    monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", "0")
    monkeypatch.setenv("CUGRAPH_AMD_XS_TILE_EDGES", str(tile_edges))
    monkeypatch.setenv("CUGRAPH_AMD_XS_PIECE", str(piece))
    # single-GPU compact source space (slots only for vertices with out-edges) and the wider tiles
    monkeypatch.setenv("CUGRAPH_AMD_COMPACT", compact)
    monkeypatch.setenv("CUGRAPH_AMD_XS_ROWBITS", row_bits)
    # plan build of unweighted graphs: the units sorted whole, or as 256 sub-unit columns each (default 16)
    if sub_bits is not None:
        monkeypatch.setenv("CUGRAPH_AMD_XS_SORT_SUBBITS", sub_bits)
    nv, ne = 1 << scale, 16 << scale
    src, dst = oracle.rmat_edges(scale, ne, seed=7)
    wgt = oracle.rmat_weights(ne, seed=8).astype(dtype) if weighted else None
    g, h = make_sg_graph(src, dst, wgt, vertices=np.arange(nv, dtype=np.int32), multigraph=True,
                         renumber=True)
    eps = nv * 1e-8
    v, r, conv = plc.pagerank(h, g, None, None, None, None, 0.85, eps, 500, False,
                              fail_on_nonconvergence=False)
    o_off, o_idx, perm = oracle.compress(nv, dst, src)
    o_pr, o_it, o_conv = oracle.pagerank(nv, o_off, o_idx, None if wgt is None else wgt[perm],
                                         dtype=np.float64, alpha=0.85, epsilon=eps,
                                         max_iterations=500)
    assert conv and o_conv
    got = np.empty(nv, dtype=np.float64)
    got[v.cpu().numpy()] = r.cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got - o_pr)) <= (1e-6 if dtype == np.float32 else 1e-10)
    top = np.argsort(-o_pr, kind="stable")[:100]
    assert np.max(np.abs(got[top] - o_pr[top]) / o_pr[top]) <= (2e-5 if dtype == np.float32 else 1e-9)


@pytest.mark.parametrize("dead_rows", ["0", "1"])
@pytest.mark.parametrize("dtype,weighted", [(np.float32, False), (np.float64, True)])
def test_dead_rows_are_skipped_correctly(dead_rows, dtype, weighted, monkeypatch):
    """Vertices no edge touches (a third of this graph) are visited by the first iteration only;
    their share of the dangling mass and of the L1 delta is added analytically and their ranks are
    filled in at the end.  With a non-uniform initial guess on exactly those vertices, the ranks,
    the iteration count and the convergence flag must equal the oracle's, with the skip on and off."""
    import oracle
    from cugraph_amd import pylibcugraph as plc

    monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", "0")
    monkeypatch.setenv("CUGRAPH_AMD_COMPACT", "1")
    monkeypatch.setenv("CUGRAPH_AMD_DEAD_ROWS", dead_rows)
    scale = 14
    nv, ne = (1 << scale) + 5000, 16 << scale  # 5000 extra ids only the vertex list mentions
    src, dst = oracle.rmat_edges(scale, ne, seed=11)
    wgt = oracle.rmat_weights(ne, seed=12).astype(dtype) if weighted else None
    g, h = make_sg_graph(src, dst, wgt, vertices=np.arange(nv, dtype=np.int32), multigraph=True, renumber=True)
    rng = np.random.default_rng(3)
    guess = rng.random(nv).astype(dtype)
    guess /= guess.sum()
    eps = 1e-7 if dtype == np.float32 else 1e-12
    o_off, o_idx, perm = oracle.compress(nv, dst, src)
    for max_it in (1, 2, 500):
        v, r, conv = plc.pagerank(h, g, None, None, to_dev(np.arange(nv, dtype=np.int32)), to_dev(guess),
                                  0.85, eps, max_it, False, fail_on_nonconvergence=False)
        o_pr, o_it, o_conv = oracle.pagerank(nv, o_off, o_idx, None if wgt is None else wgt[perm],
                                             dtype=np.float64, alpha=0.85, epsilon=eps,
                                             max_iterations=max_it, nstart=guess.astype(np.float64))
        got = np.empty(nv, dtype=np.float64)
        got[v.cpu().numpy()] = r.cpu().numpy().astype(np.float64)
        assert bool(conv) == o_conv
        assert np.max(np.abs(got - o_pr)) <= (1e-6 if dtype == np.float32 else 1e-10)
        dead = (np.bincount(dst, minlength=nv) == 0) & (np.bincount(src, minlength=nv) == 0)
        assert dead.sum() >= 5000
        assert np.max(np.abs(got[dead] - o_pr[dead]) / o_pr[dead]) <= (1e-5 if dtype == np.float32 else 1e-9)


def _plain_tie_order(deg, transposed):
    """where equal degrees are tied by ascending external id: everywhere, except the rows of
    in-degree 1-3 when a run of the whole suite forces their source ordering on small graphs
    (CUGRAPH_AMD_RENUMBER_LOCALITY=1; by default it starts at 4 M edges)"""
    import os

    if os.environ.get("CUGRAPH_AMD_RENUMBER_LOCALITY") == "1" and transposed:
        return ~((deg >= 1) & (deg <= 3))
    return np.ones(len(deg), bool)


@pytest.mark.parametrize("name", ["karate.csv", "dolphins.csv", "netscience.csv"])
@pytest.mark.parametrize("id_dtype", [np.int32, np.int64])
@pytest.mark.parametrize("transposed", [True, False])
def test_renumber_true_external_ids(name, id_dtype, transposed):
    """renumber=TRUE (cugraph.Graph's default, simpleGraph.py:1134): arbitrary, non-contiguous,
    negative external ids; result vertices are the external ids; ranks equal the oracle's on the
    relabelled graph; the graph itself is stored in descending in-degree order."""
    import oracle
    from cugraph_amd import pylibcugraph as plc

    src, dst, wgt = load_csv(name)
    nv = int(max(src.max(), dst.max())) + 1
    ext = (np.arange(nv, dtype=np.int64) * 7 - 50).astype(id_dtype)  # gaps and negative ids
    g, h = make_sg_graph(ext[src], ext[dst], wgt, store_transposed=transposed, renumber=True,
                         vertices=ext)  # the vertex list keeps ids that no edge mentions
    info = g.info()
    assert info["renumbered"] == 1 and info["num_vertices"] == nv
    off, idx, w, nmap = compressed_arrays(g, h)
    assert nmap.dtype == id_dtype and sorted(nmap.tolist()) == sorted(ext.tolist())
    deg = np.diff(off)
    assert np.all(deg[:-1] >= deg[1:])  # descending major degree
    # ties by ascending external id (stable); among the vertices without major edges those that
    # still occur as a minor come first (the ones no edge touches end up last)
    touched = np.zeros(nv, bool)
    touched[(dst if transposed else src)] = True  # rows are dst when transposed: minors are src ...
    is_minor = np.zeros(nv, bool)
    is_minor[(src if transposed else dst)] = True
    key2 = np.where(deg > 0, 1, is_minor[[int((e + 50) // 7) for e in nmap]].astype(int))
    same = (deg[:-1] == deg[1:]) & (key2[:-1] == key2[1:]) & _plain_tie_order(deg[:-1], transposed)
    assert np.all(nmap[:-1][same] < nmap[1:][same])
    assert np.all(key2[:-1][deg[:-1] == deg[1:]] >= key2[1:][deg[:-1] == deg[1:]])
    # stored structure == oracle CSC of the relabelled edge list
    inv = {int(e): i for i, e in enumerate(nmap)}
    s2 = np.array([inv[int(e)] for e in ext[src]], np.int32)
    d2 = np.array([inv[int(e)] for e in ext[dst]], np.int32)
    major, minor = (d2, s2) if transposed else (s2, d2)
    o_off, o_idx, o_w = oracle.csc_lexsort(nv, minor, major, wgt)
    assert np.array_equal(off.astype(np.int64), o_off) and np.array_equal(idx, o_idx)
    assert np.array_equal(w, o_w)
    # PageRank with a personalization and an initial guess given in EXTERNAL ids
    pv = ext[np.array([3, 1, 7])]
    pvals = np.array([0.2, 0.5, 0.3], np.float32)
    gv = ext[np.arange(nv)]
    gvals = np.full(nv, 1.0 / nv, np.float32)
    v, r = plc.personalized_pagerank(h, g, None, None, to_dev(gv), to_dev(gvals), to_dev(pv),
                                     to_dev(pvals), 0.85, 1e-7, 500, False)
    c_off, c_idx, perm = oracle.compress(nv, dst, src)
    o_pr, _, _ = oracle.pagerank(nv, c_off, c_idx, wgt[perm], dtype=np.float64, alpha=0.85,
                                 epsilon=1e-7, max_iterations=500, nstart=gvals.astype(np.float64),
                                 pers_vertices=np.array([3, 1, 7]), pers_values=pvals)
    v = v.cpu().numpy()
    assert v.dtype == id_dtype and np.array_equal(v, nmap)
    got = np.empty(nv)
    got[[int((e + 50) // 7) for e in v]] = r.cpu().numpy()
    assert np.max(np.abs(got - o_pr)) <= 2e-6


@pytest.mark.parametrize("ids", ["forced", "wide64", "wide64_no_list", "full_int32"])
@pytest.mark.parametrize("transposed", [True, False])
def test_renumber_ids_spanning_more_than_int31(ids, transposed, monkeypatch):
    """External ids whose span does not fit a direct-address table (hashed 64-bit ids; int32 ids from
    one end of the type to the other) are renumbered through the hash set + sorted-id search
    (renumber.hip, renumber_sparse_ids): same numbering rule as the table path (descending major
    degree, ties by ascending external id), number_map and results in external ids, and
    personalization / initial guess given in external ids."""
    import oracle
    from cugraph_amd import pylibcugraph as plc

    src, dst, wgt = load_csv("netscience.csv")
    nv = int(max(src.max(), dst.max())) + 1
    rng = np.random.default_rng(17)
    if ids == "forced":  # the table path's ids, pushed through the other path: identical numbering
        ext = (np.arange(nv, dtype=np.int64) * 7 - 50).astype(np.int32)
        g0, h0 = make_sg_graph(ext[src], ext[dst], wgt, store_transposed=transposed, renumber=True, vertices=ext)
        want_nmap = compressed_arrays(g0, h0)[3]
        monkeypatch.setenv("CUGRAPH_AMD_RENUMBER_SPARSE", "1")
    elif ids == "full_int32":
        ext = np.unique(rng.integers(-2**31, 2**31 - 1, size=4 * nv, dtype=np.int64))[:nv]
        ext = rng.permutation(ext).astype(np.int32)
        ext[0], ext[1] = -2**31, 2**31 - 1
    else:
        ext = np.unique(rng.integers(-2**62, 2**62, size=4 * nv, dtype=np.int64))[:nv]
        ext = rng.permutation(ext)
        ext[0], ext[1] = -2**63 + 1, 2**63 - 1
    no_list = ids == "wide64_no_list"
    g, h = make_sg_graph(ext[src], ext[dst], wgt, store_transposed=transposed, renumber=True,
                         vertices=None if no_list else ext)
    touched = np.zeros(nv, bool)
    touched[src] = True
    touched[dst] = True
    n_expected = int(touched.sum()) if no_list else nv
    assert g.info()["renumbered"] == 1 and g.info()["num_vertices"] == n_expected
    off, idx, w, nmap = compressed_arrays(g, h)
    assert nmap.dtype == ext.dtype
    assert sorted(nmap.tolist()) == sorted((ext[touched] if no_list else ext).tolist())
    if ids == "forced" and _plain_tie_order(np.array([1]), transposed)[0]:
        assert np.array_equal(nmap, want_nmap)  # (the hash-set path always keeps the plain order)
    inv_ext = {int(e): i for i, e in enumerate(ext)}
    orig = np.array([inv_ext[int(e)] for e in nmap])  # original id of each stored row
    deg = np.diff(off)
    assert np.all(deg[:-1] >= deg[1:])
    is_minor = np.zeros(nv, bool)
    is_minor[(src if transposed else dst)] = True
    key2 = np.where(deg > 0, 1, is_minor[orig].astype(int))
    same = (deg[:-1] == deg[1:]) & (key2[:-1] == key2[1:])
    assert np.all(nmap[:-1][same] < nmap[1:][same])
    # stored structure == oracle CSC of the relabelled edge list
    inv = {int(e): i for i, e in enumerate(nmap)}
    s2 = np.array([inv[int(e)] for e in ext[src]], np.int32)
    d2 = np.array([inv[int(e)] for e in ext[dst]], np.int32)
    major, minor = (d2, s2) if transposed else (s2, d2)
    o_off, o_idx, o_w = oracle.csc_lexsort(n_expected, minor, major, wgt)
    assert np.array_equal(off.astype(np.int64), o_off) and np.array_equal(idx, o_idx)
    assert np.array_equal(w, o_w)
    # PageRank, personalization and initial guess in external ids
    pv_orig = np.unique(np.concatenate([src[:50], dst[:50]]))[:3]
    pvals = np.array([0.2, 0.5, 0.3], np.float32)
    gv_orig = np.flatnonzero(touched) if no_list else np.arange(nv)
    gvals = np.full(len(gv_orig), 1.0 / len(gv_orig), np.float32)
    v, r = plc.personalized_pagerank(h, g, None, None, to_dev(ext[gv_orig]), to_dev(gvals), to_dev(ext[pv_orig]),
                                     to_dev(pvals), 0.85, 1e-7, 500, False)
    # the oracle runs on the graph in stored order (vertices no edge mentions do not exist without a list)
    c_off, c_idx, perm = oracle.compress(n_expected, d2, s2)
    pers_rows = np.array([inv[int(e)] for e in ext[pv_orig]])
    # equal personalization ids add up (pagerank.cpp:127-156 routes pairs, the kernel accumulates)
    o_pr, _, _ = oracle.pagerank(n_expected, c_off, c_idx, wgt[perm], dtype=np.float64, alpha=0.85,
                                 epsilon=1e-7, max_iterations=500,
                                 nstart=np.full(n_expected, 1.0 / len(gv_orig)),
                                 pers_vertices=pers_rows, pers_values=pvals)
    v = v.cpu().numpy()
    assert v.dtype == ext.dtype and np.array_equal(v, nmap)
    assert np.max(np.abs(r.cpu().numpy() - o_pr)) <= 2e-6
    # an id that is not a vertex is refused, on both sides of the sorted range
    for bad in (int(ext.max()) - 1, int(ext.min()) + 1):
        if bad in inv:
            continue
        with pytest.raises(ValueError):
            plc.personalized_pagerank(h, g, None, None, None, None, to_dev(np.array([bad], ext.dtype)),
                                      to_dev(np.array([1.0], np.float32)), 0.85, 1e-6, 100, False)
    if not no_list:  # an endpoint the vertex list does not hold
        with pytest.raises(ValueError, match="missing from the vertex list"):
            make_sg_graph(ext[src], ext[dst], wgt, store_transposed=transposed, renumber=True, vertices=ext[:-5])


@pytest.mark.parametrize("weighted", [False, True])
def test_build_through_lds_windows_equals_global_atomics(weighted, monkeypatch):
    """The graph and plan builds count and scatter hub majors / long columns through LDS windows
    (graph_build.hip: hist_lds_kernel, scatter_lds_kernel).  The compressed graph must be the same
    array for array as with global atomics alone, and equal to the oracle's CSC; PageRank through
    the plan built either way agrees with the oracle."""
    import oracle
    from cugraph_amd import pylibcugraph as plc

    monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", "0")
    scale = 16
    nv, ne = 1 << scale, 16 << scale
    src, dst = oracle.rmat_edges(scale, ne, seed=21, scramble=True)
    wgt = oracle.rmat_weights(ne, seed=22).astype(np.float32) if weighted else None
    got = {}
    for mode in ("lds", "global"):
        if mode == "global":
            monkeypatch.setenv("CUGRAPH_AMD_BUILD_NO_LDS", "1")
        g, h = make_sg_graph(src, dst, wgt, multigraph=True, renumber=True, vertices=np.arange(nv, dtype=np.int32))
        off, idx, w, nmap = compressed_arrays(g, h)
        v, r, conv = plc.pagerank(h, g, None, None, None, None, 0.85, 1e-7, 500, False,
                                  fail_on_nonconvergence=False)
        pr = np.empty(nv)
        pr[v.cpu().numpy()] = r.cpu().numpy()
        got[mode] = (off, idx, w, nmap, pr, bool(conv))
    for a, b in zip(got["lds"][:4], got["global"][:4]):
        assert (a is None and b is None) or np.array_equal(a, b)
    off, idx, w, nmap = got["lds"][:4]
    inv = np.empty(nv, np.int32)
    inv[nmap] = np.arange(len(nmap), dtype=np.int32)
    o_off, o_idx, o_w = oracle.csc_lexsort(len(nmap), inv[src], inv[dst], wgt)
    assert np.array_equal(off.astype(np.int64), o_off) and np.array_equal(idx, o_idx)
    if weighted:  # equal (dst, src) pairs keep their input order, weights follow
        assert np.array_equal(w, o_w)
    c_off, c_idx, perm = oracle.compress(nv, dst, src)
    o_pr, _, o_conv = oracle.pagerank(nv, c_off, c_idx, None if wgt is None else wgt[perm], dtype=np.float64,
                                      alpha=0.85, epsilon=1e-7, max_iterations=500)
    for mode in got:
        assert got[mode][5] == o_conv
        assert np.max(np.abs(got[mode][4][:len(o_pr)] - o_pr)) <= 1e-6


@pytest.mark.parametrize("xs_tiles", [False, True])
@pytest.mark.parametrize("dtype,weighted", [(np.float32, False), (np.float64, True)])
def test_sixty_four_bit_offsets(xs_tiles, dtype, weighted, monkeypatch):
    """Graphs of 2^31 - 1 edges or more keep 64-bit offsets (graph_sg.cpp:568-572); that side of the
    build, the plan and every iteration kernel is a separate instantiation which only an 80 GB graph
    would reach.  CUGRAPH_AMD_FORCE_OFF64 takes it on a small graph: same CSC, same ranks, same
    iteration count as the oracle, with and without XS tiles, plus the transpose and a Katz run."""
    import oracle
    from cugraph_amd import pylibcugraph as plc

    monkeypatch.setenv("CUGRAPH_AMD_FORCE_OFF64", "1")
    monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", "0" if xs_tiles else str(1 << 40))
    monkeypatch.setenv("CUGRAPH_AMD_COMPACT", "1" if xs_tiles else "0")
    scale = 14
    nv, ne = 1 << scale, 16 << scale
    src, dst = oracle.rmat_edges(scale, ne, seed=31, scramble=True)
    wgt = oracle.rmat_weights(ne, seed=32).astype(dtype) if weighted else None
    for transposed in (True, False):
        g, h = make_sg_graph(src, dst, wgt, vertices=np.arange(nv, dtype=np.int32), multigraph=True,
                             renumber=True, store_transposed=transposed)
        off, idx, w, nmap = compressed_arrays(g, h)
        assert off.dtype == np.int64
        inv = np.empty(nv, np.int32)
        inv[nmap] = np.arange(nv, dtype=np.int32)
        major, minor = (inv[dst], inv[src]) if transposed else (inv[src], inv[dst])
        o_off, o_idx, o_w = oracle.csc_lexsort(nv, minor, major, wgt)
        assert np.array_equal(off, o_off) and np.array_equal(idx, o_idx)
        if weighted:
            assert np.array_equal(w, o_w)
        pv = np.array([5, 77, 4000], np.int32)
        pvals = np.array([0.3, 0.3, 0.4], dtype if weighted else np.float32)
        eps = 1e-7 if dtype == np.float32 else 1e-11
        # (a graph stored the other way round is transposed in place by the call, pagerank.cpp:105-110)
        v, r, conv = plc.personalized_pagerank(h, g, None, None, None, None, to_dev(pv), to_dev(pvals),
                                               0.85, eps, 500, False, fail_on_nonconvergence=False)
        c_off, c_idx, perm = oracle.compress(nv, dst, src)
        o_pr, o_it, o_conv = oracle.pagerank(nv, c_off, c_idx, None if wgt is None else wgt[perm],
                                             dtype=np.float64, alpha=0.85, epsilon=eps, max_iterations=500,
                                             pers_vertices=pv, pers_values=pvals)
        got = np.empty(nv)
        got[v.cpu().numpy()] = r.cpu().numpy()
        assert bool(conv) == o_conv
        assert np.max(np.abs(got - o_pr)) <= (1e-6 if dtype == np.float32 else 1e-10)
        if transposed:
            from oracle import np_centrality as npc

            w64 = None if wgt is None else wgt.astype(np.float64)
            alpha = 0.5 / np.bincount(dst, weights=w64, minlength=nv).max()
            kv, kr = plc.katz_centrality(h, g, None, alpha, 1.0, 1e-9, 1000, False)
            want, _ = npc.katz_centrality(nv, src, dst, wgt, alpha=alpha, beta=1.0, epsilon=1e-9, max_iterations=1000)
            kz = np.empty(nv)
            kz[kv.cpu().numpy()] = kr.cpu().numpy()
            assert np.max(np.abs(kz - want)) < (2e-5 if dtype == np.float32 else 1e-10)


@pytest.mark.parametrize("weighted", [False, True])
def test_low_degree_rows_ordered_by_their_sources(weighted, monkeypatch):
    """On big graphs the renumbering orders the rows of in-degree 1, 2 and 3 -- inside each class,
    where the order is free -- by their smallest source (renumber.hip), so that the threads of the
    thread-per-row bucket gather from neighbouring slots.  Forced on a small graph here: still a
    descending-degree numbering, the rows of degree >= 4 and of degree 0 keep the plain tie order,
    the stored graph is the oracle's CSC of the relabelled edges, the smallest sources of a class
    ascend (up to the few rows whose sources are low-degree rows themselves and moved with them),
    and PageRank equals the oracle's by vertex id."""
    import oracle
    from cugraph_amd import pylibcugraph as plc

    scale = 14
    nv, ne = 1 << scale, 16 << scale
    src, dst = oracle.rmat_edges(scale, ne, seed=41, scramble=True)
    wgt = oracle.rmat_weights(ne, seed=42).astype(np.float32) if weighted else None
    maps = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CUGRAPH_AMD_RENUMBER_LOCALITY", mode)
        g, h = make_sg_graph(src, dst, wgt, vertices=np.arange(nv, dtype=np.int32), multigraph=True, renumber=True)
        off, idx, w, nmap = compressed_arrays(g, h)
        maps[mode] = nmap
        deg = np.diff(off)
        assert np.all(deg[:-1] >= deg[1:]) and sorted(nmap.tolist()) == list(range(nv))
        inv = np.empty(nv, np.int32)
        inv[nmap] = np.arange(nv, dtype=np.int32)
        o_off, o_idx, o_w = oracle.csc_lexsort(nv, inv[src], inv[dst], wgt)
        assert np.array_equal(off.astype(np.int64), o_off) and np.array_equal(idx, o_idx)
        if weighted:
            assert np.array_equal(w, o_w)
        v, r, conv = plc.pagerank(h, g, None, None, None, None, 0.85, 1e-7, 500, False, fail_on_nonconvergence=False)
        c_off, c_idx, perm = oracle.compress(nv, dst, src)
        o_pr, _, o_conv = oracle.pagerank(nv, c_off, c_idx, None if wgt is None else wgt[perm], dtype=np.float64,
                                          alpha=0.85, epsilon=1e-7, max_iterations=500)
        got = np.empty(nv)
        got[v.cpu().numpy()] = r.cpu().numpy()
        assert bool(conv) == o_conv and np.max(np.abs(got - o_pr)) <= 1e-6
        if mode == "1":
            low = (deg >= 1) & (deg <= 3)
            assert np.array_equal(maps["0"][~low], nmap[~low])  # the other classes are untouched
            assert not np.array_equal(maps["0"][low], nmap[low])
            first_src = np.array([idx[off[r]] for r in np.flatnonzero(low)])  # columns are sorted: the smallest
            for d in (1, 2, 3):
                fs = first_src[deg[low] == d]
                assert len(fs) > 100 and np.mean(fs[:-1] <= fs[1:]) > 0.9
                assert sorted(maps["0"][deg == d].tolist()) == sorted(nmap[deg == d].tolist())


@pytest.mark.parametrize("name", [
    "test_pagerank", "test_pagerank_4", "test_pagerank_non_convergence",
    "test_personalized_pagerank", "test_personalized_pagerank_non_convergence"])
def test_mg_c_api_goldens_over_rccl(golden, name):
    """cpp/tests/c_api/mg_pagerank_test.c:396-560 (the multi-GPU C-API test holds the same vectors as
    pagerank_test.c): through cugraph_amd.mg.pagerank -- HIP engine, RCCL collectives, one rank --
    with the personalization handed over as a per-rank share; the reference's 0.001 relative."""
    import os
    import torch
    import torch.distributed as dist
    from cugraph_amd import mg, pylibcugraph as plc

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        block = golden["c_api"]
        case = block["cases"][name]
        src, dst, wgt = case_arrays(case)
        pers = None
        if "pers_vertices" in case:
            pers = (to_dev(np.array(case["pers_vertices"], np.int32)), to_dev(np.array(case["pers_values"], np.float32)))
        h = plc.ResourceHandle(rank=0, comm_size=1)
        verts, vals, conv = mg.pagerank(h, plc.GraphProperties(is_multigraph=False), to_dev(src), to_dev(dst),
                                        to_dev(wgt), case["num_vertices"], alpha=case["alpha"], tol=case["epsilon"],
                                        max_iter=case["max_iterations"], personalization=pers, sharded=pers is not None,
                                        fail_on_nonconvergence=False)
        assert bool(conv) == (not case["allow_nonconvergence"])
        _expect_close(vals.cpu().numpy(), np.array(case["expected"])[verts.cpu().numpy()], block["rel_tol"])
        if case["allow_nonconvergence"]:
            with pytest.raises(plc.FailedToConvergeError):
                mg.pagerank(h, plc.GraphProperties(is_multigraph=False), to_dev(src), to_dev(dst), to_dev(wgt),
                            case["num_vertices"], alpha=case["alpha"], tol=case["epsilon"],
                            max_iter=case["max_iterations"], personalization=pers)
    finally:
        if own:
            dist.destroy_process_group()


def test_unknown_vertex_ids_rejected():
    from cugraph_amd import pylibcugraph as plc

    src, dst, wgt = load_csv("karate.csv")
    g, h = make_sg_graph(src, dst, wgt)
    with pytest.raises(ValueError):
        plc.personalized_pagerank(h, g, None, None, None, None, to_dev(np.array([99], np.int32)),
                                  to_dev(np.array([1.0], np.float32)), 0.85, 1e-6, 100, False)
    with pytest.raises(ValueError):  # dtype of the values must match the graph's weights
        plc.pagerank(h, g, None, None, to_dev(np.arange(34, dtype=np.int32)),
                     to_dev(np.full(34, 1 / 34, np.float64)), 0.85, 1e-6, 100, False)


@pytest.mark.parametrize("relabel", ["0", "1"])
def test_nstart_and_precomputed_out_weights(relabel, monkeypatch):
    """nstart is used verbatim (pagerank_impl.cuh:435-439), missing vertices start at 0
    (pagerank.cpp:202-211); precomputed out-weight sums replace the computed ones."""
    import oracle
    from cugraph_amd import pylibcugraph as plc

    monkeypatch.setenv("CUGRAPH_AMD_NO_RELABEL", relabel)
    src, dst, wgt = load_csv("dolphins.csv")
    nv = 62
    g, h = make_sg_graph(src, dst, wgt)
    rng = np.random.default_rng(3)
    gv = rng.permutation(nv)[:40].astype(np.int32)
    gvals = rng.random(40).astype(np.float32)
    nstart = np.zeros(nv)
    nstart[gv] = gvals
    c_off, c_idx, perm = oracle.compress(nv, dst, src)
    out_w = np.bincount(src, weights=wgt.astype(np.float64), minlength=nv)
    fake_w = (out_w * 2.0).astype(np.float32)  # any consistent scaling must pass through
    v, r, conv = plc.pagerank(h, g, to_dev(np.arange(nv, dtype=np.int32)), to_dev(fake_w),
                              to_dev(gv), to_dev(gvals), 0.85, 1e-7, 3, False,
                              fail_on_nonconvergence=False)
    o_pr, o_it, o_conv = oracle.pagerank(nv, c_off, c_idx, wgt[perm], dtype=np.float64, alpha=0.85,
                                         epsilon=1e-7, max_iterations=3, nstart=nstart,
                                         out_weight_sums=fake_w.astype(np.float64))
    assert conv == o_conv
    assert np.array_equal(v.cpu().numpy(), np.arange(nv))
    assert np.max(np.abs(r.cpu().numpy() - o_pr)) <= 1e-6


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("xs_tiles", [False, True])
def test_partitioned_sessions_emulated_ranks(world, weighted, xs_tiles, monkeypatch):
    """The HIP side of the multi-GPU path for every rank of a 2- and 3-way partition, all run in
    this one process: per-rank degree relabel (cugraph_amd_partition_degree_order), local graph
    with vertex_first != 0, XS tiles on a partition, and the exchange buffers; the all-gather and
    the scalar all-reduce are done by hand between the sessions.  (The collectives themselves are
    covered by tests/test_mg_cpu.py over gloo and by test_mg_driver_on_one_gpu over RCCL.)"""
    import torch
    import oracle
    from cugraph_amd import mg, pylibcugraph as plc

    monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", "0" if xs_tiles else str(1 << 40))
    scale = 12
    nv, ne = (1 << scale) - 5, 16 << scale  # V not a multiple of the world size
    src, dst = oracle.rmat_edges(scale, ne, seed=11, scramble=True)
    keep = (src < nv) & (dst < nv)
    src, dst = src[keep], dst[keep]
    wgt = oracle.rmat_weights(len(src), seed=12).astype(np.float32) if weighted else None
    props = plc.GraphProperties(is_multigraph=True)
    chunk = (nv + world - 1) // world
    engines, orders, ranks_l, bounds = [], [], [], []
    counts = torch.zeros(nv, dtype=torch.float64, device="cuda")
    sums = torch.zeros(nv, dtype=torch.float64, device="cuda")
    for r in range(world):
        first, last, _ = mg.partition_bounds(nv, world, r)
        eng = mg.HipEngine(plc.ResourceHandle(rank=r, comm_size=world), props)
        sel = (dst >= first) & (dst < last)
        order, rank_l = eng.degree_order(to_dev(dst[sel]), first, last)
        deg = np.bincount(dst[sel] - first, minlength=last - first)
        assert np.array_equal(order.cpu().numpy(), np.argsort(-deg, kind="stable"))
        counts += eng.source_weight_sums(to_dev(src[sel]), None, nv)
        sums += eng.source_weight_sums(to_dev(src[sel]), None if wgt is None else to_dev(wgt[sel]), nv)
        engines.append(eng), orders.append(order), ranks_l.append(rank_l), bounds.append((first, last, sel))
    assert np.array_equal(counts.cpu().numpy(), np.bincount(src, minlength=nv))
    # among a rank's rows without in-edges the ones some edge leaves come first, as the driver orders
    # them (mg.py): the vertices no edge touches are then one run at the end of every rank, which the
    # sessions skip after the first iteration and account for analytically (dead rows)
    for r in range(world):
        first, last, sel = bounds[r]
        n_with = int(np.count_nonzero(np.bincount(dst[sel] - first, minlength=last - first)))
        tail = orders[r][n_with:]
        keep = torch.argsort((counts[first:last][tail.long()] == 0).to(torch.int8), stable=True)
        orders[r] = torch.cat([orders[r][:n_with], tail[keep]])
        ranks_l[r] = torch.empty_like(ranks_l[r])
        ranks_l[r][orders[r].long()] = torch.arange(last - first, dtype=torch.int32, device="cuda")
        assert int((counts[first:last][orders[r][n_with:].long()] == 0).sum()) > 0  # there are such vertices
    # compact source space: vertices with out-edges, numbered per rank in row order
    actives = [counts[b[0]:b[1]][o.long()] > 0 for b, o in zip(bounds, orders)]
    slot_chunk = max(int(a.sum()) for a in actives)
    slot_of = torch.full((nv,), -1, dtype=torch.int32, device="cuda")
    x_poss = []
    for r in range(world):
        a = actives[r]
        x_pos = torch.where(a, torch.cumsum(a, 0) - 1 + r * slot_chunk,
                            torch.full_like(orders[r].long(), -1)).to(torch.int32)
        slot_of[bounds[r][0] + orders[r].long()] = x_pos
        x_poss.append(x_pos)
    graphs = []
    for r in range(world):
        first, last, sel = bounds[r]
        g = engines[r].make_local_graph(
            slot_of[to_dev(src[sel]).long()], ranks_l[r][(to_dev(dst[sel]) - first).long()] + first,
            None if wgt is None else to_dev(wgt[sel]), first, last, nv,
            num_source_slots=slot_chunk * world, x_pos=x_poss[r])
        graphs.append(g)
    eps = nv * 1e-8
    sessions = []
    for r in range(world):
        first, last, _ = bounds[r]
        ids = torch.arange(first, last, dtype=torch.int32, device="cuda")
        sessions.append(engines[r].make_session(
            graphs[r], alpha=0.85, epsilon=eps, precomputed_vertex_out_weight_vertices=ids,
            precomputed_vertex_out_weight_sums=sums[first:last][orders[r].long()].to(torch.float32)))

    def exchange():
        bufs = [s.exchange_buffers() for s in sessions]
        x = torch.zeros_like(bufs[0][0])
        scal = torch.zeros(2, dtype=torch.float64, device="cuda")
        for r, (xf, sc) in enumerate(bufs):
            assert xf.numel() == slot_chunk * world
            x[r * slot_chunk:(r + 1) * slot_chunk] = xf[r * slot_chunk:(r + 1) * slot_chunk]
            scal += sc[2:4]  # this rank's own {dangling mass, L1 delta}
        for xf, sc in bufs:
            xf.copy_(x), sc[0:2].copy_(scal)  # what the next iteration's kernels read

    exchange()
    for _ in range(500):
        for s in sessions:
            s.iterate(1)
        exchange()
        st = [s.status() for s in sessions]
        assert len({(a[0], a[1]) for a in st}) == 1  # every rank agrees on iteration and stop
        if st[0][1]:
            break
    full = np.zeros(nv)
    for r, s in enumerate(sessions):
        first, last, _ = bounds[r]
        verts, vals, iters, conv = s.result(500)
        assert np.array_equal(verts.cpu().numpy(), np.arange(first, last))
        full[first + orders[r].cpu().numpy()] = vals.cpu().numpy()
    off, idx, perm = oracle.compress(nv, dst, src)
    o_pr, o_it, _ = oracle.pagerank(nv, off, idx, None if wgt is None else wgt[perm], dtype=np.float64,
                                    alpha=0.85, epsilon=eps, max_iterations=500)
    assert iters == o_it and conv
    assert np.max(np.abs(full - o_pr)) <= 1e-6


@pytest.mark.parametrize("xs_tiles", [False, True])
def test_mg_driver_on_one_gpu(xs_tiles, monkeypatch):
    """The multi-GPU driver with its HIP engine and RCCL collectives, one rank: 1D-partitioned
    local graph, out-weight all-reduce, in-place all-gather of x, 2-scalar all-reduce; nstart and
    personalization in the caller's ids."""
    import os
    monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", "0" if xs_tiles else str(1 << 40))
    import torch
    import torch.distributed as dist
    import oracle
    from cugraph_amd import mg, pylibcugraph as plc

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        scale = 13
        nv, ne = 1 << scale, 16 << scale
        h = plc.ResourceHandle(rank=0, comm_size=1)
        drv = mg.MGPagerank.from_rmat(h, plc.GraphProperties(is_multigraph=True), scale, ne,
                                      seed=42, scramble_vertex_ids=True)
        drv.always_exchange = True
        eps = nv * 1e-8
        sess = drv.session(alpha=0.85, epsilon=eps)
        iters, stopped, _ = drv.run_to_convergence(sess, 500)
        full, it2, conv = drv.gathered_result(sess, 500)
        src, dst = oracle.rmat_edges(scale, ne, seed=42, scramble=True)
        off, idx, _ = oracle.compress(nv, dst, src)
        o_pr, o_it, _ = oracle.pagerank(nv, off, idx, None, dtype=np.float64, alpha=0.85,
                                        epsilon=eps, max_iterations=500)
        assert stopped and conv and iters == o_it == it2
        assert np.max(np.abs(full.cpu().numpy() - o_pr)) <= 1e-6
        verts, vals, _, _ = drv.result(sess, 500)
        assert np.array_equal(verts.cpu().numpy(), np.arange(nv))
        assert np.array_equal(vals.cpu().numpy(), full.cpu().numpy())
        # personalization (un-normalised) and a partial initial guess, ids as the caller knows them
        rng = np.random.default_rng(5)
        pers = (rng.permutation(nv)[:50].astype(np.int32), (rng.random(50) + 0.1).astype(np.float32))
        nstart = (rng.permutation(nv)[:3000].astype(np.int32), (rng.random(3000) / 1500).astype(np.float32))
        sess = drv.session(alpha=0.85, epsilon=1e-30, nstart=nstart, personalization=pers)
        drv.run_to_convergence(sess, 5, check_every=2)
        full, it2, conv = drv.gathered_result(sess, 5)
        ns = np.zeros(nv)
        ns[nstart[0]] = nstart[1]
        o_pr, o_it, o_conv = oracle.pagerank(nv, off, idx, None, dtype=np.float64, alpha=0.85,
                                             epsilon=1e-30, max_iterations=5, nstart=ns,
                                             pers_vertices=pers[0], pers_values=pers[1])
        assert it2 == o_it == 5 and not conv and not o_conv
        assert np.max(np.abs(full.cpu().numpy() - o_pr)) <= 1e-6
    finally:
        if own:
            dist.destroy_process_group()
