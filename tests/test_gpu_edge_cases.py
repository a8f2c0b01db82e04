"""Edge cases of the PageRank path through the C ABI, against the CPU oracle: degenerate graphs
(no edges, one vertex, only self-loops, a star, every vertex dangling, zero weights), the limits
of alpha and max_iterations (pagerank_impl.cuh:89-99, 328-337), personalization on vertices
without edges, and empty optional arrays."""
import numpy as np
import pytest

from helpers import make_sg_graph, to_dev

pytestmark = pytest.mark.gpu


def _run(src, dst, wgt, nv, *, alpha=0.85, eps=1e-10, max_it=200, pers=None, renumber=False, dtype=np.float64):
    import oracle
    from cugraph_amd import pylibcugraph as plc

    src = np.asarray(src, np.int32)
    dst = np.asarray(dst, np.int32)
    w = None if wgt is None else np.asarray(wgt, dtype)
    g, h = make_sg_graph(src, dst, w, vertices=np.arange(nv, dtype=np.int32), multigraph=True, renumber=renumber)
    if pers is None:
        v, r, conv = plc.pagerank(h, g, None, None, None, None, alpha, eps, max_it, False,
                                  fail_on_nonconvergence=False)
    else:
        v, r, conv = plc.personalized_pagerank(h, g, None, None, None, None, to_dev(np.asarray(pers[0], np.int32)),
                                               to_dev(np.asarray(pers[1], dtype if w is not None else np.float32)),
                                               alpha, eps, max_it, False, fail_on_nonconvergence=False)
    off, idx, perm = oracle.compress(nv, dst, src)
    kw = {}
    if pers is not None:
        kw = dict(pers_vertices=pers[0], pers_values=pers[1])
    o_pr, o_it, o_conv = oracle.pagerank(nv, off, idx, None if w is None else w[perm],
                                         dtype=np.float64, alpha=alpha, epsilon=eps, max_iterations=max_it, **kw)
    got = np.empty(nv, dtype=np.float64)
    got[v.cpu().numpy()] = r.cpu().numpy().astype(np.float64)
    return got, bool(conv), o_pr, o_conv


@pytest.mark.parametrize("renumber", [False, True])
@pytest.mark.parametrize("nv", [1, 5, 1000])
def test_graph_without_edges(nv, renumber):
    got, conv, ref, rconv = _run([], [], None, nv, renumber=renumber)
    assert conv == rconv
    assert np.allclose(got, 1.0 / nv, rtol=0, atol=1e-7)
    assert np.max(np.abs(got - ref)) <= 1e-7


@pytest.mark.parametrize("renumber", [False, True])
def test_only_self_loops_and_one_vertex(renumber):
    got, conv, ref, rconv = _run([0], [0], None, 1, renumber=renumber)
    assert conv == rconv and abs(got[0] - ref[0]) <= 1e-7 and abs(got[0] - 1.0) <= 1e-6
    got, conv, ref, rconv = _run([0, 1, 2, 2], [0, 1, 2, 2], None, 4, renumber=renumber)
    assert conv == rconv
    assert np.max(np.abs(got - ref)) <= 1e-7


@pytest.mark.parametrize("renumber", [False, True])
@pytest.mark.parametrize("n_leaves", [3, 70, 5000, 40000])
def test_star_into_one_hub(n_leaves, renumber):
    """every leaf points at vertex 0 (a single column longer than the hub thresholds at 40000),
    which points back at leaf 1; the other vertices dangle"""
    src = list(range(1, n_leaves + 1)) + [0]
    dst = [0] * n_leaves + [1]
    # unweighted graphs compute in fp32 (graph_sg.cpp:247): an epsilon fp32 can reach
    got, conv, ref, rconv = _run(src, dst, None, n_leaves + 3, renumber=renumber, eps=1e-6)
    assert conv and rconv
    assert np.max(np.abs(got - ref)) <= 1e-6
    assert abs(got.sum() - 1.0) <= 1e-4


def test_zero_weights_make_a_vertex_dangle():
    src, dst = [0, 0, 1, 2, 3], [1, 2, 2, 0, 0]
    wgt = [0.0, 0.0, 1.0, 2.0, 0.5]  # vertex 0 has out-edges of weight 0: out-weight sum 0 -> dangling
    got, conv, ref, rconv = _run(src, dst, wgt, 4)
    assert conv == rconv
    assert np.max(np.abs(got - ref)) <= 1e-10


@pytest.mark.parametrize("alpha", [0.0, 0.5, 1.0])
def test_alpha_limits(alpha):
    from helpers import load_csv

    src, dst, w = load_csv("karate.csv")
    got, conv, ref, rconv = _run(src, dst, w.astype(np.float64), 34, alpha=alpha, eps=1e-9, max_it=300)
    assert conv == rconv
    assert np.max(np.abs(got - ref)) <= 1e-9
    if alpha == 0.0:
        assert np.allclose(got, 1.0 / 34, atol=1e-12)


@pytest.mark.parametrize("max_it", [0, 1, 2])
def test_iteration_limits(max_it):
    """at least one iteration always runs and converged = iterations < max_iterations
    (pagerank_impl.cuh:328-337)"""
    from helpers import load_csv

    src, dst, w = load_csv("karate.csv")
    got, conv, ref, rconv = _run(src, dst, w.astype(np.float64), 34, eps=1e-12, max_it=max_it)
    assert conv is False and rconv is False
    assert np.max(np.abs(got - ref)) <= 1e-12


def test_personalization_on_vertices_without_edges():
    """personalization mass on an isolated vertex and on a dangling one; values not normalised"""
    src, dst = [0, 1, 2, 2, 4], [1, 2, 0, 3, 2]  # 3 dangles, 5 and 6 are isolated
    pers = ([5, 3, 0], [2.0, 0.5, 0.25])
    got, conv, ref, rconv = _run(src, dst, [1.0, 2.0, 3.0, 1.0, 1.0], 7, pers=pers)
    assert conv == rconv
    assert np.max(np.abs(got - ref)) <= 1e-10
    assert got[6] <= 1e-12 and got[5] > 0.1


def test_empty_personalization_is_rejected():
    from cugraph_amd import pylibcugraph as plc

    g, h = make_sg_graph(np.array([0, 1], np.int32), np.array([1, 0], np.int32), None,
                         vertices=np.arange(2, dtype=np.int32))
    with pytest.raises((ValueError, RuntimeError)):
        plc.personalized_pagerank(h, g, None, None, None, None, to_dev(np.zeros(0, np.int32)),
                                  to_dev(np.zeros(0, np.float32)), 0.85, 1e-6, 100, False)


def _build(src, dst, *, symmetric=False, multigraph=False, renumber=False, vertices=None, check=True,
           transposed=True, input_type="COO", id_dtype=np.int32):
    from cugraph_amd import pylibcugraph as plc

    h = plc.ResourceHandle()
    props = plc.GraphProperties(is_symmetric=symmetric, is_multigraph=multigraph)
    return plc.SGGraph(h, props, to_dev(np.asarray(src, id_dtype)), to_dev(np.asarray(dst, id_dtype)),
                       store_transposed=transposed, renumber=renumber, do_expensive_check=check,
                       input_array_format=input_type,
                       vertices_array=None if vertices is None else to_dev(np.asarray(vertices, id_dtype)))


@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("renumber", [False, True])
def test_expensive_check_symmetry(renumber, transposed):
    """create_graph_from_edgelist_impl.cuh:902-910: a graph flagged symmetric whose edge list is not
    is refused under do_expensive_check (and only then)."""
    sym_s, sym_d = [0, 1, 1, 2, 3], [1, 0, 2, 1, 3]  # a self-loop is its own reverse
    _build(sym_s, sym_d, symmetric=True, renumber=renumber, transposed=transposed)
    with pytest.raises(RuntimeError, match="is_symmetric is true but the input edge list is not symmetric"):
        _build(sym_s + [0], sym_d + [3], symmetric=True, renumber=renumber, transposed=transposed)
    # every edge has a reverse, but not as many copies of it
    with pytest.raises(RuntimeError, match="not symmetric"):
        _build([0, 0, 1], [1, 1, 0], symmetric=True, multigraph=True, renumber=renumber, transposed=transposed)
    _build([0, 0, 1, 1], [1, 1, 0, 0], symmetric=True, multigraph=True, renumber=renumber, transposed=transposed)
    _build(sym_s + [0], sym_d + [3], symmetric=True, renumber=renumber, transposed=transposed, check=False)
    _build(sym_s + [0], sym_d + [3], symmetric=False, renumber=renumber, transposed=transposed)


@pytest.mark.parametrize("renumber", [False, True])
def test_expensive_check_parallel_edges(renumber):
    """create_graph_from_edgelist_impl.cuh:912-920"""
    rng = np.random.default_rng(5)
    n = 3000
    pairs = np.unique(rng.integers(0, n, size=(40000, 2)), axis=0)
    rng.shuffle(pairs)
    src, dst = pairs[:, 0], pairs[:, 1]
    _build(src, dst, renumber=renumber)
    s2, d2 = np.append(src, src[1234]), np.append(dst, dst[1234])
    with pytest.raises(RuntimeError, match="is_multigraph is false but the input edge list has parallel edges"):
        _build(s2, d2, renumber=renumber)
    _build(s2, d2, multigraph=True, renumber=renumber)
    _build(s2, d2, renumber=renumber, check=False)
    # symmetry is tested before parallel edges (:902 before :912)
    with pytest.raises(RuntimeError, match="not symmetric"):
        _build([0, 0, 1], [1, 1, 0], symmetric=True, renumber=renumber)


@pytest.mark.parametrize("id_dtype", [np.int32, np.int64])
def test_expensive_check_vertex_list(id_dtype):
    """create_graph_from_edgelist_impl.cuh:79-108"""
    src, dst = [0, 1, 2], [1, 2, 0]
    _build(src, dst, vertices=[0, 1, 2, 3], id_dtype=id_dtype)
    _build([10, 30], [30, 50], vertices=[50, 30, 10, 70], renumber=True, id_dtype=id_dtype)
    for renumber in (False, True):
        with pytest.raises(RuntimeError, match="vertices should not have duplicates"):
            _build(src, dst, vertices=[0, 1, 2, 2], renumber=renumber, id_dtype=id_dtype)
        with pytest.raises(RuntimeError, match="vertices should not have duplicates"):
            _build(src, dst, vertices=[0, 1, 2, 900, 900], renumber=renumber, id_dtype=id_dtype)
    if id_dtype == np.int64:  # ids too far apart for a table: found by the renumbering's distinct count
        big = 1 << 60
        _build([-big, 5], [5, big], vertices=[-big, 5, big], renumber=True, id_dtype=id_dtype)
        with pytest.raises(RuntimeError, match="vertices should not have duplicates"):
            _build([-big, 5], [5, big], vertices=[-big, 5, big, 5], renumber=True, id_dtype=id_dtype)
        with pytest.raises(RuntimeError, match="consecutive integers starting from 0"):
            _build([0, 1], [1, 0], vertices=[0, 1, big], id_dtype=id_dtype)
    with pytest.raises(RuntimeError, match="consecutive integers starting from 0"):
        _build(src, dst, vertices=[0, 1, 2, 4], id_dtype=id_dtype)
    with pytest.raises(RuntimeError, match="consecutive integers starting from 0"):
        _build([1, 2], [2, 3], vertices=[1, 2, 3], id_dtype=id_dtype)
    with pytest.raises(RuntimeError, match=r"vertex IDs should be in \[0, "):
        _build(src, dst, vertices=[-1, 0, 1, 2], id_dtype=id_dtype)


def test_expensive_check_csr_input():
    """graph_sg.cpp:262-430: the CSR path hands do_expensive_check to the same build"""
    off, idx = [0, 2, 3, 3], [1, 1, 2]  # row 0 holds 0->1 twice
    with pytest.raises(RuntimeError, match="parallel edges"):
        _build(off, idx, input_type="CSR", transposed=False)
    _build(off, idx, input_type="CSR", transposed=False, multigraph=True)
    _build(off, idx, input_type="CSR", transposed=False, check=False)


def test_arguments_the_algorithm_rejects_are_unknown_errors_with_the_reference_messages():
    """cugraph::pagerank throws on these and run_algorithm maps every exception to CUGRAPH_UNKNOWN_ERROR
    (cpp/src/c_api/utils.hpp:53-55), which pylibcugraph raises as RuntimeError (utils.pyx:37-83); the texts are
    pagerank_impl.cuh:89-99, 223-225.  CUGRAPH_INVALID_INPUT (ValueError) stays for the C layer's own type
    checks (pagerank.cpp:268-299)."""
    from cugraph_amd import pylibcugraph as plc

    g, h = make_sg_graph(np.array([0, 1, 2], np.int32), np.array([1, 2, 0], np.int32), np.ones(3, np.float32),
                         vertices=np.arange(3, dtype=np.int32))
    with pytest.raises(RuntimeError, match=r"alpha should be in \[0.0, 1.0\]"):
        plc.pagerank(h, g, None, None, None, None, 1.5, 1e-6, 100, False)
    with pytest.raises(RuntimeError, match="epsilon should be non-negative"):
        plc.pagerank(h, g, None, None, None, None, 0.85, -1e-6, 100, False)
    pv = to_dev(np.array([0, 1], np.int32))
    with pytest.raises(RuntimeError, match="sum of personalization valuese should be positive"):
        plc.personalized_pagerank(h, g, None, None, None, None, pv, to_dev(np.zeros(2, np.float32)), 0.85, 1e-6,
                                  100, False)
    with pytest.raises(RuntimeError, match="personalization vector size should not be 0"):
        plc.personalized_pagerank(h, g, None, None, None, None, to_dev(np.zeros(0, np.int32)),
                                  to_dev(np.zeros(0, np.float32)), 0.85, 1e-6, 100, False)
    with pytest.raises(ValueError):  # a value array of the wrong type: the C layer's check
        plc.personalized_pagerank(h, g, None, None, None, None, pv, to_dev(np.ones(2, np.float64)), 0.85, 1e-6,
                                  100, False)


def test_do_expensive_check_of_the_pagerank_call():
    """pagerank_impl.cuh:101-187: negative edge weights, negative out-weight sums, negative or repeated
    personalization entries -- only looked for under do_expensive_check."""
    from cugraph_amd import pylibcugraph as plc

    src, dst = np.array([0, 1, 2, 2], np.int32), np.array([1, 2, 0, 1], np.int32)
    g_neg, h = make_sg_graph(src, dst, np.array([1, -2, 1, 1], np.float32), vertices=np.arange(3, dtype=np.int32))
    with pytest.raises(RuntimeError, match="input edge weights should have non-negative values"):
        plc.pagerank(h, g_neg, None, None, None, None, 0.85, 1e-6, 100, True)
    plc.pagerank(h, g_neg, None, None, None, None, 0.85, 1e-6, 5, False, fail_on_nonconvergence=False)  # unchecked
    g, h = make_sg_graph(src, dst, np.ones(4, np.float32), vertices=np.arange(3, dtype=np.int32), handle=h)
    ids = to_dev(np.arange(3, dtype=np.int32))
    with pytest.raises(RuntimeError, match="outgoing edge weight sum values should be non-negative"):
        plc.pagerank(h, g, ids, to_dev(np.array([1, -1, 2], np.float32)), None, None, 0.85, 1e-6, 100, True)
    rep_v, rep_x = to_dev(np.array([0, 2, 0], np.int32)), to_dev(np.array([0.2, 0.3, 0.5], np.float32))
    with pytest.raises(RuntimeError, match="personalization vertices not contain duplicate entries"):
        plc.personalized_pagerank(h, g, None, None, None, None, rep_v, rep_x, 0.85, 1e-6, 100, True)
    # unchecked, repeated entries add up (the accumulate scatter): the same as {0: 0.7, 2: 0.3}
    _, a = plc.personalized_pagerank(h, g, None, None, None, None, rep_v, rep_x, 0.85, 1e-7, 200, False)
    _, b = plc.personalized_pagerank(h, g, None, None, None, None, to_dev(np.array([0, 2], np.int32)),
                                     to_dev(np.array([0.7, 0.3], np.float32)), 0.85, 1e-7, 200, False)
    assert np.max(np.abs(a.cpu().numpy() - b.cpu().numpy())) < 1e-6
    with pytest.raises(RuntimeError, match="peresonalization values should be non-negative"):
        plc.personalized_pagerank(h, g, None, None, None, None, to_dev(np.array([0, 1], np.int32)),
                                  to_dev(np.array([1.0, -0.25], np.float32)), 0.85, 1e-6, 100, True)


def test_multi_gpu_c_entry_points_reject_what_they_do_not_do():
    """cugraph_mg_graph_create / cugraph_graph_create_mg (cugraph_amd/csrc/mg_capi.hip): the documented
    differences from the reference are errors, not silent behaviour."""
    import ctypes

    import torch

    from cugraph_amd import _capi, pylibcugraph as plc
    from cugraph_amd._capi import GraphPropertiesStruct
    from cugraph_amd.pylibcugraph.utils import create_view_from_py_obj, free_view

    lib = _capi.load()
    props = GraphPropertiesStruct(False, True)
    src = torch.tensor([0, 1, 2, 2], dtype=torch.int32, device="cuda")
    dst = torch.tensor([1, 2, 0, 1], dtype=torch.int32, device="cuda")
    v_src, v_dst = create_view_from_py_obj(src), create_view_from_py_obj(dst)

    def create(handle, s=v_src, d=v_dst, drop=0):
        graph, err = ctypes.c_void_p(), ctypes.c_void_p()
        arr = lambda v: (ctypes.c_void_p * 1)(v)  # noqa: E731
        code = lib.cugraph_graph_create_mg(handle.c_resource_handle_ptr, ctypes.byref(props), None, arr(s), arr(d), None,
                                           None, None, 1, 1, drop, 0, 0, ctypes.byref(graph), ctypes.byref(err))
        _capi.check(code, err, "cugraph_graph_create_mg")
        return graph

    # more than one rank needs the collectives of the handle
    with pytest.raises(Exception, match="cugraph_amd_resource_handle_set_comm"):
        create(plc.ResourceHandle(rank=0, comm_size=2))
    h = plc.ResourceHandle(rank=0, comm_size=1)
    with pytest.raises(Exception, match="drop_self_loops / drop_multi_edges"):
        create(h, drop=1)
    neg = torch.tensor([0, -1, 2, 2], dtype=torch.int32, device="cuda")
    v_neg = create_view_from_py_obj(neg)
    with pytest.raises(Exception, match="non-negative"):
        create(h, s=v_neg)
    # a good graph, then: precomputed sums are not taken, a vertex outside the graph is refused
    graph = create(h)
    ids = torch.arange(3, dtype=torch.int32, device="cuda")
    ones = torch.ones(3, dtype=torch.float32, device="cuda")
    v_ids, v_ones = create_view_from_py_obj(ids), create_view_from_py_obj(ones)
    result, err = ctypes.c_void_p(), ctypes.c_void_p()
    code = lib.cugraph_pagerank(h.c_resource_handle_ptr, graph, v_ids, v_ones, None, None, 0.85, 1e-6, 20, 0,
                                ctypes.byref(result), ctypes.byref(err))
    with pytest.raises(Exception, match="precomputed out-weight sums on a multi-GPU graph"):
        _capi.check(code, err, "cugraph_pagerank")
    far = torch.tensor([7], dtype=torch.int32, device="cuda")
    one = torch.ones(1, dtype=torch.float32, device="cuda")
    v_far, v_one = create_view_from_py_obj(far), create_view_from_py_obj(one)
    code = lib.cugraph_personalized_pagerank(h.c_resource_handle_ptr, graph, None, None, None, None, v_far, v_one, 0.85,
                                             1e-6, 20, 0, ctypes.byref(result), ctypes.byref(err))
    with pytest.raises(Exception, match="not in the graph"):
        _capi.check(code, err, "cugraph_personalized_pagerank")
    # and the plain call works: 3-cycle with a chord, ranks sum to one
    code = lib.cugraph_pagerank(h.c_resource_handle_ptr, graph, None, None, None, None, 0.85, 1e-9, 200, 0,
                                ctypes.byref(result), ctypes.byref(err))
    _capi.check(code, err, "cugraph_pagerank")
    x_view = lib.cugraph_centrality_result_get_values(result)
    vals = np.empty(3, dtype=np.float32)
    code = lib.cugraph_type_erased_device_array_view_copy_to_host(h.c_resource_handle_ptr, vals.ctypes.data, x_view,
                                                                  ctypes.byref(err))
    _capi.check(code, err, "copy_to_host")
    assert abs(float(vals.sum()) - 1.0) < 1e-5
    lib.cugraph_type_erased_device_array_view_free(x_view)
    lib.cugraph_centrality_result_free(result)
    lib.cugraph_mg_graph_free(graph)
    for v in (v_src, v_dst, v_neg, v_ids, v_ones, v_far, v_one):
        free_view(v)
