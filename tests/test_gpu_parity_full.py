"""GPU parity at BASELINE.json's full sizes.

configs[1] (R-MAT scale 22, fp32, unweighted) and configs[2] (scale 24, fp64, weighted, 1k-vertex
personalization) are small enough for the CPU oracle: ranks within north_star's max-abs bounds
(1e-6 fp32 / 1e-10 fp64) at tol = 1e-8 and the same top-1000 ranking.  configs[3]'s graph (scale 26)
is checked through size-independent properties: the CSC against checksums of the edge list, mass
conservation, the fixed point, and agreement of the two independent SpMV paths (XS tiles vs
degree buckets)."""
import os

import numpy as np
import pytest

from helpers import compressed_tensors, to_dev

HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.gpu


def _rmat_graph(scale, *, weighted=False, dtype=None, renumber=False, seed=42):
    import torch
    from cugraph_amd import generators, pylibcugraph as plc

    h = plc.ResourceHandle()
    nv, ne = 1 << scale, 16 << scale
    src, dst = generators.rmat(h, scale, ne, seed=seed, scramble_vertex_ids=True)
    wgt = None
    if weighted:
        wgt = generators.rmat_weights(h, ne, seed=seed + 1, dtype=dtype)
    g = plc.SGGraph(h, plc.GraphProperties(is_multigraph=True), src, dst, weight_array=wgt,
                    store_transposed=True, renumber=renumber,
                    vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
    return h, g, src, dst, wgt


def _top_k_agree(got, want, k, tol):
    """Same top-k ranking; two vertices may swap only where the oracle's values differ by < tol."""
    g = np.argsort(-got, kind="stable")[:k]
    w = np.argsort(-want, kind="stable")[:k]
    diff = g != w
    assert np.all(np.abs(want[g[diff]] - want[w[diff]]) < tol), "top-k ranking differs beyond ties"


def _relative_bounds(got, want, nv, rel_all, rel_top):
    """Relative error, in the style of the reference's own check (|a - b| < max(a, b) * tol,
    cpp/tests/link_analysis/pagerank_test.cpp:338-344): on every vertex whose rank is at least the mean
    rank 1/V (below that the max-abs bound is the tighter one) and on the top 1000."""
    err = np.abs(got - want)
    big = want >= 1.0 / nv
    assert float(np.max(err[big] / want[big])) <= rel_all
    top = np.argpartition(-want, 1000)[:1000]
    assert float(np.max(err[top] / want[top])) <= rel_top


def test_config1_scale22_fp32_unweighted_vs_oracle():
    import oracle
    from cugraph_amd import pylibcugraph as plc

    scale = 22
    nv = 1 << scale
    h, g, src, dst, _ = _rmat_graph(scale)
    del src, dst
    off, idx, _, _ = compressed_tensors(g)
    off, idx = off.cpu().numpy().astype(np.int64), idx.cpu().numpy()
    eps = nv * 1e-8
    v, r, conv = plc.pagerank(h, g, None, None, None, None, 0.85, eps, 500, False,
                              fail_on_nonconvergence=False)
    o_pr, o_it, o_conv = oracle.pagerank(nv, off, idx, None, dtype=np.float64, alpha=0.85,
                                         epsilon=eps, max_iterations=500)
    r = r.cpu().numpy().astype(np.float64)
    assert conv and o_conv
    assert np.max(np.abs(r - o_pr)) <= 1e-6  # north_star: fp32 max-abs
    _top_k_agree(r, o_pr, 1000, 1e-7)
    _relative_bounds(r, o_pr, nv, 1e-4, 2e-5)
    # a tighter epsilon, so that the comparison is not about where each side stopped
    v, r, conv = plc.pagerank(h, g, None, None, None, None, 0.85, 1e-5, 500, False,
                              fail_on_nonconvergence=False)
    o_pr, o_it, o_conv = oracle.pagerank(nv, off, idx, None, dtype=np.float64, alpha=0.85,
                                         epsilon=1e-5, max_iterations=500)
    r = r.cpu().numpy().astype(np.float64)
    assert conv and o_conv
    assert np.max(np.abs(r - o_pr)) <= 1e-6
    _top_k_agree(r, o_pr, 1000, 1e-7)
    _relative_bounds(r, o_pr, nv, 1e-4, 2e-5)


def test_config2_scale24_fp64_weighted_personalized_vs_oracle():
    import torch
    import oracle
    from cugraph_amd import pylibcugraph as plc

    scale = 24
    nv = 1 << scale
    h, g, src, dst, wgt = _rmat_graph(scale, weighted=True, dtype=torch.float64)
    del src, dst, wgt
    off, idx, w, _ = compressed_tensors(g)
    off, idx, w = off.cpu().numpy().astype(np.int64), idx.cpu().numpy(), w.cpu().numpy()
    rng = np.random.default_rng(24)
    pv = rng.permutation(nv)[:1000].astype(np.int32)
    px = rng.random(1000) + 0.05
    eps = nv * 1e-8
    for e in (eps, 1e-4):
        v, r = plc.personalized_pagerank(h, g, None, None, None, None, to_dev(pv), to_dev(px), 0.85,
                                         e, 500, False)
        o_pr, o_it, o_conv = oracle.pagerank(nv, off, idx, w, dtype=np.float64, alpha=0.85, epsilon=e,
                                             max_iterations=500, pers_vertices=pv, pers_values=px)
        r = r.cpu().numpy()
        assert r.dtype == np.float64 and o_conv
        assert np.max(np.abs(r - o_pr)) <= 1e-10  # north_star: fp64 max-abs
        _top_k_agree(r, o_pr, 1000, 1e-12)
        _relative_bounds(r, o_pr, nv, 1e-9, 1e-10)
        assert abs(r.sum() - 1.0) < 1e-9


@pytest.mark.parametrize("renumber", [False, True])
def test_config3_scale26_build_properties(renumber):
    """COO -> CSC at 1.07 G edges: degree histogram, column order, and order-independent
    checksums of the edge multiset (through number_map when renumbered)."""
    import torch

    scale = 26
    nv, ne = 1 << scale, 16 << scale
    h, g, src, dst, _ = _rmat_graph(scale, renumber=renumber)
    off, idx, _, nmap = compressed_tensors(g)
    assert off.numel() == nv + 1 and idx.numel() == ne
    assert int(off[0]) == 0 and int(off[-1]) == ne
    deg = off[1:] - off[:-1]
    assert int(deg.min()) >= 0
    if renumber:
        assert bool((deg[:-1] >= deg[1:]).all())  # descending in-degree
        assert nmap is not None and bool((torch.sort(nmap).values == torch.arange(nv, device="cuda", dtype=nmap.dtype)).all())
        ext_of = nmap.long()
        want_deg = torch.bincount(dst, minlength=nv)[ext_of]
    else:
        assert nmap is None
        want_deg = torch.bincount(dst, minlength=nv)
    assert bool((deg.long() == want_deg).all())
    del want_deg
    # sources ascend inside every column (by internal id)
    starts = torch.zeros(ne + 1, dtype=torch.bool, device="cuda")
    starts[off.long()] = True
    assert not bool(((idx[1:] < idx[:-1]) & ~starts[1:ne]).any())
    del starts
    # checksum of the edge multiset: sum over edges of a mixed hash of (src, dst), mod 2^64
    def checksum(s, d):
        x = s.long() * 0x9E3779B1 + d.long() * 0x85EBCA77
        x = (x ^ (x >> 15)) * 0x2C1B3C6D
        return int(x.sum()), int((x ^ (x >> 13)).sum())

    want = checksum(src, dst)
    del src, dst
    rows = torch.repeat_interleave(torch.arange(nv, device="cuda", dtype=torch.int32), deg.long())
    if renumber:
        got = checksum(nmap[idx.long()], nmap[rows.long()])
    else:
        got = checksum(idx, rows)
    assert got == want


def test_config3_scale26_pagerank_properties(monkeypatch):
    """Full-size run: mass conservation, monotone L1 delta, fixed point, and the XS-tile path
    against the degree-bucket path (independent kernels, same input) within the fp32 bound."""
    import torch
    from cugraph_amd import pylibcugraph as plc
    from cugraph_amd.session import PagerankSession

    scale = 26
    nv = 1 << scale
    results = {}
    for path, xs in (("xs", "1"), ("buckets", "0")):
        monkeypatch.setenv("CUGRAPH_AMD_XS", xs)
        h2, g2, s2, d2, _ = _rmat_graph(scale, renumber=True)  # the plan is cached per graph
        del s2, d2
        sess = PagerankSession(h2, g2, alpha=0.85, epsilon=0.0)
        deltas = []
        for _ in range(12):
            sess.iterate(1)
            deltas.append(sess.status()[2])
        assert all(b < a for a, b in zip(deltas, deltas[1:]))  # the power iteration contracts
        assert deltas[-1] < deltas[0] * 0.85 ** 8
        v, r, it, _ = sess.result(100)
        assert it == 12
        full = torch.empty(nv, dtype=torch.float64, device="cuda")
        full[v.long()] = r.double()
        results[path] = (full, deltas)
        assert abs(float(full.sum()) - 1.0) < 1e-4  # dangling mass is redistributed, none lost
        assert float(full.min()) > 0.0
        del sess, g2
    a, b = results["xs"][0], results["buckets"][0]
    assert float((a - b).abs().max()) <= 1e-6
    assert np.allclose(results["xs"][1], results["buckets"][1], rtol=1e-5)


def test_config3_scale26_fp32_vs_oracle():
    """The headline configuration against the CPU oracle itself (OpenMP C restatement of the
    reference's pagerank_reference, a few seconds per run at 1.07 G edges): absolute epsilon 1e-6,
    about 21 iterations; same iteration count, north_star's max-abs bound, top-1000 ranking and
    relative bounds."""
    import oracle
    from cugraph_amd import pylibcugraph as plc

    scale = 26
    nv = 1 << scale
    h, g, src, dst, _ = _rmat_graph(scale)  # renumber=False: the CSC is in the caller's ids
    del src, dst
    off, idx, _, _ = compressed_tensors(g)
    off, idx = off.cpu().numpy().astype(np.int64), idx.cpu().numpy()
    eps = 1e-6
    v, r, conv = plc.pagerank(h, g, None, None, None, None, 0.85, eps, 500, False, fail_on_nonconvergence=False)
    assert conv and bool((v[:1000].cpu() == __import__("torch").arange(1000)).all())
    r = r.cpu().numpy().astype(np.float64)
    o_pr, o_it, o_conv = oracle.pagerank(nv, off, idx, None, dtype=np.float64, alpha=0.85, epsilon=eps,
                                         max_iterations=500)
    assert o_conv and 15 <= o_it <= 30
    # the same number of iterations: one session run to the same epsilon
    from cugraph_amd.session import PagerankSession
    sess = PagerankSession(h, g, alpha=0.85, epsilon=eps)
    it, stopped = 0, False
    while not stopped and it < 500:
        sess.iterate(1)
        it, stopped, _, _ = sess.status()
    assert it == o_it
    assert np.max(np.abs(r - o_pr)) <= 1e-6
    _top_k_agree(r, o_pr, 1000, 1e-7)
    _relative_bounds(r, o_pr, nv, 1e-4, 2e-5)
    assert abs(r.sum() - 1.0) < 1e-4


@pytest.mark.parametrize("regions", ["0", "1"])
def test_config3_scale26_one_rank_of_eight(regions, monkeypatch):
    """configs[3]'s partitioned path at full size: the local graph of ONE rank of an 8-way partition
    of the scale-26 graph (per-rank degree order, compact source slots of all ranks, with and
    without the two regions), fed the gathered vector and scalars of iteration 3 of the single-GPU
    run, must produce this rank's slice of iteration 4 -- ranks, local dangling mass and local L1
    delta -- and hold as many edges as have their destination in its range."""
    import torch
    from cugraph_amd import pylibcugraph as plc
    from cugraph_amd.session import PagerankSession
    from rank_emulation import build_rank

    monkeypatch.setenv("CUGRAPH_AMD_MG_REGIONS", regions)
    scale, world, me = 26, 8, 5
    nv = 1 << scale
    h, g, src, dst, _ = _rmat_graph(scale)
    ref = PagerankSession(h, g, alpha=0.85, epsilon=0.0)
    ref.iterate(3)
    v3, pr3, it3, _ = ref.result(100)
    dangling3 = ref.status()[3]
    ref.iterate(1)
    v4, pr4, it4, _ = ref.result(100)
    assert (it3, it4) == (3, 4) and bool((v3[:100].cpu() == torch.arange(100)).all())
    del ref, g, v3, v4
    torch.cuda.empty_cache()

    R = build_rank(plc.ResourceHandle(rank=me, comm_size=world), plc.GraphProperties(is_multigraph=True),
                   src, dst, nv, world, me)
    first, last, order = R["first"], R["last"], R["order"]
    assert (R["regions"] is not None) == (regions == "1")
    assert R["n_local_edges"] == int(((dst >= first) & (dst < last)).sum())
    del src, dst
    outdeg = R["outdeg"]
    ids = torch.arange(first, last, dtype=torch.int32, device="cuda")
    sess = R["engine"].make_session(
        R["graph"], alpha=0.85, epsilon=0.0, precomputed_vertex_out_weight_vertices=ids,
        precomputed_vertex_out_weight_sums=outdeg[first:last][order].to(torch.float32),
        initial_guess_vertices=ids, initial_guess_values=pr3[first:last][order].contiguous())
    info = sess.info()
    assert info["edges"] == R["n_local_edges"] and info["mid_degree_vertices"] > 0  # XS tiles on this rank
    # the exchange, by hand: every slot's source value of iteration 3, and its global dangling mass
    x_full, scal = sess.exchange_buffers()
    assert x_full.numel() == R["n_slots"]
    x3 = pr3 / torch.clamp(outdeg, min=1).to(pr3.dtype)
    has = R["slot_of"] >= 0
    mine = R["x_pos"][R["x_pos"] >= 0].long()
    own = x_full[mine].clone()
    x_full[R["slot_of"][has].long()] = x3[has]
    assert bool((x_full[mine] == own).all())  # what the rank wrote itself is what the others would send
    assert abs(float(scal[2]) - float(pr3[first:last][outdeg[first:last] == 0].double().sum())) < 1e-9
    scal[0:2].copy_(torch.tensor([dangling3, 1e300], dtype=torch.float64))
    if regions == "1":
        sess.iterate_phase(0), sess.iterate_phase(1)
    else:
        sess.iterate(1)
    verts, vals, iters, _ = sess.result(100)
    assert iters == 1 and bool((verts.cpu() == torch.arange(first, last)).all())
    got = torch.empty_like(vals)
    got[order] = vals
    want = pr4[first:last]
    err = (got.double() - want.double()).abs()
    assert float(err.max()) <= 1e-8
    assert float((err / want.double()).max()) <= 1e-5
    _, scal = sess.exchange_buffers()
    dang = outdeg[first:last] == 0
    assert abs(float(scal[2]) - float(want[dang].double().sum())) <= 1e-7
    assert abs(float(scal[3]) - float((want.double() - pr3[first:last].double()).abs().sum())) <= 1e-6


def test_config4_scale28_build_and_three_iterations():
    """configs[4]'s graph on ONE GPU (4.29 G edges: int64 offsets, 52 source windows per XS unit):
    the COO -> CSC build against checksums of the edge list, then three iterations -- mass
    conservation, a contracting L1 delta, and the first iteration of a sample of rows recomputed
    from the CSC in plain torch fp64."""
    import torch
    from cugraph_amd import _capi, generators, pylibcugraph as plc
    from cugraph_amd.session import PagerankSession
    import ctypes

    scale = 28
    nv, ne = 1 << scale, 16 << scale
    h = plc.ResourceHandle()
    src, dst = generators.rmat(h, scale, ne, seed=42, scramble_vertex_ids=True)

    def checksum(s, d):
        x = s.long() * 0x9E3779B1 + d.long() * 0x85EBCA77
        x = (x ^ (x >> 15)) * 0x2C1B3C6D
        return int(x.sum()), int((x ^ (x >> 13)).sum())

    def add(a, b):
        return tuple((p + q) & 0xFFFFFFFFFFFFFFFF for p, q in zip(a, b))

    step = 1 << 28
    want, indeg, outdeg = (0, 0), torch.zeros(nv, dtype=torch.int32, device="cuda"), torch.zeros(nv, dtype=torch.int32, device="cuda")
    one = torch.ones(step, dtype=torch.int32, device="cuda")
    for b in range(0, ne, step):
        want = add(want, checksum(src[b:b + step], dst[b:b + step]))
        indeg.index_add_(0, dst[b:b + step].long(), one)
        outdeg.index_add_(0, src[b:b + step].long(), one)
    del one
    g = plc.SGGraph(h, plc.GraphProperties(is_multigraph=True), src, dst, store_transposed=True, renumber=False,
                    vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
    del src, dst
    torch.cuda.empty_cache()
    info = (ctypes.c_int64 * 8)()
    err = ctypes.c_void_p()
    _capi.check(_capi.load().cugraph_amd_graph_info(g.c_graph_ptr, info, ctypes.byref(err)), err, "graph_info")
    assert info[0] == nv and info[1] == ne and info[5] == 1  # offsets are int64 from 2^31 - 1 edges
    off, idx, _, nmap = compressed_tensors(g)
    assert off.dtype == torch.int64 and nmap is None and int(off[0]) == 0 and int(off[-1]) == ne
    assert bool(((off[1:] - off[:-1]) == indeg.long()).all())
    got = (0, 0)
    for b in range(0, nv, 1 << 24):  # rows in pieces of 16 M: their edges are one slice of indices
        e = min(b + (1 << 24), nv)
        lo, hi = int(off[b]), int(off[e])
        rows = torch.repeat_interleave(torch.arange(b, e, device="cuda", dtype=torch.int32), indeg[b:e].long())
        seg = idx[lo:hi]
        got = add(got, checksum(seg, rows))
        first_of_row = torch.zeros(hi - lo + 1, dtype=torch.bool, device="cuda")
        first_of_row[(off[b:e] - lo)] = True
        assert not bool(((seg[1:] < seg[:-1]) & ~first_of_row[1:hi - lo]).any())  # sources ascend in a column
        del rows, seg, first_of_row
    assert got == want
    # ---- three iterations
    sess = PagerankSession(h, g, alpha=0.85, epsilon=0.0)
    sinfo = sess.info()
    assert sinfo["edges"] == ne and sinfo["mid_degree_vertices"] > 0
    sess.iterate(1)
    v1, r1, it1, _ = sess.result(100)
    d1 = sess.status()[2]
    sess.iterate(2)
    v3, r3, it3, _ = sess.result(100)
    d3 = sess.status()[2]
    assert (it1, it3) == (1, 3) and d3 < d1 * 0.85
    assert abs(float(r3.double().sum()) - 1.0) < 1e-4 and float(r3.min()) > 0.0
    # iteration 1 of a sample of rows, from the CSC: pr1[v] = (alpha * d0 + 1 - alpha) / V + alpha * sum x0[u]
    gen = torch.Generator(device="cpu").manual_seed(3)
    hubs = torch.topk(indeg, 8).indices.cpu()
    sample = torch.cat([hubs, torch.randint(0, nv, (2000,), generator=gen)]).cuda()
    d0 = float((outdeg == 0).sum()) / nv
    base = (0.85 * d0 + 0.15) / nv
    x0 = (1.0 / nv) / torch.clamp(outdeg, min=1).double()
    want1 = torch.empty(sample.numel(), dtype=torch.float64, device="cuda")
    for i, vtx in enumerate(sample.tolist()):
        want1[i] = base + 0.85 * float(x0[idx[int(off[vtx]):int(off[vtx + 1])].long()].sum())
    got1 = r1[sample].double()
    assert float(((got1 - want1).abs() / want1).max()) <= 1e-5


def test_two_regions_of_the_gathered_vector(monkeypatch):
    """Multi-GPU layout with TWO regions of the gathered vector (hot slots first) at a scale where
    the library really cuts its work items at the region boundary (scale 24: four windows of 2 M
    slots): (a) the driver over RCCL with one rank, phases 0 / 1 and the two all-gathers, against
    the single-GPU run of the same graph; (b) every rank of a 2-way partition in one process,
    exchange by hand region by region, against the same."""
    import os
    import torch
    import torch.distributed as dist
    from cugraph_amd import generators, mg, pylibcugraph as plc

    monkeypatch.setenv("CUGRAPH_AMD_MG_REGIONS", "1")  # by default only from 4 ranks
    scale = 24
    nv, ne = 1 << scale, 16 << scale
    eps = 1e-6
    h0, g0, src, dst, _ = _rmat_graph(scale)
    v, r, conv = plc.pagerank(h0, g0, None, None, None, None, 0.85, eps, 100, False, fail_on_nonconvergence=False)
    want = r.cpu().numpy().astype(np.float64)
    assert conv and np.array_equal(v.cpu().numpy(), np.arange(nv))
    del g0
    # ---- (a) the driver, one rank, collectives on
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29519")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        h = plc.ResourceHandle(rank=0, comm_size=1)
        drv = mg.MGPagerank.from_edges(mg.HipEngine(h, plc.GraphProperties(is_multigraph=True)), src.clone(),
                                       dst.clone(), None, nv, rank=0, world=1)
        assert drv.regions is not None and drv.regions[0] == mg.HipEngine(h, None).window_slots(sum(drv.regions))
        drv.always_exchange = True
        sess = drv.session(alpha=0.85, epsilon=eps)
        iters, stopped, _ = drv.run_to_convergence(sess, 100, check_every=3)
        full, it2, conv2 = drv.gathered_result(sess, 100)
        assert stopped and conv2
        assert np.max(np.abs(full.cpu().numpy().astype(np.float64) - want)) <= 1e-7
        del sess, drv
    finally:
        if own:
            dist.destroy_process_group()
    # ---- (b) two ranks emulated in this process
    world = 2
    indeg = torch.bincount(dst.long(), minlength=nv)
    outdeg = torch.bincount(src.long(), minlength=nv)
    bounds = [mg.partition_bounds(nv, world, r) for r in range(world)]
    engines = [mg.HipEngine(plc.ResourceHandle(rank=r, comm_size=world), plc.GraphProperties(is_multigraph=True))
               for r in range(world)]
    orders, actives = [], []
    for r in range(world):
        first, last, _ = bounds[r]
        order, _ = engines[r].degree_order(dst[(dst >= first) & (dst < last)], first, last)
        orders.append(order.long())
        actives.append(outdeg[first:last][orders[r]] > 0)
    regions = mg.MGPagerank._choose_regions(engines[0], max(int(a.sum()) for a in actives), world)
    assert regions is not None
    c0, c1 = regions
    n_slots, hot = world * (c0 + c1), world * c0
    slot_of = torch.full((nv,), -1, dtype=torch.int32, device="cuda")
    x_poss = []
    for r in range(world):
        j = torch.cumsum(actives[r], 0) - 1
        slot = torch.where(j < c0, j + r * c0, j - c0 + hot + r * c1)
        x_pos = torch.where(actives[r], slot, torch.full_like(slot, -1)).to(torch.int32)
        slot_of[bounds[r][0] + orders[r]] = x_pos
        x_poss.append(x_pos)
    sessions = []
    graphs = []
    for r in range(world):
        first, last, _ = bounds[r]
        sel = (dst >= first) & (dst < last)
        rank_l = torch.empty(last - first, dtype=torch.int32, device="cuda")
        rank_l[orders[r]] = torch.arange(last - first, dtype=torch.int32, device="cuda")
        g = engines[r].make_local_graph(slot_of[src[sel].long()], rank_l[(dst[sel] - first).long()] + first, None,
                                        first, last, nv, num_source_slots=n_slots, x_pos=x_poss[r], hot_slots=hot)
        graphs.append(g)
        ids = torch.arange(first, last, dtype=torch.int32, device="cuda")
        sessions.append(engines[r].make_session(
            g, alpha=0.85, epsilon=eps, precomputed_vertex_out_weight_vertices=ids,
            precomputed_vertex_out_weight_sums=outdeg[first:last][orders[r]].to(torch.float32)))

    def exchange(region):
        bufs = [s.exchange_buffers() for s in sessions]
        lo, c = (0, c0) if region == 0 else (hot, c1)
        for r in range(world):
            piece = bufs[r][0][lo + r * c:lo + (r + 1) * c]
            for q in range(world):
                if q != r:
                    bufs[q][0][lo + r * c:lo + (r + 1) * c] = piece
        if region == 1:
            scal = sum(b[1][2:4] for b in bufs)  # every rank's own {dangling mass, L1 delta} ...
            for b in bufs:
                b[1][0:2].copy_(scal)  # ... summed into what the next iteration reads

    exchange(0), exchange(1)
    for _ in range(100):
        # region 0 is complete, region 1 still "in flight": phase 0 must not need it
        for s in sessions:
            s.iterate_phase(0)
        for s in sessions:
            s.iterate_phase(1)
        exchange(0), exchange(1)
        st = [s.status() for s in sessions]
        assert len({(a[0], a[1]) for a in st}) == 1
        if st[0][1]:
            break
    got = np.zeros(nv)
    for r, s in enumerate(sessions):
        first, last, _ = bounds[r]
        verts, vals, iters_b, conv_b = s.result(100)
        got[first + orders[r].cpu().numpy()] = vals.cpu().numpy()
    assert conv_b and iters_b == iters
    assert np.max(np.abs(got - want)) <= 1e-7
    # the timing-aid build (make AIDS=1, CUGRAPH_AMD_LIBRARY) counts the gathers of region-0 launches
    # that read a slot of region 1 -- whose all-gather may still be running then: there must be none
    import ctypes
    from cugraph_amd import _capi

    lib = _capi.load()
    if hasattr(lib, "cugraph_amd_aids_region_check"):
        out = (ctypes.c_ulonglong * 2)()
        assert lib.cugraph_amd_aids_region_check(out) == 0
        print(f"region check: {out[0]} gathers beyond region 0 in {out[1]} region-0 launches")
        assert out[1] > 0 and out[0] == 0


def test_phase_0_reads_only_the_hot_region():
    """The test above under the timing-aid library, whose xs_kernel counts every gather of a
    region-0 launch that lands at or beyond hot_slots (the advisor's debug assertion: phase 0 runs
    while region 1 is still being gathered)."""
    import os
    import subprocess
    import sys

    aids = os.path.join(os.path.dirname(HERE), "cugraph_amd", "libcugraph_amd_c_aids.so")
    if not os.path.exists(aids):
        pytest.skip("timing-aid library not built (make -C cugraph_amd/csrc AIDS=1)")
    env = dict(os.environ, CUGRAPH_AMD_LIBRARY=aids)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", "--timeout", "300", "--timeout-method",
                        "thread", f"{os.path.abspath(__file__)}::test_two_regions_of_the_gathered_vector"],
                       capture_output=True, text=True, timeout=400, env=env, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "region check: 0 gathers beyond region 0" in r.stdout


def test_personalised_exchange_over_rccl_with_one_rank(monkeypatch):
    """The default exchange from 2 ranks on (private slots + one all_to_all_single into library-owned memory
    behind the own sources) forced onto the driver with ONE rank over RCCL, collectives on: the foreign
    block is empty, so the all-to-all moves nothing -- what this covers is what one GPU can cover of that
    path with the real backend (the tensors that alias the session's vector, split lists, stream order);
    results against the single-GPU run.  Scale 22: XS tiles, xs2_kernel."""
    import os
    import torch
    import torch.distributed as dist
    from cugraph_amd import mg, pylibcugraph as plc

    monkeypatch.setenv("CUGRAPH_AMD_MG_EXCHANGE", "personal")
    scale = 22
    nv = 1 << scale
    eps = 1e-6
    h0, g0, src, dst, _ = _rmat_graph(scale)
    v, r, conv = plc.pagerank(h0, g0, None, None, None, None, 0.85, eps, 100, False, fail_on_nonconvergence=False)
    want = r.cpu().numpy().astype(np.float64)
    assert conv
    del g0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29521")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        h = plc.ResourceHandle(rank=0, comm_size=1)
        drv = mg.MGPagerank.from_edges(mg.HipEngine(h, plc.GraphProperties(is_multigraph=True)), src.clone(),
                                       dst.clone(), None, nv, rank=0, world=1)
        assert drv.idx_give is not None and drv.n_foreign == 0 and drv.n_own > 0
        drv.always_exchange = True
        sess = drv.session(alpha=0.85, epsilon=eps)
        iters, stopped, _ = drv.run_to_convergence(sess, 100, check_every=3)
        full, it2, conv2 = drv.gathered_result(sess, 100)
        assert stopped and conv2
        assert np.max(np.abs(full.cpu().numpy().astype(np.float64) - want)) <= 1e-7
    finally:
        if own:
            dist.destroy_process_group()
