"""world_size-2 (and 3) runs of the multi-GPU driver on CPU over gloo.

The local compute is the numpy test double (tests/mg_test_double.py); what is under test is
cugraph_amd/mg.py: 1D partition bounds, routing of edges to the owner of their destination, the
all-reduce of out-weight sums, the per-iteration all-gather + 2-scalar all-reduce, the stop rule
(same iteration count on every rank, equal to the single-process oracle) and result assembly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_dir):
    import torch.distributed as dist

    from cugraph_amd import mg
    from mg_test_double import OracleEngine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CUGRAPH_AMD_MG_REGIONS"] = case.get("regions", "0")  # two regions of the gathered vector
    # rows of in-degree 1-3 in the order of their smallest source slot (default: big graphs only)
    os.environ["CUGRAPH_AMD_RENUMBER_LOCALITY"] = case.get("locality", "0")
    os.environ["CUGRAPH_AMD_MG_PARTITION"] = case.get("partition", "edges")
    os.environ["CUGRAPH_AMD_MG_EXCHANGE"] = case.get("exchange", "gather")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src, dst, wgt, nv = case["src"], case["dst"], case["wgt"], case["nv"]
        # every rank starts with an arbitrary share of the edge list (round-robin here)
        sel = np.arange(len(src)) % world == rank
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a[sel]))  # noqa: E731
        drv = mg.MGPagerank.from_edges(OracleEngine(rank, world), t(src), t(dst), t(wgt), nv,
                                       rank=rank, world=world)
        # contiguous ranges that cover the vertices, the same on every rank; equal ones on request
        assert drv.bounds[0] == 0 and drv.bounds[-1] == nv and len(drv.bounds) == world + 1
        assert all(a <= b for a, b in zip(drv.bounds, drv.bounds[1:]))
        first, last = drv.first, drv.last
        assert (first, last) == (drv.bounds[rank], drv.bounds[rank + 1])
        if case.get("partition") == "equal":
            assert (first, last) == mg.partition_bounds(nv, world, rank)[:2]
        if "want_regions" in case:
            assert (drv.regions is not None) == case["want_regions"]
        if case.get("exchange") == "personal":
            # a slot numbering of this rank's own: its sources, then exactly the foreign sources its edges read
            assert drv.idx_give is not None and drv.regions is None
            assert drv.graph.slots == max(drv.n_own + drv.n_foreign, 1)
            used = np.unique(drv.graph.src)
            assert np.array_equal(used[used >= drv.n_own], np.arange(drv.n_own, drv.n_own + drv.n_foreign))
            assert sum(drv.want_split) == drv.n_foreign and drv.want_split[rank] == 0 and drv.give_split[rank] == 0
        assert drv.graph.first == first and len(drv.out_w) == last - first
        nstart, pers = case.get("nstart"), case.get("pers")
        if case.get("sharded"):
            # every rank holds an arbitrary share of each list (one of them none of the
            # personalization at all), not the pairs of the vertices it owns
            share = lambda pairs, r: None if pairs is None else tuple(a[r::world][::-1].copy() for a in pairs)  # noqa: E731
            nstart, pers = share(nstart, (rank + 1) % world), share(pers, rank)
            if rank == world - 1 and pers is not None:
                pers = tuple(np.concatenate([a[r::world] for r in (rank, 0)]) for a in case["pers"])
            if rank == 0 and pers is not None:
                pers = (pers[0][:0], pers[1][:0])
        sess = drv.session(alpha=case.get("alpha", 0.85), epsilon=case["eps"], nstart=nstart, personalization=pers,
                           sharded=bool(case.get("sharded")))
        iters, stopped, delta = drv.run_to_convergence(sess, case["max_iter"], check_every=case["check"])
        dangling = sess.inner.status()[3]
        full, it2, conv = drv.gathered_result(sess, case["max_iter"])
        verts, vals, _, _ = drv.result(sess, case["max_iter"])
        assert verts.tolist() == list(range(first, last)) and len(vals) == last - first
        np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
        np.save(os.path.join(out_dir, f"meta{rank}.npy"),
                np.array([iters, it2, int(stopped), int(conv), delta, dangling, len(drv.graph.dst), first, last],
                         dtype=np.float64))
    finally:
        dist.destroy_process_group()


def _run(world, case, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    fulls = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    metas = [np.load(tmp_path / f"meta{r}.npy") for r in range(world)]
    return fulls, metas


def _oracle(case):
    import oracle

    nv = case["nv"]
    off, idx, perm = oracle.compress(nv, case["dst"], case["src"])
    w = None if case["wgt"] is None else case["wgt"][perm]
    kw = {}
    if case.get("nstart") is not None:
        ns = np.zeros(nv)
        ns[case["nstart"][0]] = case["nstart"][1]
        kw["nstart"] = ns
    if case.get("pers") is not None:
        kw["pers_vertices"], kw["pers_values"] = case["pers"]
    return oracle.pagerank(nv, off, idx, w, dtype=np.float64, alpha=0.85, epsilon=case["eps"],
                           max_iterations=case["max_iter"], **kw)


def _rmat_case(scale, weighted, **extra):
    import oracle

    nv, ne = 1 << scale, 8 << scale
    src, dst = oracle.rmat_edges(scale, ne, seed=7, scramble=True)
    wgt = oracle.rmat_weights(ne, seed=8) if weighted else None
    case = dict(src=src.astype(np.int64), dst=dst.astype(np.int64), wgt=wgt, nv=nv, eps=nv * 1e-9,
                max_iter=200, check=3)
    case.update(extra)
    return case


@pytest.mark.parametrize("world,regions,want_regions", [(2, "auto", False), (2, "1", True), (3, "1", True),
                                                        (4, "auto", True), (4, "0", False), (4, None, False)])
@pytest.mark.parametrize("weighted", [False, True])
def test_mg_matches_single_process_oracle(world, regions, want_regions, weighted, tmp_path):
    """regions: the gathered vector as two regions (hot slots first), two all-gathers per iteration and
    the iteration in two phases -- opt-in ("1", or "auto" = from 4 ranks; the stand-in's window is 48
    slots); off when nothing is said (None)"""
    case = _rmat_case(9, weighted, want_regions=want_regions, locality="1" if world >= 3 else "0")
    if regions is not None:
        case["regions"] = regions
    fulls, metas = _run(world, case, tmp_path)
    o_pr, o_it, o_conv = _oracle(case)
    for f, m in zip(fulls, metas):
        assert m[0] == o_it and m[1] == o_it  # same iteration on every rank as the oracle
        assert bool(m[2]) and bool(m[3]) == o_conv
        assert np.max(np.abs(f - o_pr)) < 1e-12
    assert abs(fulls[0].sum() - 1.0) < 1e-9


@pytest.mark.parametrize("world,locality", [(2, "0"), (3, "1"), (5, "1")])
@pytest.mark.parametrize("weighted", [False, True])
def test_mg_personalised_exchange(world, locality, weighted, tmp_path):
    """CUGRAPH_AMD_MG_EXCHANGE=personal: every rank numbers its slots privately (own sources, then
    per owner exactly the sources its edges read) and the per-iteration exchange is one all-to-all
    of what each reader gathers instead of an all-gather of everything: same ranks, same iteration
    count, personalization and an initial guess included."""
    rng = np.random.default_rng(3)
    nv = 1 << 9
    pers = (rng.permutation(nv)[:23].astype(np.int64), rng.random(23) + 0.1)
    nstart = (rng.permutation(nv)[:200].astype(np.int64), rng.random(200))
    case = _rmat_case(9, weighted, exchange="personal", locality=locality, pers=pers, nstart=nstart,
                      eps=1e-12, check=4)
    fulls, metas = _run(world, case, tmp_path)
    o_pr, o_it, o_conv = _oracle(case)
    for f, m in zip(fulls, metas):
        assert m[0] == o_it and m[1] == o_it and bool(m[2]) and bool(m[3]) == o_conv
        # (an unweighted graph takes its personalization and initial guess as fp32 at the API)
        assert np.max(np.abs(f - o_pr)) < (1e-12 if weighted else 1e-8)


@pytest.mark.parametrize("sharded", [False, True])
def test_mg_personalization_nstart_and_nonconvergence(sharded, tmp_path):
    """sharded: the (vertex, value) lists arrive as per-rank shares in any order, as the reference's
    MG entry takes them (mg_pagerank_test.c:67-94), instead of replicated on every rank"""
    rng = np.random.default_rng(0)
    nv = 1 << 9
    pers = (rng.permutation(nv)[:37].astype(np.int64), rng.random(37) + 0.1)  # un-normalised
    nstart = (rng.permutation(nv)[:300].astype(np.int64), rng.random(300))
    case = _rmat_case(9, True, pers=pers, nstart=nstart, max_iter=4, eps=1e-30, check=1, regions="1",
                      want_regions=True, sharded=sharded, locality="1")
    fulls, metas = _run(3 if sharded else 2, case, tmp_path)
    o_pr, o_it, o_conv = _oracle(case)
    assert o_it == 4 and not o_conv
    for f, m in zip(fulls, metas):
        assert m[0] == 4 and not bool(m[3])
        assert np.max(np.abs(f - o_pr)) < 1e-12


@pytest.mark.parametrize("world", [2, 4])
def test_mg_scalars_after_a_stop_inside_a_batch(world, tmp_path):
    """The stop lands inside a batch of 7 iterations: the skipped iterations of the batch still run
    the collectives, which must leave the reported L1 delta and dangling mass what the converging
    iteration made them (they were once multiplied by the world size per skipped iteration)."""
    case = _rmat_case(9, False, check=7)
    fulls, metas = _run(world, case, tmp_path)
    o_pr, o_it, o_conv = _oracle(case)
    assert o_it % 7 not in (0, 6), "pick another batch size: the stop must land inside a batch"
    dang = float(o_pr[np.bincount(case["src"], minlength=case["nv"]) == 0].sum())
    for f, m in zip(fulls, metas):
        assert m[0] == o_it and bool(m[2])
        assert 0.0 < m[4] < case["eps"]
        assert abs(m[5] - dang) < 1e-9
        assert np.max(np.abs(f - o_pr)) < 1e-12


@pytest.mark.parametrize("world", [2, 4, 8])
def test_edge_balanced_partition_of_unscrambled_rmat(world, tmp_path):
    """Natural R-MAT ids put the hubs at the low ids: equal vertex ranges would give rank 0 most of
    the edges (more than twice the mean at 8 ranks).  The driver cuts the ranges by in-edges instead:
    max / mean of the local edge counts stays within 1.1, and the ranks are still the oracle's."""
    import oracle

    scale = 12
    nv, ne = 1 << scale, 16 << scale
    src, dst = oracle.rmat_edges(scale, ne, seed=7, scramble=False)
    case = dict(src=src.astype(np.int64), dst=dst.astype(np.int64), wgt=None, nv=nv, eps=nv * 1e-9, max_iter=200,
                check=4)
    equal = np.bincount(np.minimum(dst // -(-nv // world), world - 1), minlength=world)
    assert equal.max() / equal.mean() > 1.5  # what equal ranges would do to this graph
    fulls, metas = _run(world, case, tmp_path)
    local_edges = np.array([m[6] for m in metas])
    assert local_edges.sum() == ne
    assert local_edges.max() / local_edges.mean() <= 1.1
    assert [int(m[7]) for m in metas][1:] == [int(m[8]) for m in metas][:-1]  # the ranges tile the vertices
    o_pr, o_it, _ = _oracle(case)
    for f, m in zip(fulls, metas):
        assert m[0] == o_it and np.max(np.abs(f - o_pr)) < 1e-7  # (fp32 at the API for an unweighted graph)


def test_equal_ranges_on_request(tmp_path):
    case = _rmat_case(9, False, partition="equal")
    fulls, metas = _run(3, case, tmp_path)
    o_pr, o_it, _ = _oracle(case)
    assert all(m[0] == o_it and np.max(np.abs(f - o_pr)) < 1e-12 for f, m in zip(fulls, metas))


def test_partition_bounds_cover_everything():
    from cugraph_amd import mg

    for nv in (0, 1, 7, 8, 9, 1000):
        for world in (1, 2, 3, 8):
            spans = [mg.partition_bounds(nv, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nv
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0] and a[2] == b[2]
            assert all(last - first <= chunk for first, last, chunk in spans)


@pytest.mark.parametrize("name", [
    "test_pagerank", "test_pagerank_4", "test_pagerank_non_convergence",
    "test_personalized_pagerank", "test_personalized_pagerank_non_convergence"])
@pytest.mark.parametrize("world", [2, 3])
def test_mg_c_api_goldens(name, world, tmp_path):
    """The golden vectors of the reference's multi-GPU C-API test (cpp/tests/c_api/mg_pagerank_test.c:
    396-560, the same data as pagerank_test.c) through the driver: every rank holds a share of the
    edges and gets its slice back; values within the reference's own 0.001 relative."""
    import json

    with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
        c = json.load(f)["c_api"]["cases"][name]
    case = dict(src=np.array(c["src"], np.int64), dst=np.array(c["dst"], np.int64),
                wgt=np.array(c["wgt"], np.float64), nv=c["num_vertices"], eps=c["epsilon"],
                max_iter=c["max_iterations"], check=1, alpha=c["alpha"])
    if "pers_vertices" in c:
        case["pers"] = (np.array(c["pers_vertices"], np.int64), np.array(c["pers_values"], np.float64))
    fulls, metas = _run(world, case, tmp_path)
    want = np.array(c["expected"])
    for f, m in zip(fulls, metas):
        assert np.all(np.abs(f - want) <= 1e-3 * np.maximum(np.abs(f), np.abs(want)))
        assert bool(m[3]) == (not c["allow_nonconvergence"])


@pytest.mark.parametrize("seed,world", [(0, 2), (1, 3), (2, 4), (3, 5)])
def test_mg_random_small_graphs(seed, world, tmp_path):
    """Small random graphs through the driver -- isolated vertices, self-loops, parallel edges,
    ranks that own few or no vertices or receive no edges -- with regions and the source ordering
    of the low-degree rows forced on, sharded personalization and initial guess."""
    rng = np.random.default_rng(500 + seed)
    nv = int(rng.integers(world - 1, 60))  # may leave the last rank(s) without vertices
    ne = int(rng.integers(1, 6 * nv + 2))
    live = rng.choice(nv, max(1, nv // 2), replace=False)
    src, dst = rng.choice(live, ne).astype(np.int64), rng.choice(live, ne).astype(np.int64)
    wgt = rng.random(ne) + 0.05 if seed % 2 else None
    pers = (rng.permutation(nv)[: max(1, nv // 5)].astype(np.int64), rng.random(max(1, nv // 5)) + 0.1)
    nstart = (rng.permutation(nv)[: max(1, nv // 2)].astype(np.int64), rng.random(max(1, nv // 2)) / nv)
    case = dict(src=src, dst=dst, wgt=wgt, nv=nv, eps=1e-12, max_iter=60, check=2, regions="1", locality="1",
                pers=pers, nstart=nstart, sharded=True)
    fulls, metas = _run(world, case, tmp_path)
    o_pr, o_it, o_conv = _oracle(case)
    for f, m in zip(fulls, metas):
        assert bool(m[3]) == o_conv and (m[0] == o_it or wgt is None)
        # (an unweighted graph's values are fp32 at the API: the lists are rounded on their way in)
        assert np.max(np.abs(f - o_pr)) < (1e-12 if wgt is not None else 1e-7)


def _bench(*argv):
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), *argv],
                          capture_output=True, text=True, timeout=300, env=env)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it (the driver's command): the parent
    spawns the N ranks before anything touches a GPU and rank 0's JSON line comes out of it.
    Rehearsed on CPU ranks over gloo with the numpy stand-in engine."""
    import json

    r = _bench("--gpus", "2", "--backend", "gloo", "--engine", "test-double", "--scale", "9", "--steps", "3",
               "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["engine"] == "test-double"
    assert d["iterations"] == 4 and abs(d["rank_sum"] - 1.0) < 1e-9


def test_bench_says_what_it_needs_without_the_devices():
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has the devices")
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "needs 2 devices" in r.stderr and "WORLD_SIZE" not in r.stderr.split("needs 2 devices")[0][-200:]
