"""Seeded random sweep of the PageRank path through the C ABI against the CPU oracle: small random
graphs (uniform, star-heavy, with isolated vertices, self-loops and parallel edges) crossed with
the options of the call -- weights and their type, id type, renumber, stored orientation, alpha,
initial guess, personalization, precomputed out-weight sums, iteration limit -- with and without
XS tiles / compact sources / the low-degree source ordering forced onto the small graph."""
import numpy as np
import pytest

from helpers import make_sg_graph, to_dev

pytestmark = pytest.mark.gpu


def _random_graph(rng, kind):
    nv = int(rng.integers(2, 400))
    ne = int(rng.integers(1, 12 * nv))
    if kind == "uniform":
        src, dst = rng.integers(0, nv, ne), rng.integers(0, nv, ne)
    elif kind == "hubs":  # a few vertices take most edges, on either end
        hubs = rng.integers(0, nv, 3)
        src = np.where(rng.random(ne) < 0.5, rng.choice(hubs, ne), rng.integers(0, nv, ne))
        dst = np.where(rng.random(ne) < 0.5, rng.choice(hubs, ne), rng.integers(0, nv, ne))
    else:  # "sparse": most vertices untouched, some self-loops and repeated edges
        live = rng.choice(nv, max(2, nv // 5), replace=False)
        src, dst = rng.choice(live, ne), rng.choice(live, ne)
        src[: ne // 10] = dst[: ne // 10]
    return nv, src.astype(np.int64), dst.astype(np.int64)


@pytest.mark.parametrize("engine", ["buckets", "xs", "xs_forced_layouts", "xs2", "xs2_forced_layouts"])
@pytest.mark.parametrize("seed", range(12))
def test_random_graphs_and_options(seed, engine, monkeypatch):
    import oracle
    from cugraph_amd import pylibcugraph as plc

    if engine == "buckets":
        monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", str(1 << 40))
    else:
        monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", "0")
    if engine.startswith("xs2"):  # xs2_kernel instead of the barrier kernel small plans get by default
        monkeypatch.setenv("CUGRAPH_AMD_XS_ASYNC", "1")
    if engine.endswith("forced_layouts"):
        monkeypatch.setenv("CUGRAPH_AMD_COMPACT", "1")
        monkeypatch.setenv("CUGRAPH_AMD_RENUMBER_LOCALITY", "1")
        monkeypatch.setenv("CUGRAPH_AMD_XS_TILE_EDGES", "64")
        monkeypatch.setenv("CUGRAPH_AMD_XS_PIECE", "16")
    rng = np.random.default_rng(1000 + seed)
    for case in range(6):
        nv, src, dst = _random_graph(rng, ["uniform", "hubs", "sparse"][case % 3])
        weighted = bool(rng.integers(0, 2))
        wdtype = np.float64 if (weighted and rng.integers(0, 2)) else np.float32
        id_dtype = np.int64 if rng.integers(0, 2) else np.int32
        renumber = bool(rng.integers(0, 2))
        transposed = bool(rng.integers(0, 4))  # mostly stored the way PageRank wants it
        alpha = float(rng.choice([0.5, 0.85, 0.95, 1.0]))
        max_it = int(rng.choice([1, 3, 500]))
        wgt = (rng.random(len(src)) + 0.05).astype(wdtype) if weighted else None
        val_dtype = wdtype if weighted else np.float32
        g, h = make_sg_graph(src.astype(id_dtype), dst.astype(id_dtype), wgt, store_transposed=transposed,
                             renumber=renumber, vertices=np.arange(nv, dtype=id_dtype), multigraph=True)
        kw, args = {}, [None, None, None, None]
        if rng.integers(0, 2):  # partial initial guess
            gv = rng.permutation(nv)[: max(1, nv // 2)]
            gx = rng.random(len(gv)) / len(gv)
            ns = np.zeros(nv)
            ns[gv] = gx.astype(val_dtype)
            kw["nstart"] = ns
            args[2], args[3] = to_dev(gv.astype(id_dtype)), to_dev(gx.astype(val_dtype))
        if rng.integers(0, 3) == 0:  # the caller's out-weight sums (true sums scaled: any consistent scaling goes through)
            ow = np.bincount(src, weights=None if wgt is None else wgt.astype(np.float64), minlength=nv) * 1.5
            kw["out_weight_sums"] = ow.astype(val_dtype).astype(np.float64)
            args[0], args[1] = to_dev(np.arange(nv, dtype=id_dtype)), to_dev(ow.astype(val_dtype))
        eps = 1e-7 if val_dtype == np.float32 else 1e-12
        if rng.integers(0, 2):
            pv = rng.permutation(nv)[: max(1, nv // 7)]
            px = (rng.random(len(pv)) + 0.1).astype(val_dtype)
            kw["pers_vertices"], kw["pers_values"] = pv, px
            out = plc.personalized_pagerank(h, g, *args, to_dev(pv.astype(id_dtype)), to_dev(px), alpha, eps, max_it,
                                            False, fail_on_nonconvergence=False)
        else:
            out = plc.pagerank(h, g, *args, alpha, eps, max_it, False, fail_on_nonconvergence=False)
        c_off, c_idx, perm = oracle.compress(nv, dst, src)
        o_pr, o_it, o_conv = oracle.pagerank(nv, c_off, c_idx, None if wgt is None else wgt[perm], dtype=np.float64,
                                             alpha=alpha, epsilon=eps, max_iterations=max_it, **kw)
        got = np.empty(nv)
        got[out[0].cpu().numpy()] = out[1].cpu().numpy()
        tol = 2e-6 if val_dtype == np.float32 else 1e-10
        what = f"seed {seed} case {case}: nv {nv} ne {len(src)} w {weighted}/{wdtype.__name__} ids {id_dtype.__name__} " \
               f"renumber {renumber} transposed {transposed} alpha {alpha} max_it {max_it} opts {sorted(kw)}"
        assert np.max(np.abs(got - o_pr)) <= tol * max(1.0, float(np.abs(o_pr).max())), what
        if max_it < 500:
            assert bool(out[2]) == o_conv, what


@pytest.mark.parametrize("engine", ["buckets", "xs_forced_layouts", "xs2_forced_layouts"])
@pytest.mark.parametrize("seed", range(6))
def test_random_graphs_katz_and_hits(seed, engine, monkeypatch):
    """The sibling power iterations on the same random graphs: Katz (always convergent for alpha
    under 1 / the largest weighted in-degree) with and without per-vertex betas, and HITS where
    its iteration settles, against oracle/np_centrality.py."""
    from oracle import np_centrality as npc
    from cugraph_amd import pylibcugraph as plc

    monkeypatch.setenv("CUGRAPH_AMD_XS_MIN_EDGES", str(1 << 40) if engine == "buckets" else "0")
    if engine.startswith("xs2"):
        monkeypatch.setenv("CUGRAPH_AMD_XS_ASYNC", "1")
    if engine != "buckets":
        monkeypatch.setenv("CUGRAPH_AMD_COMPACT", "1")
        monkeypatch.setenv("CUGRAPH_AMD_RENUMBER_LOCALITY", "1")
        monkeypatch.setenv("CUGRAPH_AMD_XS_TILE_EDGES", "64")
        monkeypatch.setenv("CUGRAPH_AMD_XS_PIECE", "16")
    rng = np.random.default_rng(2000 + seed)
    for case in range(4):
        nv, src, dst = _random_graph(rng, ["uniform", "hubs", "sparse"][case % 3])
        dtype = np.float64 if rng.integers(0, 2) else np.float32
        wgt = (rng.random(len(src)) + 0.05).astype(dtype)
        renumber = bool(rng.integers(0, 2))
        g, h = make_sg_graph(src.astype(np.int32), dst.astype(np.int32), wgt, renumber=renumber,
                             vertices=np.arange(nv, dtype=np.int32), multigraph=True)
        tol = 2e-5 if dtype == np.float32 else 1e-9
        alpha = 0.5 / np.bincount(dst, weights=wgt.astype(np.float64), minlength=nv).max()
        betas = (rng.random(nv) + 0.5).astype(dtype) if rng.integers(0, 2) else None
        v, x = plc.katz_centrality(h, g, None if betas is None else to_dev(betas), alpha, 1.0, 1e-9, 1000, False)
        want, _ = npc.katz_centrality(nv, src, dst, wgt, alpha=alpha, beta=1.0, betas=betas, epsilon=1e-9,
                                      max_iterations=1000)
        got = np.empty(nv)
        got[v.cpu().numpy()] = x.cpu().numpy()
        assert np.max(np.abs(got - want)) < tol, f"katz seed {seed} case {case} nv {nv} renumber {renumber}"
        try:
            wh, wa, _, _ = npc.hits(nv, src, dst, epsilon=1e-7, max_iterations=300, normalize=True)
        except RuntimeError:
            continue  # this graph's HITS iteration does not settle: nothing to compare
        try:
            v, hubs, auth = plc.hits(h, g, 1e-7, 300, None, None, True, False)
        except RuntimeError:
            # the device run carries fp32 / atomics-order noise: a run that converges on the limit may differ
            continue
        gh, ga = np.empty(nv), np.empty(nv)
        gh[v.cpu().numpy()], ga[v.cpu().numpy()] = hubs.cpu().numpy(), auth.cpu().numpy()
        assert np.max(np.abs(gh - wh)) < 50 * tol and np.max(np.abs(ga - wa)) < 50 * tol, \
            f"hits seed {seed} case {case} nv {nv} renumber {renumber}"


@pytest.mark.parametrize("seed", range(8))
def test_random_graph_builds(seed):
    """The build alone on random edge lists and options: COO or CSR input, either orientation,
    int32 / int64 ids, renumbering (plain ids or ids with gaps and negatives), weights of either
    type, drop_self_loops / drop_multi_edges, do_expensive_check.  The stored graph is the
    oracle's (major, minor)-sorted compression of the expected edge list, weights in input order
    among equal edges; with renumbering: in descending major degree through the number map."""
    import torch
    import oracle
    from helpers import compressed_arrays
    from cugraph_amd import pylibcugraph as plc

    rng = np.random.default_rng(3000 + seed)
    dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    for case in range(8):
        nv, src, dst = _random_graph(rng, ["uniform", "hubs", "sparse"][case % 3])
        id_dtype = np.int64 if rng.integers(0, 2) else np.int32
        transposed, renumber = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        weighted = bool(rng.integers(0, 2))
        wdtype = np.float64 if rng.integers(0, 2) else np.float32
        wgt = (rng.random(len(src)) + 0.05).astype(wdtype) if weighted else None
        drop_loops, drop_multi = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        csr = bool(rng.integers(0, 3) == 0) and not renumber
        ext = np.arange(nv, dtype=np.int64)
        if renumber and rng.integers(0, 2):
            ext = ext * 5 - 17  # gaps and negative ids
        what = f"seed {seed} case {case}: nv {nv} ne {len(src)} ids {id_dtype.__name__} transposed {transposed} " \
               f"renumber {renumber} weighted {weighted} drop {drop_loops}/{drop_multi} csr {csr}"
        h = plc.ResourceHandle()
        props = plc.GraphProperties(is_symmetric=False, is_multigraph=not drop_multi)
        if csr:  # CSR input: offsets over sources; the edge list it stands for is (src, dst) sorted by src
            o = np.argsort(src, kind="stable")
            src, dst, wgt = src[o], dst[o], (None if wgt is None else wgt[o])
            offs = np.zeros(nv + 1, np.int64)
            np.cumsum(np.bincount(src, minlength=nv), out=offs[1:])
            g = plc.SGGraph(h, plc.GraphProperties(is_symmetric=False, is_multigraph=True), dev(offs.astype(id_dtype)),
                            dev(dst.astype(id_dtype)), weight_array=dev(wgt), store_transposed=transposed,
                            renumber=False, do_expensive_check=True, input_array_format="CSR")
            drop_loops_eff = drop_multi_eff = False
        else:
            g = plc.SGGraph(h, props, dev(ext[src].astype(id_dtype)), dev(ext[dst].astype(id_dtype)),
                            weight_array=dev(wgt), store_transposed=transposed, renumber=renumber,
                            do_expensive_check=True, vertices_array=dev(ext.astype(id_dtype)),
                            drop_self_loops=drop_loops, drop_multi_edges=drop_multi)
            drop_loops_eff, drop_multi_eff = drop_loops, drop_multi
        s, d, x = src, dst, wgt
        if drop_loops_eff:
            keep = s != d
            s, d, x = s[keep], d[keep], (None if x is None else x[keep])
        if drop_multi_eff:  # the first copy in input order survives (not flagged symmetric)
            _, first_pos = np.unique(np.stack([s, d], 1), axis=0, return_index=True)
            first_pos.sort()
            s, d, x = s[first_pos], d[first_pos], (None if x is None else x[first_pos])
        off, idx, w, nmap = compressed_arrays(g, h)
        assert g.info()["num_vertices"] == nv and int(off[-1]) == len(s), what
        if renumber:
            assert nmap is not None and nmap.dtype == id_dtype and sorted(nmap.tolist()) == sorted(ext.tolist()), what
            inv = {int(e): i for i, e in enumerate(nmap)}
            s2 = np.array([inv[int(e)] for e in ext[s]], np.int64)
            d2 = np.array([inv[int(e)] for e in ext[d]], np.int64)
            deg = np.diff(off)
            assert np.all(deg[:-1] >= deg[1:]), what
        else:
            s2, d2 = s, d
        major, minor = (d2, s2) if transposed else (s2, d2)
        o_off, o_idx, o_w = oracle.csc_lexsort(nv, minor, major, x)
        assert np.array_equal(off.astype(np.int64), o_off) and np.array_equal(idx, o_idx), what
        if weighted:
            assert w.dtype == wdtype and np.array_equal(w, o_w), what
