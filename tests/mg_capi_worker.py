"""One rank of the multi-GPU path BELOW the C ABI (cugraph_mg_graph_create / cugraph_graph_create_mg +
cugraph_[personalized_]pagerank on an MG handle, cugraph_amd/csrc/mg_capi.hip): launched by
tests/test_gpu_mg_world.py through torch.distributed.run.  The three collectives the library asks its
handle for are ctypes callbacks here, over gloo and staged through the host (ranks share GPU 0), so the
test covers everything but RCCL: the edge shuffle, per-range degree order, the table all-gather, the
out-weight all-reduce, the per-iteration exchange, and the (vertex, value) lists given in shares.
Checked against the CPU oracle: this rank's (vertex, rank) pairs and the iteration count."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

AR = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p)
AG = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p)
A2A = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t),
                       ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p)


class Comm(ctypes.Structure):
    _fields_ = [("context", ctypes.c_void_p), ("all_reduce_f64", AR), ("all_gather_v", AG), ("all_to_all_v", A2A)]


def make_comm(rank, world):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]

    def d2h(ptr, nbytes):
        out = np.empty(max(nbytes, 1), dtype=np.uint8)
        if nbytes:
            assert hip.hipMemcpy(out.ctypes.data, ptr, nbytes, 2) == 0
        return out[:nbytes]

    def h2d(ptr, arr):
        if arr.nbytes:
            arr = np.ascontiguousarray(arr)
            assert hip.hipMemcpy(ptr, arr.ctypes.data, arr.nbytes, 1) == 0

    def gather_bytes(mine, sizes):
        """every rank's byte string (sizes known to all), through an all_gather of padded tensors"""
        longest = max(max(sizes), 1)
        pad = np.zeros(longest, dtype=np.uint8)
        pad[:mine.size] = mine
        outs = [torch.zeros(longest, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(outs, torch.from_numpy(pad))
        return [o.numpy()[:n] for o, n in zip(outs, sizes)]

    def all_reduce(_ctx, buf, count, op, stream):
        try:
            hip.hipStreamSynchronize(stream)
            t = torch.from_numpy(d2h(buf, count * 8).view(np.float64).copy())
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
            h2d(buf, t.numpy())
            return 0
        except Exception as e:  # an exception must not cross the C boundary
            print("all_reduce callback:", repr(e), flush=True)
            return 1

    def all_gather_v(_ctx, send, recv, sizes_p, stream):
        try:
            hip.hipStreamSynchronize(stream)
            sizes = [int(sizes_p[r]) for r in range(world)]
            parts = gather_bytes(d2h(send, sizes[rank]), sizes)
            h2d(recv, np.concatenate(parts) if sum(sizes) else np.empty(0, np.uint8))
            return 0
        except Exception as e:
            print("all_gather_v callback:", repr(e), flush=True)
            return 1

    def all_to_all_v(_ctx, send, send_p, recv, recv_p, stream):
        try:
            hip.hipStreamSynchronize(stream)
            mine = [int(send_p[r]) for r in range(world)]
            want = [int(recv_p[r]) for r in range(world)]
            table = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(table, torch.tensor(mine, dtype=torch.int64))
            table = [t.tolist() for t in table]  # table[r][q]: bytes rank r sends to rank q
            bufs = gather_bytes(d2h(send, sum(mine)), [sum(t) for t in table])
            out = []
            for r in range(world):
                at = sum(table[r][:rank])
                assert table[r][rank] == want[r]
                out.append(bufs[r][at:at + want[r]])
            h2d(recv, np.concatenate(out) if sum(want) else np.empty(0, np.uint8))
            return 0
        except Exception as e:
            print("all_to_all_v callback:", repr(e), flush=True)
            return 1

    cbs = (AR(all_reduce), AG(all_gather_v), A2A(all_to_all_v))
    return Comm(None, *cbs), cbs  # the caller keeps cbs alive


def main():
    import oracle
    from cugraph_amd import _capi, pylibcugraph as plc
    from cugraph_amd._capi import GraphPropertiesStruct
    from cugraph_amd.pylibcugraph.utils import create_view_from_py_obj, free_view

    scale = int(sys.argv[1])
    weighted = sys.argv[2] == "weighted"
    share = sys.argv[3]  # "round-robin": every rank a share; "rank0": rank 0 everything (mg_test_utils.cpp:163)
    what = sys.argv[4] if len(sys.argv) > 4 else "plain"
    personalized = what == "personalized"
    guess = what == "guess"  # + the graph through cugraph_graph_create_mg with its share in two arrays
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("CUGRAPH_AMD_TEST_DUMP_AFTER"):
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["CUGRAPH_AMD_TEST_DUMP_AFTER"]), exit=True)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = _capi.load()
        nv, ne = 1 << scale, 16 << scale
        src, dst = oracle.rmat_edges(scale, ne, seed=9, scramble=True)
        wgt = oracle.rmat_weights(ne, seed=10) if weighted else None
        sel = (np.arange(ne) % world == rank) if share == "round-robin" else np.full(ne, rank == 0)
        id_dtype = torch.int64 if world == 3 else torch.int32  # both id types get a run
        t_src = torch.from_numpy(np.ascontiguousarray(src[sel])).to(id_dtype).cuda()
        t_dst = torch.from_numpy(np.ascontiguousarray(dst[sel])).to(id_dtype).cuda()
        t_wgt = None if wgt is None else torch.from_numpy(np.ascontiguousarray(wgt[sel])).cuda()
        if share == "rank0" and rank != 0:
            t_wgt = None  # (graph.h:221-228: a rank without edges may pass a NULL weights pointer)
        h = plc.ResourceHandle(rank=rank, comm_size=world)
        comm, keep = make_comm(rank, world)
        err = ctypes.c_void_p()
        _capi.check(lib.cugraph_amd_resource_handle_set_comm(h.c_resource_handle_ptr, ctypes.byref(comm),
                                                             ctypes.byref(err)), err, "set_comm")
        props = GraphPropertiesStruct(False, True)
        graph = ctypes.c_void_p()
        if guess:
            cut = t_src.numel() // 3
            parts = [(t_src[:cut].contiguous(), t_dst[:cut].contiguous(), None if t_wgt is None else t_wgt[:cut].contiguous()),
                     (t_src[cut:].contiguous(), t_dst[cut:].contiguous(), None if t_wgt is None else t_wgt[cut:].contiguous())]
            views = [create_view_from_py_obj(t) for part in parts for t in part]
            arr = lambda k: (ctypes.c_void_p * 2)(views[k], views[3 + k])  # noqa: E731
            lists = [arr(0), arr(1), arr(2) if t_wgt is not None else None]
            code = lib.cugraph_graph_create_mg(h.c_resource_handle_ptr, ctypes.byref(props), None, lists[0], lists[1],
                                               lists[2], None, None, 1, 2, 0, 0, 0, ctypes.byref(graph), ctypes.byref(err))
            _capi.check(code, err, "cugraph_graph_create_mg")
        else:
            views = [create_view_from_py_obj(t) for t in (t_src, t_dst, t_wgt)]
            code = lib.cugraph_mg_graph_create(h.c_resource_handle_ptr, ctypes.byref(props), views[0], views[1], views[2],
                                               None, None, 1, ne, 0, ctypes.byref(graph), ctypes.byref(err))
            _capi.check(code, err, "cugraph_mg_graph_create")
        eps = 1e-7
        result = ctypes.c_void_p()
        pers = None
        if personalized:  # a list of 50 vertices, given in shares that ignore who owns what; rank 0 an empty one
            rng = np.random.default_rng(3)
            pv = rng.choice(nv, 50, replace=False)
            px = rng.random(50) + 0.1
            mine = (np.arange(50) % max(world - 1, 1) == rank - 1) if world > 1 else np.full(50, True)
            t_pv = torch.from_numpy(pv[mine]).to(id_dtype).cuda()
            t_px = torch.from_numpy(px[mine]).to(torch.float64 if weighted else torch.float32).cuda()
            pviews = [create_view_from_py_obj(t_pv), create_view_from_py_obj(t_px)]
            if mine.sum() == 0:
                pviews = [None, None]
            code = lib.cugraph_personalized_pagerank(h.c_resource_handle_ptr, graph, None, None, None, None, pviews[0],
                                                     pviews[1], 0.85, eps, 200, 0, ctypes.byref(result), ctypes.byref(err))
            pers = (pv, px / px.sum())
        elif guess:  # an initial guess for every vertex, all of it handed in by the LAST rank
            rng = np.random.default_rng(4)
            g0 = rng.random(nv) + 0.5
            g0 /= g0.sum()
            vdt = torch.float64 if weighted else torch.float32
            if rank == world - 1:
                t_gv = torch.arange(nv).to(id_dtype).cuda()
                t_gx = torch.from_numpy(g0).to(vdt).cuda()
                gviews = [create_view_from_py_obj(t_gv), create_view_from_py_obj(t_gx)]
            else:
                gviews = [None, None]
            code = lib.cugraph_pagerank(h.c_resource_handle_ptr, graph, None, None, gviews[0], gviews[1], 0.85, eps, 200, 0,
                                        ctypes.byref(result), ctypes.byref(err))
            nstart = torch.from_numpy(g0).to(vdt).numpy().astype(np.float64)  # as the library saw it
        else:
            fn = lib.cugraph_pagerank_allow_nonconvergence if os.environ.get("MG_CAPI_DEBUG") else lib.cugraph_pagerank
            code = fn(h.c_resource_handle_ptr, graph, None, None, None, None, 0.85, eps, 200, 0,
                      ctypes.byref(result), ctypes.byref(err))
        _capi.check(code, err, "cugraph_pagerank on a multi-GPU graph")
        v_view = lib.cugraph_centrality_result_get_vertices(result)
        x_view = lib.cugraph_centrality_result_get_values(result)
        n = lib.cugraph_type_erased_device_array_view_size(v_view)
        h_v = np.empty(n, dtype=np.int64 if id_dtype == torch.int64 else np.int32)
        h_x = np.empty(n, dtype=np.float64 if weighted else np.float32)
        for view, host in ((v_view, h_v), (x_view, h_x)):
            code = lib.cugraph_type_erased_device_array_view_copy_to_host(h.c_resource_handle_ptr, host.ctypes.data,
                                                                          view, ctypes.byref(err))
            _capi.check(code, err, "copy_to_host")
        iters = lib.cugraph_centrality_result_get_num_iterations(result)
        conv = lib.cugraph_centrality_result_converged(result)
        off, idx, perm = oracle.compress(nv, dst, src)
        kw = {}
        if guess:
            kw = dict(nstart=nstart)
        if pers is not None:
            kw = dict(pers_vertices=pers[0], pers_values=pers[1])
        o_pr, o_it, o_conv = oracle.pagerank(nv, off, idx, None if wgt is None else wgt[perm], dtype=np.float64,
                                             alpha=0.85, epsilon=eps, max_iterations=200, **kw)
        chunk = (nv + world - 1) // world
        first, last = min(nv, rank * chunk), min(nv, (rank + 1) * chunk)
        if os.environ.get("MG_CAPI_DEBUG"):
            d = np.abs(h_x.astype(np.float64) - o_pr[h_v])
            print(f"DEBUG rank {rank}: iters {iters} oracle {o_it} conv {conv}; max-abs {d.max():.3e} at vertex {h_v[d.argmax()]}; "
                  f"sum got {h_x.sum():.6f} want {o_pr[first:last].sum():.6f}; ids ok {sorted(h_v.tolist()) == list(range(first, last))}", flush=True)
        assert sorted(h_v.tolist()) == list(range(first, last)), "a rank returns exactly the vertices it owns"
        assert conv and o_conv and iters == o_it, (iters, o_it)
        tol = 1e-10 if weighted else 1e-6
        worst = float(np.max(np.abs(h_x.astype(np.float64) - o_pr[h_v])))
        assert worst <= tol, worst
        print(f"rank {rank} of {world}: {last - first} vertices, {iters} iterations, max-abs {worst:.2e}", flush=True)
        lib.cugraph_type_erased_device_array_view_free(v_view)
        lib.cugraph_type_erased_device_array_view_free(x_view)
        lib.cugraph_centrality_result_free(result)
        lib.cugraph_mg_graph_free(graph)
        for v in views:
            if v is not None:
                free_view(v)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
