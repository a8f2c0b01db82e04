#!/usr/bin/env python3
"""bench.py -- PageRank iteration throughput on synthetic R-MAT (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W            (one process)
    python bench.py --gpus N --steps K --warmup W            (starts its N ranks itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE PageRank iteration over the whole graph (pull SpMV with the fused rank update,
source-value update, dangling-mass and L1-delta partial sums, plus the scalar finalize; for N>1
also the all-gather of the source vector and the 2-scalar all-reduce).  The workload is the
configuration the metric is quoted on: R-MAT scale 26, edge factor 16 (67M vertices / 1.07B
edges), unweighted fp32, inputs resident in HBM before the timed region; with N ranks the SAME
graph is 1D-partitioned by destination (strong scaling).  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--scale", type=int, default=26)
    p.add_argument("--edge-factor", type=int, default=16)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--scramble", type=int, default=1, help="scramble R-MAT vertex ids (0/1)")
    p.add_argument("--cpu-scale", type=int, default=24, help="R-MAT scale of the CPU-baseline sample")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget of that sample")
    p.add_argument("--renumber", type=int, default=1,
                   help="build the graph with renumber=True, cugraph.Graph's default (0/1)")
    p.add_argument("--force-mg", action="store_true",
                   help="run the multi-GPU driver (and its collectives) even with one rank")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-convergence", action="store_true")
    p.add_argument("--no-extra-configs", action="store_true",
                   help="skip BASELINE configs C2 (scale 22) and C3 (scale 24 fp64 weighted personalized)")
    p.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                   help="gloo: CPU ranks, only with --engine test-double (a rehearsal of the launcher and driver)")
    p.add_argument("--share-device", action="store_true",
                   help="REHEARSAL of the multi-rank path on a one-GPU machine: every rank runs its HIP sessions on "
                        "device 0 and the collectives go over gloo; the line it prints is marked as not a measurement")
    p.add_argument("--engine", default="hip", choices=("hip", "test-double"),
                   help="test-double: the numpy stand-in of tests/mg_test_double.py instead of the HIP library; "
                        "measures nothing -- it exists so that the multi-rank launch path can be run without GPUs")
    return p.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of THIS process
    (which has not touched the GPU) through torch.distributed.run, pass their output through and
    exit with their status.  Rank 0 prints the JSON line."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def _graph_to_host(graph):
    """(offsets, indices) of a graph's compressed arrays as numpy (inputs of the CPU baseline)."""
    import ctypes

    import numpy as np

    from cugraph_amd import _capi
    from cugraph_amd.session import _alias_device_memory

    lib = _capi.load()
    views = [ctypes.c_void_p() for _ in range(4)]
    err = ctypes.c_void_p()
    code = lib.cugraph_amd_graph_get_compressed(graph.c_graph_ptr, *[ctypes.byref(v) for v in views],
                                                ctypes.byref(err))
    _capi.check(code, err, "cugraph_amd_graph_get_compressed")
    out = []
    for v, dt in zip(views[:2], (None, np.int32)):
        n = lib.cugraph_type_erased_device_array_view_size(v)
        t = lib.cugraph_type_erased_device_array_view_type(v)
        ptr = lib.cugraph_type_erased_device_array_view_pointer(v)
        out.append(_alias_device_memory(ptr, n, np.int64 if t == 1 else np.int32).cpu().numpy())
    for v in views:
        if v:
            lib.cugraph_type_erased_device_array_view_free(v)
    return out[0].astype(np.int64), out[1]


def cpu_baseline(args, handle, props):
    """The C oracle (a port of the reference's CPU pagerank_reference, OpenMP over the host cores
    of this box) on a bounded sample of the same workload -- same generator, parameters and edge
    factor at a smaller scale -- sized to about 10-20 s of CPU work.  The sample's edges and CSC
    are produced on the GPU (inputs only); what is timed is the oracle's iterations.  Reported,
    not a target."""
    import numpy as np
    import torch

    import oracle
    from cugraph_amd import generators, pylibcugraph as plc

    scale = args.cpu_scale
    nv, ne = 1 << scale, args.edge_factor << scale
    src, dst = generators.rmat(handle, scale, ne, seed=args.seed, scramble_vertex_ids=bool(args.scramble))
    g = plc.SGGraph(handle, props, src, dst, store_transposed=True, renumber=False,
                    vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
    del src, dst
    off, idx = _graph_to_host(g)
    del g
    torch.cuda.empty_cache()

    def run(iters):
        t0 = time.perf_counter()
        oracle.pagerank(nv, off, idx, None, dtype=np.float32, alpha=0.85, epsilon=0.0,
                        max_iterations=iters)
        return time.perf_counter() - t0

    t2, t6 = run(2), run(6)  # the difference is 4 iterations without the fixed set-up cost
    iters = int(max(8, min(400, args.cpu_seconds / max((t6 - t2) / 4, 1e-3))))
    dt = run(iters)
    return {
        "value": ne * iters / dt / 1e9,
        "unit": "GTEPS",
        "cores": oracle.num_threads(),
        "kind": "port",
        "sample": f"R-MAT scale {scale} ef {args.edge_factor} ({ne} edges, natural vertex order), "
                  f"{iters} iterations of the OpenMP C oracle in {dt:.1f} s incl. its out-degree pass; "
                  "CSC build not timed",
    }


def other_cpu_baselines(args, handle):
    """north_star asks for NetworkX / SciPy PageRank timed on this box's host cores beside the GPU
    number: (i) networkx.pagerank itself at a scale its Python graph object can take, (ii) the
    SciPy restatement of its `_pagerank_scipy` loop (single-threaded CSR mat-vec) on a larger
    sample.  Reported, not targets."""
    import numpy as np

    import oracle
    from cugraph_amd import generators

    out = []
    try:
        scale = 20
        nv, ne = 1 << scale, args.edge_factor << scale
        src, dst = generators.rmat(handle, scale, ne, seed=args.seed, scramble_vertex_ids=bool(args.scramble))
        src, dst = src.cpu().numpy(), dst.cpu().numpy()

        tm = {}
        try:
            oracle.scipy_pagerank(nv, src, dst, alpha=0.85, tol=0.0, max_iter=20, timings=tm)
        except RuntimeError:  # tol = 0 never converges: exactly 20 iterations ran
            pass
        per = tm["loop_s"] / tm["iterations"]
        out.append({"value": ne / per / 1e9, "unit": "GTEPS", "cores": 1, "kind": "port",
                    "sample": f"SciPy restatement of networkx _pagerank_scipy, R-MAT scale {scale} ef "
                              f"{args.edge_factor} ({ne} edges): {tm['iterations']} iterations, "
                              f"{per * 1e3:.1f} ms each (CSR mat-vec is single-threaded; "
                              f"{os.cpu_count()} cores available); CSR build + normalisation "
                              f"{tm['build_s']:.2f} s not counted"})
        scale = 14
        nv, ne = 1 << scale, args.edge_factor << scale
        src, dst = generators.rmat(handle, scale, ne, seed=args.seed, scramble_vertex_ids=bool(args.scramble))
        src, dst = src.cpu().numpy(), dst.cpu().numpy()
        t0 = time.perf_counter()
        oracle.nx_pagerank(nv, src, dst, alpha=0.85, tol=1e-8, max_iter=500)
        dt = time.perf_counter() - t0
        _, iters = oracle.scipy_pagerank(nv, src, dst, alpha=0.85, tol=1e-8, max_iter=500, return_info=True)
        out.append({"value": ne * iters / dt / 1e9, "unit": "GTEPS", "cores": 1, "kind": "reference-oracle",
                    "sample": f"networkx.pagerank (the oracle north_star names) on a MultiDiGraph, R-MAT scale "
                              f"{scale} ef {args.edge_factor} ({ne} edges), tol 1e-8: {dt:.2f} s for graph "
                              f"object + {iters} iterations"})
    except Exception as exc:  # a reported extra must never cost the bench line
        out.append({"error": repr(exc)})
    return out


def extra_configs(args, handle):
    """BASELINE.json's other single-GPU configurations as driver-timed numbers beside the headline
    one: C2 = R-MAT scale 22 unweighted fp32, C3 = scale 24 fp64 weighted with a 1000-vertex
    personalization (natural vertex ids, weights from stream seed + 1, first 1000 of a
    seeded permutation with U(0,1) values).  Each: 3 warm-up + 10 event-timed iterations."""
    import torch

    from cugraph_amd import generators, pylibcugraph as plc
    from cugraph_amd.session import PagerankSession

    out = {}
    for name, scale, weighted in (("C2_scale22_fp32_unweighted", 22, False),
                                  ("C3_scale24_fp64_weighted_personalized_1k", 24, True)):
        try:
            nv, ne = 1 << scale, args.edge_factor << scale
            src, dst = generators.rmat(handle, scale, ne, seed=args.seed, scramble_vertex_ids=False)
            w = generators.rmat_weights(handle, ne, seed=args.seed + 1, dtype=torch.float64) if weighted else None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g = plc.SGGraph(handle, plc.GraphProperties(is_multigraph=True), src, dst, w, store_transposed=True,
                            renumber=True, vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
            torch.cuda.synchronize()
            t_csc = time.perf_counter() - t0
            del src, dst, w
            kw = {}
            if weighted:
                gen = torch.Generator(device="cpu").manual_seed(7)
                kw = dict(personalization_vertices=torch.randperm(nv, generator=gen)[:1000].to(torch.int32).cuda(),
                          personalization_values=torch.rand(1000, generator=gen, dtype=torch.float64).cuda())
            t0 = time.perf_counter()
            s = PagerankSession(handle, g, alpha=0.85, epsilon=0.0, **kw)
            torch.cuda.synchronize()
            t_plan = time.perf_counter() - t0
            s.iterate(3)
            it_ms, k_ms = s.timed_iterate(10)
            info = s.info()
            it, km = float(it_ms.mean()), float(k_ms.mean())
            ach = info["dominant_kernel_bytes"] / (km * 1e-3) / 1e9
            out[name] = {
                "workload": f"R-MAT scale {scale} ef {args.edge_factor}, natural ids, "
                            + ("fp64 weighted, 1000-vertex personalization" if weighted else "fp32 unweighted"),
                "dtype": "f64" if weighted else "f32",
                "ms_per_step": it, "value": ne / (it * 1e-3) / 1e9, "unit": "GTEPS",
                "algorithmic_GBps_whole_iteration": info["bytes_per_iteration"] / (it * 1e-3) / 1e9,
                "roofline": {"bound": "hbm", "kernel": xs_name(info) if info["mid_degree_vertices"] else "spmv_kernel",
                             "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                             "traffic": pmc_traffic(scale, args.edge_factor, 1, config=name),
                             "traffic_source": f"{PMC_FILE} (commit {pmc_file().get('commit')})",
                             "algorithmic_bytes_per_launch": info["dominant_kernel_bytes"],
                             "kernel_ms_avg": km},
                "graph_build": {"coo_to_csc_s": t_csc, "plan_build_s": t_plan},
            }
            del s, g
            torch.cuda.empty_cache()
        except Exception as exc:  # a reported extra must never cost the bench line
            out[name] = {"error": repr(exc)}
    return out


def main_test_double(args):
    """Rehearsal of the multi-rank path without GPUs: the same launcher, process group, driver
    (cugraph_amd.mg: partition, edge routing, collectives, stop rule) and JSON line, with the numpy
    stand-in of tests/mg_test_double.py (test infrastructure) as the local engine.  The line says so
    ("engine": "test-double"): its numbers measure nothing."""
    import numpy as np
    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from cugraph_amd import mg as mgmod
    from mg_test_double import OracleEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scale = min(args.scale, 12)
    nv, ne = 1 << scale, args.edge_factor << scale
    src, dst = oracle.rmat_edges(scale, ne, seed=args.seed, scramble=bool(args.scramble))
    sel = np.arange(ne) % world == rank
    drv = mgmod.MGPagerank.from_edges(OracleEngine(rank, world), torch.from_numpy(src[sel].astype(np.int64)),
                                      torch.from_numpy(dst[sel].astype(np.int64)), None, nv, rank=rank, world=world)
    drv.always_exchange = True
    sess = drv.session(alpha=0.85, epsilon=0.0)
    drv.iterate(sess, args.warmup)
    dist.barrier()
    t0 = time.perf_counter()
    drv.iterate(sess, args.steps)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    full, iters, _ = drv.gathered_result(sess, args.steps + args.warmup + 1)
    if rank == 0:
        print(json.dumps({
            "metric": "PageRank GTEPS/iter + achieved HBM GB/s, R-MAT scale 26 at 1/2/4/8 MI355X",
            "value": ne * args.steps / elapsed / 1e9, "unit": "GTEPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "engine": "test-double",
            "config": {"workload": f"launcher rehearsal on CPU ranks over gloo: R-MAT scale {scale}, numpy stand-in "
                                   "engine (tests/mg_test_double.py); not a measurement"},
            "iterations": int(iters), "rank_sum": float(full.sum()),
        }), flush=True)
    dist.barrier()
    dist.destroy_process_group()


PMC_FILE = "profiles/r3_pmc_xs.json"


def xs_name(info):
    """the XS-tile kernel the library runs: xs2_kernel (two accumulator banks, no item barriers) for plans from
    40 M entries on (build_xs, kXs2MinEntries), the barrier kernel xs_kernel below that or under
    CUGRAPH_AMD_XS_ASYNC=0.  (The entry count is estimated from the kernel's algorithmic bytes: 4 per entry plus
    the vectors; the two only disagree within a few per cent of the threshold.)"""
    env = os.environ.get("CUGRAPH_AMD_XS_ASYNC")
    if env is not None:
        return "xs_kernel" if env == "0" else "xs2_kernel"
    return "xs2_kernel" if info["dominant_kernel_bytes"] / 4.2 >= 40e6 else "xs_kernel"


def pmc_file():
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def pmc_traffic(scale, ef, n_gpus, config=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/, tools/profile_r3.sh + tools/r3_pmc_fold.py), or None when they do not describe this
    workload.  bench.py cannot run under the profiler itself; the file names the commit it was taken at."""
    d = pmc_file()
    if config is not None:
        return d.get("other_configs", {}).get(config, {}).get("traffic_bytes_per_launch")
    if d.get("scale") != scale or d.get("edge_factor") != ef or n_gpus != 1:
        return None
    return d.get("traffic_bytes_per_launch")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if os.environ.get("CUGRAPH_AMD_BENCH_DUMP_AFTER"):  # a rank that stops making progress says where, and exits
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["CUGRAPH_AMD_BENCH_DUMP_AFTER"]), exit=True)
    if args.engine == "test-double":
        return main_test_double(args)
    if args.backend != "nccl" and not args.share_device:
        raise SystemExit("--backend gloo runs CPU ranks: it needs --engine test-double (or --share-device)")
    # libraries (RCCL's banner, ...) write to the C-level stdout: keep fd 1 for the one JSON line
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: let bench.py start its ranks itself "
                         f"(no WORLD_SIZE in the environment) or launch {args.gpus} ranks")
    n_dev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if n_dev < world and not args.share_device:
        raise SystemExit(f"bench.py --gpus {world} needs {world} devices, one per rank: this machine has {n_dev}")
    device_index = local_rank if world > 1 and not args.share_device else 0
    torch.cuda.set_device(device_index)

    from cugraph_amd import generators, pylibcugraph as plc
    from cugraph_amd.session import PagerankSession

    scale, ef = args.scale, args.edge_factor
    nv, ne = 1 << scale, ef << scale

    dist = None
    use_mg = world > 1 or args.force_mg
    if use_mg:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if args.share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", device_index))

    handle = plc.ResourceHandle(rank=rank, comm_size=world) if use_mg else plc.ResourceHandle()
    props = plc.GraphProperties(is_symmetric=False, is_multigraph=True)

    # ---------------- inputs (untimed): R-MAT edges generated on the device, then the CSC build
    t_build0 = time.perf_counter()
    if not use_mg:
        src, dst = generators.rmat(handle, scale, ne, seed=args.seed,
                                   scramble_vertex_ids=bool(args.scramble))
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t_build0
        t0 = time.perf_counter()
        graph = plc.SGGraph(handle, props, src, dst, store_transposed=True,
                            renumber=bool(args.renumber),
                            vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
        torch.cuda.synchronize()
        t_csc = time.perf_counter() - t0
        del src, dst
        torch.cuda.empty_cache()
        mg = None
    else:
        from cugraph_amd import mg as mgmod

        mg = mgmod.MGPagerank.from_rmat(handle, props, scale, ne, seed=args.seed,
                                        scramble_vertex_ids=bool(args.scramble))
        mg.always_exchange = True
        torch.cuda.synchronize()
        t_gen, t_csc = mg.t_generate, mg.t_build
        graph = mg.graph

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- the first session on a graph builds its pull plan (degree-sorted copy of a
    # renumber=False graph, compact source slots, XS tiles), once per graph: timed on its own
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not use_mg:
        sess = PagerankSession(handle, graph, alpha=0.85, epsilon=0.0)
        step = lambda n: sess.iterate(n)  # noqa: E731
    else:
        sess = mg.session(alpha=0.85, epsilon=0.0)
        step = lambda n: mg.iterate(sess, n)  # noqa: E731
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    # ---------------- timed region: K iterations, epsilon = 0 so that none is skipped
    step(args.warmup)
    # more untimed warm-up until the machine has settled: batches of 10 iterations until two in a row take the
    # same time to 2 % (3 s at most).  Seen on the pool: the first tens of milliseconds of a process that
    # starts right after another one freed tens of GB run 10-15 % slow while the driver still reclaims that
    # memory (2.18 against 1.92 ms per iteration for the same build, profiles/r3_pmc_xs.md).
    settle_steps, prev, t_stop = 0, None, time.perf_counter() + 3.0
    while True:
        barrier()
        t0 = time.perf_counter()
        step(10)
        barrier()
        dt = time.perf_counter() - t0
        settle_steps += 10
        stop = (prev is not None and abs(dt - prev) <= 0.02 * prev) or time.perf_counter() > t_stop
        if dist is not None:  # every rank takes rank 0's decision
            flag = torch.tensor([int(stop)], dtype=torch.int64, device="cpu" if args.share_device else "cuda")
            dist.broadcast(flag, 0)
            stop = bool(int(flag.item()))
        prev = dt
        if stop:
            break
    barrier()
    t0 = time.perf_counter()
    step(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.share_device else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    gteps = ne / (elapsed / args.steps) / 1e9

    # ---------------- dominant kernel, live HIP events on the launch stream (rank-local)
    inner = sess if not use_mg else sess.inner
    info = inner.info()
    n_probe = min(args.steps, 10)
    iter_ms, kern_ms = inner.timed_iterate(n_probe) if not use_mg else mg.timed_iterate(sess, n_probe)
    kern_avg_ms = float(kern_ms.mean())
    bytes_per_launch = info["dominant_kernel_bytes"]  # this rank's share for N > 1
    achieved = bytes_per_launch / (kern_avg_ms * 1e-3) / 1e9
    roofline = {
        "bound": "hbm",
        "kernel": xs_name(info) if info["mid_degree_vertices"] and os.environ.get("CUGRAPH_AMD_XS") != "0"
                  else "spmv_kernel",
        "achieved": achieved,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS,
        "traffic": pmc_traffic(scale, ef, world),
        "traffic_source": f"{PMC_FILE} (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command, tools/profile_r3.sh; "
                          f"taken at commit {pmc_file().get('commit')})",
        "algorithmic_bytes_per_launch": bytes_per_launch,
        "kernel_ms_avg": kern_avg_ms,
        "iteration_ms_event_avg": float(iter_ms.mean()),
    }

    # a timing-aid build (make AIDS=1, CUGRAPH_AMD_LIBRARY) checks every address xs_kernel computes
    # against its arrays and records violations instead of faulting: say what it saw
    from cugraph_amd import _capi as _capi_mod
    _lib = _capi_mod.load()
    if hasattr(_lib, "cugraph_amd_aids_bounds_check"):
        import ctypes
        chk = (ctypes.c_ulonglong * 8)()
        _lib.cugraph_amd_aids_bounds_check(chk)
        print("xs_kernel bounds check: violations %d first{kind %d v0 %d v1 %d workgroup %d tile %d slice %d}"
              % (chk[0], chk[1], chk[2], chk[3], chk[4], chk[5] & 0xffffffff, chk[5] >> 32), file=sys.stderr)

    # ---------------- time to convergence at NetworkX tol = 1e-8 (epsilon = N * tol)
    conv = conv_abs = None
    if not args.no_convergence:
        sess = inner = None
        eps = nv * 1e-8
        barrier()
        t0 = time.perf_counter()
        if not use_mg:
            s2 = PagerankSession(handle, graph, alpha=0.85, epsilon=eps)
            it, stopped = 0, False
            while not stopped and it < 500:
                s2.iterate(4)
                it, stopped, delta, _ = s2.status()
        else:
            s2 = mg.session(alpha=0.85, epsilon=eps)
            it, stopped, delta = mg.run_to_convergence(s2, 500)
        barrier()
        dt = time.perf_counter() - t0
        conv = {"iterations": it, "converged": bool(stopped), "seconds": dt,
                "seconds_including_build": t_csc + t_plan + dt,
                "l1_delta": delta, "epsilon": eps,
                "note": f"NetworkX's stop rule, sum|dx| < N*tol with tol 1e-8: at N = 2^{args.scale} the threshold is "
                        f"{eps:.3g}" + (", so the first iteration already meets it (networkx.pagerank stops there too)"
                                        if it <= 1 else "")}
        del s2
        # the same run to an ABSOLUTE L1 threshold (cugraph's own epsilon semantics, pagerank_impl.cuh:328-335)
        eps_abs = 1e-6
        barrier()
        t0 = time.perf_counter()
        if not use_mg:
            s3 = PagerankSession(handle, graph, alpha=0.85, epsilon=eps_abs)
            it, stopped = 0, False
            while not stopped and it < 500:
                s3.iterate(8)
                it, stopped, delta, _ = s3.status()
        else:
            s3 = mg.session(alpha=0.85, epsilon=eps_abs)
            it, stopped, delta = mg.run_to_convergence(s3, 500, check_every=8)
        barrier()
        dt = time.perf_counter() - t0
        conv_abs = {"iterations": it, "converged": bool(stopped), "seconds": dt,
                    "seconds_including_build": t_csc + t_plan + dt,
                    "l1_delta": delta, "epsilon": eps_abs,
                    "note": "seconds: session set-up + iterations on a graph whose plan exists; "
                            "seconds_including_build adds the one-off COO->CSC build and plan build "
                            "(SURVEY 8d: time to convergence = build + iterations)"}
        del s3

    out = {
        "metric": "PageRank GTEPS/iter + achieved HBM GB/s, R-MAT scale 26 at 1/2/4/8 MI355X",
        "value": gteps,
        "unit": "GTEPS",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "settle_steps": settle_steps,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"R-MAT scale {scale} ef {ef} (a=.57 b=.19 c=.19, seed {args.seed}, "
                        f"{'scrambled' if args.scramble else 'natural'} ids), {nv} vertices / {ne} edges, "
                        "unweighted fp32 PageRank alpha=0.85, one iteration per step",
            "partition": "single GPU" if not use_mg else f"1D by destination over {world} GPUs",
            "renumber": bool(args.renumber) if not use_mg else False,
        },
        # ALGORITHMIC bytes of one iteration (E*4 + (V+1)*4 + 5*V*4) over its time: what the
        # metric calls achieved GB/s.  Not bytes moved: dead rows are not even visited, gathers re-read
        # lines (roofline.traffic has the counter bytes of the dominant kernel).
        "algorithmic_GBps_whole_iteration": info["bytes_per_iteration"] * (world if world > 1 else 1)
        / (ms_per_step * 1e-3) / 1e9,
        "roofline": roofline,
        "graph_build": {"generate_s": t_gen, "coo_to_csc_s": t_csc, "plan_build_s": t_plan,
                        "edges_per_s": ne / t_csc if t_csc > 0 else None,
                        "edges_per_s_including_plan": ne / (t_csc + t_plan) if t_csc + t_plan > 0 else None},
        "time_to_convergence": conv,
        "time_to_convergence_abs_eps": conv_abs,
        "buckets": {k: info[k] for k in ("low_degree_vertices", "mid_degree_vertices", "hub_vertices",
                                         "spmv_workgroups")},
    }
    if args.share_device:
        out["rehearsal"] = (f"{world} ranks SHARING one GPU, collectives over gloo through the host: exercises the "
                            "multi-rank code path, its timings are not a measurement")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sess = None
        graph = None
        torch.cuda.empty_cache()
        if not args.no_extra_configs:
            out["other_configs"] = extra_configs(args, handle)
        out["cpu_baseline"] = cpu_baseline(args, handle, props)
        out["cpu_baselines_networkx_scipy"] = other_cpu_baselines(args, handle)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
