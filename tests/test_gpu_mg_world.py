"""The multi-GPU driver with MORE than one rank.  On any machine: 2 and 3 ranks SHARING GPU 0, collectives
over gloo -- every rank runs the real HIP sessions on its share of the graph (edge-balanced ragged ranges,
compact slots, the shuffle, both all-gathers), only RCCL itself is missing.  And the driver over RCCL with MORE than one rank, one process per GPU.  Skipped on a machine with
one GPU (the boxes this repository is developed on); it fires on the first multi-GPU lease.  What it covers
that nothing else can: the in-place all_gather_into_tensor on tensors aliasing library-owned device memory,
the NCCL all-to-all edge shuffle, and -- with regions on -- the asynchronous all-gather of region 1 running
beside xs_kernel's phase 0."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("regions", ["0", "1"])
@pytest.mark.parametrize("mode", ["unweighted", "weighted"])
def test_driver_over_rccl_with_every_gpu_of_the_machine(mode, regions):
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    n = min(n, 8)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CUGRAPH_AMD_MG_REGIONS=regions, HSA_ENABLE_IPC_MODE_LEGACY="0",
               CUGRAPH_AMD_XS_MIN_EDGES="65536")  # XS tiles (and their regions) also on this small graph
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(HERE, "mg_world_worker.py"), "18", mode],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("layout", ["one-region", "two-regions", "personal"])
@pytest.mark.parametrize("mode", ["unweighted", "weighted"])
def test_driver_with_ranks_sharing_one_gpu_over_gloo(mode, layout, world):
    """layout: the gathered vector as P chunks (one all-gather), as two regions (two all-gathers, phase 0 on
    the hot one), or the PERSONALISED exchange -- private slot numbering per rank, one all-to-all of exactly
    the values each rank's edges read, landing in the library-owned vector behind its own sources."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CUGRAPH_AMD_MG_REGIONS="1" if layout == "two-regions" else "0", CUGRAPH_AMD_TEST_SHARED_GPU="1",
               CUGRAPH_AMD_XS_MIN_EDGES="65536", CUGRAPH_AMD_TEST_DUMP_AFTER="100")  # a stuck rank dumps its stack and exits
    env.pop("CUGRAPH_AMD_MG_EXCHANGE", None)
    if layout == "personal":
        env["CUGRAPH_AMD_MG_EXCHANGE"] = "personal"
    if layout != "one-region":  # xs2_kernel, the kernel of full-size plans (small plans default to the barrier kernel)
        env["CUGRAPH_AMD_XS_ASYNC"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(HERE, "mg_world_worker.py"), "17", mode],
                       capture_output=True, text=True, timeout=170, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.count("iterations, max-abs") == world


@pytest.mark.parametrize("world,mode,share,what", [
    (2, "unweighted", "round-robin", "plain"),
    (2, "weighted", "rank0", "personalized"),
    (3, "unweighted", "rank0", "personalized"),
    (3, "weighted", "round-robin", "plain"),
    (1, "unweighted", "round-robin", "personalized"),
    (2, "unweighted", "round-robin", "guess"),
    (3, "weighted", "round-robin", "guess"),
])
def test_multi_gpu_graph_and_pagerank_through_the_c_names(world, mode, share, what):
    """cugraph_mg_graph_create + cugraph_[personalized_]pagerank on an MG handle (cugraph_amd/csrc/mg_capi.hip),
    the way cpp/tests/c_api/mg_pagerank_test.c:67-94 drives the reference: ranks sharing GPU 0, the library's
    three collectives supplied as ctypes callbacks over gloo (tests/mg_capi_worker.py), every rank's pairs and
    the iteration count against the CPU oracle.  share "rank0": rank 0 passes every edge and the others none,
    as create_mg_test_graph does (mg_test_utils.cpp:157-163)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CUGRAPH_AMD_XS_MIN_EDGES="65536", CUGRAPH_AMD_TEST_DUMP_AFTER="100")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(HERE, "mg_capi_worker.py"), "16", mode, share, what],
                       capture_output=True, text=True, timeout=170, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.count("iterations, max-abs") == world


@pytest.mark.parametrize("id_dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("world", [1, 3, 8])
def test_group_edges_by_owner_matches_a_stable_sort(world, weighted, id_dtype):
    """cugraph_amd_partition_group_edges (the send side of the edge shuffle) against numpy's stable argsort by
    owner, bit-exact: ragged bounds with an empty range, more edges than one tile of the multisplit."""
    import numpy as np

    from cugraph_amd import mg, pylibcugraph as plc

    rng = np.random.default_rng(world * 7 + weighted)
    nv, ne = 5000, 70001
    cuts = np.sort(rng.integers(0, nv, size=world - 1)) if world > 1 else np.array([], dtype=np.int64)
    if world > 2:
        cuts[1] = cuts[0]  # an empty range
    bounds = [0, *[int(c) for c in cuts], nv]
    src = rng.integers(0, nv, size=ne)
    dst = rng.integers(0, nv, size=ne)
    wgt = rng.random(ne) if weighted else None
    eng = mg.HipEngine(plc.ResourceHandle(), plc.GraphProperties(is_multigraph=True))
    t = lambda a, dt: None if a is None else torch.from_numpy(a).to(dt).cuda()  # noqa: E731
    g_src, g_dst, g_wgt, counts = eng.group_edges(t(src, id_dtype), t(dst, id_dtype), t(wgt, torch.float64), bounds)
    owner = np.searchsorted(np.asarray(bounds[1:-1], dtype=np.int64), dst, side="right")
    order = np.argsort(owner, kind="stable")
    assert counts == np.bincount(owner, minlength=world).tolist()
    assert np.array_equal(g_src.cpu().numpy(), src[order]) and np.array_equal(g_dst.cpu().numpy(), dst[order])
    if weighted:
        assert np.array_equal(g_wgt.cpu().numpy(), wgt[order])
    with pytest.raises(ValueError):  # a destination outside the bounds
        eng.group_edges(t(src, id_dtype), t(dst + nv, id_dtype), None, bounds)
