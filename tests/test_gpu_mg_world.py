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
@pytest.mark.parametrize("regions", ["0", "1"])
@pytest.mark.parametrize("mode", ["unweighted", "weighted"])
def test_driver_with_ranks_sharing_one_gpu_over_gloo(mode, regions, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CUGRAPH_AMD_MG_REGIONS=regions, CUGRAPH_AMD_TEST_SHARED_GPU="1",
               CUGRAPH_AMD_XS_MIN_EDGES="65536", CUGRAPH_AMD_TEST_DUMP_AFTER="100")  # a stuck rank dumps its stack and exits
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(HERE, "mg_world_worker.py"), "17", mode],
                       capture_output=True, text=True, timeout=170, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.count("iterations, max-abs") == world
