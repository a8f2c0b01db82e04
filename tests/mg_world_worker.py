"""One rank of a real multi-GPU run of the driver (RCCL): launched by tests/test_gpu_mg_world.py through
torch.distributed.run, one process per GPU.  Every rank holds a round-robin share of an R-MAT edge list,
runs cugraph_amd.mg.pagerank and checks its slice (and the iteration count) against the CPU oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import oracle
    from cugraph_amd import mg, pylibcugraph as plc

    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    weighted = len(sys.argv) > 2 and sys.argv[2] == "weighted"
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    # rehearsal on a one-GPU box: every rank on device 0, collectives over gloo (staged through the host)
    shared = os.environ.get("CUGRAPH_AMD_TEST_SHARED_GPU") == "1"
    if os.environ.get("CUGRAPH_AMD_TEST_DUMP_AFTER"):  # where a rank is stuck, instead of a silent hang
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["CUGRAPH_AMD_TEST_DUMP_AFTER"]), exit=True)
    if shared:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    try:
        nv, ne = 1 << scale, 16 << scale
        src, dst = oracle.rmat_edges(scale, ne, seed=9, scramble=True)
        wgt = oracle.rmat_weights(ne, seed=10) if weighted else None
        sel = np.arange(ne) % world == rank
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a[sel])).cuda()  # noqa: E731
        h = plc.ResourceHandle(rank=rank, comm_size=world)
        drv = mg.MGPagerank.from_edges(mg.HipEngine(h, plc.GraphProperties(is_multigraph=True)),
                                       t(src), t(dst), t(wgt), nv, rank=rank, world=world)
        eps = 1e-7
        sess = drv.session(alpha=0.85, epsilon=eps)
        iters, stopped, delta = drv.run_to_convergence(sess, 200, check_every=5)
        verts, vals, it2, conv = drv.result(sess, 200)
        off, idx, perm = oracle.compress(nv, dst, src)
        o_pr, o_it, o_conv = oracle.pagerank(nv, off, idx, None if wgt is None else wgt[perm], dtype=np.float64,
                                             alpha=0.85, epsilon=eps, max_iterations=200)
        first, last = drv.first, drv.last
        assert verts.cpu().numpy().tolist() == list(range(first, last))
        got = vals.cpu().numpy().astype(np.float64)
        assert stopped and conv and o_conv and iters == o_it, (iters, o_it)
        assert 0.0 < delta < eps
        tol = 1e-10 if weighted else 1e-6
        assert np.max(np.abs(got - o_pr[first:last])) <= tol
        print(f"rank {rank} of {world}: {last - first} vertices, {iters} iterations, max-abs "
              f"{np.max(np.abs(got - o_pr[first:last])):.2e}", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
