"""Set-up cost of the multi-GPU driver at ONE rank's share of scale 28 over 8 ranks, on one GPU:
(a) the send side of the edge shuffle -- this rank's 1/8 of the R-MAT stream (537 M edges, destinations
all over the graph) grouped by owner with the library's multisplit, against round 2's torch version
(bucketize + one nonzero per owner + bincount + three gathers);  (b) everything after the shuffle is timed
by `bench.py --force-mg --scale 25` with CUGRAPH_AMD_MG_PROFILE=1 (same edge and row counts per rank).
usage: mg_setup_time.py [scale] [world]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cugraph_amd import generators, mg, pylibcugraph as plc

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 28
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nv, ne = 1 << scale, 16 << scale
n = ne // world
h = plc.ResourceHandle()
src, dst = generators.rmat(h, scale, n, seed=42, scramble_vertex_ids=True, first_edge=0)
torch.cuda.synchronize()
bounds = mg.equal_bounds(nv, world)
eng = mg.HipEngine(h, plc.GraphProperties(is_multigraph=True))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.group_edges(src, dst, None, bounds)
    torch.cuda.synchronize(); t_lib = time.perf_counter() - t0
    counts = out[3]
    del out
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    inner = torch.as_tensor(bounds[1:-1], dtype=torch.int64, device="cuda")
    owner = torch.bucketize(dst.to(torch.int64), inner, right=True)
    order = torch.cat([torch.nonzero(owner == p).flatten() for p in range(world)])
    send = torch.bincount(owner, minlength=world)
    a, b = src[order].contiguous(), dst[order].contiguous()
    torch.cuda.synchronize(); t_torch = time.perf_counter() - t0
    del inner, owner, order, a, b
print(f"scale {scale}, one rank of {world}: {n} edges grouped by owner: library multisplit {t_lib * 1e3:.1f} ms "
      f"({n * 8 * 2 / t_lib / 1e9:.0f} GB/s of read + write), torch (round 2) {t_torch * 1e3:.1f} ms; counts {counts[:3]}...")
