"""Iteration time of BASELINE config C3: R-MAT scale 24, fp64, weighted, 1k-vertex personalization."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cugraph_amd import generators, pylibcugraph as plc
from cugraph_amd.session import PagerankSession
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
nv, ne = 1 << scale, 16 << scale
h = plc.ResourceHandle()
src, dst = generators.rmat(h, scale, ne, seed=42, scramble_vertex_ids=False)
w = generators.rmat_weights(h, ne, seed=43, dtype=torch.float64)
g = plc.SGGraph(h, plc.GraphProperties(is_multigraph=True), src, dst, w, store_transposed=True, renumber=True,
                vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
del src, dst, w
gen = torch.Generator(device="cpu").manual_seed(7)
pv = torch.randperm(nv, generator=gen)[:1000].to(torch.int32).cuda()
px = torch.rand(1000, generator=gen, dtype=torch.float64).cuda()
s = PagerankSession(h, g, alpha=0.85, epsilon=0.0, personalization_vertices=pv, personalization_values=px)
s.iterate(3)
torch.cuda.synchronize()
it, km = s.timed_iterate(10)
info = s.info()
print(f"C3 scale {scale} fp64 weighted personalized: iteration {it.mean():.3f} ms, dominant kernel {km.mean():.3f} ms, "
      f"{ne / it.mean() / 1e6:.1f} GTEPS, B_iter {info['bytes_per_iteration'] / 1e9:.2f} GB -> "
      f"{info['bytes_per_iteration'] / it.mean() / 1e6:.0f} GB/s")
import json
print("C3JSON " + json.dumps({"scale": scale, "iteration_ms": float(it.mean()), "kernel_ms": float(km.mean()),
                              "dominant_kernel_bytes": int(info["dominant_kernel_bytes"]),
                              "bytes_per_iteration": int(info["bytes_per_iteration"])}))
