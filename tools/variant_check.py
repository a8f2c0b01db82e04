"""PageRank at R-MAT scale S through the C ABI under whatever CUGRAPH_AMD_* knobs are set, against the
C oracle: max-abs and max-relative error, iteration counts.  usage: variant_check.py [scale]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from cugraph_amd import generators, pylibcugraph as plc  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nv, ne = 1 << scale, 16 << scale
h = plc.ResourceHandle()
src, dst = generators.rmat(h, scale, ne, seed=42, scramble_vertex_ids=True)
g = plc.SGGraph(h, plc.GraphProperties(is_multigraph=True), src, dst, store_transposed=True, renumber=True,
                vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
verts, ranks = plc.pagerank(h, g, None, None, None, None, 0.85, 1e-7, 100, False)
order = torch.argsort(verts)
got = ranks[order].cpu().numpy().astype(np.float64)
s, d = src.cpu().numpy(), dst.cpu().numpy()
off, idx, _ = oracle.compress(nv, d, s)
ref, it, conv = oracle.pagerank(nv, off, idx, None, dtype=np.float64, alpha=0.85, epsilon=1e-7, max_iterations=100)
err = np.abs(got - ref)
big = ref >= 1.0 / nv
rel = float(np.max(err[big] / ref[big]))
top = np.argsort(-ref)[:1000]
print(f"scale {scale} iters(oracle) {it} max-abs {err.max():.3e} max-rel(rank>=1/V) {rel:.3e} "
      f"max-rel(top1000) {float(np.max(err[top] / ref[top])):.3e} sum {got.sum():.9f}")
assert err.max() <= 1e-6 and rel <= 1e-4, "variant check failed"
