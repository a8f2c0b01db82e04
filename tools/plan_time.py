"""Time of the one-off steps at scale 26: edge generation, COO->CSC (+renumber), pull-plan build
(first session on the graph), a second session (plan cached)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cugraph_amd import generators, pylibcugraph as plc
from cugraph_amd.session import PagerankSession
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
weighted = len(sys.argv) > 2 and sys.argv[2] == "w"  # fp32 weights
nv, ne = 1 << scale, 16 << scale
h = plc.ResourceHandle()
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, time.perf_counter() - t0
(src, dst), t_gen = timed(lambda: generators.rmat(h, scale, ne, seed=42, scramble_vertex_ids=True))
wgt = generators.rmat_weights(h, ne, seed=43) if weighted else None
g, t_csc = timed(lambda: plc.SGGraph(h, plc.GraphProperties(is_multigraph=True), src, dst, weight_array=wgt, store_transposed=True, renumber=True,
                                      vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda")))
del src, dst
s1, t_plan = timed(lambda: PagerankSession(h, g, alpha=0.85, epsilon=0.0))
s2, t_sess = timed(lambda: PagerankSession(h, g, alpha=0.85, epsilon=0.0))
_, t_it = timed(lambda: s2.iterate(20))
print(f"scale {scale}{' weighted' if weighted else ''}: generate {t_gen:.3f} s, COO->CSC+renumber {t_csc:.3f} s, first session (plan build) {t_plan:.3f} s, "
      f"second session {t_sess:.3f} s, 20 iterations {t_it*1e3:.1f} ms; peak memory {torch.cuda.max_memory_allocated()/1e9:.1f} GB (torch only)")
free, total = torch.cuda.mem_get_info()
print(f"device memory in use now: {(total-free)/1e9:.1f} GB")
