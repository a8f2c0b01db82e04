"""Wall time of whole calls on a tiny graph (karate): graph creation, first pagerank call (builds the
plan), repeated calls -- the per-call overhead a user looping over small graphs sees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from cugraph_amd import pylibcugraph as plc
from helpers import load_csv, to_dev

src, dst, wgt = load_csv("karate.csv")
h = plc.ResourceHandle()
s, d, w = to_dev(src), to_dev(dst), to_dev(wgt)
torch.cuda.synchronize()
def timed(f, n=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return r, (time.perf_counter() - t0) / n
mk = lambda: plc.SGGraph(h, plc.GraphProperties(is_multigraph=False), s, d, weight_array=w, store_transposed=True,
                         renumber=True, do_expensive_check=False)
g, t_first_graph = timed(mk)
_, t_graph = timed(mk, 50)
run = lambda: plc.pagerank(h, g, None, None, None, None, 0.85, 1e-6, 500, False)
_, t_first = timed(run)
(v, r), t_call = timed(run, 200)
print(f"karate: graph create {t_first_graph*1e3:.2f} ms first, {t_graph*1e3:.3f} ms after; pagerank {t_first*1e3:.2f} ms first call, "
      f"{t_call*1e3:.3f} ms per call after (converges in tens of iterations)")
