"""How many gathers of the scale-26 XS layout go to 'cold' sources, and how many distinct
(tile, 128-B line) pairs they make (= compulsory fabric requests if nothing survives between units)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cugraph_amd import generators, pylibcugraph as plc, _capi
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
nv, ne = 1 << scale, 16 << scale
h = plc.ResourceHandle()
src, dst = generators.rmat(h, scale, ne, seed=42, scramble_vertex_ids=True)
g = plc.SGGraph(h, plc.GraphProperties(is_multigraph=True), src, dst, store_transposed=True, renumber=True,
                vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
del src, dst
lib = _capi.load()
ov, iv, wv, nm = (ctypes.c_void_p() for _ in range(4)); err = ctypes.c_void_p()
lib.cugraph_amd_graph_get_compressed(g.c_graph_ptr, ctypes.byref(ov), ctypes.byref(iv), ctypes.byref(wv), ctypes.byref(nm), ctypes.byref(err))
from cugraph_amd.session import _alias_device_memory
import numpy as np
off = _alias_device_memory(lib.cugraph_type_erased_device_array_view_pointer(ov), nv + 1, np.int32).long()
idx = _alias_device_memory(lib.cugraph_type_erased_device_array_view_pointer(iv), ne, np.int32)
deg = off[1:] - off[:-1]
n_xs = int((deg >= 4).sum())
print("xs rows", n_xs, "edges", int(off[n_xs]))
outdeg = torch.bincount(idx.long(), minlength=nv)
print("sources with out-degree 0:", int((outdeg == 0).sum()))
tot = int(outdeg.sum())
for thr in (1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 17, 1 << 18, 1 << 19):
    print("share of edges from the top", thr, "sources:", float(outdeg[:thr].sum()) / tot)
for thr in (1 << 20, 1 << 22, 6 << 20, 1 << 23, 12 << 20, 1 << 24):
    print("edges from sources of rank >=", thr, ":", int(outdeg[thr:].sum()), " such sources with outdeg>0:", int((outdeg[thr:] > 0).sum()))
