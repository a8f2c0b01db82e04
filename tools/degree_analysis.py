"""In-degree structure of the bench graph: edges by row-degree class, tile sizes of the XS layout."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cugraph_amd import generators, pylibcugraph as plc
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
nv, ne = 1 << scale, 16 << scale
h = plc.ResourceHandle()
src, dst = generators.rmat(h, scale, ne, seed=42, scramble_vertex_ids=True)
indeg = torch.bincount(dst.long(), minlength=nv)
outdeg = torch.bincount(src.long(), minlength=nv)
del src, dst
d, order = torch.sort(indeg, descending=True)
cum = torch.cumsum(d, 0)
print("V", nv, "E", ne)
for thr in (1 << 20, 1 << 16, 16384, 4096, 2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 1):
    n = int((d >= thr).sum())
    print(f"rows with in-degree >= {thr:8d}: {n:10d} rows, {int(cum[n-1]) if n else 0:12d} edges ({(int(cum[n-1]) if n else 0)/ne:.3f})")
# tiles: <= 4095 rows, ~8M edges
rows_xs = int((d >= 4).sum())
offs = torch.cat([torch.zeros(1, dtype=torch.long, device=d.device), cum[:rows_xs]])
# correlation of in-degree rank and out-degree: share of out-edges by in-degree rank
od = outdeg[order]
oc = torch.cumsum(od, 0)
for thr in (1 << 13, 1 << 15, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24):
    print(f"out-edges of the top {thr} rows by in-degree: {int(oc[thr-1])/ne:.4f}")
iso = indeg == 0
print("rows without in-edges:", int(iso.sum()), " of those with out-edges:", int((iso & (outdeg > 0)).sum()),
      " with neither (dead):", int((iso & (outdeg == 0)).sum()))
low = (indeg >= 1) & (indeg <= 3)
print("rows of in-degree 1-3:", int(low.sum()), " of those with out-edges:", int((low & (outdeg > 0)).sum()))
