"""Local compute of ONE rank of a P-way 1D partition, on one GPU and without collectives: the same
local graph (per-rank degree order, compact source slots of all ranks) the multi-GPU driver builds
(tests/rank_emulation.py), then timed iterations.  usage: rank_emulate.py <scale> <P> [rank]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from cugraph_amd import generators, pylibcugraph as plc
from rank_emulation import build_rank

scale, world = int(sys.argv[1]), int(sys.argv[2])
me = int(sys.argv[3]) if len(sys.argv) > 3 else 0
private = len(sys.argv) > 4 and sys.argv[4] == "private"  # the personalised exchange's slot numbering
nv, ne = 1 << scale, 16 << scale
h = plc.ResourceHandle(rank=me, comm_size=world)
src, dst = generators.rmat(h, scale, ne, seed=42, scramble_vertex_ids=True)
R = build_rank(h, plc.GraphProperties(is_multigraph=True), src, dst, nv, world, me, private=private)
del src, dst
first, last, n_slots, hot, regions = R["first"], R["last"], R["n_slots"], R["hot"], R["regions"]
# how much of the gathered vector this rank's edges ever read (the case for a personalised exchange)
ref = torch.zeros(n_slots, dtype=torch.bool, device="cuda")
ref[R["local_src_slots"].long()] = True
n_ref = int(ref.sum())
mine = R["x_pos"][R["x_pos"] >= 0].long()
n_ref_remote = n_ref - int(ref[mine].sum())
n_own = int(mine.numel())
if private:
    print(f"rank {me} of {world}, private slots: {R['n_own']} own + {R['n_foreign']} foreign sources read = {n_slots} slots; "
          f"received per iteration {R['n_foreign'] * 4 / 1e6:.1f} MB", flush=True)
else:
  print(f"rank {me} of {world} reads {n_ref} of {n_slots} slots ({100.0 * n_ref / n_slots:.1f} %); "
        f"{n_ref_remote} of the {n_slots - n_own} slots of other ranks ({100.0 * n_ref_remote / max(n_slots - n_own, 1):.1f} %)",
        flush=True)
del ref
ids = torch.arange(first, last, dtype=torch.int32, device="cuda")
sums = R["outdeg"][first:last][R["order"]].to(torch.float32)
sess = R["engine"].make_session(R["graph"], alpha=0.85, epsilon=0.0, precomputed_vertex_out_weight_vertices=ids,
                                precomputed_vertex_out_weight_sums=sums)
sess.iterate(3)
torch.cuda.synchronize()
it, km = sess.timed_iterate(10)
print(f"scale {scale}, rank {me} of {world}: {last - first} rows, {R['n_local_edges']} edges, {n_slots} source slots "
      f"({'one region' if regions is None else f'regions of {hot} + {n_slots - hot} slots'}); "
      f"local iteration {it.mean():.3f} ms (dominant kernel {km.mean():.3f} ms) -> {ne / it.mean() / 1e6:.0f} GTEPS if every "
      f"rank took this long and the exchange were free; gathered vector {n_slots * 4 / 1e6:.0f} MB")
