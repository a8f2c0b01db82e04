"""Local compute of ONE rank of a P-way 1D partition, on one GPU and without collectives: the same
local graph (per-rank degree order, compact source slots of all ranks) the multi-GPU driver builds,
then timed iterations.  usage: rank_emulate.py <scale> <P> [rank]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cugraph_amd import generators, mg, pylibcugraph as plc

scale, world = int(sys.argv[1]), int(sys.argv[2])
me = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nv, ne = 1 << scale, 16 << scale
h = plc.ResourceHandle(rank=me, comm_size=world)
props = plc.GraphProperties(is_multigraph=True)
src, dst = generators.rmat(h, scale, ne, seed=42, scramble_vertex_ids=True)
indeg = torch.zeros(nv, dtype=torch.int32, device="cuda")
outdeg = torch.zeros(nv, dtype=torch.int32, device="cuda")
one = torch.ones(1 << 28, dtype=torch.int32, device="cuda")
for b in range(0, ne, 1 << 28):  # in pieces: index_add_ wants int64 indices
    e = min(b + (1 << 28), ne)
    indeg.index_add_(0, dst[b:e].long(), one[:e - b])
    outdeg.index_add_(0, src[b:e].long(), one[:e - b])
del one
chunk = (nv + world - 1) // world
slot_of = torch.full((nv,), -1, dtype=torch.int32, device="cuda")
orders, actives = [], []
for r in range(world):
    first, last, _ = mg.partition_bounds(nv, world, r)
    order = torch.sort(indeg[first:last], descending=True, stable=True).indices
    # among the rows without in-edges, those some edge leaves first (the driver does the same: dead rows last)
    n_with = int(torch.count_nonzero(indeg[first:last]))
    tail = order[n_with:]
    order = torch.cat([order[:n_with], tail[torch.argsort((outdeg[first:last][tail] == 0).to(torch.int8), stable=True)]])
    orders.append(order)
    actives.append(outdeg[first:last][order] > 0)
slot_chunk = max(int(a.sum()) for a in actives)
slot_chunk = (slot_chunk + 3) // 4 * 4
eng = mg.HipEngine(h, props)
regions = mg.MGPagerank._choose_regions(eng, slot_chunk, world)  # CUGRAPH_AMD_MG_REGIONS=0/1, default from 4 ranks
n_slots, hot = (slot_chunk * world, 0) if regions is None else (world * sum(regions), world * regions[0])
for r in range(world):
    first, last, _ = mg.partition_bounds(nv, world, r)
    a = actives[r]
    j = torch.cumsum(a, 0) - 1
    if regions is None:
        slot = j + r * slot_chunk
    else:
        c0, c1 = regions
        slot = torch.where(j < c0, j + r * c0, j - c0 + hot + r * c1)
    x_pos = torch.where(a, slot, torch.full_like(orders[r], -1)).to(torch.int32)
    slot_of[first + orders[r]] = x_pos
    if r == me:
        my_x_pos = x_pos
first, last, _ = mg.partition_bounds(nv, world, me)
parts_s, parts_d = [], []
for b in range(0, ne, 1 << 28):  # in pieces: one mask over 4 G edges is more than torch indexes
    e = min(b + (1 << 28), ne)
    sel = (dst[b:e] >= first) & (dst[b:e] < last)
    parts_s.append(src[b:e][sel])
    parts_d.append(dst[b:e][sel])
s_l, d_l = torch.cat(parts_s), torch.cat(parts_d)
del src, dst, sel, parts_s, parts_d
rank_l = torch.empty_like(orders[me], dtype=torch.int32)
rank_l[orders[me]] = torch.arange(last - first, dtype=torch.int32, device="cuda")
if os.environ.get("CUGRAPH_AMD_RENUMBER_LOCALITY", "1") != "0":
    # this rank's rows of in-degree 1-3 in the order of their smallest source slot; its slots follow its rows
    # (mg.py: _order_low_degree_rows_by_source, then the slots are assigned again)
    rows = rank_l[(d_l - first).long()].long()
    deg = indeg[first:last][orders[me]].long()
    idx = torch.nonzero((deg >= 1) & (deg <= 3)).reshape(-1)
    minsrc = torch.full((last - first,), torch.iinfo(torch.int64).max, dtype=torch.int64, device="cuda")
    minsrc.scatter_reduce_(0, rows, slot_of[s_l.long()].long(), reduce="amin", include_self=True)
    key = (3 - deg[idx]) * (int(minsrc[idx].max()) + 1) + minsrc[idx]
    perm = torch.argsort(key, stable=True)
    f0 = int(idx[0])
    new_order = orders[me].clone()
    new_order[f0:f0 + perm.numel()] = orders[me][f0 + perm]
    orders[me] = new_order
    rank_l[new_order] = torch.arange(last - first, dtype=torch.int32, device="cuda")
    a = outdeg[first:last][new_order] > 0
    j = torch.cumsum(a, 0) - 1
    slot = j + me * slot_chunk if regions is None else torch.where(j < regions[0], j + me * regions[0],
                                                                    j - regions[0] + hot + me * regions[1])
    my_x_pos = torch.where(a, slot, torch.full_like(new_order, -1)).to(torch.int32)
    slot_of[first + new_order] = my_x_pos
    del rows, deg, idx, minsrc, key, perm
g = eng.make_local_graph(slot_of[s_l.long()], rank_l[(d_l - first).long()] + first, None, first, last, nv,
                         num_source_slots=n_slots, x_pos=my_x_pos, hot_slots=hot)
n_local_edges = int(s_l.numel())
# how much of the gathered vector this rank's edges ever read (the case for a personalised exchange)
ref = torch.zeros(n_slots, dtype=torch.bool, device="cuda")
ref[slot_of[s_l.long()].long()] = True
n_ref = int(ref.sum())
n_ref_remote = n_ref - int(ref[my_x_pos[my_x_pos >= 0].long()].sum())
n_own = int((my_x_pos >= 0).sum())
print(f"rank {me} of {world} reads {n_ref} of {n_slots} slots ({100.0 * n_ref / n_slots:.1f} %); "
      f"{n_ref_remote} of the {n_slots - n_own} slots of other ranks ({100.0 * n_ref_remote / max(n_slots - n_own, 1):.1f} %)",
      flush=True)
del ref, s_l, d_l
ids = torch.arange(first, last, dtype=torch.int32, device="cuda")
sums = outdeg[first:last][orders[me]].to(torch.float32)
sess = eng.make_session(g, alpha=0.85, epsilon=0.0, precomputed_vertex_out_weight_vertices=ids,
                        precomputed_vertex_out_weight_sums=sums)
sess.iterate(3)
torch.cuda.synchronize()
it, km = sess.timed_iterate(10)
print(f"scale {scale}, rank {me} of {world}: {last - first} rows, {n_local_edges} edges, {n_slots} source slots "
      f"({'one region' if regions is None else f'regions of {hot} + {n_slots - hot} slots'}); "
      f"local iteration {it.mean():.3f} ms (dominant kernel {km.mean():.3f} ms) -> {ne / it.mean() / 1e6:.0f} GTEPS if every "
      f"rank took this long and the exchange were free; gathered vector {n_slots * 4 / 1e6:.0f} MB")
