"""The local graph of ONE rank of a P-way 1D partition, built on one GPU without collectives exactly as
cugraph_amd.mg.MGPagerank.from_edges builds it (per-rank degree order with the untouched vertices last,
rows of in-degree 1-3 by their smallest source slot, compact source slots of ALL ranks, two regions from 4
ranks).  Test infrastructure, also used by tools/rank_emulate.py."""
import os

import torch


def _count(ids, nv):
    out = torch.zeros(nv, dtype=torch.int32, device="cuda")
    one = torch.ones(min(ids.numel(), 1 << 28), dtype=torch.int32, device="cuda")
    for b in range(0, ids.numel(), 1 << 28):  # in pieces: index_add_ wants int64 indices
        e = min(b + (1 << 28), ids.numel())
        out.index_add_(0, ids[b:e].long(), one[:e - b])
    return out


def build_rank(handle, props, src, dst, nv, world, me, *, locality=None, private=False):
    """src, dst: the WHOLE edge list (device int32).  Returns a dict: engine, graph, first, last,
    order (row -> offset of the caller's id in the range), x_pos, slot_of (caller's id -> slot, all ranks),
    indeg, outdeg, n_slots, hot, regions, n_local_edges."""
    from cugraph_amd import mg

    ne = src.numel()
    indeg, outdeg = _count(dst, nv), _count(src, nv)
    slot_of = torch.full((nv,), -1, dtype=torch.int32, device="cuda")
    orders, actives = [], []
    for r in range(world):
        first, last, _ = mg.partition_bounds(nv, world, r)
        order = torch.sort(indeg[first:last], descending=True, stable=True).indices
        # among the rows without in-edges, those some edge leaves first (dead rows last)
        n_with = int(torch.count_nonzero(indeg[first:last]))
        tail = order[n_with:]
        order = torch.cat([order[:n_with],
                           tail[torch.argsort((outdeg[first:last][tail] == 0).to(torch.int8), stable=True)]])
        orders.append(order)
        actives.append(outdeg[first:last][order] > 0)
    slot_chunk = max(int(a.sum()) for a in actives)
    slot_chunk = (slot_chunk + 3) // 4 * 4
    eng = mg.HipEngine(handle, props)
    regions = mg.MGPagerank._choose_regions(eng, slot_chunk, world)
    n_slots, hot = (slot_chunk * world, 0) if regions is None else (world * sum(regions), world * regions[0])

    def slots_of(r, active):
        j = torch.cumsum(active, 0) - 1
        if regions is None:
            slot = j + r * slot_chunk
        else:
            c0, c1 = regions
            slot = torch.where(j < c0, j + r * c0, j - c0 + hot + r * c1)
        return torch.where(active, slot, torch.full_like(slot, -1)).to(torch.int32)

    my_x_pos = None
    for r in range(world):
        first, last, _ = mg.partition_bounds(nv, world, r)
        x_pos = slots_of(r, actives[r])
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
    del parts_s, parts_d
    rank_l = torch.empty_like(orders[me], dtype=torch.int32)
    rank_l[orders[me]] = torch.arange(last - first, dtype=torch.int32, device="cuda")
    if locality is None:
        locality = os.environ.get("CUGRAPH_AMD_RENUMBER_LOCALITY", "1") != "0"
    if locality:
        # this rank's rows of in-degree 1-3 in the order of their smallest source slot; its slots follow its
        # rows (mg.py: _order_low_degree_rows_by_source, then the slots are assigned again)
        rows = rank_l[(d_l - first).long()].long()
        deg = indeg[first:last][orders[me]].long()
        idx = torch.nonzero((deg >= 1) & (deg <= 3)).reshape(-1)
        if idx.numel() > 1:
            minsrc = torch.full((last - first,), torch.iinfo(torch.int64).max, dtype=torch.int64, device="cuda")
            minsrc.scatter_reduce_(0, rows, slot_of[s_l.long()].long(), reduce="amin", include_self=True)
            key = (3 - deg[idx]) * (int(minsrc[idx].max()) + 1) + minsrc[idx]
            perm = torch.argsort(key, stable=True)
            f0 = int(idx[0])
            new_order = orders[me].clone()
            new_order[f0:f0 + perm.numel()] = orders[me][f0 + perm]
            orders[me] = new_order
            rank_l[new_order] = torch.arange(last - first, dtype=torch.int32, device="cuda")
            my_x_pos = slots_of(me, outdeg[first:last][new_order] > 0)
            slot_of[first + new_order] = my_x_pos
    n_own = n_foreign = 0
    if private:
        # the PERSONALISED exchange's numbering (mg.py: assign_private_slots): this rank's own sources
        # first, in row order, then the sources of the other ranks that its edges read, owner by owner
        # in the owner's row order -- the order of the shared slot numbers (one region), so: mark the
        # shared slots read, number the marked ones
        assert regions is None, "private slots have one region"
        shared = slot_of[s_l.long()].long()
        lo, hi = me * slot_chunk, (me + 1) * slot_chunk
        read = torch.zeros(n_slots + 1, dtype=torch.int32, device="cuda")
        read[shared] = 1
        read[lo:hi] = 0
        pos = torch.cumsum(read, 0, dtype=torch.int32) - read  # index among the foreign slots read
        n_own = int((my_x_pos >= 0).sum())
        n_foreign = int(read.sum())
        own = (shared >= lo) & (shared < hi)
        local_slots = torch.where(own, shared - lo, n_own + pos[shared].long()).to(torch.int32)
        del shared, own, read, pos
        my_x_pos = torch.where(my_x_pos >= 0, my_x_pos - lo, my_x_pos).to(torch.int32)
        n_slots = max(n_own + n_foreign, 1)
    else:
        local_slots = slot_of[s_l.long()]
    g = eng.make_local_graph(local_slots, rank_l[(d_l - first).long()] + first, None, first, last, nv,
                             num_source_slots=n_slots, x_pos=my_x_pos, hot_slots=hot)
    return dict(engine=eng, graph=g, first=first, last=last, order=orders[me], x_pos=my_x_pos, slot_of=slot_of,
                indeg=indeg, outdeg=outdeg, n_slots=n_slots, hot=hot, regions=regions,
                n_local_edges=int(s_l.numel()), local_src_slots=local_slots, n_own=n_own, n_foreign=n_foreign)
