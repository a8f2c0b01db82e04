"""Multi-GPU PageRank: 1D destination-vertex partition over the GPUs of one node.

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Replaces, for this path, the reference's Dask layer
(python/cugraph/cugraph/dask/link_analysis/pagerank.py:174-440), its 2D edge partitioning
(cpp/include/cugraph/partition_manager.hpp:32-145) and the raft-comms collectives listed in
SURVEY.md section 2.3:

  build      : edges are routed to the owner of their DESTINATION with one all-to-all
               (reference: shuffle over the 2D grid + per-block renumbering);
               out-weight sums are per SOURCE, so one all-reduce of a V-length vector gives every
               rank the sums of its own vertices (reference: device_reduce over major_comm,
               per_v_transform_reduce_incoming_outgoing_e.cuh:1094-1108);
  iteration  : local pull SpMV with the fused epilogue (same kernels as one GPU; a 1D
               destination partition has no partial sums, so the reference's reduce over
               minor_comm, :955-1013, disappears), then
               ONE all-gather of the source-value vector x (reference: grouped device_bcast,
               update_edge_src_dst_property.cuh:559-588) and
               ONE all-reduce of {dangling mass, L1 delta} (reference: two blocking
               host_scalar_allreduce, reduce_v.cuh:141-144);
  stop       : decided on the device from the all-reduced delta, identically on every rank.

Rank r owns the destinations [bounds[r], bounds[r+1]): contiguous ranges cut so that every rank
receives about E / P in-edges (`edge_balanced_bounds`: bisection on the global in-degree prefix, one
all-reduced coarse histogram plus an exact one inside the P-1 buckets that hold a cut; SURVEY 8e --
the reference balances through its 2D grid, partition_manager.hpp:32-145).  Ownership of a vertex is
a search of the P+1 bounds.  CUGRAPH_AMD_MG_PARTITION=equal keeps equal vertex ranges
(chunk = ceil(V / P)), which is also the fallback for a graph without edges.

Inside its range every rank relabels its vertices by descending in-degree (the reference renumbers
by degree inside each partition too, renumber_edgelist_impl.cuh); the ranks all-gather the
old -> new table once and relabel their sources with it.  Ownership does not change, so the
collectives stay what they were, and the local SpMV runs on contiguous degree buckets and XS
tiles exactly like the single-GPU path.  Callers keep seeing the ids they passed in.

Compact source space: only a vertex with out-edges is ever gathered (in an R-MAT scale-26 graph
27 M of 67 M).  Each rank numbers its vertices that have out-edges 0..n_r-1 in row order; the
vector the ranks all-gather has P chunks of max_r n_r slots, edge sources are stored as slots,
and a rank writes vertex v's value to slot x_pos[v] (or nowhere).  The all-gather shrinks by the
share of vertices without out-edges and the gathered vector gets denser.

The local compute is reached through an *engine* object so that the driver logic (partition,
routing, collectives, stop rule, result assembly) can be exercised at world_size 2 on CPU with
a test double; the product engine is `HipEngine` and there is no other in this package.
"""
import os
import time

import numpy as np


def partition_bounds(num_vertices, world_size, rank):
    """[first, last) of the destinations rank owns; chunk = ceil(V / P)."""
    chunk = (num_vertices + world_size - 1) // world_size
    first = min(rank * chunk, num_vertices)
    last = min(first + chunk, num_vertices)
    return first, last, chunk


def equal_bounds(num_vertices, world_size):
    """[0, chunk, 2*chunk, ..., V] clipped to V: P+1 bounds of equal vertex ranges."""
    chunk = (num_vertices + world_size - 1) // world_size
    return [min(r * chunk, num_vertices) for r in range(world_size + 1)]


def edge_balanced_bounds(dst, num_vertices, world, group=None, coarse_buckets=1 << 16):
    """P+1 bounds of contiguous destination ranges that hold about E / P in-edges each.  dst: this
    rank's share of the destination ids (any share: the histograms are all-reduced).  Two levels:
    in-edge counts of at most `coarse_buckets` equal runs of vertices find the run every cut falls
    into; per-vertex counts inside those P-1 runs place it exactly.  The same on every rank."""
    import torch

    nv = int(num_vertices)
    if world == 1 or nv == 0:
        return equal_bounds(nv, world)
    dist = _dist()
    width = max(1, -(-nv // coarse_buckets))
    nb = -(-nv // width)
    d64 = dst.to(torch.int64)
    hist = torch.bincount(torch.div(d64, width, rounding_mode="floor"), minlength=nb).to(torch.int64)
    all_reduce(hist, group)
    prefix = torch.cumsum(hist, 0).cpu().numpy()  # edges of buckets 0..b
    total = int(prefix[-1]) if nb else 0
    if total == 0:
        return equal_bounds(nv, world)
    targets = [total * r // world for r in range(1, world)]
    cut_bucket = [int(np.searchsorted(prefix, t, side="left")) for t in targets]  # first bucket reaching t
    if width == 1:
        fine = None
    else:
        fine = torch.zeros((world - 1, width), dtype=torch.int64, device=dst.device)
        for i, b in enumerate(cut_bucket):
            lo = b * width
            inside = d64[(d64 >= lo) & (d64 < lo + width)] - lo
            fine[i] = torch.bincount(inside, minlength=width)
        all_reduce(fine, group)
        fine = fine.cpu().numpy()
    bounds = [0]
    for i, (t, b) in enumerate(zip(targets, cut_bucket)):
        before = int(prefix[b - 1]) if b > 0 else 0
        counts = np.array([int(prefix[b]) - before]) if fine is None else fine[i]
        upto = before + np.cumsum(counts)  # edges of the vertices up to and including lo + j
        j = int(np.searchsorted(upto, t, side="left"))  # first vertex with which the target is reached
        # cut after that vertex or before it, whichever leaves the smaller error
        after = int(upto[j]) if j < len(upto) else int(upto[-1])
        ahead = int(upto[j - 1]) if j > 0 else before
        cut = b * width + (j + 1 if abs(after - t) <= abs(t - ahead) else j)
        bounds.append(min(max(cut, bounds[-1]), nv))
    bounds.append(nv)
    return bounds


def all_gather_ragged(local, lengths, rank, world, group=None, fill=0):
    """Concatenation of every rank's `local` (rank r's has lengths[r] elements): padded to the
    longest, one all-gather, the pads dropped."""
    import torch

    longest = max(max(lengths), 1)
    buf = torch.full((world * longest,), fill, dtype=local.dtype, device=local.device)
    buf[rank * longest:rank * longest + lengths[rank]] = local
    if world > 1:
        all_gather_in_place(buf, longest, rank, world, group)
    return torch.cat([buf[r * longest:r * longest + lengths[r]] for r in range(world)])


def _dist():
    import torch.distributed as dist

    return dist


def exchange_mode(world):
    """How the source values travel each iteration.  "personal" (the default from 2 ranks on): every
    rank numbers its slots privately -- its own sources, then exactly the foreign ones its edges read
    -- and one all-to-all delivers those; at 8 ranks a rank reads 47 % of the others' sources, so it
    receives half the bytes of an all-gather and gathers from a vector half as long (one rank of 8 at
    scale 28: 160 instead of 350 MB, 1.44 instead of 1.52 ms of compute; profiles/r3_mg_table.md).
    "gather": the all-gather of every rank's chunk (CUGRAPH_AMD_MG_EXCHANGE=gather; implied by two
    regions, CUGRAPH_AMD_MG_REGIONS, which only the all-gather layout has)."""
    env = os.environ.get("CUGRAPH_AMD_MG_EXCHANGE")
    if env in ("personal", "gather"):
        return env
    return "gather" if os.environ.get("CUGRAPH_AMD_MG_REGIONS", "0") in ("1", "auto") else "personal"


def _gather_through_host(x_full, chunk, rank, world, group):
    """gloo with device tensors (ranks of a rehearsal sharing one GPU): its flat in-place form
    never completes there, so the chunks travel as host copies."""
    import torch

    mine = x_full[rank * chunk:(rank + 1) * chunk].cpu()
    parts = [torch.empty_like(mine) for _ in range(world)]
    _dist().all_gather(parts, mine, group=group)
    x_full[:chunk * world].copy_(torch.cat(parts))


def _gloo_on_device(t, group):
    return t.is_cuda and _dist().get_backend(group) == "gloo"


def all_reduce(t, group=None, op=None):
    """In-place all-reduce (sum by default); device tensors over gloo travel as host copies."""
    dist = _dist()
    op = dist.ReduceOp.SUM if op is None else op
    if _gloo_on_device(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def all_to_all_single(out, a, out_split=None, in_split=None, group=None):
    dist = _dist()
    if _gloo_on_device(a, group):
        h = out.cpu()
        dist.all_to_all_single(h, a.cpu(), output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        out.copy_(h)
    else:
        dist.all_to_all_single(out, a, output_split_sizes=out_split, input_split_sizes=in_split, group=group)


def all_gather_in_place(x_full, chunk, rank, world, group=None):
    """x_full[r*chunk:(r+1)*chunk] of every rank -> x_full of every rank."""
    dist = _dist()
    if _gloo_on_device(x_full, group):
        return _gather_through_host(x_full, chunk, rank, world, group)
    mine = x_full[rank * chunk:(rank + 1) * chunk]
    try:
        dist.all_gather_into_tensor(x_full, mine, group=group)
    except (RuntimeError, NotImplementedError):  # backends without the flat form (gloo)
        parts = list(x_full.split(chunk))
        dist.all_gather(parts, mine.clone(), group=group)


class _Done:
    """Stand-in for the work handle of a collective that already ran (backends without the async flat form)."""

    def wait(self):
        return True


def all_gather_region_async(region, sub_chunk, rank, world, group=None):
    """In-place all-gather of one region of the gathered vector (P equal sub-chunks, rank r's is
    the r-th), started without waiting: returns a handle whose wait() orders the current stream
    behind it."""
    dist = _dist()
    if _gloo_on_device(region, group):
        _gather_through_host(region, sub_chunk, rank, world, group)
        return _Done()
    mine = region[rank * sub_chunk:(rank + 1) * sub_chunk]
    try:
        return dist.all_gather_into_tensor(region, mine, group=group, async_op=True)
    except (RuntimeError, NotImplementedError):  # gloo: no flat form
        parts = list(region.split(sub_chunk))
        dist.all_gather(parts, mine.clone(), group=group)
        return _Done()


def shuffle_edges_to_owner(engine, src, dst, wgt, bounds, world, group=None):
    """Route every edge to the rank that owns its destination: the engine groups this rank's share by
    owner (the library's stable multisplit, cugraph_amd_partition_group_edges: one counting pass, one
    scan, one scatter of the edge arrays themselves -- round 2 did this with a bucketize, one
    torch.nonzero per owner and three E-sized gathers), then one all-to-all per array.
    bounds: the P+1 bounds of the destination ranges."""
    import torch

    src, dst, wgt, in_split = engine.group_edges(src, dst, wgt, bounds)
    send_counts = torch.tensor(in_split, dtype=torch.int64, device=dst.device)
    recv_counts = torch.empty_like(send_counts)
    all_to_all_single(recv_counts, send_counts, group=group)
    out_split = [int(c) for c in recv_counts.tolist()]

    def route(a):
        out = torch.empty(sum(out_split), dtype=a.dtype, device=a.device)
        all_to_all_single(out, a, out_split, in_split, group)
        return out

    return route(src), route(dst), (None if wgt is None else route(wgt))


def reduce_to_owners(full, bounds, rank, world, group=None):
    """Sum over the ranks of a V-length vector, every rank keeping only ITS range [bounds[rank],
    bounds[rank + 1]): one reduce-scatter over ranges padded to the longest (RCCL), where round 2
    all-reduced the whole vector to every rank -- 1/P of the bytes received.  Backends without a
    reduce-scatter (gloo) fall back to the all-reduce."""
    import torch

    dist = _dist()
    first, last = bounds[rank], bounds[rank + 1]
    if world == 1:
        return full[first:last].clone()
    if dist.get_backend(group) != "nccl":
        all_reduce(full, group)
        return full[first:last].clone()
    longest = max(max(b - a for a, b in zip(bounds, bounds[1:])), 1)
    padded = torch.zeros(world * longest, dtype=full.dtype, device=full.device)
    for r in range(world):
        padded[r * longest:r * longest + bounds[r + 1] - bounds[r]] = full[bounds[r]:bounds[r + 1]]
    mine = torch.empty(longest, dtype=full.dtype, device=full.device)
    dist.reduce_scatter_tensor(mine, padded, group=group)
    return mine[:last - first].clone()


class HipEngine:
    """Local compute on one MI355X through the C ABI."""

    def __init__(self, resource_handle, graph_properties):
        self.handle = resource_handle
        self.props = graph_properties
        self.device = "cuda"

    def degree_order(self, dst, first, last):
        """(order, rank) int32 tensors of the local destinations, descending in-degree."""
        import ctypes

        import torch

        from . import _capi
        from .pylibcugraph.utils import create_view_from_py_obj, free_view

        n = last - first
        order = torch.empty(n, dtype=torch.int32, device="cuda")
        rank = torch.empty(n, dtype=torch.int32, device="cuda")
        views = [create_view_from_py_obj(t) for t in (dst.contiguous(), order, rank)]
        err = ctypes.c_void_p()
        try:
            code = _capi.load().cugraph_amd_partition_degree_order(
                self.handle.c_resource_handle_ptr, views[0], first, last, views[1], views[2],
                ctypes.byref(err))
            _capi.check(code, err, "cugraph_amd_partition_degree_order")
        finally:
            for v in views:
                free_view(v)
        return order, rank

    def group_edges(self, src, dst, wgt, bounds):
        """(src, dst, wgt, counts): the edges grouped by the owner of their destination, input order kept
        inside a group; counts[r] = edges for rank r (cugraph_amd_partition_group_edges)."""
        import ctypes

        import torch

        from . import _capi
        from .pylibcugraph.utils import create_view_from_py_obj, free_view

        world = len(bounds) - 1
        src, dst = src.contiguous(), dst.contiguous()
        wgt = None if wgt is None else wgt.contiguous()
        outs = [torch.empty_like(src), torch.empty_like(dst), None if wgt is None else torch.empty_like(wgt)]
        views = [None if t is None else create_view_from_py_obj(t) for t in (src, dst, wgt, *outs)]
        c_bounds = (ctypes.c_int64 * (world + 1))(*[int(b) for b in bounds])
        c_counts = (ctypes.c_int64 * world)()
        err = ctypes.c_void_p()
        try:
            code = _capi.load().cugraph_amd_partition_group_edges(
                self.handle.c_resource_handle_ptr, views[0], views[1], views[2], c_bounds, world, views[3],
                views[4], views[5], c_counts, ctypes.byref(err))
            _capi.check(code, err, "cugraph_amd_partition_group_edges")
        finally:
            for v in views:
                if v is not None:
                    free_view(v)
        return outs[0], outs[1], outs[2], [int(c) for c in c_counts]

    def make_local_graph(self, src, dst, wgt, first, last, num_global, num_source_slots=None,
                         x_pos=None, hot_slots=0):
        from . import pylibcugraph as plc

        return plc.MGGraph(self.handle, self.props, src, dst, wgt, vertex_first=first,
                           vertex_last=last, num_global_vertices=num_global,
                           num_source_slots=num_source_slots, x_pos=x_pos, hot_slots=hot_slots)

    def window_slots(self, num_source_slots):
        """Granule of a region of the gathered vector (cugraph_amd_xs_window_slots)."""
        from . import _capi

        return int(_capi.load().cugraph_amd_xs_window_slots(int(num_source_slots)))

    def source_weight_sums(self, src, wgt, num_global):
        """float64[num_global]: per-source sum of the weights (edge counts when wgt is None) of
        this rank's edges."""
        import ctypes

        import torch

        from . import _capi
        from .pylibcugraph.utils import create_view_from_py_obj, free_view

        sums = torch.zeros(num_global, dtype=torch.float64, device="cuda")
        views = [create_view_from_py_obj(t) for t in (src.contiguous(),
                                                     None if wgt is None else wgt.contiguous(), sums)]
        err = ctypes.c_void_p()
        try:
            code = _capi.load().cugraph_amd_edgelist_source_weight_sums(
                self.handle.c_resource_handle_ptr, views[0], views[1], views[2], ctypes.byref(err))
            _capi.check(code, err, "cugraph_amd_edgelist_source_weight_sums")
        finally:
            for v in views:
                free_view(v)
        return sums

    def make_session(self, graph, **kw):
        from .session import PagerankSession

        return PagerankSession(self.handle, graph, **kw)


class MGSession:
    def __init__(self, inner):
        self.inner = inner
        self.pending = None  # (region 0, region 1) all-gathers of the current vector still in flight


class MGPagerank:
    """Driver of a 1D-partitioned PageRank.  Construct with from_edges / from_rmat."""

    def __init__(self, engine, num_vertices, rank, world, group=None):
        self.engine = engine
        self.always_exchange = False  # run the collectives even at world size 1 (tests)
        self.num_vertices = int(num_vertices)
        self.rank, self.world, self.group = rank, world, group
        self.set_bounds(equal_bounds(self.num_vertices, world))
        self.graph = None
        self.order = None  # int32[n_local]: internal position -> offset of the caller's id in the range
        self.rank_local = None  # int32[n_local]: offset of the caller's id in the range -> internal position
        self.slot_chunk = 0  # slots per rank of the gathered vector (compact source space)
        # two regions of the gathered vector: (C0, C1) = per-rank sub-chunk of the hot region
        # [0, P*C0) and of the rest [P*C0, P*C0 + P*C1); None: one region of P chunks
        self.regions = None
        self.out_w = None  # out-weight sums of the local vertices (float64), internal order
        # personalised exchange (CUGRAPH_AMD_MG_EXCHANGE=personal): own sources, foreign sources read,
        # per-rank split sizes of the all-to-all and the indices (among the own sources) to pack
        self.n_own = self.n_foreign = 0
        self.want_split = self.give_split = self.idx_give = None
        self.value_dtype = None
        self.vertex_dtype = None
        self.t_generate = 0.0
        self.t_build = 0.0

    def set_bounds(self, bounds):
        self.bounds = [int(b) for b in bounds]
        self.lengths = [b - a for a, b in zip(self.bounds, self.bounds[1:])]
        self.first, self.last = self.bounds[self.rank], self.bounds[self.rank + 1]

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_edges(cls, engine, src, dst, wgt, num_vertices, *, rank=None, world=None, group=None):
        """src/dst/wgt: this rank's ARBITRARY share of the global edge list (as each Dask worker
        holds in the reference)."""
        import torch

        dist = _dist()
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
        self = cls(engine, num_vertices, rank, world, group)
        t0 = time.perf_counter()
        marks = [("start", t0)]

        def mark(name):  # CUGRAPH_AMD_MG_PROFILE=1: where the set-up time goes (synchronises at every mark)
            if os.environ.get("CUGRAPH_AMD_MG_PROFILE") == "1":
                if engine.device == "cuda":
                    torch.cuda.synchronize()
                marks.append((name, time.perf_counter()))

        self.vertex_dtype = src.dtype
        self.value_dtype = torch.float32 if wgt is None else wgt.dtype
        if os.environ.get("CUGRAPH_AMD_MG_PARTITION", "edges") != "equal":
            self.set_bounds(edge_balanced_bounds(dst, self.num_vertices, world, group))
        mark("bounds")
        if world > 1:
            src, dst, wgt = shuffle_edges_to_owner(engine, src, dst, wgt, self.bounds, world, group)
        mark("shuffle")
        # out-edge counts and out-weight sums per source: every rank tallies the sources of ITS edges
        # over the whole vertex range and the owners receive the sums of their own vertices
        counts = engine.source_weight_sums(src, None, self.num_vertices)
        own_sums = None
        if wgt is not None:
            own_sums = reduce_to_owners(engine.source_weight_sums(src, wgt, self.num_vertices), self.bounds, rank,
                                        world, group)
        own_counts = reduce_to_owners(counts, self.bounds, rank, world, group)
        del counts
        if own_sums is None:
            own_sums = own_counts
        mark("source sums")
        # rows: descending in-degree inside the owned range
        self.order, self.rank_local = engine.degree_order(dst, self.first, self.last)
        mark("degree order")
        self._sources_first_among_rows_without_in_edges(dst, own_counts)
        mark("sources first")

        def assign_slots():
            """slots: this rank's vertices with out-edges, numbered in row order (hottest first);
            returns x_pos (row -> slot or -1), the caller's id -> slot table of every rank's range
            (one all-gather of the per-rank tables), the slot count and the hot slots"""
            order = self.order.long()
            active = own_counts[order] > 0
            n_active = torch.count_nonzero(active).reshape(1).to(torch.int64)
            if world > 1:
                all_reduce(n_active, group, dist.ReduceOp.MAX)
            self.slot_chunk = max(int(n_active.item()), 1)
            self.regions = self._choose_regions(engine, self.slot_chunk, world)
            j = torch.cumsum(active, 0) - 1  # index of a row among this rank's sources
            if self.regions is None:
                slot = j + rank * self.slot_chunk
                n_slots, hot_slots = self.slot_chunk * world, 0
            else:
                c0, c1 = self.regions
                slot = torch.where(j < c0, j + rank * c0, j - c0 + world * c0 + rank * c1)
                n_slots, hot_slots = world * (c0 + c1), world * c0
            x_pos = torch.where(active, slot, torch.full_like(order, -1)).to(torch.int32)
            mine = torch.full((self.last - self.first,), -1, dtype=torch.int32, device=src.device)
            mine[order] = x_pos
            slot_of = all_gather_ragged(mine, self.lengths, rank, world, group, fill=-1)
            return x_pos, slot_of, n_slots, hot_slots

        def assign_private_slots():
            """The PERSONALISED exchange (CUGRAPH_AMD_MG_EXCHANGE=personal): a slot numbering of this
            rank's own -- its sources first, in row order, then one block per other rank holding
            exactly the sources of that rank which this rank's edges read, in that rank's row
            order.  Returns x_pos, the slot of every local edge's source, the number of own sources
            and the sorted (owner << 32 | index among the owner's sources) keys of the foreign ones."""
            order = self.order.long()
            active = own_counts[order] > 0
            n_own = int(torch.count_nonzero(active))
            j = torch.cumsum(active, 0) - 1
            x_pos = torch.where(active, j, torch.full_like(order, -1)).to(torch.int32)
            mine = torch.full((self.last - self.first,), -1, dtype=torch.int32, device=src.device)
            mine[order] = x_pos
            index_at_owner = all_gather_ragged(mine, self.lengths, rank, world, group, fill=-1)
            # the sources this rank's edges read, as a mark per vertex (V-sized; round 2 sorted the
            # E-sized key list with torch.unique): the marked foreign vertices, numbered owner by
            # owner in the owner's row order, are the slots behind this rank's own sources
            read = torch.zeros(self.num_vertices, dtype=torch.bool, device=src.device)
            read[src] = True
            read[self.first:self.last] = False
            ids = torch.nonzero(read).reshape(-1)  # ascending vertex id: owner-major
            del read
            inner = torch.as_tensor(self.bounds[1:-1], dtype=torch.int64, device=src.device)
            owner = torch.bucketize(ids, inner, right=True)
            j_ids = index_at_owner[ids].long()
            key = (owner << 32) + j_ids  # V-sized at most: the owner's row order inside each owner
            uniq, perm = torch.sort(key)
            slot_of_vertex = torch.full((self.num_vertices,), -1, dtype=torch.int64, device=src.device)
            slot_of_vertex[ids[perm]] = n_own + torch.arange(ids.numel(), dtype=torch.int64, device=src.device)
            slot_of_vertex[self.first:self.last] = index_at_owner[self.first:self.last].long()
            slot = slot_of_vertex[src]
            return x_pos, slot, n_own, uniq

        # (with one rank only when asked for by name: a test of the path itself, its foreign block is empty)
        personal = exchange_mode(world) == "personal" and (world > 1 or os.environ.get("CUGRAPH_AMD_MG_EXCHANGE") == "personal")
        if personal:
            x_pos, src_slot, n_own, uniq = assign_private_slots()
        else:
            x_pos, slot_of, n_slots, hot_slots = assign_slots()
        mark("slots")
        # rows of in-degree 1-3 in the order of their smallest source (decided for all ranks alike:
        # the slots are assigned again afterwards, which is a collective)
        n_edges_all = torch.tensor([src.numel()], dtype=torch.int64, device=src.device)
        if world > 1:
            all_reduce(n_edges_all, group)
        env = os.environ.get("CUGRAPH_AMD_RENUMBER_LOCALITY")
        if env == "1" or (env is None and int(n_edges_all.item()) >= (1 << 22)):
            rows = self.rank_local[(dst - self.first).long()].long()
            self._order_low_degree_rows_by_source(src_slot if personal else slot_of[src].long(), rows)
            del rows, x_pos
            # slots follow the rows, so that a rank still writes its values in row order
            if personal:
                x_pos, src_slot, n_own, uniq = assign_private_slots()
            else:
                del slot_of
                x_pos, slot_of, n_slots, hot_slots = assign_slots()
        mark("low rows by source + slots again")
        if personal:
            # every owner learns which of its sources each rank reads (counts, then the index lists)
            want = torch.bincount(uniq >> 32, minlength=world)
            give = torch.empty_like(want)
            all_to_all_single(give, want, group=group)
            self.want_split = [int(c) for c in want.tolist()]
            self.give_split = [int(c) for c in give.tolist()]
            self.idx_give = torch.empty(sum(self.give_split), dtype=torch.int64, device=src.device)
            all_to_all_single(self.idx_give, (uniq & 0xFFFFFFFF).contiguous(), self.give_split, self.want_split, group)
            self.n_own, self.n_foreign = n_own, int(uniq.numel())
            n_slots, hot_slots = max(n_own + self.n_foreign, 1), 0
            self.regions, self.slot_chunk = None, 0
            engine.private_slots = True  # (the CPU stand-in checks the chunk layout otherwise)
            src = src_slot.to(self.vertex_dtype)
            del src_slot, uniq
        else:
            src = slot_of[src].to(self.vertex_dtype)
            del slot_of
        dst = (self.rank_local[dst - self.first] + self.first).to(self.vertex_dtype)
        order = self.order.long()
        mark("relabel")
        self.graph = engine.make_local_graph(src, dst, wgt, self.first, self.last, self.num_vertices,
                                             num_source_slots=n_slots, x_pos=x_pos, hot_slots=hot_slots)
        del src, dst, wgt, x_pos
        self.out_w = own_sums[order].clone()  # internal (row) order
        del own_sums, own_counts
        mark("local graph")
        self.t_build = time.perf_counter() - t0
        if len(marks) > 1:
            import sys
            print("MGPROF rank %d: " % rank + ", ".join("%s %.3f s" % (n, t - marks[i][1]) for i, (n, t) in enumerate(marks[1:])),
                  file=sys.stderr, flush=True)
        return self

    def _sources_first_among_rows_without_in_edges(self, dst, own_counts):
        """Among the rows without in-edges (the tail of the degree order), those some edge leaves
        come first: what is left at the very end are the vertices no edge touches, which the
        iteration does not visit (pagerank.hip: dead rows; renumber.hip does the same on one GPU).
        Updates self.order / self.rank_local."""
        import torch

        n_local = self.last - self.first
        if n_local == 0:
            return
        rows = self.rank_local[(dst - self.first).long()].long()
        deg = torch.bincount(rows, minlength=n_local)
        n_with = int(torch.count_nonzero(deg))
        if n_local - n_with < 2:
            return
        tail = self.order[n_with:]
        keep = torch.argsort((own_counts[tail.long()] == 0).to(torch.int8), stable=True)
        order = self.order.clone()
        order[n_with:] = tail[keep]
        self.order = order
        rank_local = torch.empty_like(self.rank_local)
        rank_local[order.long()] = torch.arange(n_local, dtype=self.rank_local.dtype, device=order.device)
        self.rank_local = rank_local

    def _order_low_degree_rows_by_source(self, src_slots, rows):
        """Rows of in-degree 1, 2 and 3 in the order of their smallest source slot, inside each of
        the three classes (the single-GPU renumbering does the same, renumber.hip
        refine_low_degree_order: the thread-per-row bucket then gathers from neighbouring slots).
        Rows only change places inside their degree class, so the degree order stands.
        src_slots / rows: (provisional) source slot and current row of every local edge.
        Updates self.order / self.rank_local."""
        import torch

        n_local = self.last - self.first
        if n_local == 0 or rows.numel() == 0:
            return
        deg = torch.bincount(rows, minlength=n_local)
        low = (deg >= 1) & (deg <= 3)
        idx = torch.nonzero(low).reshape(-1)
        if idx.numel() < 2:
            return
        assert int(idx[-1] - idx[0]) + 1 == idx.numel(), "low-degree rows must be one run of the degree order"
        big = torch.iinfo(torch.int64).max
        minsrc = torch.full((n_local,), big, dtype=torch.int64, device=rows.device)
        minsrc.scatter_reduce_(0, rows, src_slots, reduce="amin", include_self=True)
        # rows are in descending degree order already: a stable sort of the low rows by
        # (degree descending, smallest source ascending) only moves rows inside their class
        key = (3 - deg[idx]) * (int(minsrc[idx].max()) + 1) + minsrc[idx]
        perm = torch.argsort(key, stable=True)  # new position (inside the run) -> old position
        first = int(idx[0])
        order = self.order.clone()
        order[first:first + perm.numel()] = self.order[first + perm]
        self.order = order
        rank_local = torch.empty_like(self.rank_local)
        rank_local[order.long()] = torch.arange(n_local, dtype=self.rank_local.dtype, device=rows.device)
        self.rank_local = rank_local

    @classmethod
    def from_rmat(cls, resource_handle, graph_properties, scale, num_edges, *, seed=42,
                  scramble_vertex_ids=True, weighted=False, group=None):
        """Every rank generates its slice of the R-MAT edge stream on its own GPU."""
        import torch

        from . import generators

        dist = _dist()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        per = (num_edges + world - 1) // world
        e0 = min(rank * per, num_edges)
        n = min(per, num_edges - e0)
        t0 = time.perf_counter()
        src, dst = generators.rmat(resource_handle, scale, n, seed=seed,
                                   scramble_vertex_ids=scramble_vertex_ids, first_edge=e0)
        wgt = generators.rmat_weights(resource_handle, n, seed=seed + 1, first_edge=e0) if weighted else None
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        self = cls.from_edges(HipEngine(resource_handle, graph_properties), src, dst, wgt,
                              1 << scale, rank=rank, world=world, group=group)
        self.t_generate = t_gen
        return self

    # ------------------------------------------------------------------ sessions
    def _local_pairs(self, vertices, values, scale=1.0):
        """Keep the (vertex, value) pairs this rank owns; inputs are replicated on every rank."""
        import torch

        v = torch.as_tensor(vertices, device=self.engine.device).to(self.vertex_dtype)
        x = torch.as_tensor(values, device=self.engine.device).to(torch.float64)
        keep = (v >= self.first) & (v < self.last)
        v = (self.rank_local[(v[keep] - self.first).long()] + self.first).to(self.vertex_dtype)
        return v.contiguous(), (x[keep] / scale).to(self.value_dtype).contiguous()

    def _replicated(self, pairs):
        """Per-rank shards of (vertex, value) pairs -> the whole list on every rank: the sizes are
        all-gathered, then the shards padded to the longest.  The reference shuffles each pair to
        the rank that owns its vertex (c_api/pagerank.cpp:140-144); the lists are short (or V-sized
        at most), so here every rank sees them all and _local_pairs keeps its own."""
        import torch

        dist = _dist()
        dev = self.engine.device
        v = torch.as_tensor(pairs[0], device=dev).to(torch.int64).reshape(-1)
        x = torch.as_tensor(pairs[1], device=dev).to(torch.float64).reshape(-1)
        if v.numel() != x.numel():
            raise ValueError("vertices and values of a (vertex, value) list must have the same length")
        if self.world == 1:
            return v, x
        sizes = torch.zeros(self.world, dtype=torch.int64, device=dev)
        sizes[self.rank] = v.numel()
        all_reduce(sizes, self.group)
        longest = int(sizes.max())
        if longest == 0:
            return v, x
        all_v = torch.zeros(self.world * longest, dtype=torch.int64, device=dev)
        all_x = torch.zeros(self.world * longest, dtype=torch.float64, device=dev)
        all_v[self.rank * longest:self.rank * longest + v.numel()] = v
        all_x[self.rank * longest:self.rank * longest + x.numel()] = x
        all_gather_in_place(all_v, longest, self.rank, self.world, self.group)
        all_gather_in_place(all_x, longest, self.rank, self.world, self.group)
        keep = (torch.arange(longest, device=dev)[None, :] < sizes[:, None]).reshape(-1)
        return all_v[keep], all_x[keep]

    def session(self, *, alpha=0.85, epsilon=1e-6, nstart=None, personalization=None, sharded=False):
        """nstart / personalization: optional (vertices, values) pairs -- the SAME on every rank,
        or with sharded=True any share of the list on each rank (a vertex need not be given to the
        rank that owns it; every rank must make the call, an empty share is fine), as the
        reference's MG entry takes them (mg_pagerank_test.c:67-94).
        Personalization values are divided by their global sum here (the library normalises only
        what it can see, pagerank_impl.cuh:213-226)."""
        import torch

        if sharded:
            nstart = None if nstart is None else self._replicated(nstart)
            personalization = None if personalization is None else self._replicated(personalization)

        ids = torch.arange(self.first, self.last, device=self.engine.device).to(self.vertex_dtype)
        kw = dict(alpha=alpha, epsilon=epsilon,
                  precomputed_vertex_out_weight_vertices=ids,
                  precomputed_vertex_out_weight_sums=self.out_w.to(self.value_dtype))
        if nstart is not None:
            gv, gx = self._local_pairs(*nstart)
            kw.update(initial_guess_vertices=gv, initial_guess_values=gx)
        if personalization is not None:
            total = float(torch.as_tensor(personalization[1]).to(torch.float64).sum())
            if not total > 0.0:
                raise ValueError("Invalid input argument: sum of personalization values should be positive.")
            pv, px = self._local_pairs(personalization[0], personalization[1], scale=total)
            kw.update(personalization_vertices=pv, personalization_values=px)
        sess = MGSession(self.engine.make_session(self.graph, **kw))
        self._exchange(sess)  # x0 and the dangling mass of pr0
        return sess

    @staticmethod
    def _choose_regions(engine, max_sources, world):
        """(C0, C1) or None.  Region 0 is ONE window of the XS layout (about 2 M slots: the hottest
        C0 sources of every rank, most of all gathers), region 1 the rest; each region is P equal
        sub-chunks, so each is one all-gather, and the second can run beside the part of the
        iteration that needs only the first (MGPagerank.iterate).  Two regions cost a second set
        of partial sums per row (one rank at scale 26: +0.2 ms per iteration), which can only pay
        once the exchange is a sizeable part of the iteration.  OFF by default: the overlap has
        never run on more than one GPU (the async all-gather beside xs_kernel's phase 0 is covered
        by emulated ranks and gloo only), so it stays an opt-in until a multi-GPU run has timed it.
        CUGRAPH_AMD_MG_REGIONS=1: whenever the layout allows; "auto": from 4 ranks."""
        import os

        mode = os.environ.get("CUGRAPH_AMD_MG_REGIONS", "0")
        if mode not in ("1", "auto") or (mode == "auto" and world < 4) or not hasattr(engine, "window_slots"):
            return None
        total = world * max_sources
        for _ in range(3):  # the window size depends on the number of slots, which the padding changes
            win = engine.window_slots(total)
            if win <= 0 or win % world:
                return None
            c0 = win // world
            if max_sources <= c0:
                return None  # everything fits the first window
            c1 = (max_sources - c0 + 3) // 4 * 4
            if world * (c0 + c1) == total:
                return c0, c1
            total = world * (c0 + c1)
        return None

    def _region_views(self, x_full):
        c0, c1 = self.regions
        hot = self.world * c0
        return (x_full[:hot], c0), (x_full[hot:hot + self.world * c1], c1)

    def _exchange(self, sess):
        """All-gather of the source vector and all-reduce of the two scalars, waited for."""
        if self.world == 1 and not self.always_exchange:
            return
        for w in self._exchange_start(sess):
            w.wait()

    def _exchange_start(self, sess):
        """The scalars first (every kernel of the next iteration tests them), then the all-gathers,
        not waited for: one handle per region."""
        x_full, scal = sess.inner.exchange_buffers()
        # scal[2:4]: this rank's {dangling mass, L1 delta}; their sum over the ranks goes to scal[0:2],
        # which the next iteration's kernels read.  An iteration skipped after the stop leaves
        # scal[2:4] alone, so summing again gives the same scal[0:2] (never P times the last one).
        total = scal[2:4].clone()
        all_reduce(total, self.group)
        scal[0:2].copy_(total)
        if self.idx_give is not None:
            # personalised exchange: every rank packs, per reader, the values that reader's edges
            # gather (one index list) and one all-to-all lands them in the readers' blocks
            packed = x_full[:self.n_own].index_select(0, self.idx_give)
            all_to_all_single(x_full[self.n_own:self.n_own + self.n_foreign], packed, self.want_split,
                              self.give_split, self.group)
            return ()
        if self.regions is None:
            return (all_gather_region_async(x_full[:self.slot_chunk * self.world], self.slot_chunk, self.rank,
                                            self.world, self.group),)
        return tuple(all_gather_region_async(v, c, self.rank, self.world, self.group)
                     for v, c in self._region_views(x_full))

    def iterate(self, sess, n=1):
        """n iterations.  With two regions, an iteration starts on the hot region as soon as that
        one has arrived (phase 0: the XS tiles over the hot slots, most of the work) while the
        all-gather of the other region is still running on the collective's stream; the rest of
        the iteration (phase 1) waits for it.  Nothing is left in flight when this returns."""
        if self.world == 1 and not self.always_exchange:
            sess.inner.iterate(n)
            return
        for _ in range(n):
            pend, sess.pending = sess.pending, None
            if pend and len(pend) == 2:
                pend[0].wait()
                sess.inner.iterate_phase(0)
                pend[1].wait()
                sess.inner.iterate_phase(1)
            else:
                for w in pend or ():
                    w.wait()
                sess.inner.iterate(1)
            sess.pending = self._exchange_start(sess)
        for w in sess.pending or ():
            w.wait()
        sess.pending = None

    def timed_iterate(self, sess, n):
        it, km = [], []
        for _ in range(n):
            a, b = sess.inner.timed_iterate(1)
            self._exchange(sess)
            it.append(a[0])
            km.append(b[0])
        return np.array(it), np.array(km)

    def run_to_convergence(self, sess, max_iterations, check_every=4):
        """Returns (iterations, stopped, l1_delta).  At least one iteration runs
        (pagerank_impl.cuh:235-335)."""
        budget = max(int(max_iterations), 1)
        done = 0
        iters, stopped, delta = 0, False, float("inf")
        while done < budget:
            n = min(check_every, budget - done)
            self.iterate(sess, n)
            done += n
            iters, stopped, delta, _ = sess.inner.status()
            if stopped:
                break
        return iters, stopped, delta

    def result(self, sess, max_iterations):
        """(local vertex ids, local ranks, iterations, converged) -- per-rank slices, as the
        reference returns per worker (dask/link_analysis/pagerank.py:426-427); ids are the
        caller's, ascending."""
        import torch

        verts, vals, iters, conv = sess.inner.result(max_iterations)
        out = torch.empty_like(vals)
        out[self.order.long()] = vals  # internal position -> the caller's id order
        return verts, out, iters, conv

    def gathered_result(self, sess, max_iterations):
        """Every rank gets the full rank vector (tests / small graphs)."""
        import torch

        verts, vals, iters, conv = self.result(sess, max_iterations)
        if self.world == 1:
            return vals, iters, conv
        return all_gather_ragged(vals, self.lengths, self.rank, self.world, self.group), iters, conv


def pagerank(resource_handle, graph_properties, src, dst, wgt, num_vertices, *, alpha=0.85,
             tol=1.0e-5, max_iter=100, nstart=None, personalization=None, sharded=False,
             fail_on_nonconvergence=True, group=None):
    """cugraph.dask.pagerank-shaped convenience call (dask/link_analysis/pagerank.py:174-183):
    returns this rank's (vertices, pageranks[, converged])."""
    from .pylibcugraph.exceptions import FailedToConvergeError

    drv = MGPagerank.from_edges(HipEngine(resource_handle, graph_properties), src, dst, wgt,
                                num_vertices, group=group)
    sess = drv.session(alpha=alpha, epsilon=tol, nstart=nstart, personalization=personalization,
                       sharded=sharded)
    drv.run_to_convergence(sess, max_iter)
    verts, vals, iters, conv = drv.result(sess, max_iter)
    if fail_on_nonconvergence:
        if not conv:
            raise FailedToConvergeError
        return verts, vals
    return verts, vals, conv
