"""CPU stand-in for cugraph_amd.mg.HipEngine, used ONLY by the world_size-2 gloo tests.

It restates, with numpy on one partition, what the HIP session does between two collectives
(cugraph_amd/csrc/pagerank.hip: prep/spmv/pers/finalize kernels and the device-side stop rule), so
that the distributed driver -- partition bounds, edge routing, the out-weight all-reduce, the
all-gather / all-reduce per iteration, personalization routing, result assembly -- runs end to end
without a GPU.  Test infrastructure; never imported by the package.
"""
import numpy as np
import torch


class _Graph:
    def __init__(self, src, dst, wgt, first, last, num_global, num_source_slots, x_pos, hot_slots=0,
                 private_slots=False):
        self.private_slots = private_slots  # personalised exchange: a slot numbering of this rank's own
        self.first, self.last, self.num_global = first, last, num_global
        self.slots = num_source_slots
        self.hot_slots = hot_slots  # slots below it: the first region of the gathered vector
        self.x_pos = x_pos.numpy().astype(np.int64)  # slot of each local row, -1: never gathered
        self.src = src.numpy().astype(np.int64)
        self.dst = dst.numpy().astype(np.int64)
        self.wgt = None if wgt is None else wgt.numpy().astype(np.float64)
        assert np.all((self.dst >= first) & (self.dst < last)), "edge routed to the wrong rank"


class _Session:
    def __init__(self, g, world, rank, alpha, epsilon, out_w_v, out_w, guess_v=None, guess_x=None,
                 pers_v=None, pers_x=None):
        self.g, self.alpha, self.eps = g, alpha, epsilon
        nl = g.last - g.first
        assert np.all(g.src < g.slots)
        mine = g.x_pos[g.x_pos >= 0]
        if g.private_slots:  # own sources first, in row order
            assert np.array_equal(mine, np.arange(len(mine)))
        else:
            self._check_chunks(g, world, rank, mine)
        self.wr = g.x_pos >= 0
        self.out_w = np.zeros(nl)
        self.out_w[out_w_v.numpy() - g.first] = out_w.numpy()
        self.pr = np.full(nl, 1.0 / g.num_global)
        if guess_v is not None:
            self.pr = np.zeros(nl)
            self.pr[guess_v.numpy() - g.first] = guess_x.numpy()
        self.pers = None
        if pers_v is not None:
            self.pers = np.zeros(nl)
            np.add.at(self.pers, pers_v.numpy() - g.first, pers_x.numpy())  # already normalised
        self.x = [torch.zeros(g.slots, dtype=torch.float64) for _ in range(2)]
        # [0:2] what the next iteration reads (summed over the ranks by the driver), [2:4] this rank's own
        self.scal = torch.zeros(4, dtype=torch.float64)
        self.enqueued, self.iters, self.done = 0, 0, False
        dang = self.out_w == 0
        xs = np.where(dang, self.pr, self.pr / np.where(dang, 1.0, self.out_w))
        self.x[0][g.x_pos[self.wr]] = torch.from_numpy(xs[self.wr])
        self.scal[0] = self.scal[2] = float(self.pr[dang].sum())
        self.scal[1] = self.scal[3] = 1e300

    @staticmethod
    def _check_chunks(g, world, rank, mine):
        # this rank writes only its own sub-chunk of each region (one region: P equal chunks)
        assert g.slots % world == 0
        assert g.hot_slots % world == 0 and (g.slots - g.hot_slots) % world == 0
        c0, c1 = g.hot_slots // world, (g.slots - g.hot_slots) // world
        hot = mine[mine < g.hot_slots]
        rest = mine[mine >= g.hot_slots] - g.hot_slots
        assert np.all((hot >= rank * c0) & (hot < (rank + 1) * c0))
        assert np.all((rest >= rank * c1) & (rest < (rank + 1) * c1))
        if g.hot_slots:
            assert len(hot) == min(c0, len(mine))  # the hot region takes a rank's first (hottest) sources

    def iterate(self, n=1):
        g = self.g
        for _ in range(n):
            cur, nxt = self.enqueued & 1, (self.enqueued & 1) ^ 1
            self.enqueued += 1
            if self.done:
                continue
            if float(self.scal[1]) < self.eps:
                self.done = True
                continue
            x = self.x[cur].numpy()
            contrib = x[g.src] * (1.0 if g.wgt is None else g.wgt)
            acc = np.bincount(g.dst - g.first, weights=contrib, minlength=g.last - g.first)
            teleport = float(self.scal[0]) * self.alpha + (1.0 - self.alpha)
            base = 0.0 if self.pers is not None else teleport / g.num_global
            pr = base + self.alpha * acc
            if self.pers is not None:
                pr = pr + teleport * self.pers
            dang = self.out_w == 0
            self.scal[1] = self.scal[3] = float(np.abs(pr - self.pr).sum())
            self.scal[0] = self.scal[2] = float(pr[dang].sum())
            xs = np.where(dang, pr, pr / np.where(dang, 1.0, self.out_w))
            self.x[nxt][g.x_pos[self.wr]] = torch.from_numpy(xs[self.wr])
            self.pr = pr
            self.iters += 1

    def iterate_phase(self, phase):
        # the stand-in has no use for the hot region alone: everything happens in phase 1
        if phase == 1:
            self.iterate(1)

    def exchange_buffers(self):
        return self.x[self.enqueued & 1], self.scal

    def status(self):
        stopped = self.done or (self.iters > 0 and float(self.scal[1]) < self.eps)
        return self.iters, stopped, float(self.scal[1]), float(self.scal[0])

    def result(self, max_iterations):
        verts = torch.arange(self.g.first, self.g.last, dtype=torch.int64)
        return verts, torch.from_numpy(self.pr.copy()), self.iters, self.iters < max_iterations


class OracleEngine:
    device = "cpu"

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def degree_order(self, dst, first, last):
        deg = np.bincount(dst.numpy().astype(np.int64) - first, minlength=last - first)
        order = np.argsort(-deg, kind="stable").astype(np.int32)
        rank = np.empty_like(order)
        rank[order] = np.arange(len(order), dtype=np.int32)
        return torch.from_numpy(order), torch.from_numpy(rank)

    def group_edges(self, src, dst, wgt, bounds):
        inner = np.asarray(bounds[1:-1], dtype=np.int64)
        owner = np.searchsorted(inner, dst.numpy().astype(np.int64), side="right")
        order = torch.from_numpy(np.argsort(owner, kind="stable"))
        counts = np.bincount(owner, minlength=len(bounds) - 1)
        return (src[order].contiguous(), dst[order].contiguous(), None if wgt is None else wgt[order].contiguous(),
                [int(c) for c in counts])

    def make_local_graph(self, src, dst, wgt, first, last, num_global, num_source_slots=None,
                         x_pos=None, hot_slots=0):
        return _Graph(src, dst, wgt, first, last, num_global, num_source_slots, x_pos, hot_slots,
                      getattr(self, "private_slots", False))

    def window_slots(self, num_source_slots):
        # small enough for the test graphs to get two regions (the library's are 2^21..2^23 slots)
        return 48

    def source_weight_sums(self, src, wgt, num_global):
        w = np.ones(len(src)) if wgt is None else wgt.numpy().astype(np.float64)
        return torch.from_numpy(np.bincount(src.numpy().astype(np.int64), weights=w,
                                            minlength=num_global))

    def make_session(self, graph, *, alpha, epsilon, precomputed_vertex_out_weight_vertices,
                     precomputed_vertex_out_weight_sums, initial_guess_vertices=None,
                     initial_guess_values=None, personalization_vertices=None,
                     personalization_values=None):
        return _Session(graph, self.world, self.rank, alpha, epsilon,
                        precomputed_vertex_out_weight_vertices, precomputed_vertex_out_weight_sums,
                        initial_guess_vertices, initial_guess_values, personalization_vertices,
                        personalization_values)
