// PageRank power iteration on gfx950: pull SpMV with a fused epilogue.
//
// Replaces, for this path, cugraph::detail::pagerank (reference:
// cpp/src/link_analysis/pagerank_impl.cuh:50-338), per_v_transform_reduce_incoming_e
// (cpp/src/prims/per_v_transform_reduce_incoming_outgoing_e.cuh:254-565, 577-1111),
// transform_reduce_v / reduce_v (cpp/src/prims/transform_reduce_v.cuh:80-108,
// reduce_v.cuh:62-136), update_edge_src_property (update_edge_src_dst_property.cuh:656-671),
// compute_out_weight_sums (cpp/src/structure/graph_weight_utils_impl.cuh:41-115) and
// collect_local_vertex_values_from_ext_vertex_value_pairs (cpp/src/detail/collect_local_vertex_values.cu:27-83).
//
// The reference runs ~7 V-sized passes and two blocking scalar read-backs around its SpMV each
// iteration.  Here one iteration is
//     spmv_kernel      : for every destination v   s = sum_{u->v} x[u] * w(u,v)
//                          pr'[v] = base + alpha*s (+ teleport * p[v])
//                          x'[v]  = pr'[v] / out_w[v]          (the vector the next SpMV gathers)
//                          partial sums of |pr'[v]-pr[v]| and of pr'[v] over dangling v
//     [pers_fix_kernel]: sparse personalization add-on for |P| << V
//     finalize_kernel  : one workgroup folds the partials into {dangling mass, L1 delta}
// with pr/pr' and x/x' ping-ponged, so HBM traffic is E*(4+b_w) + (V+1)*b_off + 5*V*b_val, the
// figure of BASELINE.md section 3.  Convergence is decided ON THE DEVICE: every kernel of
// iteration k first looks at the delta of iteration k-1 and returns at once when it is below
// epsilon, so the host can enqueue iterations ahead and read {iterations, stopped} back only now
// and then.  Degree-bucketed load balancing (north_star): G lanes per destination with
// G = 1 (thread per vertex), 16, 64 (wave64 per vertex, DPP/shuffle reduction) and one workgroup
// per hub with LDS-staged partial sums, all in one spmv_kernel launch -- on a graph that is too
// small for them or not degree-sorted.  Otherwise the rows of in-degree >= 4 run as XS tiles in a
// launch of their own (xs_kernel / xs2_kernel + xs_finish_kernel) and spmv_kernel keeps the
// thread-per-row and isolated buckets.  The path is a sparse HBM-bound contraction: no MFMA.
#include "common.hpp"

#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace cga {

namespace {

constexpr int kThreads     = 256;
constexpr int kNumBuckets  = 5;  // hub items, wave64, 16-lane, thread per vertex, isolated
constexpr int kMaxPartials = 16384;
constexpr int64_t kHubChunk = 32768;  // edges of one hub work item (longer columns are split)

// XS ("XCD-sliced, source-sorted") tiles for the mid-degree rows: see the comment at xs_bucket
constexpr int kXsSlices       = 8;      // one slice of the source ids per XCD
constexpr int64_t kXsTileEdges = 1 << 26;  // a tile is a run of rows holding about this many edges,
constexpr int kXsEntriesPerLane = 8;   // gathers a lane keeps in flight per pass
constexpr int64_t kXsPieceEntries = 1 << 18;  // entries of one work item: longer (tile, slice) units are cut
// rows per tile = fp64 LDS accumulators per workgroup: 2^12 (32 KiB, 256 threads, 4 workgroups per
// CU), 2^13 (64 KiB, 512 threads, 2 per CU) or 2^14 (128 KiB, 1024 threads, 1 per CU) -- the widest
// whose entry format still covers the source space with the windows a record carries
constexpr int kXsRowBits      = 12;
constexpr int kXsRowBitsMax   = 14;
constexpr int kXsMinDegree    = kLowDegreeMax + 1;
// (measured and dropped, profiles/r3_xs_diag.md: "dense" passes -- one coalesced load of the few dozen
// consecutive source positions a pass over a hot stretch names, and ds_bpermute reads from it)
// entries the first pass of an item is aligned down to: 64 = every wave load of 256 bytes is two
// whole cache lines
#ifndef CGA_XS_ALIGN
#define CGA_XS_ALIGN 64
#endif
constexpr bool kXsWideDefault = false;
constexpr int64_t kXsMaxDegree = 16384;  // without XS tiles: longer columns are hub work items

// a packed entry is (slice-local source & (2^(32 - row_bits) - 1)) << row_bits | row in tile

// slice of a source id: interleaved in granules of 2^shift ids (default 128 ids = four 128-B lines)
__host__ __device__ inline uint32_t xs_slice_of(uint32_t src, int shift)
{
  return (src >> shift) & (kXsSlices - 1);
}
// position of a source inside its slice (the slice bits squeezed out) and back
__host__ __device__ inline uint32_t xs_slice_local(uint32_t src, int shift)
{
  return ((src >> (shift + 3)) << shift) | (src & ((1u << shift) - 1u));
}
__host__ __device__ inline uint32_t xs_global_id(uint32_t local, uint32_t slice, int shift)
{
  return ((local >> shift) << (shift + 3)) | (slice << shift) | (local & ((1u << shift) - 1u));
}

template <typename ValT>
struct XsArgs {
  // packed entries [n_edges + 4], ascending source inside each (tile, slice, window) unit:
  // (low 20 bits of the source's slice-local position) << 12 | row relative to the tile's first
  uint32_t const* ent;
  ValT const* w;             // [n_edges + 4] or nullptr
  int64_t const* unit_off;   // [n_tiles * 8 * n_win + 1]
  // one 512-byte record per (slice, work item), see xs_records_kernel: everything a workgroup needs
  // to start on a piece of a (tile, slice) unit, fetched with one load behind the queue's atomic
  int64_t const* rec;        // [8][n_items][64]
  int64_t const* pbase;      // [n_tiles + 1] first partial of each tile (8 * pieces * rows per tile)
  int n_items;               // work items per slice queue: the pieces of all tiles, biggest tiles first
  int n_win;                 // windows of 2^(32 - row_bits) slice-local positions per slice
  int slice_shift;
  int row_bits;
  int32_t const* tile_row;   // [n_tiles + 1] first row of each tile, relative to row_first
  ValT* partial;             // per (tile, slice, piece): one partial sum per row of the tile
  int* counters;             // [8] next item of each slice's queue
  int64_t row_first, n_rows;
  int n_tiles;
#ifdef CGA_TIMING_AIDS
  int affinity_mode;     // 0: queue of the XCD we run on; 1, 2: blockIdx based
  int profile;           // per-workgroup times into xs_prof_buf
  uint32_t gather_mask;  // narrows every gathered index (results are then wrong)
  int no_atomics;        // row sums are not accumulated (one add per lane and pass keeps the loads alive)
  int no_gather;         // nothing is gathered (every source value reads as 1)
  int no_writeback;      // the partials are not stored (the accumulators are still cleared)
  uint32_t slot_limit;   // region 0 of a partitioned graph: gathers at or beyond it are counted (0: no check)
  // bounds of the three arrays xs_kernel addresses with computed indices: an access outside them is
  // recorded in xs_check_buf (and redirected) instead of faulting the GPU
  int64_t ent_len, partial_len;
  uint32_t x_len;
#endif
};

// Timing aids (CUGRAPH_AMD_GATHER_MASK, _BUCKET_MASK, _XS_AFFINITY, _XS_PROFILE) exist only in a
// library built with `make AIDS=1`; the shipped kernels carry none of them.
#ifdef CGA_TIMING_AIDS
#define CGA_GMASK(args, i) ((i) & (args).gather_mask)
#else
#define CGA_GMASK(args, i) (i)
#endif

constexpr int kXsRecWords   = 64;  // one load of the first wave
constexpr int kXsRecWindows = kXsRecWords - 4;  // window starts a record carries: wide tiles up to this many

// one piece of a hub column; its partial sum lands in item_partial[item]
struct HubItem {
  int64_t beg;
  int64_t end;
};

struct Ctrl {
  unsigned long long iters;
  int done;
  int pad;
  double base_prev;  // rank every vertex without in-edges got in the last iteration (dead rows)
};

struct Bucket {
  int32_t const* list;  // ascending local vertex ids, or nullptr for the identity
  int64_t first;        // used when list == nullptr
  int64_t count;
  int block_begin;
  int n_blocks;
};

template <typename OffT, typename ValT>
struct IterArgs {
  OffT const* offsets;
  int32_t const* indices;
  ValT const* weights;
  ValT const* x_cur;
  ValT* x_next;
  ValT const* pr_old;
  ValT* pr_new;
  ValT const* out_w;
  ValT const* pers_dense;
  double const* scal;  // {dangling mass, L1 delta} of the previous iteration
  Ctrl* ctrl;
  double* partials;  // [2 * gridDim.x]
  ValT alpha;
  ValT one_minus_alpha;
  ValT num_vertices;  // global vertex count as a value
  double epsilon;
  int personalized;
#ifdef CGA_TIMING_AIDS
  uint32_t gather_mask;
#endif
  int64_t vertex_first;
  ValT* raw_out;         // not nullptr: plain SpMV, row sums go here and nothing else happens
  int32_t const* x_pos;  // slot of row v in the gathered vector, -1: nobody gathers it; nullptr: vertex_first + v
  HubItem const* hub_items;  // bucket 0 works on these instead of whole rows
  int64_t n_hub_items;
  double* hub_item_partial;
  XsArgs<ValT> xs;
  Bucket buckets[kNumBuckets];
};

__device__ inline uint32_t xcc_id()
{
  // HW_REG_XCC_ID (20), bits [3:0]: which XCD this wave runs on.  Used for L2 affinity only.
  return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
}

// a value every lane of the wave holds (read from LDS, say) moved to scalar registers
__device__ inline int64_t uniform64(int64_t v)
{
  uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(static_cast<uint64_t>(v) >> 32));
  return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}

template <typename T>
__device__ inline T group_sum(T v, int width)
{
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, width);
  return v;
}

// returns true when this iteration must not run (already stopped, or the previous delta is small)
__device__ inline bool stop_check(Ctrl* ctrl, double const* scal, double epsilon, bool may_flag)
{
  if (ctrl->done) return true;
  if (scal[1] < epsilon) {
    if (may_flag && blockIdx.x == 0 && threadIdx.x == 0) ctrl->done = 1;
    return true;
  }
  return false;
}

template <typename OffT, typename ValT>
struct Epilogue {
  IterArgs<OffT, ValT> const& a;
  ValT teleport, base;
  double delta{0.0}, dangling{0.0};
  __device__ Epilogue(IterArgs<OffT, ValT> const& args) : a(args)
  {
    ValT d   = static_cast<ValT>(a.scal[0]);
    teleport = d * a.alpha + a.one_minus_alpha;  // pagerank_impl.cuh:266-269, 316
    base     = a.personalized ? ValT(0) : teleport / a.num_vertices;
  }
  __device__ inline void finish(int64_t v, ValT weighted_sum)
  {
    if (a.raw_out) {
      a.raw_out[v] = weighted_sum;
      return;
    }
    ValT old = __builtin_nontemporal_load(a.pr_old + v);
    ValT ow  = __builtin_nontemporal_load(a.out_w + v);
    finish(v, weighted_sum, old, ow);
  }
  // old rank and out-weight sum of v already loaded (callers that want those loads in flight early)
  __device__ inline void finish(int64_t v, ValT weighted_sum, ValT old, ValT ow)
  {
    if (a.raw_out) {
      a.raw_out[v] = weighted_sum;
      return;
    }
    ValT pr = base + a.alpha * weighted_sum;
    if (a.pers_dense) pr += teleport * a.pers_dense[v];
    delta += fabs(static_cast<double>(pr) - static_cast<double>(old));
    ValT x = pr;
    if (ow == ValT(0)) {
      dangling += static_cast<double>(pr);
    } else {
      x = pr / ow;
    }
    __builtin_nontemporal_store(pr, a.pr_new + v);
    if (a.x_pos) {
      int32_t slot = a.x_pos[v];
      if (slot >= 0) a.x_next[slot] = x;
    } else {
      __builtin_nontemporal_store(x, a.x_next + a.vertex_first + v);
    }
  }
};

__device__ inline void write_partials(double dangling, double delta, double* partials)
{
  __shared__ double lds[2 * (kThreads / 64)];
  dangling = group_sum(dangling, 64);
  delta    = group_sum(delta, 64);
  int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    lds[2 * wave]     = dangling;
    lds[2 * wave + 1] = delta;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double d = 0.0, l = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) {
      d += lds[2 * w];
      l += lds[2 * w + 1];
    }
    partials[2 * blockIdx.x]     = d;
    partials[2 * blockIdx.x + 1] = l;
  }
}

__device__ inline void load4(float const* p, float* w)
{
  float4 v = *reinterpret_cast<float4 const*>(p);
  w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}
__device__ inline void load4(double const* p, double* w)
{
  double2 v0 = *reinterpret_cast<double2 const*>(p);
  double2 v1 = *reinterpret_cast<double2 const*>(p + 2);
  w[0] = v0.x; w[1] = v0.y; w[2] = v1.x; w[3] = v1.y;
}

// Partial sum of x[src]*w over edges [beg, end) of one column for one of G cooperating lanes.
// Every lane loads 16-byte vectors of 4 consecutive indices at 16-byte aligned positions (lane
// stride 16 B: a wave reads 1 KiB per instruction, fully coalesced) and issues the 4 (x2) gathers
// back to back, so each lane keeps 8 independent gathers in flight.  The arrays are padded so the
// aligned over-read past the last edge stays inside the allocation (graph_build.hip).
template <int G, typename OffT, typename ValT, bool WEIGHTED>
__device__ inline double column_partial(IterArgs<OffT, ValT> const& a, OffT beg, OffT end, int lane)
{
  double acc     = 0.0;
  OffT const step = static_cast<OffT>(G) * 4;
  for (OffT j = (beg & ~static_cast<OffT>(3)) + static_cast<OffT>(lane) * 4; j < end; j += 2 * step) {
    OffT j2  = j + step;
    bool two = j2 < end;
    int4 q   = *reinterpret_cast<int4 const*>(a.indices + j);
    int4 r   = two ? *reinterpret_cast<int4 const*>(a.indices + j2) : int4{0, 0, 0, 0};
    ValT w[8];
    if constexpr (WEIGHTED) {
      load4(a.weights + j, w);
      if (two) {
        load4(a.weights + j2, w + 4);
      } else {
        w[4] = w[5] = w[6] = w[7] = ValT(0);
      }
    }
    bool p[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      p[k]     = (j + k >= beg) && (j + k < end);
      p[4 + k] = two && (j2 + k < end);  // j2 + k > beg always
    }
    uint32_t gi[8] = {(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w,
                      (uint32_t)r.x, (uint32_t)r.y, (uint32_t)r.z, (uint32_t)r.w};
    ValT xv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] = p[k] ? a.x_cur[CGA_GMASK(a, gi[k])] : ValT(0);
    if constexpr (WEIGHTED) {
#pragma unroll
      for (int k = 0; k < 8; ++k) xv[k] *= w[k];
    }
    ValT s0 = (xv[0] + xv[1]) + (xv[2] + xv[3]);
    ValT s1 = (xv[4] + xv[5]) + (xv[6] + xv[7]);
    acc += static_cast<double>(s0) + static_cast<double>(s1);
  }
  return acc;
}

// G lanes cooperate on one destination vertex (G = 64: one wave64 per vertex)
template <int G, typename OffT, typename ValT, bool WEIGHTED>
__device__ inline void group_bucket(IterArgs<OffT, ValT> const& a, Bucket const& b,
                                    Epilogue<OffT, ValT>& ep)
{
  int lane         = threadIdx.x & (G - 1);
  int64_t group    = ((int64_t)(blockIdx.x - b.block_begin) * kThreads + threadIdx.x) / G;
  int64_t n_groups = ((int64_t)b.n_blocks * kThreads) / G;
  for (int64_t i = group; i < b.count; i += n_groups) {
    int64_t v = b.list ? static_cast<int64_t>(b.list[i]) : b.first + i;
    OffT beg = a.offsets[v], end = a.offsets[v + 1];
    double acc = column_partial<G, OffT, ValT, WEIGHTED>(a, beg, end, lane);
    acc        = group_sum(acc, G);
    if (lane == 0) ep.finish(v, static_cast<ValT>(acc));
  }
}

// thread per vertex, for the shortest columns (1 <= deg < kXsMinDegree).  Two rows per thread and
// trip, every load of a step issued for both before anything waits: offsets -> the (up to) three
// indices together with the row's old rank and out-weight -> the gathers.  Loads carry no
// predicates: a position past the column's end re-reads its last edge and is dropped.
template <typename OffT, typename ValT, bool WEIGHTED>
__device__ inline void thread_bucket(IterArgs<OffT, ValT> const& a, Bucket const& b,
                                     Epilogue<OffT, ValT>& ep)
{
  constexpr int D = static_cast<int>(kXsMinDegree) > 1 ? static_cast<int>(kXsMinDegree) - 1 : 1;  // longest column of this bucket
  constexpr int R = 2;
  int64_t t = (int64_t)(blockIdx.x - b.block_begin) * kThreads + threadIdx.x;
  int64_t n = (int64_t)b.n_blocks * kThreads;
  bool const raw = a.raw_out != nullptr;
  for (int64_t i = t; i < b.count; i += R * n) {
    int64_t v[R];
    bool have[R];
    OffT beg[R], end[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int64_t ir = i + r * n;
      have[r]    = ir < b.count;
      ir         = have[r] ? ir : i;
      v[r]       = b.list ? static_cast<int64_t>(b.list[ir]) : b.first + ir;
      beg[r]     = a.offsets[v[r]];
      end[r]     = a.offsets[v[r] + 1];
    }
    ValT old[R], ow[R];
    uint32_t idx[R][D];
    ValT w[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      old[r] = raw ? ValT(0) : __builtin_nontemporal_load(a.pr_old + v[r]);
      ow[r]  = raw ? ValT(0) : __builtin_nontemporal_load(a.out_w + v[r]);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        OffT j    = beg[r] + d < end[r] ? beg[r] + d : end[r] - 1;
        idx[r][d] = static_cast<uint32_t>(a.indices[j]);
        if constexpr (WEIGHTED) w[r][d] = a.weights[j];
      }
    }
    ValT xv[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int d = 0; d < D; ++d) xv[r][d] = a.x_cur[CGA_GMASK(a, idx[r][d])];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      ValT acc = 0;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        ValT term = xv[r][d];
        if constexpr (WEIGHTED) term *= w[r][d];
        acc += beg[r] + d < end[r] ? term : ValT(0);
      }
      if (have[r]) ep.finish(v[r], acc, old[r], ow[r]);
    }
  }
}

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ inline void stream_load_vals4(float const* p, float* r)
{
  v4f_t v = __builtin_nontemporal_load(reinterpret_cast<v4f_t const*>(p));
  r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
}
__device__ inline void stream_load_vals4(double const* p, double* r)
{
  v2d_t v0 = __builtin_nontemporal_load(reinterpret_cast<v2d_t const*>(p));
  v2d_t v1 = __builtin_nontemporal_load(reinterpret_cast<v2d_t const*>(p) + 1);
  r[0] = v0.x; r[1] = v0.y; r[2] = v1.x; r[3] = v1.y;
}
__device__ inline void stream_store_vals4(float* p, float const* r)
{
  v4f_t v = {r[0], r[1], r[2], r[3]};
  __builtin_nontemporal_store(v, reinterpret_cast<v4f_t*>(p));
}
__device__ inline void stream_store_vals4(double* p, double const* r)
{
  v2d_t v0 = {r[0], r[1]}, v1 = {r[2], r[3]};
  __builtin_nontemporal_store(v0, reinterpret_cast<v2d_t*>(p));
  __builtin_nontemporal_store(v1, reinterpret_cast<v2d_t*>(p) + 1);
}

// vertices without in-edges (about half of an R-MAT graph): no offsets, no gathers, the rank
// update streamed four vertices per lane when the rows are a contiguous, aligned range
template <typename OffT, typename ValT>
__device__ inline void isolated_bucket(IterArgs<OffT, ValT> const& a, Bucket const& b,
                                       Epilogue<OffT, ValT>& ep)
{
  int64_t t = (int64_t)(blockIdx.x - b.block_begin) * kThreads + threadIdx.x;
  int64_t n = (int64_t)b.n_blocks * kThreads;
  // (with a compact source space the new source value goes to the row's slot, if it has one)
  bool vec  = b.list == nullptr && a.pers_dense == nullptr && a.raw_out == nullptr &&
             (a.vertex_first & 3) == 0;
  if (!vec) {
    for (int64_t i = t; i < b.count; i += n)
      ep.finish(b.list ? static_cast<int64_t>(b.list[i]) : b.first + i, ValT(0));
    return;
  }
  int64_t head = (4 - (b.first & 3)) & 3;
  if (head > b.count) head = b.count;
  int64_t nvec = (b.count - head) >> 2;
  if (t < head) ep.finish(b.first + t, ValT(0));
  for (int64_t i = head + 4 * nvec + t; i < b.count; i += n) ep.finish(b.first + i, ValT(0));
  for (int64_t q = t; q < nvec; q += n) {
    int64_t v = b.first + head + 4 * q;
    ValT old[4], ow[4], prn[4], x[4];
    stream_load_vals4(a.pr_old + v, old);
    stream_load_vals4(a.out_w + v, ow);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ValT pe = ep.base + a.alpha * ValT(0);  // as Epilogue::finish with an empty sum
      ep.delta += fabs(static_cast<double>(pe) - static_cast<double>(old[e]));
      if (ow[e] == ValT(0)) {
        ep.dangling += static_cast<double>(pe);
        x[e] = pe;
      } else {
        x[e] = pe / ow[e];
      }
      prn[e] = pe;
    }
    stream_store_vals4(a.pr_new + v, prn);
    if (a.x_pos) {
      v4i_t slot = __builtin_nontemporal_load(reinterpret_cast<v4i_t const*>(a.x_pos + v));
      if (slot.x >= 0) a.x_next[slot.x] = x[0];
      if (slot.y >= 0) a.x_next[slot.y] = x[1];
      if (slot.z >= 0) a.x_next[slot.z] = x[2];
      if (slot.w >= 0) a.x_next[slot.w] = x[3];
    } else {
      stream_store_vals4(a.x_next + a.vertex_first + v, x);
    }
  }
}

// hubs: one workgroup per work item (a column, or a 32K-edge piece of a longer one); threads
// stride the piece coalesced, partial sums staged through LDS; hub_finish_kernel adds the pieces
template <typename OffT, typename ValT, bool WEIGHTED>
__device__ inline void hub_bucket(IterArgs<OffT, ValT> const& a, Bucket const& b)
{
  __shared__ double hub_lds[kThreads / 64];
  for (int64_t i = blockIdx.x - b.block_begin; i < a.n_hub_items; i += b.n_blocks) {
    HubItem it = a.hub_items[i];
    double acc = column_partial<kThreads, OffT, ValT, WEIGHTED>(
      a, static_cast<OffT>(it.beg), static_cast<OffT>(it.end), threadIdx.x);
    acc = group_sum(acc, 64);
    if ((threadIdx.x & 63) == 0) hub_lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = 0.0;
      for (int w = 0; w < kThreads / 64; ++w) s += hub_lds[w];
      a.hub_item_partial[i] = s;
    }
    __syncthreads();
  }
}


// streamed-once data (XS entries): non-temporal loads, so that they do not push the gathered
// vector out of the XCD's L2
__device__ inline void stream_load4(uint32_t const* p, uint32_t* r)
{
  v4i_t v = __builtin_nontemporal_load(reinterpret_cast<v4i_t const*>(p));
  r[0] = (uint32_t)v.x; r[1] = (uint32_t)v.y; r[2] = (uint32_t)v.z; r[3] = (uint32_t)v.w;
}

// XS tiles.  The gathers of a pull SpMV are bound by two request rates, not by bytes: ~240 G/s
// for 4-byte gathers that hit L2 and ~55 G/s for the ones that go to the fabric (measured,
// tools/gather_bench.hip).  Two things cut both counts for the mid-degree rows:
//   * source ids are cut into 8 interleaved slices (128-byte granules) and the entries of slice k
//     are processed by workgroups running on XCD k, so each XCD's 4 MiB L2 only ever sees 1/8 of
//     the gathered vector: the 8 private L2s add up to 32 MiB of distinct hot sources;
//   * inside a (tile, slice) unit the entries of MANY rows are stored together in ascending
//     source order, so the 64 lanes of a gather instruction touch neighbouring sources and the
//     texture path merges them into a few cache-line requests.
// Row sums are accumulated with LDS atomics (fp64) per tile and written as one partial per
// (slice, row); xs_finish_kernel adds the 8 partials and runs the epilogue.  Workgroups pull
// tiles from the queue of the XCD they run on (HW_REG_XCC_ID) and steal from the other queues
// when theirs is empty, so the result never depends on where a workgroup was placed.
// what xs_kernel needs of an iteration (the tiles carry their own layout: no offsets type)
template <typename ValT>
struct XsLaunch {
  XsArgs<ValT> xs;
  ValT const* x_cur;
  double const* scal;
  Ctrl* ctrl;
  double epsilon;
};

#ifdef CGA_TIMING_AIDS
// per-workgroup times (make AIDS=1; CUGRAPH_AMD_XS_PROFILE=1)
__device__ unsigned long long xs_prof_buf[1024 * 8];
// gathers of a region-0 launch that read a slot of region 1 (the all-gather of region 1 may still
// be running then: there must be none), and the region-0 launches checked
__device__ unsigned long long xs_region_check[2];
#define CGA_XSP(stmt) do { if (a.xs.profile) { stmt; } } while (0)
// bounds checks of xs_kernel's computed addresses: [0] violations, [1] kind of the first (1 entry
// stream, 2 gather, 3 partials), [2..3] two values of it, [4] its workgroup, [5] tile | slice << 32
__device__ unsigned long long xs_check_buf[8];
#define CGA_XS_CHECK(cond, kind, v0, v1)                                                            \
  do {                                                                                              \
    if (cond) {                                                                                     \
      if (atomicAdd(&xs_check_buf[0], 1ull) == 0ull) {                                              \
        xs_check_buf[1] = (kind);                                                                   \
        xs_check_buf[2] = static_cast<unsigned long long>(v0);                                      \
        xs_check_buf[3] = static_cast<unsigned long long>(v1);                                      \
        xs_check_buf[4] = blockIdx.x;                                                               \
        xs_check_buf[5] = static_cast<unsigned long long>(t) | (static_cast<unsigned long long>(k) << 32); \
      }                                                                                             \
    }                                                                                               \
  } while (0)
#else
#define CGA_XSP(stmt) do { } while (0)
#define CGA_XS_CHECK(cond, kind, v0, v1) do { } while (0)
#endif

typedef double v2f64_t __attribute__((ext_vector_type(2)));

// (make VARIANT=phase EXTRA=-DCGA_XS_PHASE_PROF) where a wave's time goes inside its passes: shader
// clock ticks summed over all waves of a launch -- [0] passes, [1] claim + window tracking,
// [2] address arithmetic + issue of gathers and next entries, [3] wait for the gathers, [4] LDS
// accumulation (to completion), [5] item boundaries (end of last pass -> first pass of next item),
// [6] whole kernel, [7] fast passes
#ifdef CGA_XS_PHASE_PROF
__device__ unsigned long long xs_phase_buf[8];
__global__ void xs_phase_dump_kernel()
{
  printf("XSPHASE passes %llu fast %llu | per wave-pass ticks: claim+windows %.0f issue %.0f gather-wait %.0f accumulate %.0f | "
         "boundaries total %llu kernel total %llu (sums over waves) pass-sum %llu\n",
         xs_phase_buf[0], xs_phase_buf[7], (double)xs_phase_buf[1] / xs_phase_buf[0], (double)xs_phase_buf[2] / xs_phase_buf[0],
         (double)xs_phase_buf[3] / xs_phase_buf[0], (double)xs_phase_buf[4] / xs_phase_buf[0], xs_phase_buf[5], xs_phase_buf[6],
         xs_phase_buf[1] + xs_phase_buf[2] + xs_phase_buf[3] + xs_phase_buf[4]);
  for (int i = 0; i < 8; ++i) xs_phase_buf[i] = 0;
}
__global__ void xs2_phase_dump_kernel()
{
  double const w = 1.0;
  printf("XS2PHASE wave-passes %llu | sums over waves (ticks): wait-record %.3g wait-bank %.3g produce %.3g setup %.3g passes %.3g "
         "leave+writeback %.3g kernel %.3g | per pass %.0f\n",
         xs_phase_buf[0], w * xs_phase_buf[1], w * xs_phase_buf[2], w * xs_phase_buf[3], w * xs_phase_buf[4], w * xs_phase_buf[5],
         w * xs_phase_buf[6], w * xs_phase_buf[7], (double)xs_phase_buf[5] / xs_phase_buf[0]);
  for (int i = 0; i < 8; ++i) xs_phase_buf[i] = 0;
}
#define CGA_PH(stmt) do { stmt; } while (0)
#else
#define CGA_PH(stmt) do { } while (0)
#endif

// wait for this wave's LDS operations, then the workgroup barrier: unlike __syncthreads() it does
// not wait for the wave's global stores (the partials just written need not have landed)
__device__ inline void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// accumulators per row of a tile, as a power of two: as many as fit the bank, 64 at most
__device__ inline int xs_rep_bits(int rows, int bank_rows)
{
  int const fit = bank_rows / (rows > 0 ? rows : 1);  // >= 1
  int const b   = 31 - __builtin_clz(static_cast<unsigned int>(fit));
  return b < 6 ? b : 6;
}

// One workgroup pulls work items (pieces of (tile, slice) units) from the queue of the XCD it runs
// on.  Inside an item the unit of work is a WAVE pass: 64 * EPL consecutive entries, entry e of
// lane l at position e * 64 + l, so that the 64 lanes of one gather instruction hold 64
// consecutive entries (ascending sources: few cache lines per instruction).  Waves CLAIM their
// passes from a counter in LDS instead of owning a fixed share of every pass: with fixed shares
// the waves of a workgroup drift apart (the older waves win the issue arbitration and run many
// passes ahead), and at the end of every item the early ones idle at the barrier -- wave 0 for
// 39 % of the kernel (profiles/r2_xs_waves.md).  With claims all waves reach the end of an item
// within one wave pass of each other.  Each wave software-pipelines its own passes: the entries
// of its next pass are requested right after the gathers of the current one were issued.
// WPE: waves per SIMD the register allocation is held to (__launch_bounds__'s second argument):
// 8 = 64 VGPRs, so that two 1024-thread workgroups are resident on a CU (8192-row tiles: 2 x 65 KiB
// of LDS) and the item boundaries of one -- drain, write-back, re-fill: ~13 us, 23 % of wave 0's
// time with one workgroup per CU (profiles/r2_xs_waves.md) -- are covered by the passes of the other.
template <typename ValT, bool WEIGHTED, int THREADS, int ROWBITS, int EPL, int WPE>
__global__ __launch_bounds__(THREADS, WPE) void xs_kernel(XsLaunch<ValT> a)
{
  constexpr int kLowBits       = 32 - ROWBITS;
  constexpr uint32_t kRowMask  = (1u << ROWBITS) - 1u;
  constexpr uint32_t kPass     = THREADS * EPL;  // granule of a piece's share of its unit (xs_records_kernel)
  constexpr uint32_t kWavePass = 64 * EPL;       // entries of one wave pass
  extern __shared__ double xs_lds[];
  double* const acc = xs_lds;  // [2^ROWBITS]
  __shared__ int64_t s_rec[2][kXsRecWords];
  __shared__ unsigned int s_claim;  // next wave pass of the current item
  if (stop_check(a.ctrl, a.scal, a.epsilon, false)) return;
  XsArgs<ValT> const& xs = a.xs;
  int const tid  = threadIdx.x;
  int const lane = tid & 63;
  bool const wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
#ifdef CGA_TIMING_AIDS
  int const my  = xs.affinity_mode == 0   ? static_cast<int>(xcc_id() & (kXsSlices - 1))
                  : xs.affinity_mode == 1 ? static_cast<int>(blockIdx.x & (kXsSlices - 1))
                                          : static_cast<int>((blockIdx.x >> 3) & (kXsSlices - 1));
  unsigned long long pt_bound = 0, pt_n = 0, pt_stolen = 0, pt_a = 0, pt_b1 = 0, pt_wbk = 0, pt_c = 0;
  unsigned long long const pt_t0 = wall_clock64();
#else
  int const my = static_cast<int>(xcc_id() & (kXsSlices - 1));
#endif
#ifdef CGA_XS_PHASE_PROF
  unsigned long long ph_n = 0, ph_claim = 0, ph_issue = 0, ph_wait = 0, ph_acc = 0, ph_bound = 0, ph_fast = 0;
  unsigned long long const ph_k0 = clock64();
  unsigned long long ph_t = ph_k0, ph_end = ph_k0;
#endif
  // the accumulators start at zero and every write-back leaves them at zero
  for (int i = tid; i < (1 << ROWBITS); i += THREADS) acc[i] = 0.0;
  if (tid == 0) s_claim = 0;
  int const shift         = xs.slice_shift;
  uint32_t const low_mask = (1u << shift) - 1u;
  int const n_win         = xs.n_win;
  // record of queue item q of slice k (the first wave, 8 bytes a lane); -1: queue empty
  auto load_record = [&](int k, int q) -> int64_t {
    int64_t r = -1;
    if (lane < kXsRecWords && q < xs.n_items)
      r = xs.rec[(static_cast<int64_t>(k) * xs.n_items + q) * kXsRecWords + lane];
    return r;
  };
  // next wave pass of the item, the same value in every lane (one LDS atomic per wave and pass)
  auto claim = [&]() -> uint32_t {
    unsigned int c = 0;
    if (lane == 0) c = atomicAdd(&s_claim, 1u);
    return __builtin_amdgcn_readfirstlane(c);
  };
  int buf = 0;
  for (int rot = 0; rot < kXsSlices; ++rot) {
    int const k = (my + rot) & (kXsSlices - 1);
    uint32_t const kbits = static_cast<uint32_t>(k) << shift;
    // the first item of a queue is fetched on the spot, every later one during the item before it
    // (not further ahead: an item reserved by a busy workgroup is an item an idle one cannot take,
    // and a small graph has only a few items per workgroup).  It goes to the buffer nobody reads:
    // the terminal record of the previous queue sits in the current one, and a wave that lags
    // behind wave 0 may not have looked at it yet (no barrier separates that read from this write).
    buf ^= 1;
    if (wave0) {
      int q = 0;
      if (lane == 0) q = atomicAdd(&xs.counters[k], 1);
      q         = __shfl(q, 0);
      int64_t r = load_record(k, q);
      if (lane < kXsRecWords) s_rec[buf][lane] = r;
    }
    __syncthreads();
    while (true) {
      int64_t const* rec = s_rec[buf];
      int64_t const head = uniform64(rec[0]);  // tile | piece << 32 | pieces << 40
      if (head < 0) break;
      CGA_XSP(pt_n++; pt_stolen += rot != 0);
      int const t       = static_cast<int>(head & 0xffffffff);
      uint32_t const piece = static_cast<uint32_t>(head >> 32) & 0xffu;
      uint32_t const n_pc  = static_cast<uint32_t>(head >> 40) & 0xffu;
      int const rows    = static_cast<int>(uniform64(rec[1]) >> 32);
      ValT* const out   = xs.partial + uniform64(rec[2]);
      // Tiles of FEW rows (the hub tiles: 131 rows share 16.8 M entries in the first tile of a
      // scale-26 graph): lanes of one atomic instruction that add to the same row serialise in the
      // LDS, and such tiles ran at half the speed of their entry stream (profiles/r3_xs_tiles.md).
      // A row therefore gets 2^rep accumulators, as many as fit (up to 64), and a lane adds into
      // the one its lane number selects; the write-back adds a row's accumulators in a fixed order.
      int const rep          = xs_rep_bits(rows, 1 << ROWBITS);
      uint32_t const rep_sel = static_cast<uint32_t>(lane) & ((1u << rep) - 1u);
      // the (tile, slice) stream is the concatenation of its windows: an entry's window follows
      // from its position.  bound(w): first position of window w relative to the aligned start
      // (0 for a window that starts before this item's entries -- an item can be the tail of a
      // unit: the second region of a partitioned graph --, 2^32 - 1 for windows past the last).
      int64_t const u0    = ((int64_t)t * kXsSlices + k) * n_win;
      int64_t const beg   = uniform64(rec[3]);
      int64_t const end   = uniform64(rec[4]);
      int64_t const abeg  = beg & ~int64_t(CGA_XS_ALIGN - 1);
      uint32_t const rbeg = static_cast<uint32_t>(beg - abeg);
      uint32_t const rend = static_cast<uint32_t>(end - abeg);
      auto bound = [&](int w) -> uint32_t {
        if (w <= 0) return 0u;
        if (w >= n_win) return 0xFFFFFFFFu;
        // a record carries the starts of windows 1 .. kXsRecWindows - 1; beyond: from memory
        int64_t const s = w < kXsRecWindows ? uniform64(rec[4 + w]) : xs.unit_off[u0 + w];
        int64_t const d = s - abeg;
        return d > 0xFFFFFFFELL ? 0xFFFFFFFFu : d < 0 ? 0u : static_cast<uint32_t>(d);
      };
      // this piece's share of the unit, in wave passes
      uint32_t const n_pass = (rend + kPass - 1) / kPass;
      uint32_t const pb_lo  = static_cast<uint32_t>(static_cast<uint64_t>(n_pass) * piece / n_pc) * kPass;
      uint32_t const pb_hi  = static_cast<uint32_t>(static_cast<uint64_t>(n_pass) * (piece + 1) / n_pc) * kPass;
      // (a piece's share is whole workgroup passes; the wave passes past the unit's end are not run)
      uint32_t const pb_end = pb_hi < rend ? pb_hi : rend;
      uint32_t const n_wp   = pb_end > pb_lo ? (pb_end - pb_lo + kWavePass - 1) / kWavePass : 0u;
      // window of the piece's first entry: the last one that starts at or before it (the starts
      // ascend; empty windows share a start with their successor).  A wave's passes ascend, so
      // its window only ever moves forward from here.
      int w_cur = 0;
      {
        uint32_t const p0 = pb_lo > rbeg ? pb_lo : rbeg;
        int lo = 0, hi = n_win - 1;
        while (lo < hi) {
          int const mid = (lo + hi + 1) >> 1;
          if (bound(mid) <= p0) lo = mid; else hi = mid - 1;
        }
        w_cur = lo;
      }
      uint32_t nb = bound(w_cur + 1);  // where the next window starts
      // Loads carry no predicates (a branch around a load defeats the compiler's vmcnt
      // bookkeeping): positions past the unit's end are clamped to its last entry and dropped.
      uint32_t const q_last = (rend > 0 ? rend : 1u) - 1u;
      uint32_t en[EPL];
      ValT w[EPL];
      // The pass's first entry is a wave-uniform address; a lane adds a small offset of its own
      // (scalar base + 32-bit lane offset: no 64-bit address arithmetic per load, no address pairs
      // in vector registers).
      auto fetch = [&](uint32_t pb, uint32_t* en_out, ValT* w_out) {
        uint32_t const pbc = pb < q_last ? pb : q_last;
        uint32_t lim       = q_last - pbc;
        int64_t first      = abeg + static_cast<int64_t>(pbc);
#ifdef CGA_TIMING_AIDS
        if (first < 0 || first + lim >= xs.ent_len) {
          CGA_XS_CHECK(true, 1, first, lim);
          first = 0;
          lim   = 0u;
        }
#endif
        uint32_t const* const eb = xs.ent + first;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          uint32_t q = static_cast<uint32_t>(lane) + e * 64;
          q          = q < lim ? q : lim;
          en_out[e]  = __builtin_nontemporal_load(eb + q);
          if constexpr (WEIGHTED) w_out[e] = __builtin_nontemporal_load(xs.w + first + q);
        }
      };
      // FAST: all entries of the pass exist and lie in one window (decided on uniform values)
      auto pass = [&](auto fast_tag, uint32_t pb, uint32_t pb_next, bool have_next, int w_lo, int w_hi) {
        constexpr bool FAST = decltype(fast_tag)::value;
        uint32_t const pos = pb + static_cast<uint32_t>(lane);
        bool p[EPL];
        uint32_t wn[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          uint32_t q = pos + e * 64;
          p[e]       = FAST || (q >= rbeg && q < rend);
          wn[e]      = static_cast<uint32_t>(w_lo);
        }
        if constexpr (!FAST) {
          // the pass crosses window starts: count the ones at or before each entry
          for (int wi = w_lo + 1; wi <= w_hi; ++wi) {
            uint32_t const b = bound(wi);
#pragma unroll
            for (int e = 0; e < EPL; ++e) wn[e] += pos + e * 64 >= b ? 1u : 0u;
          }
        }
        ValT xv[EPL];
        // The gathers of one window: its first slot (of this slice) is a wave-uniform address, an
        // entry's own bits a 32-bit offset from it (below 2^(35 - ROWBITS) elements) -- scalar base
        // + lane offset again.  A pass that crosses window starts gathers window by window, every
        // lane that has no entry in the window re-reading the window's first slot (which exists:
        // the window holds an entry of this unit); such passes are few.
        auto gather1 = [&](uint32_t wi, uint32_t off) -> ValT {
#ifdef CGA_TIMING_AIDS
          if (xs.no_gather) return ValT(1);
          uint32_t gi = ((wi << (kLowBits + 3)) | kbits) + off;
          if (a.xs.slot_limit != 0u && gi >= a.xs.slot_limit) atomicAdd(&xs_region_check[0], 1ull);
          if (gi >= xs.x_len) {
            CGA_XS_CHECK(true, 2, gi, (static_cast<uint64_t>(wi) << 32) | off);
            gi = 0u;
          }
          return a.x_cur[CGA_GMASK(a.xs, gi)];
#else
          [[maybe_unused]] ValT const* const xb = a.x_cur + ((static_cast<uint64_t>(wi) << (kLowBits + 3)) | kbits);
#if defined(CGA_XS_GATHER_SC1)
          return __hip_atomic_load(xb + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif defined(CGA_XS_ABL_NOGATHER)
          return ValT(1) + static_cast<ValT>(off & 1u);
#elif defined(CGA_XS_ABL_MASK)
          return xb[off & CGA_XS_ABL_MASK];
#else
          return xb[off];
#endif
#endif
        };
        if constexpr (FAST) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            uint32_t const lo = en[e] >> ROWBITS;
            xv[e]             = gather1(static_cast<uint32_t>(w_lo), ((lo & ~low_mask) << 3) | (lo & low_mask));
          }
        } else {
          // windows the pass may span with 32-bit byte offsets from the first one's base
          constexpr int kSpanMax = 1 << (29 - kLowBits - (sizeof(ValT) == 8 ? 3 : 2));
          if (w_hi - w_lo < kSpanMax) {
            // one gather per entry, its window folded into the offset (entries that are not the
            // item's read the window's first slot)
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              uint32_t const lo = en[e] >> ROWBITS;
              uint32_t const of = ((wn[e] - static_cast<uint32_t>(w_lo)) << (kLowBits + 3)) | ((lo & ~low_mask) << 3) | (lo & low_mask);
              ValT const v      = gather1(static_cast<uint32_t>(w_lo), p[e] ? of : 0u);
              xv[e]             = p[e] ? v : ValT(0);
            }
          } else {
#pragma unroll
          for (int e = 0; e < EPL; ++e) xv[e] = ValT(0);
          for (int wi = w_lo; wi <= w_hi; ++wi) {
            if (bound(wi + 1) <= bound(wi)) continue;  // no entry of this item in window wi
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              uint32_t const lo = en[e] >> ROWBITS;
              // (an entry that is not the item's was clamped to the item's last one: dropped)
              bool const mine   = p[e] && wn[e] == static_cast<uint32_t>(wi);
              ValT const v      = gather1(static_cast<uint32_t>(wi), mine ? ((lo & ~low_mask) << 3) | (lo & low_mask) : 0u);
              xv[e]             = mine ? v : xv[e];
            }
          }
          }
        }
        uint32_t en_next[EPL];
        ValT w_next[EPL];
        if (have_next) fetch(pb_next, en_next, w_next);
        CGA_PH(unsigned long long const t2 = clock64(); ph_issue += t2 - ph_t; ph_t = t2);
        if constexpr (WEIGHTED) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) xv[e] *= w[e];
        }
#ifdef CGA_XS_PHASE_PROF
        {
#pragma unroll
          for (int e = 0; e < EPL; ++e) asm volatile("" : "+v"(xv[e]));
          unsigned long long const t3 = clock64();
          ph_wait += t3 - ph_t;
          ph_t = t3;
        }
#endif
#ifdef CGA_TIMING_AIDS
        if (xs.no_atomics) {
          ValT sum = 0;
#pragma unroll
          for (int e = 0; e < EPL; ++e) sum += p[e] ? xv[e] + static_cast<ValT>(en[e] & kRowMask) : ValT(0);
          if (sum == ValT(-1)) atomicAdd(&acc[0], 1.0);
        } else
#endif
#if defined(CGA_XS_ABL_NOATOM)
        if (true) {
          ValT sum = 0;
#pragma unroll
          for (int e = 0; e < EPL; ++e) sum += p[e] ? xv[e] + static_cast<ValT>(en[e] & kRowMask) : ValT(0);
          if (sum == ValT(-1)) atomicAdd(&acc[0], 1.0);
        } else
#endif
        {
#pragma unroll
          for (int e = 0; e < EPL; ++e)
#if defined(CGA_XS_ABL_U64)
            if (p[e]) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (((en[e] & kRowMask) << rep) | rep_sel),
                                static_cast<unsigned long long>(__float_as_uint(static_cast<float>(xv[e]))));
#elif defined(CGA_XS_ABL_U32)
            if (p[e]) atomicAdd(reinterpret_cast<unsigned int*>(acc) + (((en[e] & kRowMask) << rep) | rep_sel),
                                __float_as_uint(static_cast<float>(xv[e])));
#else
            if (p[e]) atomicAdd(&acc[((en[e] & kRowMask) << rep) | rep_sel], static_cast<double>(xv[e]));
#endif
        }
        CGA_PH(unsigned long long const t4 = clock64(); ph_acc += t4 - ph_t; ph_t = t4; ph_n += 1; ph_fast += FAST ? 1 : 0);
        if (have_next) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            en[e] = en_next[e];
            if constexpr (WEIGHTED) w[e] = w_next[e];
          }
        }
      };
      // the next item is asked for at the start of this one and its record during the first pass,
      // so neither round trip is waited for; it is parked in LDS at the end
      int q_next     = 0;
      int64_t r_next = -1;
      bool rec_asked = false;
      if (wave0 && lane == 0) q_next = atomicAdd(&xs.counters[k], 1);
      uint32_t i0 = claim(), i1 = claim();
      if (i0 < n_wp) fetch(pb_lo + i0 * kWavePass, en, w);
      CGA_PH(unsigned long long const tb = clock64(); ph_bound += tb - ph_end; ph_t = tb);
      while (i0 < n_wp) {
        uint32_t const pb = pb_lo + i0 * kWavePass, pe = pb + kWavePass;
        // (the claim of the pass after next is asked for here and looked at when this pass is done:
        // the LDS serves it behind every queued atomic of the workgroup, which is not waited for)
        unsigned int i2_raw = 0;
        if (lane == 0) i2_raw = atomicAdd(&s_claim, 1u);
        // windows that end before this pass were another wave's
        while (nb <= pb && w_cur + 1 < n_win) {
          ++w_cur;
          nb = bound(w_cur + 1);
        }
        bool const have_next = i1 < n_wp;
        uint32_t const pb_next = pb_lo + i1 * kWavePass;
        CGA_PH(unsigned long long const t1 = clock64(); ph_claim += t1 - ph_t; ph_t = t1);
        if (pb >= rbeg && pe <= rend && pe <= nb) {
          pass(std::true_type{}, pb, pb_next, have_next, w_cur, w_cur);
        } else {
          int w_hi = w_cur;
          while (nb < pe && w_hi + 1 < n_win) {  // window w_hi + 1 starts inside this pass
            ++w_hi;
            nb = bound(w_hi + 1);
          }
          pass(std::false_type{}, pb, pb_next, have_next, w_cur, w_hi);
          w_cur = w_hi;
        }
        if (wave0 && !rec_asked) {
          r_next    = load_record(k, __shfl(q_next, 0));
          rec_asked = true;
        }
        i0 = i1;
        i1 = __builtin_amdgcn_readfirstlane(i2_raw);
        CGA_PH(ph_t = clock64());
      }
      CGA_PH(ph_end = clock64());
      if (wave0 && !rec_asked) r_next = load_record(k, __shfl(q_next, 0));
      CGA_XSP(pt_a = wall_clock64());
      __syncthreads();
      CGA_XSP(pt_c = wall_clock64(); pt_b1 += pt_c - pt_a);
      // the partials of a tile are padded to a multiple of 4 rows (build_xs): whole vectors
      int const rows4 = (rows + 3) & ~3;
      // (the thread's index made opaque here: what is derived from it -- LDS and partial addresses,
      // loop-invariant for the compiler -- would otherwise be kept in registers, or spilled, across
      // the passes of every item)
      int tw = tid;
      asm volatile("" : "+v"(tw));
      if (rep > 0) {
        // a row's 2^rep accumulators are neighbours: one row per thread and trip
        for (int i = tw; i < rows4; i += THREADS) {
          double sum = 0.0;
          for (int q = 0; q < (1 << rep); q += 2) {
            v2f64_t* const pa = reinterpret_cast<v2f64_t*>(acc + (i << rep) + q);
            v2f64_t const v   = *pa;
            sum += v.x + v.y;
            *pa = v2f64_t{0.0, 0.0};
          }
#ifdef CGA_TIMING_AIDS
          if ((out - xs.partial) + i < 0 || (out - xs.partial) + i >= xs.partial_len) {
            CGA_XS_CHECK(true, 3, out - xs.partial, i);
          } else if (xs.no_writeback) {
          } else
#endif
          __builtin_nontemporal_store(static_cast<ValT>(sum), out + i);
        }
      } else
      for (int i = tw * 4; i < rows4; i += THREADS * 4) {
        v2f64_t const s0 = *reinterpret_cast<v2f64_t const*>(acc + i);
        v2f64_t const s1 = *reinterpret_cast<v2f64_t const*>(acc + i + 2);
        ValT const r4[4] = {static_cast<ValT>(s0.x), static_cast<ValT>(s0.y), static_cast<ValT>(s1.x),
                            static_cast<ValT>(s1.y)};
#ifdef CGA_TIMING_AIDS
        if ((out - xs.partial) + i < 0 || (out - xs.partial) + i + 4 > xs.partial_len) {
          CGA_XS_CHECK(true, 3, out - xs.partial, i);
        } else if (xs.no_writeback) {
        } else
#endif
        stream_store_vals4(out + i, r4);
        v2f64_t const z = {0.0, 0.0};
        *reinterpret_cast<v2f64_t*>(acc + i)     = z;
        *reinterpret_cast<v2f64_t*>(acc + i + 2) = z;
      }
      CGA_XSP(pt_wbk += wall_clock64() - pt_c);
      if (wave0) {
        int lw = lane;
        asm volatile("" : "+v"(lw));
        if (lw < kXsRecWords) s_rec[buf ^ 1][lw] = r_next;
      }
      if (tid == 0) s_claim = 0;
      buf ^= 1;
      lds_barrier();
      CGA_XSP(pt_bound += wall_clock64() - pt_a);
    }
  }
#ifdef CGA_XS_PHASE_PROF
  if (lane == 0) {
    atomicAdd(&xs_phase_buf[0], ph_n);
    atomicAdd(&xs_phase_buf[1], ph_claim);
    atomicAdd(&xs_phase_buf[2], ph_issue);
    atomicAdd(&xs_phase_buf[3], ph_wait);
    atomicAdd(&xs_phase_buf[4], ph_acc);
    atomicAdd(&xs_phase_buf[5], ph_bound);
    atomicAdd(&xs_phase_buf[6], clock64() - ph_k0);
    atomicAdd(&xs_phase_buf[7], ph_fast);
  }
#endif
#ifdef CGA_TIMING_AIDS
  if (xs.profile && tid == 0 && blockIdx.x < 1024) {
    unsigned long long* o8 = xs_prof_buf + 8 * blockIdx.x;
    o8[0] = my; o8[1] = wall_clock64() - pt_t0; o8[2] = pt_bound; o8[3] = pt_n; o8[4] = pt_stolen;
    o8[5] = pt_b1; o8[6] = pt_wbk; o8[7] = pt_t0;
  }
#endif
}


// ---- xs_kernel without workgroup-wide item boundaries.
// In xs_kernel the 16 waves of a workgroup finish an item together: barrier, write-back, barrier,
// re-fill -- ~25 000 clock ticks (11 us) per item in which the workgroup streams nothing, 30 % of a
// wave's time at scale 26 and most of it for the 600 tiles of low-degree rows, whose units are
// three wave passes long (profiles/r3_xs_phases.md).  Here the workgroup keeps TWO banks of row
// accumulators (8192-row tiles: 2 x 64 KiB) and consecutive items alternate between them, so a wave
// that finds no pass left in item j walks on into item j + 1 while the others finish j; the LAST
// wave to leave an item writes its bank back, clears it and hands it to item j + 2.  No barrier
// after the start of the kernel.  The control words live in LDS:
//   s_ready[j & 3] == j + 1   the record of item j is in s_rec[j & 3] (written by the first wave that
//                             entered item j - 2; the queue pop and the record load are its two
//                             round trips to memory, two items ahead of their use)
//   s_bank_item[j & 1] == j   bank j & 1 is clear and its counters are reset for item j
//   s_claim[b], s_left[b]     next wave pass of the bank's item; waves that have not left it yet
// Waits are spins on those words with s_sleep; every spin gives up after kXsSpinLimit rounds and
// raises ctrl->done with the error bit (a logic error then ends the kernel instead of hanging it).
constexpr int kXs2Threads   = 1024;
constexpr int kXs2Waves     = kXs2Threads / 64;
constexpr unsigned int kXsSpinLimit = 1u << 24;

template <typename ValT, bool WEIGHTED, int ROWBITS, int EPL>
__global__ __launch_bounds__(kXs2Threads) void xs2_kernel(XsLaunch<ValT> a)
{
  constexpr int kLowBits       = 32 - ROWBITS;
  constexpr uint32_t kRowMask  = (1u << ROWBITS) - 1u;
  constexpr uint32_t kPass     = kXs2Threads * EPL;  // granule of a piece's share of its unit (xs_records_kernel)
  constexpr uint32_t kWavePass = 64 * EPL;           // entries of one wave pass
  constexpr int kRows          = 1 << ROWBITS;
  extern __shared__ double xs_lds[];  // [2][kRows]
  __shared__ int64_t s_rec[4][kXsRecWords];
  __shared__ unsigned int s_ready[4], s_enter[4], s_claim[2], s_left[2], s_bank_item[2], s_rot, s_fail;
  // write-back of a bank, shared by the waves that have nothing else to do: s_wb_claim[b] ==
  // (j + 1) << 8 | c says item j's bank is being written back and chunk c (kWbRows rows) is the next
  // to take -- taken by compare-and-swap, so that a wave that looks late never takes a chunk of a
  // later item's write-back; finished chunks are counted in s_wb_done[b], and the wave that finishes
  // the last one hands the bank to item j + 2
  __shared__ unsigned int s_wb_claim[2], s_wb_done[2];
  constexpr int kWbRows = 512;
  if (stop_check(a.ctrl, a.scal, a.epsilon, false)) return;
  XsArgs<ValT> const& xs = a.xs;
  int const tid  = threadIdx.x;
  int const lane = tid & 63;
  int const my   = static_cast<int>(xcc_id() & (kXsSlices - 1));
  for (int i = tid; i < 2 * kRows; i += kXs2Threads) xs_lds[i] = 0.0;
  if (tid < 4) {
    s_ready[tid] = 0u;
    s_enter[tid] = 0u;
  }
  if (tid < 2) {
    s_claim[tid]     = 0u;
    s_left[tid]      = kXs2Waves;
    s_bank_item[tid] = static_cast<unsigned int>(tid);
    s_wb_claim[tid]  = 0u;
    s_wb_done[tid]   = 0u;
  }
  if (tid == 0) {
    s_rot  = 0u;
    s_fail = 0u;
  }
  __syncthreads();
#ifdef CGA_XS_PHASE_PROF
  // [0] passes [1] wait for record [2] wait for bank [3] produce [4] setup [5] passes [6] leave + write-back [7] kernel
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long const ph_k0 = clock64();
  unsigned long long ph_t = ph_k0;
#define CGA_PH2(i) do { unsigned long long const t_ = clock64(); ph[i] += t_ - ph_t; ph_t = t_; } while (0)
#else
#define CGA_PH2(i) do { } while (0)
#endif
  int const shift         = xs.slice_shift;
  uint32_t const low_mask = (1u << shift) - 1u;
  int const n_win         = xs.n_win;
  auto lds_word = [&](unsigned int* p) -> unsigned int {
    return __builtin_amdgcn_readfirstlane(*reinterpret_cast<unsigned int volatile*>(p));
  };
  // wait until *p == want; false: gave up (somebody failed, or the spin limit)
  auto wait_for = [&](unsigned int* p, unsigned int want) -> bool {
    for (unsigned int spins = 0; spins < kXsSpinLimit; ++spins) {
      if (lds_word(p) == want) return true;
      if ((spins & 63u) == 63u && lds_word(&s_fail) != 0u) return false;
      __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) {
      s_fail       = 1u;
      a.ctrl->done = 3;  // stopped | internal error: session_status reports it
    }
    return false;
  };
  // Takes chunks of the bank of item jj (its record is still in s_rec[jj & 3]) until none is left:
  // partials out, accumulators cleared.  Whoever finishes the last chunk resets the bank's
  // counters and hands it to item jj + 2.
  auto write_back = [&](unsigned int jj) {
    unsigned int const b  = jj & 1u;
    int64_t const* rec    = s_rec[jj & 3u];
    int const rows        = static_cast<int>(uniform64(rec[1]) >> 32);
    ValT* const out       = xs.partial + uniform64(rec[2]);
    double* const acc     = xs_lds + static_cast<size_t>(b) * kRows;
    int const rows4       = (rows + 3) & ~3;  // partial sets are whole 4-row vectors (build_xs)
    int const rep         = xs_rep_bits(rows, kRows);
    int const chunk_rows  = kWbRows >> rep;  // a chunk is kWbRows accumulators
    unsigned int const n_chunks = static_cast<unsigned int>((rows4 + chunk_rows - 1) / chunk_rows);  // 1 .. 16
    unsigned int const tag      = (jj + 1u) << 8;
    while (true) {
      unsigned int c = 0u, got = 0u;
      if (lane == 0) {
        unsigned int v = *reinterpret_cast<unsigned int volatile*>(&s_wb_claim[b]);
        while ((v & ~255u) == tag && (v & 255u) < n_chunks) {
          unsigned int const old = atomicCAS(&s_wb_claim[b], v, v + 1u);
          if (old == v) {
            got = 1u;
            break;
          }
          v = old;
        }
        c = v & 255u;
      }
      if (__builtin_amdgcn_readfirstlane(got) == 0u) return;
      c = __builtin_amdgcn_readfirstlane(c);
      int const r0 = static_cast<int>(c) * chunk_rows;
      int const r1 = r0 + chunk_rows < rows4 ? r0 + chunk_rows : rows4;
      if (rep > 0) {
        for (int i = r0 + lane; i < r1; i += 64) {
          double sum = 0.0;
          for (int q = 0; q < (1 << rep); q += 2) {
            v2f64_t* const pa = reinterpret_cast<v2f64_t*>(acc + (i << rep) + q);
            v2f64_t const v   = *pa;
            sum += v.x + v.y;
            *pa = v2f64_t{0.0, 0.0};
          }
#ifdef CGA_TIMING_AIDS
          if ((out - xs.partial) + i < 0 || (out - xs.partial) + i >= xs.partial_len) {
            unsigned long long const t = 0, k = 0;
            CGA_XS_CHECK(true, 3, out - xs.partial, i);
          } else
#endif
          __builtin_nontemporal_store(static_cast<ValT>(sum), out + i);
        }
      } else
      for (int i = r0 + lane * 4; i < r1; i += 256) {
        v2f64_t const s0 = *reinterpret_cast<v2f64_t const*>(acc + i);
        v2f64_t const s1 = *reinterpret_cast<v2f64_t const*>(acc + i + 2);
        ValT const r4[4] = {static_cast<ValT>(s0.x), static_cast<ValT>(s0.y), static_cast<ValT>(s1.x),
                            static_cast<ValT>(s1.y)};
#ifdef CGA_TIMING_AIDS
        if ((out - xs.partial) + i < 0 || (out - xs.partial) + i + 4 > xs.partial_len) {
          unsigned long long const t = 0, k = 0;
          CGA_XS_CHECK(true, 3, out - xs.partial, i);
        } else
#endif
        stream_store_vals4(out + i, r4);
        v2f64_t const z = {0.0, 0.0};
        *reinterpret_cast<v2f64_t*>(acc + i)     = z;
        *reinterpret_cast<v2f64_t*>(acc + i + 2) = z;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the chunk is clear before it counts as done
      unsigned int d = 0u;
      if (lane == 0) d = atomicAdd(&s_wb_done[b], 1u);
      if (__builtin_amdgcn_readfirstlane(d) + 1u == n_chunks) {
        if (lane == 0) {
          s_claim[b]    = 0u;
          s_left[b]     = kXs2Waves;
          s_wb_claim[b] = 0u;
          s_wb_done[b]  = 0u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *reinterpret_cast<unsigned int volatile*>(&s_bank_item[b]) = jj + 2u;
        return;
      }
    }
  };
  // Pops the next work item of this workgroup (own queue first, then the other slices') and parks
  // its record in slot j & 3; the whole wave runs it.  Queue empty everywhere: the end marker.
  auto produce = [&](unsigned int j) {
    unsigned int rot = lds_word(&s_rot);
    int64_t r        = -1;
    while (rot < static_cast<unsigned int>(kXsSlices)) {
      int const k = (my + static_cast<int>(rot)) & (kXsSlices - 1);
      int q       = 0;
      if (lane == 0) q = atomicAdd(&xs.counters[k], 1);
      q = __builtin_amdgcn_readfirstlane(q);
      if (q < xs.n_items) {
        r = xs.rec[(static_cast<int64_t>(k) * xs.n_items + q) * kXsRecWords + lane];
        if (lane == 0) r |= static_cast<int64_t>(k) << 48;  // the slice travels in the head word
        break;
      }
      ++rot;
    }
    s_rec[j & 3u][lane] = r;
    if (lane == 0) {
      atomicMax(&s_rot, rot);
      s_enter[j & 3u] = 0u;
    }
    // (one wave's LDS operations execute in order: the record is in place before the tag says so)
    if (lane == 0) *reinterpret_cast<unsigned int volatile*>(&s_ready[j & 3u]) = j + 1u;
  };
  if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
    produce(0u);
    produce(1u);
  }
  for (unsigned int j = 0;; ++j) {
    unsigned int const slot = j & 3u, bank = j & 1u;
    if (!wait_for(&s_ready[slot], j + 1u)) return;
    CGA_PH2(1);
    int64_t const* rec = s_rec[slot];
    int64_t const head = uniform64(rec[0]);  // tile | piece << 32 | pieces << 40 | slice << 48
    if (head < 0) break;
    // the bank belongs to item j - 2 until that is written back: lend a hand, then wait
    if (j >= 2u && lds_word(&s_bank_item[bank]) != j) {
      for (unsigned int spins = 0; spins < kXsSpinLimit; ++spins) {
        if ((lds_word(&s_wb_claim[bank]) & ~255u) == (j - 1u) << 8) {
          write_back(j - 2u);
          break;
        }
        if (lds_word(&s_bank_item[bank]) == j) break;
        if ((spins & 63u) == 63u && lds_word(&s_fail) != 0u) return;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (!wait_for(&s_bank_item[bank], j)) return;
    CGA_PH2(2);
    {
      unsigned int first = 0u;
      if (lane == 0) first = atomicAdd(&s_enter[slot], 1u);
      if (__builtin_amdgcn_readfirstlane(first) == 0u) {
        // (queue pops in item order: once one item is the end marker, every later one is)
        if (!wait_for(&s_ready[(j + 1u) & 3u], j + 2u)) return;
        produce(j + 2u);
      }
    }
    CGA_PH2(3);
    double* const acc          = xs_lds + static_cast<size_t>(bank) * kRows;
    unsigned int* const claimp = &s_claim[bank];
    int const t          = static_cast<int>(head & 0xffffffff);
    uint32_t const piece = static_cast<uint32_t>(head >> 32) & 0xffu;
    uint32_t const n_pc  = static_cast<uint32_t>(head >> 40) & 0xffu;
    int const k          = static_cast<int>(head >> 48) & (kXsSlices - 1);
    uint32_t const kbits = static_cast<uint32_t>(k) << shift;
    int const rows       = static_cast<int>(uniform64(rec[1]) >> 32);
    int const rep          = xs_rep_bits(rows, kRows);  // accumulators per row: see xs_kernel
    uint32_t const rep_sel = static_cast<uint32_t>(lane) & ((1u << rep) - 1u);
    int64_t const u0     = ((int64_t)t * kXsSlices + k) * n_win;
    int64_t const beg    = uniform64(rec[3]);
    int64_t const end    = uniform64(rec[4]);
    int64_t const abeg   = beg & ~int64_t(CGA_XS_ALIGN - 1);
    uint32_t const rbeg  = static_cast<uint32_t>(beg - abeg);
    uint32_t const rend  = static_cast<uint32_t>(end - abeg);
    auto bound = [&](int w) -> uint32_t {
      if (w <= 0) return 0u;
      if (w >= n_win) return 0xFFFFFFFFu;
      int64_t const s = w < kXsRecWindows ? uniform64(rec[4 + w]) : xs.unit_off[u0 + w];
      int64_t const d = s - abeg;
      return d > 0xFFFFFFFELL ? 0xFFFFFFFFu : d < 0 ? 0u : static_cast<uint32_t>(d);
    };
    auto claim = [&]() -> uint32_t {
      unsigned int c = 0;
      if (lane == 0) c = atomicAdd(claimp, 1u);
      return __builtin_amdgcn_readfirstlane(c);
    };
    uint32_t const n_pass = (rend + kPass - 1) / kPass;
    uint32_t const pb_lo  = static_cast<uint32_t>(static_cast<uint64_t>(n_pass) * piece / n_pc) * kPass;
    uint32_t const pb_hi  = static_cast<uint32_t>(static_cast<uint64_t>(n_pass) * (piece + 1) / n_pc) * kPass;
    uint32_t const pb_end = pb_hi < rend ? pb_hi : rend;
    uint32_t const n_wp   = pb_end > pb_lo ? (pb_end - pb_lo + kWavePass - 1) / kWavePass : 0u;
    uint32_t i0 = claim();
    CGA_PH2(4);
    if (i0 < n_wp) {
      uint32_t i1 = claim();
      int w_cur = 0;
      {
        uint32_t const p0 = pb_lo + i0 * kWavePass > rbeg ? pb_lo + i0 * kWavePass : rbeg;
        int lo = 0, hi = n_win - 1;
        while (lo < hi) {
          int const mid = (lo + hi + 1) >> 1;
          if (bound(mid) <= p0) lo = mid; else hi = mid - 1;
        }
        w_cur = lo;
      }
      uint32_t nb = bound(w_cur + 1);  // where the next window starts
      uint32_t const q_last = (rend > 0 ? rend : 1u) - 1u;
      uint32_t en[EPL];
      ValT w[EPL];
      auto fetch = [&](uint32_t pb, uint32_t* en_out, ValT* w_out) {
        uint32_t const pbc = pb < q_last ? pb : q_last;
        uint32_t lim       = q_last - pbc;
        int64_t first      = abeg + static_cast<int64_t>(pbc);
#ifdef CGA_TIMING_AIDS
        if (first < 0 || first + lim >= xs.ent_len) {
          CGA_XS_CHECK(true, 1, first, lim);
          first = 0;
          lim   = 0u;
        }
#endif
        uint32_t const* const eb = xs.ent + first;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          uint32_t q = static_cast<uint32_t>(lane) + e * 64;
          q          = q < lim ? q : lim;
          en_out[e]  = __builtin_nontemporal_load(eb + q);
          if constexpr (WEIGHTED) w_out[e] = __builtin_nontemporal_load(xs.w + first + q);
        }
      };
      auto pass = [&](auto fast_tag, uint32_t pb, uint32_t pb_next, bool have_next, int w_lo, int w_hi) {
        constexpr bool FAST = decltype(fast_tag)::value;
        uint32_t const pos = pb + static_cast<uint32_t>(lane);
        bool p[EPL];
        uint32_t wn[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          uint32_t q = pos + e * 64;
          p[e]       = FAST || (q >= rbeg && q < rend);
          wn[e]      = static_cast<uint32_t>(w_lo);
        }
        if constexpr (!FAST) {
          for (int wi = w_lo + 1; wi <= w_hi; ++wi) {
            uint32_t const b = bound(wi);
#pragma unroll
            for (int e = 0; e < EPL; ++e) wn[e] += pos + e * 64 >= b ? 1u : 0u;
          }
        }
        ValT xv[EPL];
        auto gather1 = [&](uint32_t wi, uint32_t off) -> ValT {
#ifdef CGA_TIMING_AIDS
          if (xs.no_gather) return ValT(1);
          uint32_t gi = ((wi << (kLowBits + 3)) | kbits) + off;
          if (a.xs.slot_limit != 0u && gi >= a.xs.slot_limit) atomicAdd(&xs_region_check[0], 1ull);
          if (gi >= xs.x_len) {
            CGA_XS_CHECK(true, 2, gi, (static_cast<uint64_t>(wi) << 32) | off);
            gi = 0u;
          }
          return a.x_cur[CGA_GMASK(a.xs, gi)];
#else
          [[maybe_unused]] ValT const* const xb = a.x_cur + ((static_cast<uint64_t>(wi) << (kLowBits + 3)) | kbits);
#if defined(CGA_XS_ABL_NOGATHER)
          return ValT(1) + static_cast<ValT>(off & 1u);
#elif defined(CGA_XS_ABL_MASK)
          return xb[off & CGA_XS_ABL_MASK];
#else
          return xb[off];
#endif
#endif
        };
        if constexpr (FAST) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            uint32_t const lo = en[e] >> ROWBITS;
            xv[e]             = gather1(static_cast<uint32_t>(w_lo), ((lo & ~low_mask) << 3) | (lo & low_mask));
          }
        } else {
          // windows the pass may span with 32-bit byte offsets from the first one's base
          constexpr int kSpanMax = 1 << (29 - kLowBits - (sizeof(ValT) == 8 ? 3 : 2));
          if (w_hi - w_lo < kSpanMax) {
            // one gather per entry, its window folded into the offset (entries that are not the
            // item's read the window's first slot)
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              uint32_t const lo = en[e] >> ROWBITS;
              uint32_t const of = ((wn[e] - static_cast<uint32_t>(w_lo)) << (kLowBits + 3)) | ((lo & ~low_mask) << 3) | (lo & low_mask);
              ValT const v      = gather1(static_cast<uint32_t>(w_lo), p[e] ? of : 0u);
              xv[e]             = p[e] ? v : ValT(0);
            }
          } else {
#pragma unroll
          for (int e = 0; e < EPL; ++e) xv[e] = ValT(0);
          for (int wi = w_lo; wi <= w_hi; ++wi) {
            if (bound(wi + 1) <= bound(wi)) continue;  // no entry of this item in window wi
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              uint32_t const lo = en[e] >> ROWBITS;
              bool const mine   = p[e] && wn[e] == static_cast<uint32_t>(wi);
              ValT const v      = gather1(static_cast<uint32_t>(wi), mine ? ((lo & ~low_mask) << 3) | (lo & low_mask) : 0u);
              xv[e]             = mine ? v : xv[e];
            }
          }
          }
        }
        uint32_t en_next[EPL];
        ValT w_next[EPL];
        if (have_next) fetch(pb_next, en_next, w_next);
        if constexpr (WEIGHTED) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) xv[e] *= w[e];
        }
#if defined(CGA_XS_ABL_NOATOM)
        if (true) {
          ValT sum = 0;
#pragma unroll
          for (int e = 0; e < EPL; ++e) sum += p[e] ? xv[e] + static_cast<ValT>(en[e] & kRowMask) : ValT(0);
          if (sum == ValT(-1)) atomicAdd(&acc[0], 1.0);
        } else
#endif
        {
#pragma unroll
          for (int e = 0; e < EPL; ++e)
#if defined(CGA_XS_ABL_U64)
            if (p[e]) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (((en[e] & kRowMask) << rep) | rep_sel),
                                static_cast<unsigned long long>(__float_as_uint(static_cast<float>(xv[e]))));
#elif defined(CGA_XS_ABL_U32)
            if (p[e]) atomicAdd(reinterpret_cast<unsigned int*>(acc) + (((en[e] & kRowMask) << rep) | rep_sel),
                                __float_as_uint(static_cast<float>(xv[e])));
#else
            if (p[e]) atomicAdd(&acc[((en[e] & kRowMask) << rep) | rep_sel], static_cast<double>(xv[e]));
#endif
        }
        if (have_next) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            en[e] = en_next[e];
            if constexpr (WEIGHTED) w[e] = w_next[e];
          }
        }
      };
      fetch(pb_lo + i0 * kWavePass, en, w);
      while (i0 < n_wp) {
        uint32_t const pb = pb_lo + i0 * kWavePass, pe = pb + kWavePass;
        // (the claim of the pass after next is asked for here and looked at when this pass is done:
        // the LDS serves it behind every queued atomic of the workgroup, which is not waited for)
        unsigned int i2_raw = 0;
        if (lane == 0) i2_raw = atomicAdd(claimp, 1u);
        while (nb <= pb && w_cur + 1 < n_win) {  // windows that end before this pass were another wave's
          ++w_cur;
          nb = bound(w_cur + 1);
        }
        bool const have_next   = i1 < n_wp;
        uint32_t const pb_next = pb_lo + i1 * kWavePass;
        if (pb >= rbeg && pe <= rend && pe <= nb) {
          pass(std::true_type{}, pb, pb_next, have_next, w_cur, w_cur);
        } else {
          int w_hi = w_cur;
          while (nb < pe && w_hi + 1 < n_win) {  // window w_hi + 1 starts inside this pass
            ++w_hi;
            nb = bound(w_hi + 1);
          }
          pass(std::false_type{}, pb, pb_next, have_next, w_cur, w_hi);
          w_cur = w_hi;
        }
        i0 = i1;
        i1 = __builtin_amdgcn_readfirstlane(i2_raw);
        CGA_PH(ph[0] += 1);
      }
    }
    CGA_PH2(5);
    // ---- leave the item: this wave's LDS atomics have landed before it says so
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned int left = 0u;
    if (lane == 0) left = atomicSub(&s_left[bank], 1u);
    if (__builtin_amdgcn_readfirstlane(left) == 1u) {
      // the last wave out: every other wave's atomics preceded its own decrement
      if (lane == 0) *reinterpret_cast<unsigned int volatile*>(&s_wb_claim[bank]) = (j + 1u) << 8;
      write_back(j);
    }
    CGA_PH2(6);
  }
#ifdef CGA_XS_PHASE_PROF
  if (lane == 0) {
    ph[7] = clock64() - ph_k0;
    for (int i = 0; i < 8; ++i) atomicAdd(&xs_phase_buf[i], ph[i]);
  }
#endif
}

#ifdef CGA_TIMING_AIDS
__global__ void xs_region_count_kernel() { xs_region_check[1] += 1ull; }

__global__ void xs_prof_dump_kernel(int n)
{
  for (int i = 0; i < n; ++i) {
    unsigned long long* o = xs_prof_buf + 8 * i;
    printf("XSP %d %llu %llu %llu %llu %llu %llu %llu %llu\n", i, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
  }
}
#endif

// the instantiations of xs_kernel that exist, by what the plan says of its tiles
template <typename ValT>
struct XsKernelChoice {
  void (*kern)(XsLaunch<ValT>);
  int threads;
  int lds;
};

// Threads of a workgroup: 4096 threads per CU whatever the tile (256 x 4, 512 x 2, 1024 x 1) -- or,
// `wide`, two 1024-thread workgroups on 8192-row tiles (8 waves per SIMD, 64 VGPRs a lane), which
// the unweighted fp32 / fp64 kernels fit into without spilling.
constexpr int xs_threads_for(int row_bits, bool wide = false)
{
  return wide && row_bits == 13 ? 1024 : kThreads << (row_bits - kXsRowBits);
}

template <typename ValT>
XsKernelChoice<ValT> xs_pick(bool weighted, int row_bits, int epl, int threads, bool async)
{
  if (async) {
#define CGA_XS2_CASE(W, RB, EPL)                                                          \
  if (weighted == W && row_bits == RB && epl == EPL)                                      \
    return XsKernelChoice<ValT>{xs2_kernel<ValT, W, RB, EPL>, kXs2Threads,                \
                                2 * (static_cast<int>(sizeof(double)) << RB)};
    CGA_XS2_CASE(false, 12, 8)
    CGA_XS2_CASE(false, 13, 8)
    CGA_XS2_CASE(false, 12, 16)
    CGA_XS2_CASE(false, 13, 16)
    if constexpr (sizeof(ValT) == 4) {  // (fp64 weights: 12 entries a lane would spill)
      CGA_XS2_CASE(true, 12, 12)
      CGA_XS2_CASE(true, 13, 12)
    }
    CGA_XS2_CASE(true, 12, 8)
    CGA_XS2_CASE(true, 13, 8)
#undef CGA_XS2_CASE
    throw Failure(CUGRAPH_UNKNOWN_ERROR, "internal: no xs2_kernel for this tile layout");
  }
#define CGA_XS_CASE(W, RB, EPL, TH, WPE)                                                  \
  if (weighted == W && row_bits == RB && epl == EPL && threads == TH)                     \
    return XsKernelChoice<ValT>{xs_kernel<ValT, W, TH, RB, EPL, WPE>, TH,                 \
                                static_cast<int>(sizeof(double)) << RB};
  CGA_XS_CASE(false, 12, 8, 256, 1)
  CGA_XS_CASE(false, 13, 8, 512, 1)
  CGA_XS_CASE(false, 14, 8, 1024, 1)
  CGA_XS_CASE(true, 12, 8, 256, 1)
  CGA_XS_CASE(true, 13, 8, 512, 1)
  CGA_XS_CASE(true, 14, 8, 1024, 1)
  if constexpr (sizeof(ValT) == 4) {  // (the fp64 kernel needs more than 64 registers)
    CGA_XS_CASE(false, 13, 8, 1024, 8)
  }
#undef CGA_XS_CASE
  throw Failure(CUGRAPH_UNKNOWN_ERROR, "internal: no xs_kernel for this tile layout");
}

template <typename OffT, typename ValT, bool WEIGHTED>
__global__ __launch_bounds__(kThreads) void spmv_kernel(IterArgs<OffT, ValT> a)
{
  if (stop_check(a.ctrl, a.scal, a.epsilon, true)) return;
  Epilogue<OffT, ValT> ep(a);
  int blk = blockIdx.x;
  // order in the launch: hub items, wave64, 16-lane, thread per vertex, isolated (XS tiles: xs_kernel)
  if (blk < a.buckets[0].block_begin + a.buckets[0].n_blocks) {
    hub_bucket<OffT, ValT, WEIGHTED>(a, a.buckets[0]);
  } else if (blk < a.buckets[1].block_begin + a.buckets[1].n_blocks) {
    group_bucket<64, OffT, ValT, WEIGHTED>(a, a.buckets[1], ep);
  } else if (blk < a.buckets[2].block_begin + a.buckets[2].n_blocks) {
    group_bucket<16, OffT, ValT, WEIGHTED>(a, a.buckets[2], ep);
  } else if (blk < a.buckets[3].block_begin + a.buckets[3].n_blocks) {
    thread_bucket<OffT, ValT, WEIGHTED>(a, a.buckets[3], ep);
  } else {
    isolated_bucket<OffT, ValT>(a, a.buckets[4], ep);
  }
  write_partials(ep.dangling, ep.delta, a.partials);
}

// epilogue of the hub rows: add the pieces of each column in a fixed order, then the usual update
template <typename OffT, typename ValT>
__global__ __launch_bounds__(kThreads) void hub_finish_kernel(IterArgs<OffT, ValT> a,
                                                              int64_t const* __restrict__ row_item_off,
                                                              double* __restrict__ partials)
{
  if (stop_check(a.ctrl, a.scal, a.epsilon, false)) return;
  Epilogue<OffT, ValT> ep(a);
  Bucket const& b = a.buckets[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < b.count;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = b.list ? static_cast<int64_t>(b.list[i]) : b.first + i;
    double s  = 0.0;
    for (int64_t k = row_item_off[i]; k < row_item_off[i + 1]; ++k) s += a.hub_item_partial[k];
    ep.finish(v, static_cast<ValT>(s));
  }
  write_partials(ep.dangling, ep.delta, partials);
}

// epilogue of the XS rows: the partials of a row (one per slice and piece of its tile), added in a
// fixed order.  A work item is a run of up to 512 rows of one tile; its record (4 words, one
// scalar load): first row relative to row_first, rows in the run | partials per row << 32,
// offset of the run's first partial, rows of the tile (= stride between partials of a row).
constexpr int kXsFinishRows = 512;
template <typename OffT, typename ValT>
__global__ __launch_bounds__(kThreads) void xs_finish_kernel(IterArgs<OffT, ValT> a,
                                                             int64_t const* __restrict__ fin_items,
                                                             int n_fin_items,
                                                             double* __restrict__ partials)
{
  if (stop_check(a.ctrl, a.scal, a.epsilon, false)) return;
  Epilogue<OffT, ValT> ep(a);
  XsArgs<ValT> const& xs = a.xs;
  for (int it = blockIdx.x; it < n_fin_items; it += gridDim.x) {
    int64_t const* r  = fin_items + 4 * static_cast<int64_t>(it);
    int64_t const row = xs.row_first + r[0];
    int const count   = static_cast<int>(r[1] & 0xffffffff);
    int const m       = static_cast<int>(r[1] >> 32);  // 8 * pieces
    ValT const* p0    = xs.partial + r[2];
    int64_t const stride = r[3];
    // two rows per thread and eight partials per row at a time (m is a multiple of 8), so that
    // the loads of a step are in flight together
    int const i0  = threadIdx.x;
    int const i1  = i0 + kThreads;
    bool const h0 = i0 < count, h1 = i1 < count;
    // loads without predicates (rows past the run's end re-read its last row and are dropped)
    int const c0 = h0 ? i0 : count - 1, c1 = h1 ? i1 : count - 1;
    double s0 = 0.0, s1 = 0.0;
    for (int j = 0; j < m; j += kXsSlices) {
      ValT v0[kXsSlices], v1[kXsSlices];
#pragma unroll
      for (int u = 0; u < kXsSlices; ++u) {
        ValT const* pj = p0 + (j + u) * stride;
        v0[u]          = pj[c0];
        v1[u]          = pj[c1];
      }
#pragma unroll
      for (int u = 0; u < kXsSlices; ++u) {
        s0 += static_cast<double>(v0[u]);
        s1 += static_cast<double>(v1[u]);
      }
    }
    if (h0) ep.finish(row + i0, static_cast<ValT>(s0));
    if (h1) ep.finish(row + i1, static_cast<ValT>(s1));
  }
  write_partials(ep.dangling, ep.delta, partials);
}

// One record per (slice, work item): what xs_kernel needs to start on that piece of the unit.
//  [0] tile | piece << 32 | pieces << 40   [1] first row | rows << 32   [2] offset of its partials
//  [3] first entry  [4] end of the item's entries  [5..63] start of windows 1..59 (INT64_MAX: none)
// An item covers the windows [win_lo, win_hi) of its unit (all of them, or one region of a
// partitioned graph's source space) and is piece `piece` of `pieces` of that range; its partials
// are set k * tile_sets[tile] + item.pset of the tile's sets.
struct XsItem {
  int32_t tile, piece, pieces, pset;
  int32_t win_lo, win_hi;  // the windows of its unit the item covers
};
__global__ void xs_records_kernel(XsItem const* __restrict__ items, int n_items,
                                  int32_t const* __restrict__ tile_row,
                                  int64_t const* __restrict__ unit_off,
                                  int64_t const* __restrict__ pbase, int n_win,
                                  int64_t* __restrict__ rec)
{
  int64_t total = static_cast<int64_t>(kXsSlices) * n_items;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int k = static_cast<int>(i / n_items), q = static_cast<int>(i % n_items);
    XsItem it = items[q];
    int64_t t = it.tile, piece = it.piece, n_pc = it.pieces;
    int64_t row0 = tile_row[t], rows = tile_row[t + 1] - row0;
    int64_t rows_pad = (rows + 3) & ~int64_t(3);  // partials of a tile: whole 4-row vectors (build_xs)
    int64_t sets = (pbase[t + 1] - pbase[t]) / (rows_pad * kXsSlices);
    int64_t u0 = (t * kXsSlices + k) * n_win;
    int64_t* r = rec + i * kXsRecWords;
    r[0] = t | (piece << 32) | (n_pc << 40);
    r[1] = row0 | (rows << 32);
    r[2] = pbase[t] + (k * sets + it.pset) * rows_pad;
    r[3] = unit_off[u0 + it.win_lo];
    r[4] = unit_off[u0 + it.win_hi];
    for (int w = 1; w < kXsRecWindows; ++w) r[4 + w] = w < n_win ? unit_off[u0 + w] : INT64_MAX;
  }
}

// ---- XS build: unit id ((tile * 8 + slice) * n_win + window) and row of every edge of the XS row
// range.  Edge-parallel: a workgroup takes kXsLabelChunk consecutive edges, finds the rows they
// span by two binary searches of the offsets and keeps those offsets in LDS (every XS row has at
// least kXsMinDegree edges, so a chunk spans a bounded number of rows).  A row-parallel walk is
// bound by the longest row: 29 ms at scale 26, where one row has millions of edges.
constexpr int kXsLabelChunk = 4096;
constexpr int kXsLabelRows  = kXsLabelChunk + 2;  // rows of degree >= 1

template <typename OffT>
__global__ __launch_bounds__(kThreads) void xs_label_kernel(
  OffT const* __restrict__ offsets, int32_t const* __restrict__ indices, int64_t row_first, int64_t n_rows,
  int64_t e0, int64_t n_edges, int32_t const* __restrict__ tile_of_row, int slice_shift, int n_win,
  int row_bits, int32_t* __restrict__ unit, int32_t* __restrict__ row_of_edge,
  int32_t const* __restrict__ tile_row, uint32_t* __restrict__ packed, int sub_bits)
{
  __shared__ int64_t s_off[kXsLabelRows];
  __shared__ int64_t s_range[2];
  OffT const* const off = offsets + row_first;  // off[i] - e0: first edge of XS row i
  for (int64_t c0 = blockIdx.x * (int64_t)kXsLabelChunk; c0 < n_edges; c0 += (int64_t)gridDim.x * kXsLabelChunk) {
    int64_t const c1 = c0 + kXsLabelChunk < n_edges ? c0 + kXsLabelChunk : n_edges;
    if (threadIdx.x < 2) {
      // last row that starts at or before the chunk's first / last edge
      int64_t const target = e0 + (threadIdx.x == 0 ? c0 : c1 - 1);
      int64_t lo = 0, hi = n_rows;  // invariant: off[lo] <= target < off[hi]
      while (hi - lo > 1) {
        int64_t const mid = (lo + hi) >> 1;
        if (static_cast<int64_t>(off[mid]) <= target) lo = mid; else hi = mid;
      }
      s_range[threadIdx.x] = lo;
    }
    __syncthreads();
    int64_t const r_lo = s_range[0], r_hi = s_range[1];
    int const n_r      = static_cast<int>(r_hi - r_lo) + 1;  // empty rows in between count too
    bool const in_lds  = n_r + 1 <= kXsLabelRows;
    if (in_lds)
      for (int k = threadIdx.x; k <= n_r; k += blockDim.x) s_off[k] = static_cast<int64_t>(off[r_lo + k]) - e0;
    __syncthreads();
    for (int64_t j = c0 + threadIdx.x; j < c1; j += blockDim.x) {
      int lo = 0, hi = n_r;  // invariant: first edge of row r_lo + lo <= j < first edge of row r_lo + hi
      while (hi - lo > 1) {
        int const mid     = (lo + hi) >> 1;
        int64_t const beg = in_lds ? s_off[mid] : static_cast<int64_t>(off[r_lo + mid]) - e0;
        if (beg <= j) lo = mid; else hi = mid;
      }
      int64_t const i = r_lo + lo;
      int32_t const t = tile_of_row[i];
      uint32_t src    = static_cast<uint32_t>(indices[e0 + j]);
      uint32_t slice  = xs_slice_of(src, slice_shift);
      uint32_t win    = xs_slice_local(src, slice_shift) >> (32 - row_bits);
      int32_t const u = (t * kXsSlices + static_cast<int32_t>(slice)) * n_win + static_cast<int32_t>(win);
      if (packed) {
        // the finished entry (unweighted graphs: nothing else travels with an edge); the sort
        // columns are 2^sub_bits sub-units of the unit, by the top bits of the source position
        uint32_t const low = xs_slice_local(src, slice_shift) & ((1u << (32 - row_bits)) - 1u);
        packed[j]          = (low << row_bits) | static_cast<uint32_t>(i - tile_row[t]);
        unit[j]            = (u << sub_bits) | static_cast<int32_t>(low >> (32 - row_bits - sub_bits));
      } else {
        unit[j]        = u;
        row_of_edge[j] = static_cast<int32_t>(i);
      }
    }
    __syncthreads();
  }
}

// A tile is a run of rows: a new one starts whenever a row's first edge falls into a new window
// of kXsTileEdges edges, and at every 2^row_bits-th row.  So a tile has at most 2^row_bits rows
// and about kXsTileEdges edges (64 rows of 16K edges, or 4096 rows of 16); a longer column simply
// makes its tile longer.  flags mark the first row of each tile; a scan numbers the tiles.
template <typename OffT>
__global__ void xs_break_flags_kernel(OffT const* __restrict__ offsets, int64_t row_first,
                                      int64_t n_rows, int64_t e0, int64_t tile_edges, int64_t max_rows,
                                      unsigned int* __restrict__ flags)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t w  = (static_cast<int64_t>(offsets[row_first + i]) - e0) / tile_edges;
    int64_t wp = i == 0 ? -1 : (static_cast<int64_t>(offsets[row_first + i - 1]) - e0) / tile_edges;
    flags[i]   = (w != wp || (i % max_rows) == 0) ? 1u : 0u;
  }
}

__global__ void xs_tiles_kernel(unsigned int const* __restrict__ flags,
                                int64_t const* __restrict__ pos, int64_t n_rows,
                                int32_t* __restrict__ tile_of_row, int32_t* __restrict__ tile_row)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i <= n_rows;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i == n_rows) {
      tile_row[pos[n_rows]] = static_cast<int32_t>(n_rows);
      continue;
    }
    int64_t t      = pos[i] + flags[i] - 1;
    tile_of_row[i] = static_cast<int32_t>(t);
    if (flags[i]) tile_row[t] = static_cast<int32_t>(i);
  }
}

template <typename OffT, typename UOffT, typename ValT>
__global__ void xs_payload_kernel(int32_t const* __restrict__ tile_of_row, int64_t e0,
                                  int64_t n_edges, uint32_t const* __restrict__ perm,
                                  int32_t const* __restrict__ row_of_edge,
                                  int32_t const* __restrict__ tile_row,
                                  ValT const* __restrict__ w_in, UOffT const* __restrict__ unit_off_in,
                                  int64_t n_units, int32_t const* __restrict__ src_sorted,
                                  int slice_shift, int row_bits, uint32_t* __restrict__ ent,
                                  ValT* __restrict__ w_out, int64_t* __restrict__ unit_off_out)
{
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n_edges;
       k += (int64_t)gridDim.x * blockDim.x) {
    uint32_t e = perm[k];
    int32_t r  = row_of_edge[e];
    uint32_t rl  = static_cast<uint32_t>(r - tile_row[tile_of_row[r]]);
    uint32_t low = xs_slice_local(static_cast<uint32_t>(src_sorted[k]), slice_shift) &
                   ((1u << (32 - row_bits)) - 1u);
    ent[k]       = (low << row_bits) | rl;
    if (w_in) w_out[k] = w_in[e0 + e];
  }
  for (int64_t u = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; u <= n_units;
       u += (int64_t)gridDim.x * blockDim.x)
    unit_off_out[u] = static_cast<int64_t>(unit_off_in[u]);
}

template <typename UOffT>
__global__ void xs_unit_off_kernel(UOffT const* __restrict__ in, int64_t n_units, int sub_bits,
                                   int64_t* __restrict__ out)
{
  for (int64_t u = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; u <= n_units;
       u += (int64_t)gridDim.x * blockDim.x)
    out[u] = static_cast<int64_t>(in[u << sub_bits]);
}

// debug aid (CUGRAPH_AMD_XS_VERIFY=1): entries of a unit must ascend by source
__global__ void xs_verify_kernel(uint32_t const* __restrict__ ent, int64_t const* __restrict__ unit_off,
                                 int64_t n_units, int row_bits, unsigned long long* __restrict__ out)
{
  for (int64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
    int64_t b = unit_off[u], e = unit_off[u + 1];
    unsigned long long bad = 0, same = 0;
    for (int64_t k = b + 1 + threadIdx.x; k < e; k += blockDim.x) {
      uint32_t a0 = ent[k - 1] >> row_bits, a1 = ent[k] >> row_bits;
      bad += a1 < a0;
      same += a1 == a0;
    }
    if (bad) atomicAdd(out, bad);
    if (same) atomicAdd(out + 1, same);
    if (threadIdx.x == 0 && e > b) atomicAdd(out + 2, 1ull);
  }
}

// hub work items: column i of the hub bucket is cut into ceil(deg / kHubChunk) pieces
template <typename OffT>
__global__ void hub_item_count_kernel(OffT const* offsets, int32_t const* list, int64_t first,
                                      int64_t count, unsigned int* n_items)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v   = list ? static_cast<int64_t>(list[i]) : first + i;
    int64_t base = static_cast<int64_t>(offsets[v]) & ~int64_t(3);  // as hub_item_fill_kernel
    int64_t span = static_cast<int64_t>(offsets[v + 1]) - base;
    n_items[i]   = static_cast<unsigned int>((span + kHubChunk - 1) / kHubChunk);
  }
}

template <typename OffT>
__global__ void hub_item_fill_kernel(OffT const* offsets, int32_t const* list, int64_t first,
                                     int64_t count, int64_t const* row_item_off, HubItem* items)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v   = list ? static_cast<int64_t>(list[i]) : first + i;
    int64_t beg = static_cast<int64_t>(offsets[v]), end = static_cast<int64_t>(offsets[v + 1]);
    int64_t k   = row_item_off[i];
    // pieces start at multiples of 4 edges from the aligned column start, so that every piece's
    // vector loads stay 16-byte aligned without overlapping its neighbour's
    int64_t base = beg & ~int64_t(3);
    for (int64_t p = base; p < end; p += kHubChunk, ++k) {
      items[k].beg = p < beg ? beg : p;
      items[k].end = p + kHubChunk < end ? p + kHubChunk : end;
    }
  }
}

// pr' += teleport * p_i for a short personalization list, with the matching corrections of x',
// the L1 delta and the dangling mass (pagerank_impl.cuh:299-319).  Entries are unique vertices.
template <typename ValT>
__global__ __launch_bounds__(kThreads) void pers_fix_kernel(
  int32_t const* __restrict__ pers_v, ValT const* __restrict__ pers_p, int64_t n_pers,
  ValT const* __restrict__ pr_old, ValT* __restrict__ pr_new, ValT* __restrict__ x_next,
  ValT const* __restrict__ out_w, double const* __restrict__ scal, Ctrl* ctrl, double epsilon,
  ValT alpha, ValT one_minus_alpha, int64_t vertex_first, int32_t const* __restrict__ x_pos,
  double* __restrict__ partials)
{
  if (stop_check(ctrl, scal, epsilon, false)) return;
  ValT teleport = static_cast<ValT>(scal[0]) * alpha + one_minus_alpha;
  double delta = 0.0, dangling = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pers;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = pers_v[i];
    ValT pr   = pr_new[v];
    ValT pr2  = pr + teleport * pers_p[i];
    ValT old  = pr_old[v];
    ValT ow   = out_w[v];
    delta += fabs(static_cast<double>(pr2) - static_cast<double>(old)) -
             fabs(static_cast<double>(pr) - static_cast<double>(old));
    ValT x = pr2;
    if (ow == ValT(0)) {
      dangling += static_cast<double>(pr2) - static_cast<double>(pr);
    } else {
      x = pr2 / ow;
    }
    pr_new[v]                = pr2;
    if (x_pos) {
      int32_t slot = x_pos[v];
      if (slot >= 0) x_next[slot] = x;
    } else {
      x_next[vertex_first + v] = x;
    }
  }
  write_partials(dangling, delta, partials);
}

// one workgroup: fold the per-workgroup partials in a fixed order -> bitwise reproducible scalars
// Dead rows (vertices no edge touches; the last n_dead rows of a degree-sorted graph) are not
// visited after the first iteration: each gets the same rank `base`, so their share of the two
// sums is added here.  base is recomputed exactly as Epilogue does, from the dangling mass the
// iteration's kernels read.
struct DeadRows {
  int64_t n_dead;   // 0: nothing to add (no dead rows, or this iteration visited them)
  int track;        // keep ctrl->base_prev up to date
  int f64;
  double alpha, one_minus_alpha, num_vertices;  // as the kernels hold them (already rounded to ValT)
};

__global__ __launch_bounds__(kThreads) void finalize_kernel(double const* __restrict__ partials,
                                                            int n_partials, double* scal, Ctrl* ctrl,
                                                            double epsilon, int count_iteration,
                                                            DeadRows dead)
{
  if (count_iteration && stop_check(ctrl, scal, epsilon, false)) return;
  double dead_dangling = 0.0, dead_delta = 0.0, base = 0.0;
  if (dead.track) {
    if (dead.f64) {
      double teleport = scal[0] * dead.alpha + dead.one_minus_alpha;
      base            = teleport / dead.num_vertices;
    } else {
      float teleport = static_cast<float>(scal[0]) * static_cast<float>(dead.alpha) +
                       static_cast<float>(dead.one_minus_alpha);
      base           = static_cast<double>(teleport / static_cast<float>(dead.num_vertices));
    }
    dead_dangling = static_cast<double>(dead.n_dead) * base;
    dead_delta    = static_cast<double>(dead.n_dead) * fabs(base - ctrl->base_prev);
  }
  __shared__ double lds[2 * (kThreads / 64)];
  double d = 0.0, l = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += kThreads) {
    d += partials[2 * i];
    l += partials[2 * i + 1];
  }
  d = group_sum(d, 64);
  l = group_sum(l, 64);
  if ((threadIdx.x & 63) == 0) {
    lds[2 * (threadIdx.x >> 6)]     = d;
    lds[2 * (threadIdx.x >> 6) + 1] = l;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ds = 0.0, ls = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) {
      ds += lds[2 * w];
      ls += lds[2 * w + 1];
    }
    // scal[0..1]: what the next iteration's kernels read; scal[2..3]: the same sums as THIS rank
    // produced them.  A multi-GPU driver all-reduces [2..3] into [0..1] (cugraph_amd.mg): an
    // iteration skipped after the stop leaves [2..3] alone, so reducing them again changes nothing.
    double const d_out = ds + dead_dangling;
    double const l_out = count_iteration ? ls + dead_delta
                                         : 1e300;  // "no previous iteration": never below epsilon, survives an 8-rank sum
    scal[0] = scal[2] = d_out;
    scal[1] = scal[3] = l_out;
    if (count_iteration) {
      ctrl->iters += 1;
      if (dead.track) ctrl->base_prev = base;
    }
  }
}

// x0 = pr0 / out_w and the dangling mass of pr0 (pagerank_impl.cuh:239-264, first trip)
template <typename ValT>
__global__ __launch_bounds__(kThreads) void prep_kernel(ValT const* __restrict__ pr,
                                                        ValT const* __restrict__ out_w, int64_t n,
                                                        int64_t vertex_first,
                                                        int32_t const* __restrict__ x_pos,
                                                        ValT* __restrict__ x,
                                                        double* __restrict__ partials)
{
  double dangling = 0.0;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n;
       v += (int64_t)gridDim.x * blockDim.x) {
    ValT p = pr[v], ow = out_w[v];
    int64_t slot = x_pos ? static_cast<int64_t>(x_pos[v]) : vertex_first + v;
    if (ow == ValT(0)) {
      dangling += static_cast<double>(p);
      if (slot >= 0) x[slot] = p;
    } else if (slot >= 0) {
      x[slot] = p / ow;
    }
  }
  write_partials(dangling, 0.0, partials);
}

template <typename ValT>
__global__ void fill_kernel(ValT* p, int64_t n, ValT value)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = value;
}

template <typename InT, typename OutT>
__global__ void cast_kernel(InT const* in, OutT* out, int64_t n)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = static_cast<OutT>(in[i]);
}

// out[src] += w over all edges (graph_weight_utils_impl.cuh:41-79), accumulated in double
template <typename ValT>
__global__ void source_sums_kernel(int32_t const* __restrict__ indices,
                                   ValT const* __restrict__ weights, int64_t n_edges,
                                   double* __restrict__ sums)
{
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_edges;
       e += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&sums[indices[e]], weights ? static_cast<double>(weights[e]) : 1.0);
}

// the same for a degree-sorted graph, whose first ids are the sources of most edges: every
// workgroup adds the weights of its span of the edges that leave the first kSumsWindow ids in LDS
// (fp64 atomics there are an order of magnitude faster than the memory-side ones, and uncontended)
// and hands each of those sums over once; the cold rest goes to memory directly
constexpr int kSumsWindow = 16384;  // 128 of the 160 KiB of LDS

template <typename ValT>
__global__ __launch_bounds__(1024) void source_sums_window_kernel(int32_t const* __restrict__ indices,
                                                                  ValT const* __restrict__ weights,
                                                                  int64_t n_edges, double* __restrict__ sums)
{
  extern __shared__ double sums_lds[];
  for (int k = threadIdx.x; k < kSumsWindow; k += blockDim.x) sums_lds[k] = 0.0;
  __syncthreads();
  int64_t const span = (n_edges + gridDim.x - 1) / gridDim.x;
  int64_t const beg = blockIdx.x * span, end = beg + span < n_edges ? beg + span : n_edges;
  for (int64_t e = beg + threadIdx.x; e < end; e += blockDim.x) {
    int32_t const s = indices[e];
    double const w  = weights ? static_cast<double>(weights[e]) : 1.0;
    if (s < kSumsWindow)
      atomicAdd(&sums_lds[s], w);
    else
      atomicAdd(&sums[s], w);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kSumsWindow; k += blockDim.x)
    if (sums_lds[k] != 0.0) atomicAdd(&sums[k], sums_lds[k]);
}

// external id -> row of the compute graph, through up to two lookup tables
struct IdMap {
  int32_t const* lookup0;  // graph's own renumber map (external -> graph id), or nullptr
  int64_t min0, range0;
  long long const* sorted0;  // not nullptr: lookup0 is indexed by the id's position in these sorted ids
  int32_t const* lookup1;  // plan's degree-sort permutation (graph id -> plan row), or nullptr
  int64_t first;           // first global id of the local range (no lookup0)
  int64_t count;           // rows
  __device__ inline int64_t row(int64_t id) const
  {
    int64_t v;
    if (lookup0) {
      v = ExtLookup{lookup0, min0, range0, sorted0}.find(id);
      if (v < 0) return -1;
    } else {
      v = id - first;
      if (v < 0 || v >= count) return -1;
    }
    return lookup1 ? static_cast<int64_t>(lookup1[v]) : v;
  }
};

// (vertex, value) pairs -> dense vector over the compute graph's rows; unknown ids are counted
template <typename IdT, typename ValT, bool ACCUMULATE>
__global__ void scatter_pairs_kernel(IdT const* __restrict__ ids, ValT const* __restrict__ vals,
                                     int64_t n, IdMap map, ValT* __restrict__ dense,
                                     unsigned int* __restrict__ n_bad)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = map.row(static_cast<int64_t>(ids[i]));
    if (v < 0) {
      atomicAdd(n_bad, 1u);
    } else if constexpr (ACCUMULATE) {
      atomicAdd(&dense[v], vals[i]);
    } else {
      dense[v] = vals[i];
    }
  }
}

// do_expensive_check: personalization vertices must not repeat (pagerank_impl.cuh:170-187)
template <typename IdT>
__global__ void count_repeated_ids_kernel(IdT const* __restrict__ ids, int64_t n, IdMap map,
                                          unsigned int* __restrict__ seen, unsigned int* __restrict__ n_repeated)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = map.row(static_cast<int64_t>(ids[i]));
    if (v >= 0 && atomicAdd(&seen[v], 1u) > 0u) atomicAdd(n_repeated, 1u);
  }
}

template <typename ValT>
__global__ void permute_out_kernel(ValT const* __restrict__ in, int32_t const* __restrict__ order,
                                   int64_t n, ValT* __restrict__ out)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[order[i]] = in[i];
}

template <typename ValT>
__global__ void value_checks_kernel(ValT const* __restrict__ vals, int64_t n,
                                    double* __restrict__ sum, unsigned int* __restrict__ n_negative)
{
  double s = 0.0;
  unsigned int neg = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    ValT v = vals[i];
    s += static_cast<double>(v);
    neg += v < ValT(0) ? 1u : 0u;
  }
  s = group_sum(s, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(sum, s);
  if (neg) atomicAdd(n_negative, neg);
}

template <typename ValT>
__global__ void scale_kernel(ValT* p, int64_t n, double divisor)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = static_cast<ValT>(static_cast<double>(p[i]) / divisor);
}

// dense personalization -> compact (vertex, p) list of the non-zero entries, ascending vertex
template <typename ValT>
__global__ void count_nonzero_kernel(ValT const* p, int64_t n, unsigned int* out)
{
  unsigned int c = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    c += p[i] != ValT(0) ? 1u : 0u;
  if (c) atomicAdd(out, c);
}

template <typename ValT>
__global__ void flag_nonzero_kernel(ValT const* dense, int64_t n, unsigned int* flags)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    flags[i] = dense[i] != ValT(0) ? 1u : 0u;
}

template <typename ValT>
__global__ void compact_nonzero_kernel(ValT const* dense, int64_t n, unsigned int const* flags,
                                       int64_t const* pos, int32_t* out_v, ValT* out_p)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    if (flags[i]) {
      out_v[pos[i]] = static_cast<int32_t>(i);
      out_p[pos[i]] = dense[i];
    }
}

template <typename OffT>
__global__ void flag_bucket_kernel(OffT const* offsets, int64_t n, int64_t lo, int64_t hi,
                                   unsigned int* flags)
{
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n;
       v += (int64_t)gridDim.x * blockDim.x) {
    int64_t deg = static_cast<int64_t>(offsets[v + 1] - offsets[v]);
    flags[v]    = (deg >= lo && deg < hi) ? 1u : 0u;
  }
}

__global__ void compact_ids_kernel(int64_t n, unsigned int const* flags, int64_t const* pos,
                                   int32_t* out)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    if (flags[i]) out[pos[i]] = static_cast<int32_t>(i);
}

inline int grid_for(int64_t items, int cap = 256 * 16)
{
  int64_t b = (items + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return static_cast<int>(b);
}

}  // namespace

#ifdef CGA_TIMING_AIDS
// (make AIDS=1) {gathers of region-0 launches beyond the region, region-0 launches checked} so far
// (make AIDS=1) xs_check_buf: violations of xs_kernel's address bounds so far and the first one
void xs_bounds_check_read(unsigned long long* out)
{
  CGA_HIP(hipDeviceSynchronize());
  CGA_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(xs_check_buf), 8 * sizeof(unsigned long long)));
}

void xs_region_check_read(unsigned long long* out)
{
  CGA_HIP(hipDeviceSynchronize());
  CGA_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(xs_region_check), 2 * sizeof(unsigned long long)));
}

#endif

// implemented in graph_build.hip (reused here for stream compaction)
void exclusive_scan_u32_to_i64(hipStream_t stream, unsigned int const* counts, int64_t n,
                               int64_t* out /* [n+1] */);

// ------------------------------------------------------------------ plan (cached on the graph)
// Degree buckets as ascending index lists over the graph's own vertex order.
struct PullPlan {
  Graph* internal{nullptr};  // degree-sorted copy the kernels run on (nullptr: the graph itself)
  // Compact source space on one GPU (as the multi-GPU layout has it): only a vertex that some edge
  // leaves is ever gathered, so only those get a slot in the gathered vector; compact is a view
  // of the compute graph (same offsets and weights, NOT owned) whose indices are slots and whose
  // x_pos maps a row to its slot.  nullptr: every source is addressed by its vertex id.
  Graph* compact{nullptr};
  // Once the XS tiles exist, the only rows that still read compact->indices are the rows below
  // the tiles (in-degree < kXsMinDegree: the tail of a degree-sorted graph, 2 % of the edges at
  // scale 26).  The slot-numbered copy is then cut down to that tail: compact->indices points
  // tail_first edges BEFORE this buffer, so that the kernels keep indexing it with CSC offsets.
  DevBuf compact_tail;
  int64_t n_dead{0};  // rows at the very end without in-edges and without a slot (no edge touches them)
  DevBuf lists[kNumBuckets];
  int64_t counts[kNumBuckets]{};
  int64_t firsts[kNumBuckets]{};
  bool contiguous{false};  // buckets are row ranges, no index lists
  // XS tiles over the rows of buckets 1 and 2 (only on a degree-sorted compute graph)
  bool xs_on{false};
  int64_t xs_row_first{0}, xs_n_rows{0}, xs_n_edges{0};
  int xs_n_tiles{0}, xs_n_fin_items{0};
  int xs_regions{1};        // 2: region 0 (hot slots) and region 1 have item lists of their own
  int xs_n_items[2]{0, 0};  // work items per slice queue, per region
  int64_t xs_partial_len{0};
  int xs_n_win{1}, xs_slice_shift{7}, xs_row_bits{12};
  int xs_epl{kXsEntriesPerLane};  // entries a lane of xs_kernel keeps in flight per pass
  int xs_threads{kThreads};       // threads of an xs_kernel workgroup (xs_threads_for)
  bool xs_async{false};           // xs2_kernel: two accumulator banks, no item barriers
  DevBuf xs_ent, xs_w, xs_unit_off, xs_tile_row, xs_rec[2], xs_pbase, xs_fin_items;
  DevBuf hub_items;         // HubItem[n_hub_items]
  DevBuf hub_row_item_off;  // int64[counts[0] + 1]
  int64_t n_hub_items{0};
  DevBuf out_w_sums;  // double[num_global_vertices]: per-source weight sums, compute-graph order
  bool have_out_w{false};
  ~PullPlan()
  {
    if (compact) {  // offsets and weights belong to the graph it views
      compact->offsets = nullptr;
      compact->weights = nullptr;
      if (compact_tail.p) compact->indices = nullptr;  // a biased pointer into compact_tail
      delete compact;
    }
    delete internal;
  }
};

void destroy_plan(PullPlan* p) { delete p; }

void compute_source_sums(hipStream_t stream, Graph const& g, double* sums);

namespace {

// [lo, hi) in-degree ranges, launch order: hubs, wave64, 16-lane groups, thread per vertex
constexpr int64_t kBucketLo[kNumBuckets] = {kXsMaxDegree, 128, kXsMinDegree, 1, 0};
constexpr int64_t kBucketHi[kNumBuckets] = {INT64_MAX, kXsMaxDegree, 128, kXsMinDegree, 1};

template <typename OffT>
void build_plan_lists(hipStream_t stream, Graph const& g, PullPlan& plan)
{
  int64_t n = g.num_vertices;
  plan.contiguous = g.degree_sorted;
  int64_t running = 0;
  DevBuf flags(static_cast<size_t>(n > 0 ? n : 1) * sizeof(unsigned int));
  DevBuf pos(static_cast<size_t>(n + 1) * sizeof(int64_t));
  for (int b = 0; b < kNumBuckets; ++b) {
    plan.counts[b] = 0;
    if (n == 0) continue;
    hipLaunchKernelGGL((flag_bucket_kernel<OffT>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                       static_cast<OffT const*>(g.offsets), n, kBucketLo[b], kBucketHi[b],
                       flags.as<unsigned int>());
    exclusive_scan_u32_to_i64(stream, flags.as<unsigned int>(), n, pos.as<int64_t>());
    int64_t total = 0;
    CGA_HIP(hipMemcpyAsync(&total, pos.as<int64_t>() + n, sizeof(int64_t), hipMemcpyDeviceToHost,
                           stream));
    CGA_HIP(hipStreamSynchronize(stream));
    plan.counts[b] = total;
    plan.firsts[b] = plan.contiguous ? running : 0;
    running += total;
    if (!plan.contiguous && total > 0 && total < n) {
      plan.lists[b].alloc(static_cast<size_t>(total) * sizeof(int32_t));
      hipLaunchKernelGGL(compact_ids_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, n,
                         flags.as<unsigned int>(), pos.as<int64_t>(), plan.lists[b].as<int32_t>());
    }  // total == n: the identity, no list needed
  }
  // hub work items
  int64_t nh = plan.counts[0];
  plan.n_hub_items = 0;
  if (nh > 0) {
    int32_t const* list = plan.lists[0].p ? plan.lists[0].as<int32_t>() : nullptr;
    DevBuf cnt(static_cast<size_t>(nh) * sizeof(unsigned int));
    plan.hub_row_item_off.alloc(static_cast<size_t>(nh + 1) * sizeof(int64_t));
    hipLaunchKernelGGL((hub_item_count_kernel<OffT>), dim3(grid_for(nh)), dim3(kThreads), 0, stream,
                       static_cast<OffT const*>(g.offsets), list, plan.firsts[0], nh,
                       cnt.as<unsigned int>());
    exclusive_scan_u32_to_i64(stream, cnt.as<unsigned int>(), nh, plan.hub_row_item_off.as<int64_t>());
    CGA_HIP(hipMemcpyAsync(&plan.n_hub_items, plan.hub_row_item_off.as<int64_t>() + nh,
                           sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    CGA_HIP(hipStreamSynchronize(stream));
    plan.hub_items.alloc(static_cast<size_t>(plan.n_hub_items > 0 ? plan.n_hub_items : 1) * sizeof(HubItem));
    CGA_HIP(hipMemsetAsync(plan.hub_items.p, 0, plan.hub_items.bytes, stream));
    hipLaunchKernelGGL((hub_item_fill_kernel<OffT>), dim3(grid_for(nh)), dim3(kThreads), 0, stream,
                       static_cast<OffT const*>(g.offsets), list, plan.firsts[0], nh,
                       plan.hub_row_item_off.as<int64_t>(), plan.hub_items.as<HubItem>());
  }
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));
}

// rows per tile (as a power of two) for a source space of `minor_space` ids or slots: see build_xs
int xs_row_bits_for(int64_t minor_space, int slice_shift, int max_bits = kXsRowBitsMax)
{
  uint32_t const last_local = xs_slice_local(static_cast<uint32_t>(minor_space - 1), slice_shift) |
                              ((1u << slice_shift) - 1u);
  int row_bits = max_bits;
  while (row_bits > kXsRowBits && (last_local >> (32 - row_bits)) >= static_cast<uint32_t>(kXsRecWindows)) --row_bits;
  return row_bits;
}

// the two settings build_xs and xs_window_slots have to agree on (a region of the gathered vector is a
// whole number of windows, and a window is 2^(35 - row bits) slots)
int xs_row_bits_env(int row_bits)
{
  char const* rb_env = std::getenv("CUGRAPH_AMD_XS_ROWBITS");
  if (rb_env && std::atoi(rb_env) >= kXsRowBits && std::atoi(rb_env) <= kXsRowBitsMax) return std::atoi(rb_env);
  return row_bits;
}
constexpr int64_t kXs2MinEntries = 40000000;  // XS entries from which xs2_kernel is the default
constexpr int kXs2RowBitsMax = 13;  // xs2_kernel: two banks of 8192 fp64 accumulators are the LDS it has
bool xs_async_on()
{
  char const* async_env = std::getenv("CUGRAPH_AMD_XS_ASYNC");
  return async_env ? std::atoi(async_env) != 0 : true;
}

template <typename OffT, typename ValT>
void build_xs(hipStream_t stream, Graph const& g, PullPlan& plan)
{
  int64_t row_first = plan.firsts[0];
  int64_t n_rows    = plan.counts[0] + plan.counts[1] + plan.counts[2];
  if (n_rows == 0) return;
  // CUGRAPH_AMD_PLAN_PROFILE=1: seconds between the marks of this function, on stderr (synchronises at each)
  bool const prof = std::getenv("CUGRAPH_AMD_PLAN_PROFILE") != nullptr;
  auto t_last     = std::chrono::steady_clock::now();
  auto mark       = [&](char const* what) {
    if (!prof) return;
    (void)hipStreamSynchronize(stream);
    auto const now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "xs plan: %-28s %.4f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  OffT const* offsets = static_cast<OffT const*>(g.offsets);
  OffT h_off[2];
  CGA_HIP(hipMemcpyAsync(&h_off[0], offsets + row_first, sizeof(OffT), hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipMemcpyAsync(&h_off[1], offsets + row_first + n_rows, sizeof(OffT), hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  int64_t e0 = static_cast<int64_t>(h_off[0]), ne = static_cast<int64_t>(h_off[1]) - e0;
  char const* min_env = std::getenv("CUGRAPH_AMD_XS_MIN_EDGES");
  int64_t min_edges   = min_env ? std::atoll(min_env) : (int64_t(1) << 22);
  if (ne < min_edges || ne == 0 || ne > 0xffffffffLL) return;
  char const* te_env  = std::getenv("CUGRAPH_AMD_XS_TILE_EDGES");
  // Tile size follows the graph: big tiles gather from fewer distinct cache lines (scale 26: xs_kernel
  // 2.32 ms at 8 M edges, 2.17 at 16 M, 2.13 at 64 M) but a small graph still needs enough of them to
  // fill the chip.  Small scales are flat in both knobs (scale 22: 0.235 ms from 1 M to 4 M).
  int64_t tile_edges = kXsTileEdges;
  while (tile_edges > (int64_t(1) << 16) && tile_edges * 32 > ne) tile_edges >>= 1;
  // From 120 M entries on there are enough work items either way, and tiles of 16 M entries gather from
  // fewer distinct lines than tiles of ne / 32 -- what a rank of a partitioned graph needs most, whose few
  // rows face the whole source space (round 3, same box: one rank of 8 / of 4 at scale 26 0.351 -> 0.315 /
  // 0.573 -> 0.539 ms per iteration; single GPU scale 24 0.352 -> 0.347 ms; below that the smaller tiles
  // win: scale 22 0.119 against 0.136 ms with 4 M, one rank of 8 at scale 24 0.135 / 0.151).  (ne / 32
  // stays where it is the larger: 64 M from scale 28 on.)  Beyond 16 M the iteration still gains (-2.5 % at
  // scale 26 with 128 M, -7 % for one rank of 8 at scale 28 with 64 M) but the plan build -- one workgroup
  // sorts each hot column of a tile -- pays more than a run of a few hundred iterations gets back.
  if (ne >= 120000000LL) tile_edges = std::max(tile_edges, int64_t(1) << 24);
  if (te_env) tile_edges = std::atoll(te_env);
  char const* sh_env  = std::getenv("CUGRAPH_AMD_XS_SLICE_SHIFT");
  int slice_shift     = sh_env ? std::atoi(sh_env) : 7;
  // Rows per tile: as many as the entry format allows.  An entry keeps 32 - row_bits bits of its
  // source position, the rest is implied by its window, and a record carries 59 window starts (up
  // to 16 windows they live in scalar registers, beyond that an entry's window is found by binary
  // search of the record in LDS): 2^14 rows reach 123 M source ids (or slots), 2^13 rows 247 M,
  // 2^12 rows 495 M (beyond: window starts read from memory).  Twice the rows per tile = twice
  // the entries per (tile, slice) unit over the same sources, so fewer distinct lines per gather
  // instruction, and half the tiles.  Measured (xs_kernel, 12 / 13 / 14 bits): scale 22 0.190 /
  // 0.146 / 0.129 ms, scale 24 0.503 / 0.457 / 0.438, scale 26 2.14 / 1.96 / 1.86, scale 28 (52
  // windows at 14 bits) 12.99 / 11.82 / 10.86.
  char const* rb_env  = std::getenv("CUGRAPH_AMD_XS_ROWBITS");
  int row_bits = xs_row_bits_env(xs_row_bits_for(g.minor_space(), slice_shift));
  // entries a lane keeps in flight per pass: 8 in xs_kernel (4: 1.94 ms, 16: 1.77 ms against 1.75 ms at
  // scale 26); xs2_kernel on an unweighted graph 16 (scale 26: 1.573 ms against 1.639 with 8 and 1.583 with
  // 12; 24 and 32 spill registers: 1.75 / 2.43; weighted graphs: 16 would spill)
  int epl = kXsEntriesPerLane;
  // Wide workgroups (unweighted graphs): 8192-row tiles run by two 1024-thread workgroups per CU
  // instead of 16384-row tiles run by one (see xs_kernel's WPE).  CUGRAPH_AMD_XS_WIDE=0/1.
  char const* wide_env = std::getenv("CUGRAPH_AMD_XS_WIDE");
  bool wide = g.weights == nullptr && sizeof(ValT) == 4 && row_bits >= 13 && (wide_env ? std::atoi(wide_env) != 0 : kXsWideDefault);
  if (wide && !rb_env) row_bits = 13;
  wide = wide && row_bits == 13;
  // xs2_kernel (two banks of accumulators in one 1024-thread workgroup, items without barriers):
  // tiles of at most 8192 rows.  CUGRAPH_AMD_XS_ASYNC=0/1.
  // (the default -- xs_kernel with barriers against xs2_kernel: scale 22 0.122 / 0.119 ms, scale 24
  // 0.403 / 0.355, scale 26 1.666 / 1.571, profiles/r3_xs_diag.md)
  // (small plans keep the barrier kernel: their items are too few for two banks to pay -- scale 19 / 20 / 21:
  // 0.044 / 0.051 / 0.074 ms against 0.073 / 0.065 / 0.077 with xs2_kernel; scale 22, 62 M entries: 0.122 / 0.119.
  // A graph with two regions follows the setting alone, so that the window size cugraph_amd_xs_window_slots
  // announced is the one its tiles get.)
  char const* async_env = std::getenv("CUGRAPH_AMD_XS_ASYNC");
  bool const async      = xs_async_on() && (async_env != nullptr || ne >= kXs2MinEntries || g.hot_slots > 0);
  if (async) {
    wide = false;
    if (row_bits > kXs2RowBitsMax) row_bits = kXs2RowBitsMax;
  }
  int const threads = async ? kXs2Threads : xs_threads_for(row_bits, wide);
  if (async) epl = g.weights == nullptr ? 16 : sizeof(ValT) == 4 ? 12 : 8;
  if (char const* epl_env = std::getenv("CUGRAPH_AMD_XS_EPL")) epl = std::atoi(epl_env);
  DevBuf flags(static_cast<size_t>(n_rows) * 4), pos(static_cast<size_t>(n_rows + 1) * 8);
  DevBuf tile_of_row(static_cast<size_t>(n_rows) * 4);
  hipLaunchKernelGGL((xs_break_flags_kernel<OffT>), dim3(grid_for(n_rows)), dim3(kThreads), 0, stream,
                     offsets, row_first, n_rows, e0, tile_edges, int64_t(1) << row_bits,
                     flags.as<unsigned int>());
  exclusive_scan_u32_to_i64(stream, flags.as<unsigned int>(), n_rows, pos.as<int64_t>());
  int64_t n_tiles64 = 0;
  CGA_HIP(hipMemcpyAsync(&n_tiles64, pos.as<int64_t>() + n_rows, 8, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  int n_tiles     = static_cast<int>(n_tiles64);
  mark("tile cuts");
  // windows of 2^(32 - row_bits) slice-local source positions: a packed entry carries only those bits
  uint32_t max_local = xs_slice_local(static_cast<uint32_t>(g.minor_space() - 1), slice_shift) |
                       ((1u << slice_shift) - 1u);
  int n_win          = static_cast<int>(max_local >> (32 - row_bits)) + 1;
  int64_t n_units    = static_cast<int64_t>(n_tiles) * kXsSlices * n_win;
  if (n_units >= 0x7fffffffLL) return;
  plan.xs_tile_row.alloc(static_cast<size_t>(n_tiles + 1) * 4);
  hipLaunchKernelGGL(xs_tiles_kernel, dim3(grid_for(n_rows + 1)), dim3(kThreads), 0, stream,
                     flags.as<unsigned int>(), pos.as<int64_t>(), n_rows,
                     tile_of_row.as<int32_t>(), plan.xs_tile_row.as<int32_t>());
  flags.release();
  pos.release();

  // Unweighted graphs: the packed entry itself is the sort key (inside a (tile, slice, window)
  // unit it ascends with the source; 3 radix passes over 4-byte keys).  Weighted graphs sort
  // (source, input position) pairs and fetch row and weight through the permutation.
  bool const packed_keys = g.weights == nullptr && std::getenv("CUGRAPH_AMD_XS_SORT_PAIRS") == nullptr;
  // Sort columns: a unit is cut into 2^sub_bits sub-units by the top bits of the source position,
  // so the grouping pass does the first radix step (one pass less inside the columns) and the
  // longest column -- the hot window of a hub tile, millions of entries for one workgroup -- is
  // 2^sub_bits times shorter.
  char const* sb_env = std::getenv("CUGRAPH_AMD_XS_SORT_SUBBITS");
  int sub_bits       = packed_keys ? (sb_env ? std::atoi(sb_env) : 4) : 0;
  if (sub_bits < 0 || sub_bits > 8) sub_bits = 0;
  while (sub_bits > 0 && (n_units << sub_bits) >= 0x7fffffffLL) --sub_bits;
  DevBuf unit(static_cast<size_t>(ne) * 4), row_of_edge(static_cast<size_t>(ne) * 4);
  int64_t const label_chunks = (ne + kXsLabelChunk - 1) / kXsLabelChunk;
  hipLaunchKernelGGL((xs_label_kernel<OffT>), dim3(static_cast<unsigned>(std::min<int64_t>(label_chunks, 1 << 20))),
                     dim3(kThreads), 0, stream,
                     offsets, g.indices, row_first, n_rows, e0, ne, tile_of_row.as<int32_t>(), slice_shift,
                     n_win, row_bits, unit.as<int32_t>(), row_of_edge.as<int32_t>(),
                     plan.xs_tile_row.as<int32_t>(), packed_keys ? row_of_edge.as<uint32_t>() : nullptr, sub_bits);
  CGA_HIP(hipGetLastError());
  mark("labels");
  plan.xs_unit_off.alloc(static_cast<size_t>(n_units + 1) * sizeof(int64_t));
  if (packed_keys) {
    CompressedOut c = compress_coo_keys(stream, n_units << sub_bits, ne, unit.as<int32_t>(),
                                        row_of_edge.as<uint32_t>(), row_bits, 32 - sub_bits);
    unit.release();
    row_of_edge.release();
    plan.xs_ent = std::move(c.indices);  // ne + 4 entries, the last 4 zero
    if (c.off64)
      hipLaunchKernelGGL((xs_unit_off_kernel<int64_t>), dim3(grid_for(n_units + 1)), dim3(kThreads), 0, stream,
                         c.offsets.as<int64_t>(), n_units, sub_bits, plan.xs_unit_off.as<int64_t>());
    else
      hipLaunchKernelGGL((xs_unit_off_kernel<int32_t>), dim3(grid_for(n_units + 1)), dim3(kThreads), 0, stream,
                         c.offsets.as<int32_t>(), n_units, sub_bits, plan.xs_unit_off.as<int64_t>());
    CGA_HIP(hipGetLastError());
    CGA_HIP(hipStreamSynchronize(stream));
  } else {
    // (entries of one source inside a unit go to different rows: their order does not matter)
    CompressedOut c = compress_coo_perm(stream, n_units, g.minor_space(), ne, unit.as<int32_t>(),
                                        g.indices + e0, false);
    unit.release();
    plan.xs_ent.alloc(static_cast<size_t>(ne + 4) * sizeof(uint32_t));
    CGA_HIP(hipMemsetAsync(plan.xs_ent.p, 0, plan.xs_ent.bytes, stream));
    if (g.weights) {
      plan.xs_w.alloc(static_cast<size_t>(ne + 4) * sizeof(ValT));
      CGA_HIP(hipMemsetAsync(plan.xs_w.p, 0, plan.xs_w.bytes, stream));
    }
    auto launch = [&](auto tag) {
      using UOffT = decltype(tag);
      hipLaunchKernelGGL((xs_payload_kernel<OffT, UOffT, ValT>), dim3(grid_for(ne)), dim3(kThreads), 0,
                         stream, tile_of_row.as<int32_t>(), e0, ne, c.perm.as<uint32_t>(),
                         row_of_edge.as<int32_t>(), plan.xs_tile_row.as<int32_t>(),
                         static_cast<ValT const*>(g.weights), c.offsets.as<UOffT>(), n_units,
                         c.indices.as<int32_t>(), slice_shift, row_bits, plan.xs_ent.as<uint32_t>(),
                         plan.xs_w.as<ValT>(), plan.xs_unit_off.as<int64_t>());
    };
    if (c.off64)
      launch(int64_t{});
    else
      launch(int32_t{});
    CGA_HIP(hipGetLastError());
    CGA_HIP(hipStreamSynchronize(stream));
  }
  mark("group + sort");
  if (std::getenv("CUGRAPH_AMD_XS_VERIFY")) {
    DevBuf out(24);
    CGA_HIP(hipMemsetAsync(out.p, 0, 24, stream));
    hipLaunchKernelGGL(xs_verify_kernel, dim3(4096), dim3(kThreads), 0, stream,
                       plan.xs_ent.as<uint32_t>(), plan.xs_unit_off.as<int64_t>(), n_units, row_bits,
                       out.as<unsigned long long>());
    unsigned long long h[3];
    CGA_HIP(hipMemcpyAsync(h, out.p, 24, hipMemcpyDeviceToHost, stream));
    CGA_HIP(hipStreamSynchronize(stream));
    std::fprintf(stderr, "xs verify: tiles %d n_win %d units %lld non-empty %llu entries %lld descending pairs %llu equal pairs %llu\n",
                 n_tiles, n_win, (long long)n_units, h[2], (long long)ne, h[0], h[1]);
  }
  // ---- work items: a big tile is cut into pieces (runs of passes of each of its 8 units) so that
  // no item is much longer than the rest; every (slice, piece) writes partials of its own
  char const* pc_env     = std::getenv("CUGRAPH_AMD_XS_PIECE");
  // a piece is 64 passes of its workgroup (128 of a 256-thread one); small graphs: down to 16, so
  // that there are a few thousand pieces
  int64_t const pass_entries = int64_t(threads) * epl;
  int64_t piece_entries      = row_bits == kXsRowBits ? kXsPieceEntries : 64 * pass_entries;
  while (piece_entries > 16 * pass_entries && piece_entries * 1024 > ne) piece_entries >>= 1;
  if (pc_env) piece_entries = std::atoll(pc_env);
  // Regions (partitioned graphs, Graph::hot_slots): the source slots below hot_slots -- whole
  // windows -- are region 0, the rest region 1.  Entries are sorted by slot, so region 0 of a unit
  // is a prefix of it that ends on a window boundary: two item lists and record tables over the
  // same entry stream, so that xs_kernel can run on region 0 while region 1 is still being
  // all-gathered.  Without regions everything is region 0.
  int w0 = 0;
  {
    int64_t const win_slots = static_cast<int64_t>(kXsSlices) << (32 - row_bits);
    if (g.hot_slots > 0 && g.hot_slots % win_slots == 0 && g.hot_slots / win_slots < n_win)
      w0 = static_cast<int>(g.hot_slots / win_slots);
  }
  std::vector<int64_t> h_uoff(static_cast<size_t>(n_units + 1));
  std::vector<int32_t> h_row(n_tiles + 1);
  CGA_HIP(hipMemcpyAsync(h_uoff.data(), plan.xs_unit_off.p, h_uoff.size() * 8, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipMemcpyAsync(h_row.data(), plan.xs_tile_row.p, static_cast<size_t>(n_tiles + 1) * 4,
                         hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  mark("unit offsets to host");
  // Source PARTS (single GPU; big and medium tiles): a unit is cut at window boundaries into
  // parts of part_windows windows -- about 3 MiB of the gathered vector per XCD slice, what an
  // XCD's L2 holds beside the streams -- and the queue is ordered part-major: first every such
  // tile's part 0, then every part 1, ...  All workgroups of an XCD then gather from the same few
  // MiB at a time, so the colder parts, which one sweep of a unit touches line by line (one L2 miss
  // per line and unit), stay in L2 across the tiles of the phase.  No barrier: the order of the
  // queue is all there is to it, and the kernel does not know (an item is a window range of a unit
  // either way).  The price is one more set of partials and one more item boundary per part, so
  // only tiles with enough entries per part take it; the low-degree tiles follow as whole units.
  char const* pw_env = std::getenv("CUGRAPH_AMD_XS_PART_WINDOWS");
  char const* pm_env = std::getenv("CUGRAPH_AMD_XS_PHASE_MIN");
  // (3 windows of a 16384-row tile = 3 << 18 slice-local positions; 8192-row tiles: 2 windows of twice that)
  int const part_windows  = pw_env ? std::atoi(pw_env) : row_bits == kXsRowBitsMax ? 3 : 2;
  int64_t const phase_min = pm_env ? std::atoll(pm_env) : (int64_t(1) << 18);  // entries per unit
  // (only for 16384-row tiles, the layout the parts were measured and run at full size with: graphs
  // whose source space needs 4096- or 8192-row tiles keep whole-unit items)
  int const n_parts = (w0 == 0 && part_windows > 0 && n_win > part_windows && (row_bits == kXsRowBitsMax || wide || (async && row_bits == 13)))
                        ? (n_win + part_windows - 1) / part_windows : 1;
  std::vector<XsItem> h_items[2];
  std::vector<std::vector<XsItem>> part_items(n_parts);  // region 0 of phased tiles, by part
  std::vector<int64_t> h_fin;
  std::vector<int64_t> h_pbase(n_tiles + 1);
  int64_t plen = 0;
  auto pieces_of = [&](int64_t entries_all_slices) {
    int64_t const per_unit = entries_all_slices / kXsSlices;
    return static_cast<int32_t>(std::min<int64_t>(15, std::max<int64_t>(1, (per_unit + piece_entries / 2) / piece_entries)));
  };
  // timing aid (results are then wrong): CUGRAPH_AMD_XS_ONLY_TILES=lo:hi keeps the work items of
  // tiles lo <= t < hi only; the tile list (rows, entries) goes to stderr
  int only_lo = 0, only_hi = n_tiles;
  if (char const* ot = std::getenv("CUGRAPH_AMD_XS_ONLY_TILES")) {
    std::sscanf(ot, "%d:%d", &only_lo, &only_hi);
    for (int t = 0; t < n_tiles; ++t) {
      int64_t const u0 = static_cast<int64_t>(t) * kXsSlices * n_win, u1 = static_cast<int64_t>(t + 1) * kXsSlices * n_win;
      if (t < 12 || t % 50 == 0 || t == n_tiles - 1 || std::getenv("CUGRAPH_AMD_XS_TILE_LIST"))
        std::fprintf(stderr, "xs tile %d rows %d entries %lld\n", t, h_row[t + 1] - h_row[t], (long long)(h_uoff[u1] - h_uoff[u0]));
    }
  }
  for (int t = 0; t < n_tiles; ++t) {
    // entries of the tile per window, over its eight units
    auto count = [&](int lo, int hi) {
      int64_t c = 0;
      for (int k = 0; k < kXsSlices; ++k) {
        int64_t const u0 = (static_cast<int64_t>(t) * kXsSlices + k) * n_win;
        c += h_uoff[u0 + hi] - h_uoff[u0 + lo];
      }
      return c;
    };
    int32_t sets = 0;
    if (w0 > 0) {  // two regions of a partitioned graph's source space: item lists of their own
      for (int reg = 0; reg < 2; ++reg) {
        int const lo = reg == 0 ? 0 : w0, hi = reg == 0 ? w0 : n_win;
        int32_t const n_pc = pieces_of(count(lo, hi));
        for (int32_t j = 0; j < n_pc; ++j) h_items[reg].push_back(XsItem{t, j, n_pc, sets + j, lo, hi});
        sets += n_pc;
      }
    } else if (n_parts > 1 && count(0, n_win) / kXsSlices >= phase_min) {
      for (int part = 0; part < n_parts; ++part) {
        int const lo = part * part_windows, hi = std::min(n_win, lo + part_windows);
        int64_t const c = count(lo, hi);
        if (c == 0) continue;  // nothing to add, no partials
        int32_t const n_pc = pieces_of(c);
        for (int32_t j = 0; j < n_pc; ++j) part_items[part].push_back(XsItem{t, j, n_pc, sets + j, lo, hi});
        sets += n_pc;
      }
      if (sets == 0) {  // (cannot happen: the tile has entries) keep the row sums defined
        part_items[0].push_back(XsItem{t, 0, 1, 0, 0, n_win});
        sets = 1;
      }
    } else {
      int32_t const n_pc = pieces_of(count(0, n_win));
      for (int32_t j = 0; j < n_pc; ++j) h_items[0].push_back(XsItem{t, j, n_pc, sets + j, 0, n_win});
      sets += n_pc;
    }
    int64_t const rows_t = h_row[t + 1] - h_row[t];
    // a set of partials is padded to a multiple of 4 rows: xs_kernel writes it as 16-byte vectors
    int64_t const rows_pad = (rows_t + 3) & ~int64_t(3);
    h_pbase[t] = plen;
    plen += static_cast<int64_t>(kXsSlices) * sets * rows_pad;
    for (int64_t r = 0; r < rows_t; r += kXsFinishRows) {
      h_fin.push_back(h_row[t] + r);
      h_fin.push_back(std::min<int64_t>(kXsFinishRows, rows_t - r) | static_cast<int64_t>(kXsSlices * sets) << 32);
      h_fin.push_back(h_pbase[t] + r);
      h_fin.push_back(rows_pad);
    }
  }
  {  // the queue: the phased tiles part by part, then the whole units of the other tiles
    std::vector<XsItem> whole;
    whole.swap(h_items[0]);
    for (auto const& pi : part_items) h_items[0].insert(h_items[0].end(), pi.begin(), pi.end());
    h_items[0].insert(h_items[0].end(), whole.begin(), whole.end());
    if (only_lo > 0 || only_hi < n_tiles) {
      std::vector<XsItem> kept;
      for (auto const& it : h_items[0])
        if (it.tile >= only_lo && it.tile < only_hi) kept.push_back(it);
      h_items[0].swap(kept);
      std::fprintf(stderr, "xs only tiles [%d, %d) of %d: %zu items\n", only_lo, only_hi, n_tiles, h_items[0].size());
    }
  }
  h_pbase[n_tiles] = plen;
  mark("work items (host)");
  if (n_tiles >= (1 << 24) || h_items[0].size() + h_items[1].size() >= 0x7fffffffu / kXsSlices) return;
  plan.xs_pbase.alloc(static_cast<size_t>(n_tiles + 1) * 8);
  CGA_HIP(hipMemcpyAsync(plan.xs_pbase.p, h_pbase.data(), plan.xs_pbase.bytes, hipMemcpyHostToDevice, stream));
  for (int reg = 0; reg < 2; ++reg) {
    int n_items = static_cast<int>(h_items[reg].size());
    plan.xs_n_items[reg] = n_items;
    if (n_items == 0) continue;
    DevBuf items(static_cast<size_t>(n_items) * sizeof(XsItem));
    plan.xs_rec[reg].alloc(static_cast<size_t>(kXsSlices) * n_items * kXsRecWords * 8);
    CGA_HIP(hipMemcpyAsync(items.p, h_items[reg].data(), items.bytes, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(xs_records_kernel, dim3(grid_for(static_cast<int64_t>(kXsSlices) * n_items)),
                       dim3(kThreads), 0, stream, items.as<XsItem>(), n_items,
                       plan.xs_tile_row.as<int32_t>(), plan.xs_unit_off.as<int64_t>(),
                       plan.xs_pbase.as<int64_t>(), n_win, plan.xs_rec[reg].as<int64_t>());
    CGA_HIP(hipGetLastError());
    CGA_HIP(hipStreamSynchronize(stream));
  }
  plan.xs_fin_items.alloc(h_fin.size() * 8);
  CGA_HIP(hipMemcpy(plan.xs_fin_items.p, h_fin.data(), h_fin.size() * 8, hipMemcpyHostToDevice));
  plan.xs_n_fin_items = static_cast<int>(h_fin.size() / 4);
  mark("records");
  plan.xs_regions     = w0 > 0 ? 2 : 1;
  plan.xs_partial_len = plen;
  plan.xs_on        = true;
  plan.xs_row_first = row_first;
  plan.xs_n_rows    = n_rows;
  plan.xs_n_edges   = ne;
  plan.xs_n_tiles   = n_tiles;
  plan.xs_n_win     = n_win;
  plan.xs_slice_shift = slice_shift;
  plan.xs_row_bits  = row_bits;
  plan.xs_epl       = epl;
  plan.xs_threads   = threads;
  plan.xs_async     = async;
}

template <typename OffT>
__global__ void dead_rows_kernel(OffT const* __restrict__ offsets, int32_t const* __restrict__ x_pos,
                                 int64_t n, unsigned long long* __restrict__ out /* {count, first} */)
{
  unsigned long long cnt = 0, first = ~0ull;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n;
       v += (int64_t)gridDim.x * blockDim.x)
    if (offsets[v + 1] == offsets[v] && x_pos[v] < 0) {
      ++cnt;
      if (static_cast<unsigned long long>(v) < first) first = static_cast<unsigned long long>(v);
    }
  if (cnt) {
    atomicAdd(&out[0], cnt);
    atomicMin(&out[1], first);
  }
}

__global__ void mark_sources_kernel(int32_t const* __restrict__ indices, int64_t n_edges,
                                    unsigned int* __restrict__ flags)
{
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_edges;
       e += (int64_t)gridDim.x * blockDim.x)
    flags[indices[e]] = 1u;
}

__global__ void slot_of_vertex_kernel(unsigned int const* __restrict__ flags,
                                      int64_t const* __restrict__ pos, int64_t n,
                                      int32_t* __restrict__ x_pos)
{
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n;
       v += (int64_t)gridDim.x * blockDim.x)
    x_pos[v] = flags[v] ? static_cast<int32_t>(pos[v]) : -1;
}

__global__ void slots_of_indices_kernel(int32_t const* __restrict__ indices, int64_t n_edges,
                                        int32_t const* __restrict__ x_pos, int32_t* __restrict__ out)
{
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_edges;
       e += (int64_t)gridDim.x * blockDim.x)
    out[e] = x_pos[indices[e]];
}

// Single GPU, big graph: number the sources compactly (slots in vertex order, so a degree-sorted
// graph keeps its hot sources in front).  An R-MAT graph at scale 26 gathers from 27 M of its 67 M
// vertices: the cold lines of the gathered vector hold 2.5 times as many useful values, and the
// smaller source space lets the XS tiles take more rows (build_xs).  The out-weight sums are
// taken here, while the indices still are vertex ids.
// unweighted graphs: the out-weight sum of a vertex is how often it occurs as a source, and "occurs
// at all" is the source mark -- one pass of 4-byte atomics instead of fp64 atomics plus a marking pass
__global__ void counts_to_sums_kernel(unsigned int* __restrict__ counts, int64_t n, double* __restrict__ sums)
{
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    unsigned int const c = counts[v];
    sums[v]              = static_cast<double>(c);
    counts[v]            = c ? 1u : 0u;
  }
}

// dead rows (no in-edges, no slot: no edge touches them): only if they are one run at the very end
// (a degree-sorted graph built here has them there: renumber.hip, degree_order_sources_first; the
// multi-GPU driver orders each rank's rows the same way).  g: a graph with x_pos.
void detect_dead_rows(hipStream_t stream, Graph const& g, PullPlan& plan)
{
  char const* dead_env = std::getenv("CUGRAPH_AMD_DEAD_ROWS");
  int64_t const n      = g.num_vertices;
  if ((dead_env && dead_env[0] == '0') || !g.x_pos || n == 0) return;
  DevBuf out(16);
  unsigned long long h_out[2] = {0ull, ~0ull};
  CGA_HIP(hipMemcpyAsync(out.p, h_out, 16, hipMemcpyHostToDevice, stream));
  if (g.off64)
    hipLaunchKernelGGL((dead_rows_kernel<int64_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                       static_cast<int64_t const*>(g.offsets), g.x_pos, n, out.as<unsigned long long>());
  else
    hipLaunchKernelGGL((dead_rows_kernel<int32_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                       static_cast<int32_t const*>(g.offsets), g.x_pos, n, out.as<unsigned long long>());
  CGA_HIP(hipMemcpyAsync(h_out, out.p, 16, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  if (h_out[0] > 0 && h_out[1] + h_out[0] == static_cast<unsigned long long>(n))
    plan.n_dead = static_cast<int64_t>(h_out[0]);
}

void build_compact(hipStream_t stream, Graph const& cg, PullPlan& plan)
{
  char const* env = std::getenv("CUGRAPH_AMD_COMPACT");
  bool const forced = env && env[0] == '1';
  if (env && env[0] == '0') return;
  if (cg.x_pos || cg.num_global_vertices != cg.num_vertices) return;  // the caller's layout stands
  int64_t const n = cg.num_vertices, ne = cg.num_edges;
  if (n == 0 || ne == 0 || (!forced && ne < (int64_t(1) << 22))) return;
  plan.out_w_sums.alloc(static_cast<size_t>(n) * sizeof(double));
  plan.have_out_w = true;
  DevBuf flags(static_cast<size_t>(n) * 4), pos(static_cast<size_t>(n + 1) * 8);
  CGA_HIP(hipMemsetAsync(flags.p, 0, flags.bytes, stream));
  if (cg.weights) {
    compute_source_sums(stream, cg, plan.out_w_sums.as<double>());
    hipLaunchKernelGGL(mark_sources_kernel, dim3(grid_for(ne)), dim3(kThreads), 0, stream, cg.indices, ne,
                       flags.as<unsigned int>());
  } else {
    histogram_u32(stream, cg.indices, ne, flags.as<unsigned int>(), cg.degree_sorted);
    hipLaunchKernelGGL(counts_to_sums_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream,
                       flags.as<unsigned int>(), n, plan.out_w_sums.as<double>());
  }
  exclusive_scan_u32_to_i64(stream, flags.as<unsigned int>(), n, pos.as<int64_t>());
  int64_t n_slots = 0;
  CGA_HIP(hipMemcpyAsync(&n_slots, pos.as<int64_t>() + n, 8, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  if (!forced && n_slots * 10 > n * 9) return;  // (nearly) every vertex is a source: nothing to gain
  DevBuf x_pos(static_cast<size_t>(n) * 4), slots(static_cast<size_t>(ne + 8) * 4);
  CGA_HIP(hipMemsetAsync(static_cast<char*>(slots.p) + static_cast<size_t>(ne) * 4, 0, 32, stream));
  hipLaunchKernelGGL(slot_of_vertex_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream,
                     flags.as<unsigned int>(), pos.as<int64_t>(), n, x_pos.as<int32_t>());
  hipLaunchKernelGGL(slots_of_indices_kernel, dim3(grid_for(ne)), dim3(kThreads), 0, stream, cg.indices,
                     ne, x_pos.as<int32_t>(), slots.as<int32_t>());
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));
  auto c                 = std::make_unique<Graph>();
  c->vertex_type         = cg.vertex_type;
  c->edge_type           = cg.edge_type;
  c->weight_type         = cg.weight_type;
  c->store_transposed    = cg.store_transposed;
  c->is_symmetric        = cg.is_symmetric;
  c->is_multigraph       = cg.is_multigraph;
  c->renumbered          = cg.renumbered;
  c->degree_sorted       = cg.degree_sorted;
  c->num_vertices        = n;
  c->num_global_vertices = n;
  c->vertex_first        = 0;
  c->num_edges           = ne;
  c->off64               = cg.off64;
  c->offsets             = cg.offsets;   // viewed, released by ~PullPlan without freeing
  c->weights             = cg.weights;   // viewed
  c->indices             = static_cast<int32_t*>(slots.detach());
  c->x_pos               = static_cast<int32_t*>(x_pos.detach());
  c->num_source_slots    = std::max<int64_t>(n_slots, 1);
  plan.compact           = c.release();
  detect_dead_rows(stream, *plan.compact, plan);
}

PullPlan& get_plan(Handle const& h, Graph& g)
{
  hipStream_t stream = h.stream;
  if (!g.plan) {
    auto* p = new PullPlan();
    try {
      // graphs that are not already in descending in-degree order get a degree-sorted copy to
      // run on (single GPU; CUGRAPH_AMD_NO_RELABEL=1 keeps the index-list buckets instead)
      char const* no_relabel = std::getenv("CUGRAPH_AMD_NO_RELABEL");
      bool relabel = !(no_relabel && no_relabel[0] == '1') && !g.degree_sorted &&
                     g.num_global_vertices == g.num_vertices && g.num_vertices > 0;
      if (relabel) p->internal = degree_sorted_copy(h, g);
      Graph const& vg = p->internal ? *p->internal : g;
      if (vg.off64)
        build_plan_lists<int64_t>(stream, vg, *p);
      else
        build_plan_lists<int32_t>(stream, vg, *p);
      char const* xs_env = std::getenv("CUGRAPH_AMD_XS");
      bool want_xs = !(xs_env && xs_env[0] == '0') && p->contiguous;
      if (want_xs) build_compact(stream, vg, *p);
      if (p->contiguous && vg.x_pos) detect_dead_rows(stream, vg, *p);  // the caller's slots (multi-GPU)
      Graph const& cg = p->compact ? *p->compact : vg;
      if (want_xs) {
        bool f64 = cg.weight_type == FLOAT64;
        if (cg.off64) {
          if (f64) build_xs<int64_t, double>(stream, cg, *p);
          else build_xs<int64_t, float>(stream, cg, *p);
        } else {
          if (f64) build_xs<int32_t, double>(stream, cg, *p);
          else build_xs<int32_t, float>(stream, cg, *p);
        }
      }
      if (p->compact && p->xs_on && p->xs_row_first == 0 && std::getenv("CUGRAPH_AMD_KEEP_SLOT_COPY") == nullptr) {
        // keep the slot numbers of the edges below the XS rows only (see PullPlan::compact_tail)
        Graph& c = *p->compact;
        // (XS rows are the first rows, their edges the first edges.)  The copy starts at a multiple
        // of 4 edges: the biased base pointer below stays 16-byte aligned, and a reader that aligns
        // a column start DOWN to 4 entries (column_partial) still lands inside the buffer.
        int64_t const tail_first = p->xs_n_edges & ~int64_t(3);
        int64_t const n_tail     = c.num_edges - tail_first;
        DevBuf tail(static_cast<size_t>(n_tail + 8) * sizeof(int32_t));
        CGA_HIP(hipMemsetAsync(tail.p, 0, tail.bytes, stream));
        if (n_tail > 0)
          CGA_HIP(hipMemcpyAsync(tail.p, c.indices + tail_first, static_cast<size_t>(n_tail) * sizeof(int32_t),
                                 hipMemcpyDeviceToDevice, stream));
        CGA_HIP(hipStreamSynchronize(stream));
        CGA_HIP(hipFree(c.indices));
        p->compact_tail = std::move(tail);
        c.indices       = p->compact_tail.as<int32_t>() - tail_first;
      }
    } catch (...) {
      delete p;
      throw;
    }
    g.plan = p;
  }
  return *g.plan;
}

}  // namespace

// per-source weight sums of this graph's edges, double[num_global_vertices]
void compute_source_sums(hipStream_t stream, Graph const& g, double* sums)
{
  CGA_EXPECTS(g.x_pos == nullptr, CUGRAPH_INVALID_INPUT,
              "source sums of a compact-source graph are indexed by slot: compute them from the edge list");
  CGA_HIP(hipMemsetAsync(sums, 0, static_cast<size_t>(g.num_global_vertices) * sizeof(double), stream));
  bool const window = g.degree_sorted && g.num_edges >= (int64_t{1} << 20) &&
                      g.num_global_vertices >= kSumsWindow && std::getenv("CUGRAPH_AMD_BUILD_NO_LDS") == nullptr;
  if (window) {
    size_t const lds = static_cast<size_t>(kSumsWindow) * sizeof(double);
    auto launch = [&](auto kern, auto const* w) {
      CGA_HIP(hipFuncSetAttribute(reinterpret_cast<void const*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds)));
      hipLaunchKernelGGL(kern, dim3(1024), dim3(1024), lds, stream, g.indices, w, g.num_edges, sums);
    };
    if (g.weights && g.weight_type == FLOAT64)
      launch(source_sums_window_kernel<double>, static_cast<double const*>(g.weights));
    else
      launch(source_sums_window_kernel<float>, static_cast<float const*>(g.weights));
    CGA_HIP(hipGetLastError());
  } else if (g.num_edges > 0) {
    if (g.weights && g.weight_type == FLOAT64)
      hipLaunchKernelGGL((source_sums_kernel<double>), dim3(grid_for(g.num_edges)), dim3(kThreads),
                         0, stream, g.indices, static_cast<double const*>(g.weights), g.num_edges,
                         sums);
    else
      hipLaunchKernelGGL((source_sums_kernel<float>), dim3(grid_for(g.num_edges)), dim3(kThreads),
                         0, stream, g.indices, static_cast<float const*>(g.weights), g.num_edges,
                         sums);
    CGA_HIP(hipGetLastError());
  }
}

// per-source weight sums (1 per edge when weights == nullptr) of a raw edge list
void edgelist_source_sums(hipStream_t stream, ArrayView const& src, ArrayView const* weights,
                          int64_t num_vertices, double* sums)
{
  CGA_HIP(hipMemsetAsync(sums, 0, static_cast<size_t>(num_vertices > 0 ? num_vertices : 1) * sizeof(double), stream));
  int64_t ne = static_cast<int64_t>(src.size);
  if (ne == 0) return;
  DevBuf s32(static_cast<size_t>(ne) * 4);
  int64_t smin = 0;
  int64_t smax = narrow_ids(stream, src.data, src.type, src.size, s32.as<int32_t>(), &smin);
  CGA_EXPECTS(smin >= 0 && smax < num_vertices, CUGRAPH_INVALID_INPUT,
              "source ids outside [0, num_vertices)");
  if (weights && weights->type == FLOAT64)
    hipLaunchKernelGGL((source_sums_kernel<double>), dim3(grid_for(ne)), dim3(kThreads), 0, stream,
                       s32.as<int32_t>(), static_cast<double const*>(weights->data), ne, sums);
  else
    hipLaunchKernelGGL((source_sums_kernel<float>), dim3(grid_for(ne)), dim3(kThreads), 0, stream,
                       s32.as<int32_t>(), weights ? static_cast<float const*>(weights->data) : nullptr,
                       ne, sums);
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));
}

// ------------------------------------------------------------------ session
struct Session {
  Handle const* handle{nullptr};
  Graph* graph{nullptr};        // the caller's graph
  Graph const* cg{nullptr};     // the graph the kernels run on (graph, or its degree-sorted copy)
  int32_t const* order{nullptr};  // cg row -> graph row when cg is a copy
  bool f64{false};
  double alpha{0.85}, epsilon{0.0};
  int64_t n_local{0}, n_global{0};
  int64_t x_len{0};  // length of the gathered vector: n_global, padded to comm_size equal chunks
  DevBuf pr[2], x[2], out_w, pers_dense, pers_v, pers_p, partials, scal, ctrl;
  int64_t n_pers{0};       // entries of the compact list (sparse mode)
  bool personalized{false};
  bool pers_is_dense{false};
  size_t enqueued{0};
  int n_spmv_blocks{0}, n_pers_blocks{0}, n_hub_blocks{0};
  bool xs_on{false};
  int64_t n_dead{0};  // trailing rows no edge touches, skipped after the first iteration
  int xs_block_begin{0}, n_xs_blocks{0}, n_xs_finish_blocks{0};
  void* xs_kernel_fn{nullptr};  // the xs_kernel instantiation of this graph's tiles (xs_pick)
  int xs_threads{0}, xs_lds_bytes{0};
  DevBuf xs_partial, xs_counters;
  int64_t n_hub_items{0};
  HubItem const* hub_items{nullptr};
  int64_t const* hub_row_item_off{nullptr};
  DevBuf hub_item_partial;
  Bucket buckets[kNumBuckets]{};
  std::vector<hipEvent_t> events;
  // spmv_kernel beside xs_kernel on a stream of its own (see enqueue_iteration)
  hipStream_t side_stream{nullptr};
  hipEvent_t ev_fork{nullptr}, ev_join{nullptr};
  ~Session()
  {
    for (auto e : events) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side_stream) (void)hipStreamDestroy(side_stream);
  }
};

namespace {

template <typename ValT>
void dense_from_pairs(hipStream_t stream, ArrayView const* ids, ArrayView const* vals,
                      IdMap const& map, ValT* dense, bool accumulate, char const* what)
{
  CGA_EXPECTS(ids && vals, CUGRAPH_INVALID_INPUT,
              std::string(what) + ": vertices and values must both be given");
  CGA_EXPECTS(ids->size == vals->size, CUGRAPH_INVALID_INPUT,
              std::string(what) + ": vertices and values must have the same size");
  int64_t n = static_cast<int64_t>(ids->size);
  if (n == 0) return;
  DevBuf bad(sizeof(unsigned int));
  CGA_HIP(hipMemsetAsync(bad.p, 0, sizeof(unsigned int), stream));
  auto launch = [&](auto id_tag) {
    using IdT = decltype(id_tag);
    if (accumulate)
      hipLaunchKernelGGL((scatter_pairs_kernel<IdT, ValT, true>), dim3(grid_for(n)), dim3(kThreads),
                         0, stream, static_cast<IdT const*>(ids->data),
                         static_cast<ValT const*>(vals->data), n, map, dense,
                         bad.as<unsigned int>());
    else
      hipLaunchKernelGGL((scatter_pairs_kernel<IdT, ValT, false>), dim3(grid_for(n)),
                         dim3(kThreads), 0, stream, static_cast<IdT const*>(ids->data),
                         static_cast<ValT const*>(vals->data), n, map, dense,
                         bad.as<unsigned int>());
  };
  if (ids->type == INT32)
    launch(int32_t{});
  else
    launch(int64_t{});
  CGA_HIP(hipGetLastError());
  unsigned int h_bad = 0;
  CGA_HIP(hipMemcpyAsync(&h_bad, bad.p, sizeof(h_bad), hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  CGA_EXPECTS(h_bad == 0, CUGRAPH_INVALID_INPUT,
              std::string(what) + ": vertex ids outside this graph's vertex range");
}

// launch geometry of a session: bucket block ranges, XS queues and partials, hub work items
template <typename ValT>
int session_geometry(Session& s, PullPlan& plan)
{
  size_t vb  = sizeof(ValT);
  int64_t nl = s.n_local;
  // ---- launch geometry: all buckets in one grid, hubs first
  int begin = 0;
  // timing aid (make AIDS=1): CUGRAPH_AMD_BUCKET_MASK=<bits> runs only the named buckets
#ifdef CGA_TIMING_AIDS
  char const* mask_env = std::getenv("CUGRAPH_AMD_BUCKET_MASK");
#else
  char const* mask_env = nullptr;
#endif
  int bucket_mask      = mask_env ? std::atoi(mask_env) : 0x1F;
  int const group_width[kNumBuckets] = {kThreads, 64, 16, 1, 1};
  int const block_cap[kNumBuckets]   = {4096, 3072, 3072, 2048, 2048};
  s.xs_on = plan.xs_on && ((bucket_mask >> 5) & 1 || !mask_env);
  for (int b = 0; b < kNumBuckets; ++b) {
    Bucket& bk    = s.buckets[b];
    bk.count      = ((bucket_mask >> b) & 1) ? plan.counts[b] : 0;
    if (plan.xs_on && b <= 2) bk.count = 0;  // rows with >= kXsMinDegree edges: XS tiles
    bk.list       = plan.lists[b].p ? plan.lists[b].as<int32_t>() : nullptr;
    bk.first      = plan.firsts[b];
    int64_t need  = (bk.count * group_width[b] + kThreads - 1) / kThreads;
    if (b == 0) need = bk.count > 0 ? plan.n_hub_items : 0;  // one workgroup per hub work item
    bk.n_blocks   = static_cast<int>(std::min<int64_t>(need, block_cap[b]));
    bk.block_begin = begin;
    begin += bk.n_blocks;
  }
  // the slot-numbered indices of a plan with XS tiles are cut down to the rows below the tiles
  // (PullPlan::compact_tail): no bucket that reads the rows above them may run on it
  CGA_EXPECTS(!plan.compact_tail.p || (s.buckets[0].count == 0 && s.buckets[1].count == 0 && s.buckets[2].count == 0),
              CUGRAPH_UNKNOWN_ERROR, "internal: a degree bucket of XS rows on a plan whose slot copy is trimmed");
  // XS tiles run in their own launch (xs_kernel): 4096 threads per CU whatever the workgroup size
  s.xs_block_begin = 0;
  s.n_xs_blocks    = 0;
  if (s.xs_on) {
    // the kernel instantiation for this plan's tiles, and as many workgroups as are resident at
    // once (LDS- and register-bound: asked of the runtime), capped by the work items there are
    XsKernelChoice<ValT> const ch = xs_pick<ValT>(s.cg->weights != nullptr, plan.xs_row_bits, plan.xs_epl, plan.xs_threads, plan.xs_async);
    CGA_HIP(hipFuncSetAttribute(reinterpret_cast<void const*>(ch.kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, ch.lds));
    int per_cu = 0, n_cu = 0;
    CGA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<void const*>(ch.kern),
                                                         ch.threads, static_cast<size_t>(ch.lds)));
    CGA_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, s.handle->device));
    CGA_EXPECTS(per_cu > 0 && n_cu > 0, CUGRAPH_UNKNOWN_ERROR, "internal: xs_kernel does not fit a compute unit");
    s.xs_kernel_fn = reinterpret_cast<void*>(ch.kern);
    s.xs_threads   = ch.threads;
    s.xs_lds_bytes = ch.lds;
    char const* xb = std::getenv("CUGRAPH_AMD_XS_BLOCKS");
    int want       = xb ? std::atoi(xb) : per_cu * n_cu;
    s.n_xs_blocks  = static_cast<int>(std::min<int64_t>(
      static_cast<int64_t>(std::max(plan.xs_n_items[0], plan.xs_n_items[1])) * kXsSlices, want));
  }
  if (s.xs_on) {
    s.xs_partial.alloc(static_cast<size_t>(plan.xs_partial_len) * vb);
    s.xs_counters.alloc(256);
    s.n_xs_finish_blocks = std::min(plan.xs_n_fin_items, 4096);
  }
  s.n_spmv_blocks = std::max(begin, 1);
  s.n_hub_items   = s.buckets[0].count > 0 ? plan.n_hub_items : 0;
  s.n_hub_blocks  = s.n_hub_items > 0 ? grid_for(s.buckets[0].count, 256) : 0;
  s.hub_items     = plan.hub_items.p ? plan.hub_items.as<HubItem>() : nullptr;
  s.hub_row_item_off = plan.hub_row_item_off.p ? plan.hub_row_item_off.as<int64_t>() : nullptr;
  s.hub_item_partial.alloc(static_cast<size_t>(s.n_hub_items > 0 ? s.n_hub_items : 1) * sizeof(double));
  s.n_pers_blocks = (s.personalized && !s.pers_is_dense && s.n_pers > 0)
                      ? grid_for(s.n_pers, 256)
                      : 0;
  int n_prep_blocks = grid_for(nl, 1024);
  int n_partials    = std::max(s.n_spmv_blocks + s.n_hub_blocks + s.n_xs_finish_blocks + s.n_pers_blocks,
                               n_prep_blocks);
  CGA_EXPECTS(n_partials <= kMaxPartials, CUGRAPH_UNKNOWN_ERROR, "internal: too many partials");
  s.partials.alloc(static_cast<size_t>(2 * n_partials) * sizeof(double));

  return n_prep_blocks;
}

template <typename ValT>
void session_setup(Session& s, PagerankArgs const& a)
{
  hipStream_t stream = s.handle->stream;
  PullPlan& plan     = get_plan(*s.handle, *s.graph);
  s.cg               = plan.compact ? plan.compact : plan.internal ? plan.internal : s.graph;
  s.order            = plan.internal ? static_cast<int32_t const*>(plan.internal->number_map) : nullptr;
  Graph const& g     = *s.cg;
  int64_t nl = s.n_local, ng = s.n_global;
  size_t vb = sizeof(ValT);
  IdMap idmap{};
  idmap.lookup0 = s.graph->ext_lookup;
  idmap.min0    = s.graph->ext_min;
  idmap.range0  = s.graph->ext_range;
  idmap.sorted0 = s.graph->ext_sorted;
  idmap.lookup1 = plan.internal ? plan.internal->ext_lookup : nullptr;
  idmap.first   = s.graph->vertex_first;
  idmap.count   = nl;

  // 1D partition: rank r owns destinations [r*chunk, min((r+1)*chunk, V)), chunk = ceil(V / P);
  // the gathered vector is padded to P equal chunks so that one all-gather fills it
  // (with a compact source space the destination ranges may be ANY contiguous cover of the
  // vertices -- the driver balances them by in-edges --: the slots define the gathered vector)
  s.x_len = ng;
  if (!g.x_pos && (g.num_global_vertices != g.num_vertices || s.handle->comm_size > 1)) {
    int64_t P     = s.handle->comm_size;
    int64_t chunk = (ng + P - 1) / P;
    int64_t first = std::min<int64_t>(chunk * s.handle->rank, ng);
    int64_t last  = std::min<int64_t>(first + chunk, ng);
    CGA_EXPECTS(g.vertex_first == first && g.num_vertices == last - first, CUGRAPH_INVALID_INPUT,
                "partitioned graph: this rank must own destinations [rank*ceil(V/P), ...) "
                "(see cugraph_amd.mg.partition_bounds)");
    s.x_len = chunk * P;
  }
  if (g.x_pos) s.x_len = g.num_source_slots;  // compact source space: the caller sized and padded it
  for (int k = 0; k < 2; ++k) {
    s.pr[k].alloc(static_cast<size_t>(nl > 0 ? nl : 1) * vb);
    s.x[k].alloc(static_cast<size_t>(s.x_len > 0 ? s.x_len : 1) * vb);
  }
  s.out_w.alloc(static_cast<size_t>(nl > 0 ? nl : 1) * vb);
  s.scal.alloc(4 * sizeof(double));
  s.ctrl.alloc(sizeof(Ctrl));
  CGA_HIP(hipMemsetAsync(s.ctrl.p, 0, sizeof(Ctrl), stream));
  // x buffers are gathered from at every global id: start them defined
  CGA_HIP(hipMemsetAsync(s.x[0].p, 0, s.x[0].bytes, stream));
  CGA_HIP(hipMemsetAsync(s.x[1].p, 0, s.x[1].bytes, stream));

  // ---- do_expensive_check: no negative edge weight (pagerank_impl.cuh:119-131)
  if (a.do_expensive_check && g.weights && g.num_edges > 0) {
    DevBuf chk(sizeof(double) + sizeof(unsigned int) * 2);
    CGA_HIP(hipMemsetAsync(chk.p, 0, chk.bytes, stream));
    hipLaunchKernelGGL((value_checks_kernel<ValT>), dim3(grid_for(g.num_edges, 4096)), dim3(kThreads), 0, stream,
                       static_cast<ValT const*>(g.weights), g.num_edges, chk.as<double>(),
                       reinterpret_cast<unsigned int*>(chk.as<double>() + 1));
    unsigned int neg = 0;
    CGA_HIP(hipMemcpyAsync(&neg, chk.as<double>() + 1, sizeof(neg), hipMemcpyDeviceToHost, stream));
    CGA_HIP(hipStreamSynchronize(stream));
    CGA_EXPECTS(neg == 0, kAlgorithmRejects,
                "Invalid input argument: input edge weights should have non-negative values.");
  }
  // ---- out-weight sums: given (pagerank.cpp:158-184) or computed once per graph and cached
  if (a.pre_sums) {
    CGA_HIP(hipMemsetAsync(s.out_w.p, 0, s.out_w.bytes, stream));
    dense_from_pairs<ValT>(stream, a.pre_vertices, a.pre_sums, idmap, s.out_w.as<ValT>(), false,
                           "precomputed_vertex_out_weight");
    if (a.do_expensive_check && nl > 0) {
      DevBuf chk(sizeof(double) + sizeof(unsigned int) * 2);
      CGA_HIP(hipMemsetAsync(chk.p, 0, chk.bytes, stream));
      hipLaunchKernelGGL((value_checks_kernel<ValT>), dim3(grid_for(nl, 1024)), dim3(kThreads), 0,
                         stream, s.out_w.as<ValT>(), nl, chk.as<double>(),
                         reinterpret_cast<unsigned int*>(chk.as<double>() + 1));
      unsigned int neg = 0;
      CGA_HIP(hipMemcpyAsync(&neg, chk.as<double>() + 1, sizeof(neg), hipMemcpyDeviceToHost, stream));
      CGA_HIP(hipStreamSynchronize(stream));
      CGA_EXPECTS(neg == 0, kAlgorithmRejects,
                  "Invalid input argument: outgoing edge weight sum values should be non-negative.");
    }
  } else {
    CGA_EXPECTS(g.num_global_vertices == g.num_vertices, CUGRAPH_INVALID_INPUT,
                "a 1D-partitioned graph needs precomputed_vertex_out_weight_sums (all-reduced "
                "over the ranks by the multi-GPU driver)");
    if (!plan.have_out_w) {
      plan.out_w_sums.alloc(static_cast<size_t>(ng > 0 ? ng : 1) * sizeof(double));
      compute_source_sums(stream, g, plan.out_w_sums.as<double>());
      plan.have_out_w = true;
    }
    if (nl > 0)
      hipLaunchKernelGGL((cast_kernel<double, ValT>), dim3(grid_for(nl)), dim3(kThreads), 0, stream,
                         plan.out_w_sums.as<double>(), s.out_w.as<ValT>(), nl);
  }

  // ---- initial ranks: 1/V, or the caller's guess verbatim with 0 for vertices not listed
  //      (pagerank_impl.cuh:430-440; pagerank.cpp:186-212)
  if (a.guess_values) {
    CGA_HIP(hipMemsetAsync(s.pr[0].p, 0, s.pr[0].bytes, stream));
    dense_from_pairs<ValT>(stream, a.guess_vertices, a.guess_values, idmap, s.pr[0].as<ValT>(),
                           false, "initial_guess");
  } else if (nl > 0) {
    hipLaunchKernelGGL((fill_kernel<ValT>), dim3(grid_for(nl)), dim3(kThreads), 0, stream,
                       s.pr[0].as<ValT>(), nl, ValT(1) / static_cast<ValT>(ng));
  }

  // ---- personalization (pagerank.cpp:127-156; pagerank_impl.cuh:213-226, 299-319)
  s.personalized = a.pers_vertices != nullptr;
  if (s.personalized) {
    CGA_EXPECTS(a.pers_values && a.pers_vertices->size == a.pers_values->size, kAlgorithmRejects,
                "Invalid input argument: if personalization.has_value() is true, the size of "
                "vertices and values should match");
    int64_t np = static_cast<int64_t>(a.pers_vertices->size);
    // in a 1D partition a rank may hold no personalization vertex; the sum is supplied
    // pre-normalised by the driver in that case (values already divided by the global sum)
    bool local_only = g.num_global_vertices != g.num_vertices;
    CGA_EXPECTS(local_only || np > 0, kAlgorithmRejects,
                "Invalid input argument: if personalizations.has_value() is true, the input "
                "personalization vector size should not be 0.");
    s.pers_dense.alloc(static_cast<size_t>(nl > 0 ? nl : 1) * vb);
    CGA_HIP(hipMemsetAsync(s.pers_dense.p, 0, s.pers_dense.bytes, stream));
    DevBuf chk(sizeof(double) + sizeof(unsigned int) * 2);
    CGA_HIP(hipMemsetAsync(chk.p, 0, chk.bytes, stream));
    if (np > 0) {
      hipLaunchKernelGGL((value_checks_kernel<ValT>), dim3(grid_for(np, 1024)), dim3(kThreads), 0,
                         stream, static_cast<ValT const*>(a.pers_values->data), np,
                         chk.as<double>(), reinterpret_cast<unsigned int*>(chk.as<double>() + 1));
      dense_from_pairs<ValT>(stream, a.pers_vertices, a.pers_values, idmap,
                             s.pers_dense.as<ValT>(), true, "personalization");
    }
    struct {
      double sum;
      unsigned int neg;
      unsigned int pad;
    } h_chk{};
    CGA_HIP(hipMemcpyAsync(&h_chk, chk.p, sizeof(h_chk), hipMemcpyDeviceToHost, stream));
    CGA_HIP(hipStreamSynchronize(stream));
    if (a.do_expensive_check) {
      CGA_EXPECTS(h_chk.neg == 0, kAlgorithmRejects,
                  "Invalid input argument: peresonalization values should be non-negative.");
      if (np > 1 && nl > 0) {
        DevBuf seen(static_cast<size_t>(nl + 1) * sizeof(unsigned int));
        CGA_HIP(hipMemsetAsync(seen.p, 0, seen.bytes, stream));
        unsigned int* n_rep = seen.as<unsigned int>() + nl;
        if (a.pers_vertices->type == INT32)
          hipLaunchKernelGGL((count_repeated_ids_kernel<int32_t>), dim3(grid_for(np)), dim3(kThreads), 0, stream,
                             static_cast<int32_t const*>(a.pers_vertices->data), np, idmap, seen.as<unsigned int>(), n_rep);
        else
          hipLaunchKernelGGL((count_repeated_ids_kernel<int64_t>), dim3(grid_for(np)), dim3(kThreads), 0, stream,
                             static_cast<int64_t const*>(a.pers_vertices->data), np, idmap, seen.as<unsigned int>(), n_rep);
        unsigned int h_rep = 0;
        CGA_HIP(hipMemcpyAsync(&h_rep, n_rep, sizeof(h_rep), hipMemcpyDeviceToHost, stream));
        CGA_HIP(hipStreamSynchronize(stream));
        CGA_EXPECTS(h_rep == 0, kAlgorithmRejects,
                    "Invalid input argument: personalization vertices not contain duplicate entries.");
      }
    }
    if (!local_only) {
      // the reference keeps the sum in result_t
      double psum = static_cast<double>(static_cast<ValT>(h_chk.sum));
      CGA_EXPECTS(psum > 0.0, kAlgorithmRejects,
                  "Invalid input argument: sum of personalization valuese should be positive.");
      if (nl > 0)
        hipLaunchKernelGGL((scale_kernel<ValT>), dim3(grid_for(nl)), dim3(kThreads), 0, stream,
                           s.pers_dense.as<ValT>(), nl, psum);
    }
    // short lists go through the fix-up kernel instead of a V-sized read per iteration
    if (np < nl / 8) {
      DevBuf flags(static_cast<size_t>(nl) * sizeof(unsigned int));
      DevBuf pos(static_cast<size_t>(nl + 1) * sizeof(int64_t));
      hipLaunchKernelGGL((flag_nonzero_kernel<ValT>), dim3(grid_for(nl)), dim3(kThreads), 0, stream,
                         s.pers_dense.as<ValT>(), nl, flags.as<unsigned int>());
      exclusive_scan_u32_to_i64(stream, flags.as<unsigned int>(), nl, pos.as<int64_t>());
      int64_t total = 0;
      CGA_HIP(hipMemcpyAsync(&total, pos.as<int64_t>() + nl, sizeof(int64_t), hipMemcpyDeviceToHost,
                             stream));
      CGA_HIP(hipStreamSynchronize(stream));
      s.n_pers = total;
      s.pers_v.alloc(static_cast<size_t>(total > 0 ? total : 1) * sizeof(int32_t));
      s.pers_p.alloc(static_cast<size_t>(total > 0 ? total : 1) * vb);
      if (total > 0)
        hipLaunchKernelGGL((compact_nonzero_kernel<ValT>), dim3(grid_for(nl)), dim3(kThreads), 0,
                           stream, s.pers_dense.as<ValT>(), nl, flags.as<unsigned int>(),
                           pos.as<int64_t>(), s.pers_v.as<int32_t>(), s.pers_p.as<ValT>());
      CGA_HIP(hipGetLastError());
      CGA_HIP(hipStreamSynchronize(stream));
      s.pers_dense.release();
      s.pers_is_dense = false;
    } else {
      s.pers_is_dense = true;
    }
  }

  int n_prep_blocks = session_geometry<ValT>(s, plan);
  // a personalized run gives a dead row its own rank (teleport * p_v), and out-weight sums given
  // by the caller may call a dead row non-dangling: then they are all visited
  s.n_dead = (!s.personalized && plan.contiguous && plan.n_dead <= s.buckets[4].count) ? plan.n_dead : 0;
  if (s.n_dead > 0 && a.pre_sums) {
    // (the multi-GPU driver hands every rank the true sums: zero for a row no edge leaves)
    DevBuf nz(4);
    CGA_HIP(hipMemsetAsync(nz.p, 0, 4, stream));
    hipLaunchKernelGGL((count_nonzero_kernel<ValT>), dim3(grid_for(s.n_dead)), dim3(kThreads), 0, stream,
                       s.out_w.as<ValT>() + (nl - s.n_dead), s.n_dead, nz.as<unsigned int>());
    unsigned int h_nz = 0;
    CGA_HIP(hipMemcpyAsync(&h_nz, nz.p, 4, hipMemcpyDeviceToHost, stream));
    CGA_HIP(hipStreamSynchronize(stream));
    if (h_nz != 0) s.n_dead = 0;
  }

  // ---- x0 and the dangling mass of pr0
  hipLaunchKernelGGL((prep_kernel<ValT>), dim3(n_prep_blocks), dim3(kThreads), 0, stream,
                     s.pr[0].as<ValT>(), s.out_w.as<ValT>(), nl, g.vertex_first, g.x_pos,
                     s.x[0].as<ValT>(),
                     s.partials.as<double>());
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kThreads), 0, stream, s.partials.as<double>(),
                     n_prep_blocks, s.scal.as<double>(), s.ctrl.as<Ctrl>(), s.epsilon, 0, DeadRows{});
  CGA_HIP(hipGetLastError());
}

template <typename OffT, typename ValT>
// phase: -1 the whole iteration; on a graph with two regions of source slots (xs_regions == 2)
// 0 = what needs region 0 only (xs_kernel over its items), 1 = the rest of the iteration.  On a
// graph without regions phase 0 is empty and phase 1 the whole iteration.
void enqueue_iteration(Session& s, hipEvent_t ev_kernel_begin, hipEvent_t ev_kernel_end,
                       ValT const* raw_x = nullptr, ValT* raw_y = nullptr, int phase = -1)
{
  Graph const& g     = *s.cg;
  hipStream_t stream = s.handle->stream;
  bool const split   = s.xs_on && s.graph->plan->xs_regions == 2;
  if (!split) {
    if (phase == 0) return;
    phase = -1;
  }
  int cur = static_cast<int>(s.enqueued & 1), nxt = cur ^ 1;
  IterArgs<OffT, ValT> a{};
  a.offsets         = static_cast<OffT const*>(g.offsets);
  a.indices         = g.indices;
  a.weights         = static_cast<ValT const*>(g.weights);
  a.x_cur           = s.x[cur].as<ValT>();
  a.x_next          = s.x[nxt].as<ValT>();
  a.pr_old          = s.pr[cur].as<ValT>();
  a.pr_new          = s.pr[nxt].as<ValT>();
  a.out_w           = s.out_w.as<ValT>();
  a.pers_dense      = s.pers_is_dense ? s.pers_dense.as<ValT>() : nullptr;
  a.scal            = s.scal.as<double>();
  a.ctrl            = s.ctrl.as<Ctrl>();
  a.partials        = s.partials.as<double>();
  a.alpha           = static_cast<ValT>(s.alpha);
  a.one_minus_alpha = static_cast<ValT>(1.0 - static_cast<double>(static_cast<ValT>(s.alpha)));
  a.num_vertices    = static_cast<ValT>(s.n_global);
  a.epsilon         = s.epsilon;
  a.personalized    = s.personalized ? 1 : 0;
#ifdef CGA_TIMING_AIDS
  {
    char const* gm = std::getenv("CUGRAPH_AMD_GATHER_MASK");
    a.gather_mask  = gm ? static_cast<uint32_t>(std::strtoul(gm, nullptr, 0)) : 0xFFFFFFFFu;
    a.xs.gather_mask = a.gather_mask;
    char const* am     = std::getenv("CUGRAPH_AMD_XS_AFFINITY");
    a.xs.affinity_mode = am ? std::atoi(am) : 0;
    a.xs.profile       = std::getenv("CUGRAPH_AMD_XS_PROFILE") != nullptr && !g.weights;
    a.xs.no_atomics    = std::getenv("CUGRAPH_AMD_XS_NO_ATOMICS") != nullptr;
    a.xs.no_gather     = std::getenv("CUGRAPH_AMD_XS_NO_GATHER") != nullptr;
    a.xs.no_writeback  = std::getenv("CUGRAPH_AMD_XS_NO_WRITEBACK") != nullptr;
  }
#endif
  a.vertex_first    = g.vertex_first;
  a.x_pos           = g.x_pos;
  if (raw_y) {  // plain SpMV for the sibling power iterations (centrality.hip)
    a.x_cur   = raw_x;
    a.raw_out = raw_y;
  }
  a.hub_items        = s.hub_items;
  a.n_hub_items      = s.n_hub_items;
  a.hub_item_partial = s.hub_item_partial.as<double>();
  if (s.xs_on) {
    PullPlan const& plan = *s.graph->plan;
    a.xs.ent       = plan.xs_ent.as<uint32_t>();
    a.xs.n_win     = plan.xs_n_win;
    a.xs.slice_shift = plan.xs_slice_shift;
    a.xs.row_bits  = plan.xs_row_bits;
    a.xs.w         = plan.xs_w.p ? plan.xs_w.as<ValT>() : nullptr;
    a.xs.unit_off  = plan.xs_unit_off.as<int64_t>();
    a.xs.tile_row  = plan.xs_tile_row.as<int32_t>();
    a.xs.rec       = plan.xs_rec[0].as<int64_t>();
    a.xs.pbase     = plan.xs_pbase.as<int64_t>();
    a.xs.n_items   = plan.xs_n_items[0];
    a.xs.partial   = s.xs_partial.as<ValT>();
    a.xs.counters  = s.xs_counters.as<int>();
    a.xs.row_first = plan.xs_row_first;
    a.xs.n_rows    = plan.xs_n_rows;
    a.xs.n_tiles   = plan.xs_n_tiles;
#ifdef CGA_TIMING_AIDS
    a.xs.ent_len     = plan.xs_n_edges + 4;
    a.xs.partial_len = plan.xs_partial_len;
    a.xs.x_len       = static_cast<uint32_t>(std::min<int64_t>(s.x_len, 0xFFFFFFFFLL));
#endif
    if (phase != 1) CGA_HIP(hipMemsetAsync(s.xs_counters.p, 0, 256, stream));  // both regions' queues
  }
  for (int b = 0; b < kNumBuckets; ++b) a.buckets[b] = s.buckets[b];
  // dead rows (the tail of the isolated bucket): visited by the first iteration only, whose old
  // ranks are the caller's; from then on finalize_kernel adds their share (see DeadRows)
  DeadRows dead{};
  if (s.n_dead > 0 && !raw_y) {
    dead.track           = 1;
    dead.f64             = s.f64 ? 1 : 0;
    dead.alpha           = static_cast<double>(a.alpha);
    dead.one_minus_alpha = static_cast<double>(a.one_minus_alpha);
    dead.num_vertices    = static_cast<double>(a.num_vertices);
    if (s.enqueued > 0) {
      dead.n_dead = s.n_dead;
      a.buckets[4].count -= s.n_dead;
    }
  }

  if (ev_kernel_begin) CGA_HIP(hipEventRecord(ev_kernel_begin, stream));
  bool const beside = s.side_stream && s.xs_on && s.n_xs_blocks > 0 && phase == -1 && !raw_y;
  auto launch_spmv = [&](hipStream_t st) {
    if (g.weights)
      hipLaunchKernelGGL((spmv_kernel<OffT, ValT, true>), dim3(s.n_spmv_blocks), dim3(kThreads), 0, st, a);
    else
      hipLaunchKernelGGL((spmv_kernel<OffT, ValT, false>), dim3(s.n_spmv_blocks), dim3(kThreads), 0, st, a);
  };
  if (beside) {
    CGA_HIP(hipEventRecord(s.ev_fork, stream));
    CGA_HIP(hipStreamWaitEvent(s.side_stream, s.ev_fork, 0));
  }
  for (int region = 0; region < (split ? 2 : 1); ++region) {
    if (!(s.xs_on && s.n_xs_blocks > 0)) break;
    if ((region == 0 && phase == 1) || (region == 1 && phase == 0)) continue;
    PullPlan const& plan = *s.graph->plan;
    if (plan.xs_n_items[region] == 0) continue;
    a.xs.rec      = plan.xs_rec[region].as<int64_t>();
    a.xs.n_items  = plan.xs_n_items[region];
    a.xs.counters = s.xs_counters.as<int>() + kXsSlices * region;
#ifdef CGA_TIMING_AIDS
    a.xs.slot_limit = split && region == 0 ? static_cast<uint32_t>(g.hot_slots) : 0u;
    if (a.xs.slot_limit != 0u) hipLaunchKernelGGL(xs_region_count_kernel, dim3(1), dim3(1), 0, stream);
#endif
    XsLaunch<ValT> xl{a.xs, a.x_cur, a.scal, a.ctrl, a.epsilon};
    auto kern = reinterpret_cast<void (*)(XsLaunch<ValT>)>(s.xs_kernel_fn);
    hipLaunchKernelGGL(kern, dim3(s.n_xs_blocks), dim3(s.xs_threads), s.xs_lds_bytes, stream, xl);
#ifdef CGA_XS_PHASE_PROF
    if (s.enqueued >= 3 && s.enqueued < 6) {
      if (plan.xs_async)
        hipLaunchKernelGGL(xs2_phase_dump_kernel, dim3(1), dim3(1), 0, stream);
      else
        hipLaunchKernelGGL(xs_phase_dump_kernel, dim3(1), dim3(1), 0, stream);
    }
#endif
#ifdef CGA_TIMING_AIDS
    if (a.xs.profile && s.enqueued == 3)
      hipLaunchKernelGGL(xs_prof_dump_kernel, dim3(1), dim3(1), 0, stream, std::min(s.n_xs_blocks, 1024));
#endif
  }
  if (phase == 0) {
    CGA_HIP(hipGetLastError());
    return;
  }
  // the dominant kernel: xs_kernel when the graph has XS tiles, else spmv_kernel
  if (ev_kernel_end && s.xs_on) CGA_HIP(hipEventRecord(ev_kernel_end, stream));
  if (beside) {
    launch_spmv(s.side_stream);
    CGA_HIP(hipEventRecord(s.ev_join, s.side_stream));
  } else {
    launch_spmv(stream);
  }
  if (ev_kernel_end && !s.xs_on) CGA_HIP(hipEventRecord(ev_kernel_end, stream));
  if (s.n_hub_blocks > 0)
    hipLaunchKernelGGL((hub_finish_kernel<OffT, ValT>), dim3(s.n_hub_blocks), dim3(kThreads), 0,
                       stream, a, s.hub_row_item_off, s.partials.as<double>() + 2 * s.n_spmv_blocks);
  if (s.n_xs_finish_blocks > 0)
    hipLaunchKernelGGL((xs_finish_kernel<OffT, ValT>), dim3(s.n_xs_finish_blocks), dim3(kThreads), 0,
                       stream, a, s.graph->plan->xs_fin_items.as<int64_t>(), s.graph->plan->xs_n_fin_items,
                       s.partials.as<double>() + 2 * (s.n_spmv_blocks + s.n_hub_blocks));
  if (raw_y) {
    CGA_HIP(hipGetLastError());
    return;
  }
  if (beside) CGA_HIP(hipStreamWaitEvent(stream, s.ev_join, 0));
  if (s.n_pers_blocks > 0)
    hipLaunchKernelGGL((pers_fix_kernel<ValT>), dim3(s.n_pers_blocks), dim3(kThreads), 0, stream,
                       s.pers_v.as<int32_t>(), s.pers_p.as<ValT>(), s.n_pers, a.pr_old, a.pr_new,
                       a.x_next, a.out_w, a.scal, a.ctrl, a.epsilon, a.alpha, a.one_minus_alpha,
                       a.vertex_first, a.x_pos,
                       s.partials.as<double>() +
                         2 * (s.n_spmv_blocks + s.n_hub_blocks + s.n_xs_finish_blocks));
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kThreads), 0, stream, s.partials.as<double>(),
                     s.n_spmv_blocks + s.n_hub_blocks + s.n_xs_finish_blocks + s.n_pers_blocks,
                     s.scal.as<double>(),
                     s.ctrl.as<Ctrl>(), s.epsilon, 1, dead);
  CGA_HIP(hipGetLastError());
  s.enqueued += 1;
}

void enqueue_dispatch(Session& s, hipEvent_t b, hipEvent_t e, int phase = -1)
{
  bool o64 = s.cg->off64;
  if (s.f64) {
    if (o64) enqueue_iteration<int64_t, double>(s, b, e, nullptr, nullptr, phase);
    else enqueue_iteration<int32_t, double>(s, b, e, nullptr, nullptr, phase);
  } else {
    if (o64) enqueue_iteration<int64_t, float>(s, b, e, nullptr, nullptr, phase);
    else enqueue_iteration<int32_t, float>(s, b, e, nullptr, nullptr, phase);
  }
}

}  // namespace

Session* session_create(Handle const& h, Graph& g, PagerankArgs const& a)
{
  CGA_EXPECTS(a.alpha >= 0.0 && a.alpha <= 1.0, kAlgorithmRejects,
              "Invalid input argument: alpha should be in [0.0, 1.0].");
  CGA_EXPECTS(a.epsilon >= 0.0, kAlgorithmRejects,
              "Invalid input argument: epsilon should be non-negative.");
  if (!g.store_transposed) transpose_graph(h, g);  // pagerank.cpp:105-110: mutates the graph
  auto s      = new Session();
  s->handle   = &h;
  s->graph    = &g;
  s->f64      = g.weight_type == FLOAT64;
  s->alpha    = a.alpha;
  // alpha and epsilon are cast to the graph's weight type (pagerank.cpp:234-235)
  s->epsilon  = s->f64 ? a.epsilon : static_cast<double>(static_cast<float>(a.epsilon));
  s->n_local  = g.num_vertices;
  s->n_global = g.num_global_vertices;
  try {
    if (s->f64)
      session_setup<double>(*s, a);
    else
      session_setup<float>(*s, a);
    if (char const* e = std::getenv("CUGRAPH_AMD_SPMV_BESIDE"); e && std::atoi(e) != 0 && s->xs_on) {
      int lo = 0, hi = 0;
      CGA_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo: the least urgent
      CGA_HIP(hipStreamCreateWithPriority(&s->side_stream, hipStreamNonBlocking, std::atoi(e) == 2 ? lo : hi));
      CGA_HIP(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
      CGA_HIP(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    }
  } catch (...) {
    delete s;
    throw;
  }
  return s;
}

void session_iterate(Session& s, size_t n)
{
  if (s.n_global == 0) return;
  for (size_t i = 0; i < n; ++i) enqueue_dispatch(s, nullptr, nullptr);
}

// slots of one XS window for a gathered vector of num_source_slots slots: the granule of a region
int64_t xs_window_slots(int64_t num_source_slots)
{
  if (num_source_slots <= 0) return 0;
  int row_bits = xs_row_bits_env(xs_row_bits_for(num_source_slots, 7));
  if (xs_async_on() && row_bits > kXs2RowBitsMax) row_bits = kXs2RowBitsMax;
  return static_cast<int64_t>(kXsSlices) << (32 - row_bits);
}

void session_iterate_phase(Session& s, int phase)
{
  CGA_EXPECTS(phase == 0 || phase == 1, CUGRAPH_INVALID_INPUT, "phase must be 0 or 1");
  if (s.n_global == 0) return;
  enqueue_dispatch(s, nullptr, nullptr, phase);
}

void session_timed_iterate(Session& s, size_t n, float* iter_ms, float* kernel_ms)
{
  hipStream_t stream = s.handle->stream;
  while (s.events.size() < 3 * n + 1) {
    hipEvent_t e;
    CGA_HIP(hipEventCreate(&e));
    s.events.push_back(e);
  }
  for (size_t i = 0; i < n; ++i) {
    CGA_HIP(hipEventRecord(s.events[3 * i], stream));
    enqueue_dispatch(s, s.events[3 * i + 1], s.events[3 * i + 2]);
  }
  CGA_HIP(hipEventRecord(s.events[3 * n], stream));
  CGA_HIP(hipStreamSynchronize(stream));
  for (size_t i = 0; i < n; ++i) {
    if (iter_ms) CGA_HIP(hipEventElapsedTime(&iter_ms[i], s.events[3 * i], s.events[3 * (i + 1)]));
    if (kernel_ms)
      CGA_HIP(hipEventElapsedTime(&kernel_ms[i], s.events[3 * i + 1], s.events[3 * i + 2]));
  }
}

void session_status(Session& s, size_t* iters, bool* stopped, double* delta, double* dangling)
{
  hipStream_t stream = s.handle->stream;
  Ctrl c{};
  double sc[2] = {0.0, 0.0};
  if (s.n_global > 0) {
    CGA_HIP(hipMemcpyAsync(&c, s.ctrl.p, sizeof(c), hipMemcpyDeviceToHost, stream));
    CGA_HIP(hipMemcpyAsync(sc, s.scal.p, sizeof(sc), hipMemcpyDeviceToHost, stream));
  }
  CGA_HIP(hipStreamSynchronize(stream));
  if (iters) *iters = static_cast<size_t>(c.iters);
  // the stop flag is raised by the first kernel enqueued AFTER the converging iteration; when
  // nothing was enqueued after it the scalars still tell
  CGA_EXPECTS((c.done & 2) == 0, CUGRAPH_UNKNOWN_ERROR, "internal: xs2_kernel gave up waiting on a work item");
  bool st = c.done != 0 || (c.iters > 0 && sc[1] < s.epsilon);
  if (stopped) *stopped = st;
  if (delta) *delta = sc[1];
  if (dangling) *dangling = sc[0];
}

CentralityResult* session_result(Session& s, size_t max_iterations)
{
  Graph& g           = *s.graph;
  hipStream_t stream = s.handle->stream;
  size_t iters       = 0;
  bool stopped       = false;
  session_status(s, &iters, &stopped, nullptr, nullptr);
  auto res = new CentralityResult();
  try {
    size_t vb       = s.f64 ? 8 : 4;
    auto* vals      = new Array();
    res->values     = vals;
    vals->type      = g.weight_type;
    vals->size      = static_cast<size_t>(s.n_local);
    vals->nbytes    = vals->size * vb;
    auto* ids       = new Array();
    res->vertex_ids = ids;
    ids->type       = g.vertex_type;
    ids->size       = static_cast<size_t>(s.n_local);
    ids->nbytes     = ids->size * dtype_size(g.vertex_type);
    if (s.n_local > 0) {
      CGA_HIP(hipMalloc(&vals->data, vals->nbytes));
      CGA_HIP(hipMalloc(&ids->data, ids->nbytes));
      if (s.n_dead > 0 && iters >= 1) {
        // the dead rows were last written by the first iteration: they hold the rank every vertex
        // without in-edges got in the last one
        Ctrl c{};
        CGA_HIP(hipMemcpyAsync(&c, s.ctrl.p, sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
        CGA_HIP(hipStreamSynchronize(stream));
        int64_t first = s.n_local - s.n_dead;
        if (s.f64)
          hipLaunchKernelGGL((fill_kernel<double>), dim3(grid_for(s.n_dead)), dim3(kThreads), 0, stream,
                             s.pr[iters & 1].as<double>() + first, s.n_dead, c.base_prev);
        else
          hipLaunchKernelGGL((fill_kernel<float>), dim3(grid_for(s.n_dead)), dim3(kThreads), 0, stream,
                             s.pr[iters & 1].as<float>() + first, s.n_dead, static_cast<float>(c.base_prev));
        CGA_HIP(hipGetLastError());
      }
      if (s.order) {  // ranks were computed on the degree-sorted copy: back to the graph's order
        int grid = grid_for(s.n_local);
        if (s.f64)
          hipLaunchKernelGGL((permute_out_kernel<double>), dim3(grid), dim3(kThreads), 0, stream,
                             s.pr[iters & 1].as<double>(), s.order, s.n_local,
                             static_cast<double*>(vals->data));
        else
          hipLaunchKernelGGL((permute_out_kernel<float>), dim3(grid), dim3(kThreads), 0, stream,
                             s.pr[iters & 1].as<float>(), s.order, s.n_local,
                             static_cast<float*>(vals->data));
        CGA_HIP(hipGetLastError());
      } else {
        CGA_HIP(hipMemcpyAsync(vals->data, s.pr[iters & 1].p, vals->nbytes, hipMemcpyDeviceToDevice,
                               stream));
      }
      if (g.number_map)
        CGA_HIP(hipMemcpyAsync(ids->data, g.number_map, ids->nbytes, hipMemcpyDeviceToDevice, stream));
      else
        fill_sequence(stream, ids->data, g.vertex_type, ids->size, g.vertex_first);
      CGA_HIP(hipStreamSynchronize(stream));
    }
    res->num_iterations = iters;
    // pagerank_impl.cuh:328-337: ++iter; break if delta < eps, else if iter >= max;
    // converged = iter < max_iterations
    res->converged = iters < max_iterations;
    if (s.n_global == 0) res->converged = true;  // pagerank_impl.cuh:76
  } catch (...) {
    delete res;
    throw;
  }
  return res;
}

void session_info(Session& s, int64_t info[8])
{
  Graph const& g = *s.cg;
  int64_t bval = s.f64 ? 8 : 4;
  int64_t bw   = g.weights ? bval : 0;
  int64_t boff = g.off64 ? 8 : 4;
  info[0]      = g.num_edges * (4 + bw) + (g.num_vertices + 1) * boff + 5 * g.num_vertices * bval;
  info[1]      = g.num_edges;
  info[2]      = s.buckets[4].count + s.buckets[3].count + s.buckets[2].count;
  info[3]      = s.xs_on ? s.graph->plan->xs_n_rows : s.buckets[1].count;
  info[4]      = s.buckets[0].count;
  info[5]      = s.n_hub_items;
  info[6]      = s.n_spmv_blocks;
  // algorithmic bytes of the dominant kernel alone.  xs_kernel: the index (+weight) stream and
  // the offsets of the XS rows, and the one read of the gathered vector; the other four V-terms
  // of those rows belong to xs_finish_kernel.  Without XS tiles: spmv_kernel does everything
  // except the rank update of the hub rows (hub_finish_kernel).
  if (s.xs_on) {
    PullPlan const& plan = *s.graph->plan;
    info[7] = plan.xs_n_edges * (4 + bw) + plan.xs_n_rows * boff + s.x_len * bval;
  } else {
    info[7] = info[0] - 4 * s.buckets[0].count * bval;
  }
}

void session_exchange_buffers(Session& s, void** x_full, void** scalars)
{
  *x_full  = s.x[s.enqueued & 1].p;
  *scalars = s.scal.p;
}

void session_destroy(Session* s) { delete s; }

// ---- plain SpMV on the same plan and kernels: y[v] = sum over in-edges (u -> v) of x[u] * w
// x and y are in the order of the graph the kernels run on (see spmv_session_order)
Session* spmv_session_create(Handle const& h, Graph& g)
{
  CGA_EXPECTS(g.store_transposed, CUGRAPH_INVALID_INPUT, "graph must be stored transposed");
  CGA_EXPECTS(g.num_global_vertices == g.num_vertices && g.x_pos == nullptr, CUGRAPH_NOT_IMPLEMENTED,
              "partitioned graphs are only supported by PageRank");
  auto s = std::make_unique<Session>();
  s->handle   = &h;
  s->graph    = &g;
  s->f64      = g.weight_type == FLOAT64;
  s->n_local  = g.num_vertices;
  s->n_global = g.num_global_vertices;
  s->x_len    = g.num_global_vertices;
  PullPlan& plan = get_plan(h, g);
  s->cg          = plan.compact ? plan.compact : plan.internal ? plan.internal : &g;
  s->order       = plan.internal ? static_cast<int32_t const*>(plan.internal->number_map) : nullptr;
  if (s->f64) session_geometry<double>(*s, plan);
  else session_geometry<float>(*s, plan);
  if (plan.compact) {  // the caller's vector is scattered to the slots before every product
    s->x_len = plan.compact->num_source_slots;
    s->x[0].alloc(static_cast<size_t>(s->x_len) * (s->f64 ? 8 : 4));
    CGA_HIP(hipMemsetAsync(s->x[0].p, 0, s->x[0].bytes, h.stream));
  }
  s->scal.alloc(4 * sizeof(double));
  s->ctrl.alloc(sizeof(Ctrl));
  double never[2] = {0.0, 1e300};  // "previous delta" that never passes the stop check
  CGA_HIP(hipMemsetAsync(s->ctrl.p, 0, sizeof(Ctrl), h.stream));
  CGA_HIP(hipMemcpyAsync(s->scal.p, never, sizeof(never), hipMemcpyHostToDevice, h.stream));
  CGA_HIP(hipStreamSynchronize(h.stream));
  return s.release();
}

int32_t const* spmv_session_order(Session& s) { return s.order; }

template <typename ValT>
__global__ void scatter_to_slots_kernel(ValT const* __restrict__ x, int32_t const* __restrict__ x_pos,
                                        int64_t n, ValT* __restrict__ slots)
{
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n;
       v += (int64_t)gridDim.x * blockDim.x) {
    int32_t p = x_pos[v];
    if (p >= 0) slots[p] = x[v];
  }
}

void spmv_apply(Session& s, void const* x, void* y)
{
  if (s.n_local == 0) return;
  if (s.cg->x_pos) {  // compact source space: the kernels gather by slot
    if (s.f64)
      hipLaunchKernelGGL((scatter_to_slots_kernel<double>), dim3(grid_for(s.n_local)), dim3(kThreads), 0,
                         s.handle->stream, static_cast<double const*>(x), s.cg->x_pos, s.n_local, s.x[0].as<double>());
    else
      hipLaunchKernelGGL((scatter_to_slots_kernel<float>), dim3(grid_for(s.n_local)), dim3(kThreads), 0,
                         s.handle->stream, static_cast<float const*>(x), s.cg->x_pos, s.n_local, s.x[0].as<float>());
    x = s.x[0].p;
  }
  bool o64 = s.cg->off64;
  if (s.f64) {
    if (o64) enqueue_iteration<int64_t, double>(s, nullptr, nullptr, static_cast<double const*>(x), static_cast<double*>(y));
    else enqueue_iteration<int32_t, double>(s, nullptr, nullptr, static_cast<double const*>(x), static_cast<double*>(y));
  } else {
    if (o64) enqueue_iteration<int64_t, float>(s, nullptr, nullptr, static_cast<float const*>(x), static_cast<float*>(y));
    else enqueue_iteration<int32_t, float>(s, nullptr, nullptr, static_cast<float const*>(x), static_cast<float*>(y));
  }
}
bool session_is_f64(Session& s) { return s.f64; }
int64_t session_num_global(Session& s) { return s.x_len; }

CentralityResult* run_pagerank(Handle const& h, Graph& g, PagerankArgs const& a)
{
  Session* s = session_create(h, g, a);
  CentralityResult* res = nullptr;
  try {
    // the loop body of the reference runs at least once (pagerank_impl.cuh:235-335)
    size_t budget = std::max<size_t>(a.max_iterations, 1);
    size_t batch  = 4;
    while (s->enqueued < budget) {
      size_t n = std::min(batch, budget - s->enqueued);
      session_iterate(*s, n);
      size_t iters;
      bool stopped;
      session_status(*s, &iters, &stopped, nullptr, nullptr);
      if (stopped || s->n_global == 0) break;
      batch = std::min<size_t>(batch * 2, 32);
    }
    res = session_result(*s, a.max_iterations);
  } catch (...) {
    delete s;
    throw;
  }
  delete s;
  return res;
}

}  // namespace cga
