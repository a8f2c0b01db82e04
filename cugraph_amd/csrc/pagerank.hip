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
// per hub with LDS-staged partial sums; all buckets run in ONE launch, hubs first.
// The path is a sparse HBM-bound contraction: no MFMA.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace cga {

namespace {

constexpr int kThreads     = 256;
constexpr int kNumBuckets  = 4;
constexpr int kMaxPartials = 8192;

struct Ctrl {
  unsigned long long iters;
  int done;
  int pad;
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
  int64_t vertex_first;
  Bucket buckets[kNumBuckets];
};

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
    ValT pr = base + a.alpha * weighted_sum;
    if (a.pers_dense) pr += teleport * a.pers_dense[v];
    ValT old = a.pr_old[v];
    ValT ow  = a.out_w[v];
    delta += fabs(static_cast<double>(pr) - static_cast<double>(old));
    ValT x = pr;
    if (ow == ValT(0)) {
      dangling += static_cast<double>(pr);
    } else {
      x = pr / ow;
    }
    a.pr_new[v]                 = pr;
    a.x_next[a.vertex_first + v] = x;
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

// G lanes cooperate on one destination vertex
template <int G, typename OffT, typename ValT, bool WEIGHTED>
__device__ inline void group_bucket(IterArgs<OffT, ValT> const& a, Bucket const& b,
                                    Epilogue<OffT, ValT>& ep)
{
  using AccT       = typename std::conditional<(G > 1), double, ValT>::type;
  int lane         = threadIdx.x & (G - 1);
  int64_t group    = ((int64_t)(blockIdx.x - b.block_begin) * kThreads + threadIdx.x) / G;
  int64_t n_groups = ((int64_t)b.n_blocks * kThreads) / G;
  for (int64_t i = group; i < b.count; i += n_groups) {
    int64_t v = b.list ? static_cast<int64_t>(b.list[i]) : b.first + i;
    OffT beg = a.offsets[v], end = a.offsets[v + 1];
    AccT acc = 0;
#pragma unroll 4
    for (OffT j = beg + lane; j < end; j += G) {
      ValT xv = a.x_cur[a.indices[j]];
      if constexpr (WEIGHTED) xv *= a.weights[j];
      acc += static_cast<AccT>(xv);
    }
    if constexpr (G > 1) acc = group_sum(acc, G);
    if (lane == 0) ep.finish(v, static_cast<ValT>(acc));
  }
}

// one workgroup per hub: threads stride the column (coalesced), partial sums staged through LDS
template <typename OffT, typename ValT, bool WEIGHTED>
__device__ inline void hub_bucket(IterArgs<OffT, ValT> const& a, Bucket const& b,
                                  Epilogue<OffT, ValT>& ep)
{
  __shared__ double hub_lds[kThreads / 64];
  for (int64_t i = blockIdx.x - b.block_begin; i < b.count; i += b.n_blocks) {
    int64_t v = b.list ? static_cast<int64_t>(b.list[i]) : b.first + i;
    OffT beg = a.offsets[v], end = a.offsets[v + 1];
    double acc = 0.0;
#pragma unroll 4
    for (OffT j = beg + threadIdx.x; j < end; j += kThreads) {
      ValT xv = a.x_cur[a.indices[j]];
      if constexpr (WEIGHTED) xv *= a.weights[j];
      acc += static_cast<double>(xv);
    }
    acc = group_sum(acc, 64);
    if ((threadIdx.x & 63) == 0) hub_lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = 0.0;
      for (int w = 0; w < kThreads / 64; ++w) s += hub_lds[w];
      ep.finish(v, static_cast<ValT>(s));
    }
    __syncthreads();
  }
}

template <typename OffT, typename ValT, bool WEIGHTED>
__global__ __launch_bounds__(kThreads) void spmv_kernel(IterArgs<OffT, ValT> a)
{
  if (stop_check(a.ctrl, a.scal, a.epsilon, true)) return;
  Epilogue<OffT, ValT> ep(a);
  int blk = blockIdx.x;
  // bucket order in the launch: hubs, wave64, 16-lane, thread-per-vertex
  if (blk < a.buckets[0].block_begin + a.buckets[0].n_blocks) {
    hub_bucket<OffT, ValT, WEIGHTED>(a, a.buckets[0], ep);
  } else if (blk < a.buckets[1].block_begin + a.buckets[1].n_blocks) {
    group_bucket<64, OffT, ValT, WEIGHTED>(a, a.buckets[1], ep);
  } else if (blk < a.buckets[2].block_begin + a.buckets[2].n_blocks) {
    group_bucket<16, OffT, ValT, WEIGHTED>(a, a.buckets[2], ep);
  } else {
    group_bucket<1, OffT, ValT, WEIGHTED>(a, a.buckets[3], ep);
  }
  write_partials(ep.dangling, ep.delta, a.partials);
}

// pr' += teleport * p_i for a short personalization list, with the matching corrections of x',
// the L1 delta and the dangling mass (pagerank_impl.cuh:299-319).  Entries are unique vertices.
template <typename ValT>
__global__ __launch_bounds__(kThreads) void pers_fix_kernel(
  int32_t const* __restrict__ pers_v, ValT const* __restrict__ pers_p, int64_t n_pers,
  ValT const* __restrict__ pr_old, ValT* __restrict__ pr_new, ValT* __restrict__ x_next,
  ValT const* __restrict__ out_w, double const* __restrict__ scal, Ctrl* ctrl, double epsilon,
  ValT alpha, ValT one_minus_alpha, int64_t vertex_first, double* __restrict__ partials)
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
    x_next[vertex_first + v] = x;
  }
  write_partials(dangling, delta, partials);
}

// one workgroup: fold the per-workgroup partials in a fixed order -> bitwise reproducible scalars
__global__ __launch_bounds__(kThreads) void finalize_kernel(double const* __restrict__ partials,
                                                            int n_partials, double* scal, Ctrl* ctrl,
                                                            double epsilon, int count_iteration)
{
  if (count_iteration && stop_check(ctrl, scal, epsilon, false)) return;
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
    scal[0] = ds;
    if (count_iteration) {
      scal[1] = ls;
      ctrl->iters += 1;
    } else {
      scal[1] = 1e300;  // "no previous iteration": never below epsilon, survives an 8-rank sum
    }
  }
}

// x0 = pr0 / out_w and the dangling mass of pr0 (pagerank_impl.cuh:239-264, first trip)
template <typename ValT>
__global__ __launch_bounds__(kThreads) void prep_kernel(ValT const* __restrict__ pr,
                                                        ValT const* __restrict__ out_w, int64_t n,
                                                        int64_t vertex_first, ValT* __restrict__ x,
                                                        double* __restrict__ partials)
{
  double dangling = 0.0;
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n;
       v += (int64_t)gridDim.x * blockDim.x) {
    ValT p = pr[v], ow = out_w[v];
    if (ow == ValT(0)) {
      dangling += static_cast<double>(p);
      x[vertex_first + v] = p;
    } else {
      x[vertex_first + v] = p / ow;
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

// external id -> row of the compute graph, through up to two lookup tables
struct IdMap {
  int32_t const* lookup0;  // graph's own renumber map (external -> graph id), or nullptr
  int64_t min0, range0;
  int32_t const* lookup1;  // plan's degree-sort permutation (graph id -> plan row), or nullptr
  int64_t first;           // first global id of the local range (no lookup0)
  int64_t count;           // rows
  __device__ inline int64_t row(int64_t id) const
  {
    int64_t v;
    if (lookup0) {
      int64_t k = id - min0;
      if (k < 0 || k >= range0) return -1;
      v = lookup0[k];
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

// implemented in graph_build.hip (reused here for stream compaction)
void exclusive_scan_u32_to_i64(hipStream_t stream, unsigned int const* counts, int64_t n,
                               int64_t* out /* [n+1] */);

// ------------------------------------------------------------------ plan (cached on the graph)
// Degree buckets as ascending index lists over the graph's own vertex order.
struct PullPlan {
  Graph* internal{nullptr};  // degree-sorted copy the kernels run on (nullptr: the graph itself)
  DevBuf lists[kNumBuckets];
  int64_t counts[kNumBuckets]{};
  int64_t firsts[kNumBuckets]{};
  bool contiguous{false};  // buckets are row ranges, no index lists
  DevBuf out_w_sums;  // double[num_global_vertices]: per-source weight sums, compute-graph order
  bool have_out_w{false};
  ~PullPlan() { delete internal; }
};

void destroy_plan(PullPlan* p) { delete p; }

namespace {

// [lo, hi) in-degree ranges, launch order: hubs, wave64, 16-lane groups, thread per vertex
constexpr int64_t kBucketLo[kNumBuckets] = {2048, 48, 6, 0};
constexpr int64_t kBucketHi[kNumBuckets] = {INT64_MAX, 2048, 48, 6};

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
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));
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
      Graph const& cg = p->internal ? *p->internal : g;
      if (cg.off64)
        build_plan_lists<int64_t>(stream, cg, *p);
      else
        build_plan_lists<int32_t>(stream, cg, *p);
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
  CGA_HIP(hipMemsetAsync(sums, 0, static_cast<size_t>(g.num_global_vertices) * sizeof(double), stream));
  if (g.num_edges > 0) {
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
  int n_spmv_blocks{0}, n_pers_blocks{0};
  Bucket buckets[kNumBuckets]{};
  std::vector<hipEvent_t> events;
  ~Session()
  {
    for (auto e : events) (void)hipEventDestroy(e);
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

template <typename ValT>
void session_setup(Session& s, PagerankArgs const& a)
{
  hipStream_t stream = s.handle->stream;
  PullPlan& plan     = get_plan(*s.handle, *s.graph);
  s.cg               = plan.internal ? plan.internal : s.graph;
  s.order            = plan.internal ? static_cast<int32_t const*>(plan.internal->number_map) : nullptr;
  Graph const& g     = *s.cg;
  int64_t nl = s.n_local, ng = s.n_global;
  size_t vb = sizeof(ValT);
  IdMap idmap{};
  idmap.lookup0 = s.graph->ext_lookup;
  idmap.min0    = s.graph->ext_min;
  idmap.range0  = s.graph->ext_range;
  idmap.lookup1 = plan.internal ? plan.internal->ext_lookup : nullptr;
  idmap.first   = s.graph->vertex_first;
  idmap.count   = nl;

  // 1D partition: rank r owns destinations [r*chunk, min((r+1)*chunk, V)), chunk = ceil(V / P);
  // the gathered vector is padded to P equal chunks so that one all-gather fills it
  s.x_len = ng;
  if (g.num_global_vertices != g.num_vertices || s.handle->comm_size > 1) {
    int64_t P     = s.handle->comm_size;
    int64_t chunk = (ng + P - 1) / P;
    int64_t first = std::min<int64_t>(chunk * s.handle->rank, ng);
    int64_t last  = std::min<int64_t>(first + chunk, ng);
    CGA_EXPECTS(g.vertex_first == first && g.num_vertices == last - first, CUGRAPH_INVALID_INPUT,
                "partitioned graph: this rank must own destinations [rank*ceil(V/P), ...) "
                "(see cugraph_amd.mg.partition_bounds)");
    s.x_len = chunk * P;
  }
  for (int k = 0; k < 2; ++k) {
    s.pr[k].alloc(static_cast<size_t>(nl > 0 ? nl : 1) * vb);
    s.x[k].alloc(static_cast<size_t>(s.x_len > 0 ? s.x_len : 1) * vb);
  }
  s.out_w.alloc(static_cast<size_t>(nl > 0 ? nl : 1) * vb);
  s.scal.alloc(2 * sizeof(double));
  s.ctrl.alloc(sizeof(Ctrl));
  CGA_HIP(hipMemsetAsync(s.ctrl.p, 0, sizeof(Ctrl), stream));
  // x buffers are gathered from at every global id: start them defined
  CGA_HIP(hipMemsetAsync(s.x[0].p, 0, s.x[0].bytes, stream));
  CGA_HIP(hipMemsetAsync(s.x[1].p, 0, s.x[1].bytes, stream));

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
      CGA_EXPECTS(neg == 0, CUGRAPH_INVALID_INPUT,
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
    CGA_EXPECTS(a.pers_values && a.pers_vertices->size == a.pers_values->size, CUGRAPH_INVALID_INPUT,
                "Invalid input argument: if personalization.has_value() is true, the size of "
                "vertices and values should match");
    int64_t np = static_cast<int64_t>(a.pers_vertices->size);
    // in a 1D partition a rank may hold no personalization vertex; the sum is supplied
    // pre-normalised by the driver in that case (values already divided by the global sum)
    bool local_only = g.num_global_vertices != g.num_vertices;
    CGA_EXPECTS(local_only || np > 0, CUGRAPH_INVALID_INPUT,
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
    if (a.do_expensive_check)
      CGA_EXPECTS(h_chk.neg == 0, CUGRAPH_INVALID_INPUT,
                  "Invalid input argument: peresonalization values should be non-negative.");
    if (!local_only) {
      // the reference keeps the sum in result_t
      double psum = static_cast<double>(static_cast<ValT>(h_chk.sum));
      CGA_EXPECTS(psum > 0.0, CUGRAPH_INVALID_INPUT,
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

  // ---- launch geometry: all buckets in one grid, hubs first
  int begin = 0;
  // timing aid: CUGRAPH_AMD_BUCKET_MASK=<bits> runs only the named buckets (results are then wrong)
  char const* mask_env = std::getenv("CUGRAPH_AMD_BUCKET_MASK");
  int bucket_mask      = mask_env ? std::atoi(mask_env) : 0xF;
  int const group_width[kNumBuckets] = {kThreads, 64, 16, 1};
  int const block_cap[kNumBuckets]   = {1024, 2048, 2048, 2048};
  for (int b = 0; b < kNumBuckets; ++b) {
    Bucket& bk    = s.buckets[b];
    bk.count      = ((bucket_mask >> b) & 1) ? plan.counts[b] : 0;
    bk.list       = plan.lists[b].p ? plan.lists[b].as<int32_t>() : nullptr;
    bk.first      = plan.firsts[b];
    int64_t need  = (bk.count * group_width[b] + kThreads - 1) / kThreads;
    bk.n_blocks   = static_cast<int>(std::min<int64_t>(need, block_cap[b]));
    bk.block_begin = begin;
    begin += bk.n_blocks;
  }
  s.n_spmv_blocks = std::max(begin, 1);
  s.n_pers_blocks = (s.personalized && !s.pers_is_dense && s.n_pers > 0)
                      ? grid_for(s.n_pers, 256)
                      : 0;
  int n_prep_blocks = grid_for(nl, 1024);
  int n_partials    = std::max(s.n_spmv_blocks + s.n_pers_blocks, n_prep_blocks);
  CGA_EXPECTS(n_partials <= kMaxPartials, CUGRAPH_UNKNOWN_ERROR, "internal: too many partials");
  s.partials.alloc(static_cast<size_t>(2 * n_partials) * sizeof(double));

  // ---- x0 and the dangling mass of pr0
  hipLaunchKernelGGL((prep_kernel<ValT>), dim3(n_prep_blocks), dim3(kThreads), 0, stream,
                     s.pr[0].as<ValT>(), s.out_w.as<ValT>(), nl, g.vertex_first, s.x[0].as<ValT>(),
                     s.partials.as<double>());
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kThreads), 0, stream, s.partials.as<double>(),
                     n_prep_blocks, s.scal.as<double>(), s.ctrl.as<Ctrl>(), s.epsilon, 0);
  CGA_HIP(hipGetLastError());
}

template <typename OffT, typename ValT>
void enqueue_iteration(Session& s, hipEvent_t ev_kernel_begin, hipEvent_t ev_kernel_end)
{
  Graph const& g     = *s.cg;
  hipStream_t stream = s.handle->stream;
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
  a.vertex_first    = g.vertex_first;
  for (int b = 0; b < kNumBuckets; ++b) a.buckets[b] = s.buckets[b];

  if (ev_kernel_begin) CGA_HIP(hipEventRecord(ev_kernel_begin, stream));
  if (g.weights)
    hipLaunchKernelGGL((spmv_kernel<OffT, ValT, true>), dim3(s.n_spmv_blocks), dim3(kThreads), 0,
                       stream, a);
  else
    hipLaunchKernelGGL((spmv_kernel<OffT, ValT, false>), dim3(s.n_spmv_blocks), dim3(kThreads), 0,
                       stream, a);
  if (ev_kernel_end) CGA_HIP(hipEventRecord(ev_kernel_end, stream));
  if (s.n_pers_blocks > 0)
    hipLaunchKernelGGL((pers_fix_kernel<ValT>), dim3(s.n_pers_blocks), dim3(kThreads), 0, stream,
                       s.pers_v.as<int32_t>(), s.pers_p.as<ValT>(), s.n_pers, a.pr_old, a.pr_new,
                       a.x_next, a.out_w, a.scal, a.ctrl, a.epsilon, a.alpha, a.one_minus_alpha,
                       a.vertex_first, s.partials.as<double>() + 2 * s.n_spmv_blocks);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kThreads), 0, stream, s.partials.as<double>(),
                     s.n_spmv_blocks + s.n_pers_blocks, s.scal.as<double>(), s.ctrl.as<Ctrl>(),
                     s.epsilon, 1);
  CGA_HIP(hipGetLastError());
  s.enqueued += 1;
}

void enqueue_dispatch(Session& s, hipEvent_t b, hipEvent_t e)
{
  bool o64 = s.cg->off64;
  if (s.f64) {
    if (o64) enqueue_iteration<int64_t, double>(s, b, e);
    else enqueue_iteration<int32_t, double>(s, b, e);
  } else {
    if (o64) enqueue_iteration<int64_t, float>(s, b, e);
    else enqueue_iteration<int32_t, float>(s, b, e);
  }
}

}  // namespace

Session* session_create(Handle const& h, Graph& g, PagerankArgs const& a)
{
  CGA_EXPECTS(a.alpha >= 0.0 && a.alpha <= 1.0, CUGRAPH_INVALID_INPUT,
              "Invalid input argument: alpha should be in [0.0, 1.0].");
  CGA_EXPECTS(a.epsilon >= 0.0, CUGRAPH_INVALID_INPUT,
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
  info[2]      = s.buckets[3].count + s.buckets[2].count;
  info[3]      = s.buckets[1].count;
  info[4]      = s.buckets[0].count;
  info[5]      = 0;
  info[6]      = s.n_spmv_blocks;
  info[7]      = s.enqueued;
}

void session_exchange_buffers(Session& s, void** x_full, void** scalars)
{
  *x_full  = s.x[s.enqueued & 1].p;
  *scalars = s.scal.p;
}

void session_destroy(Session* s) { delete s; }
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
