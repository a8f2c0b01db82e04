// The sibling power iterations of PageRank on the same pull SpMV:
// Katz centrality, eigenvector centrality and HITS.  Reference:
//   cpp/src/centrality/katz_centrality_impl.cuh:43-196
//   cpp/src/centrality/eigenvector_centrality_impl.cuh:46-165
//   cpp/src/link_analysis/hits_impl.cuh:57-206
// Each iteration is one (HITS: two) launch(es) of the PageRank SpMV kernels in "raw" mode
// (pagerank.hip: spmv_apply -- XS tiles, degree buckets, the same plan cached on the graph) plus
// V-sized elementwise kernels whose reductions (L1 difference, sum of squares, maximum, sum) are
// folded per workgroup on the device and finished on the host in a fixed order, in fp64.
// Vectors live in the order of the graph the kernels run on and are permuted back once at the end.
#include "common.hpp"

#include <cmath>
#include <limits>

namespace cga {

struct Session;
Session* spmv_session_create(Handle const& h, Graph& g);
int32_t const* spmv_session_order(Session& s);
void spmv_apply(Session& s, void const* x, void* y);
void session_destroy(Session* s);

namespace {

constexpr int kThreads   = 256;
constexpr int kRedBlocks = 1024;

inline int grid_for(int64_t items, int cap = 256 * 16)
{
  int64_t b = (items + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return static_cast<int>(b);
}

#define GRID_STRIDE(i, n)                                                        \
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n);      \
       i += (int64_t)gridDim.x * blockDim.x)

enum class Red { kSum, kSumSq, kMax, kAbsDiff, kNegCount };

__device__ inline double block_fold(double v, bool is_max)
{
  __shared__ double lds[kThreads / 64];
  for (int o = 32; o > 0; o >>= 1) {
    double t = __shfl_xor(v, o, 64);
    v        = is_max ? (t > v ? t : v) : v + t;
  }
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = lds[0];
  for (int w = 1; w < kThreads / 64; ++w) r = is_max ? (lds[w] > r ? lds[w] : r) : r + lds[w];
  return r;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void reduce_kernel(T const* a, T const* b, int64_t n, int op,
                                                          double* partials)
{
  bool is_max = op == static_cast<int>(Red::kMax);
  double acc  = is_max ? -1.7976931348623157e308 : 0.0;
  GRID_STRIDE(i, n)
  {
    double x = static_cast<double>(a[i]);
    switch (static_cast<Red>(op)) {
      case Red::kSum: acc += x; break;
      case Red::kSumSq: acc += x * x; break;
      case Red::kMax: acc = x > acc ? x : acc; break;
      case Red::kAbsDiff: acc += fabs(x - static_cast<double>(b[i])); break;
      case Red::kNegCount: acc += x < 0.0 ? 1.0 : 0.0; break;
    }
  }
  acc = block_fold(acc, is_max);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

template <typename T>
__global__ void fill_kernel(T* p, int64_t n, T v)
{
  GRID_STRIDE(i, n) p[i] = v;
}
template <typename T>
__global__ void scale_kernel(T* p, int64_t n, T divisor)
{
  GRID_STRIDE(i, n) p[i] = p[i] / divisor;
}
// y = alpha * y + (betas ? betas[i] : beta)      (katz_centrality_impl.cuh:110-150)
template <typename T>
__global__ void katz_update_kernel(T* y, T const* betas, int64_t n, T alpha, T beta)
{
  GRID_STRIDE(i, n) y[i] = alpha * y[i] + (betas ? betas[i] : beta);
}
// y += old      (eigenvector_centrality_impl.cuh:124-129)
template <typename T>
__global__ void add_kernel(T* y, T const* old, int64_t n)
{
  GRID_STRIDE(i, n) y[i] = y[i] + old[i];
}
template <typename T>
__global__ void gather_kernel(T const* in, int32_t const* order, int64_t n, T* out)
{
  GRID_STRIDE(i, n) out[i] = in[order[i]];
}
template <typename T>
__global__ void scatter_kernel(T const* in, int32_t const* order, int64_t n, T* out)
{
  GRID_STRIDE(i, n) out[order[i]] = in[i];
}
template <typename IdT, typename T>
__global__ void scatter_pairs_kernel(IdT const* ids, T const* vals, int64_t n, ExtLookup ext,
                                     long long count, T* dense, unsigned int* bad)
{
  GRID_STRIDE(i, n)
  {
    long long v = static_cast<long long>(ids[i]);
    if (ext.lookup) v = ext.find(v);
    if (v < 0 || v >= count)
      atomicAdd(bad, 1u);
    else
      dense[v] = vals[i];
  }
}

template <typename T>
struct Ops {
  hipStream_t stream;
  int64_t n;
  DevBuf partials;
  std::vector<double> host;
  Ops(hipStream_t s, int64_t n_) : stream(s), n(n_), partials(kRedBlocks * sizeof(double)), host(kRedBlocks) {}
  double reduce(Red op, T const* a, T const* b = nullptr)
  {
    if (n == 0) return 0.0;
    int grid = grid_for(n, kRedBlocks);
    hipLaunchKernelGGL((reduce_kernel<T>), dim3(grid), dim3(kThreads), 0, stream, a, b, n,
                       static_cast<int>(op), partials.as<double>());
    CGA_HIP(hipMemcpyAsync(host.data(), partials.p, grid * sizeof(double), hipMemcpyDeviceToHost, stream));
    CGA_HIP(hipStreamSynchronize(stream));
    double r = host[0];
    for (int i = 1; i < grid; ++i) r = op == Red::kMax ? std::max(r, host[i]) : r + host[i];
    return r;
  }
  void fill(T* p, T v) { if (n) hipLaunchKernelGGL((fill_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, stream, p, n, v); }
  void divide(T* p, T d) { if (n) hipLaunchKernelGGL((scale_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, stream, p, n, d); }
  void gather(T const* in, int32_t const* order, T* out)
  {
    if (n) hipLaunchKernelGGL((gather_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, stream, in, order, n, out);
  }
  void scatter(T const* in, int32_t const* order, T* out)
  {
    if (n) hipLaunchKernelGGL((scatter_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, stream, in, order, n, out);
  }
};

struct SessionHolder {
  Session* s{nullptr};
  ~SessionHolder() { if (s) session_destroy(s); }
};

// values in the compute order of `sess` -> result arrays in the caller graph's order, ids attached
template <typename T>
Array* values_out(Handle const& h, Graph& g, Session& sess, T const* vals_cg)
{
  auto out    = std::make_unique<Array>();
  out->type   = g.weight_type;
  out->size   = static_cast<size_t>(g.num_vertices);
  out->nbytes = out->size * sizeof(T);
  if (out->size == 0) return out.release();
  CGA_HIP(hipMalloc(&out->data, out->nbytes));
  int32_t const* order = spmv_session_order(sess);
  if (order)
    hipLaunchKernelGGL((scatter_kernel<T>), dim3(grid_for(g.num_vertices)), dim3(kThreads), 0, h.stream,
                       vals_cg, order, g.num_vertices, static_cast<T*>(out->data));
  else
    CGA_HIP(hipMemcpyAsync(out->data, vals_cg, out->nbytes, hipMemcpyDeviceToDevice, h.stream));
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(h.stream));
  return out.release();
}

Array* vertex_ids_out(Handle const& h, Graph& g)
{
  auto ids    = std::make_unique<Array>();
  ids->type   = g.vertex_type;
  ids->size   = static_cast<size_t>(g.num_vertices);
  ids->nbytes = ids->size * dtype_size(g.vertex_type);
  if (ids->size == 0) return ids.release();
  CGA_HIP(hipMalloc(&ids->data, ids->nbytes));
  if (g.number_map)
    CGA_HIP(hipMemcpyAsync(ids->data, g.number_map, ids->nbytes, hipMemcpyDeviceToDevice, h.stream));
  else
    fill_sequence(h.stream, ids->data, g.vertex_type, ids->size, 0);
  CGA_HIP(hipStreamSynchronize(h.stream));
  return ids.release();
}

// dense vector in the CALLER graph's row order from (external id, value) pairs
template <typename T>
void dense_from_pairs(Handle const& h, Graph& g, ArrayView const& ids, ArrayView const& vals, T* dense,
                      char const* what)
{
  CGA_EXPECTS(ids.size == vals.size, CUGRAPH_INVALID_INPUT, std::string(what) + ": size mismatch");
  CGA_EXPECTS(ids.type == g.vertex_type, CUGRAPH_INVALID_INPUT,
              std::string("vertex type of graph and ") + what + " must match");
  CGA_EXPECTS(vals.type == g.weight_type, CUGRAPH_INVALID_INPUT,
              std::string("weight type of graph and ") + what + " must match");
  int64_t n = static_cast<int64_t>(ids.size);
  if (n == 0) return;
  DevBuf bad(4);
  CGA_HIP(hipMemsetAsync(bad.p, 0, 4, h.stream));
  if (ids.type == INT64)
    hipLaunchKernelGGL((scatter_pairs_kernel<int64_t, T>), dim3(grid_for(n)), dim3(kThreads), 0, h.stream,
                       static_cast<int64_t const*>(ids.data), static_cast<T const*>(vals.data), n,
                       ExtLookup{g.ext_lookup, g.ext_min, g.ext_range, g.ext_sorted},
                       (long long)g.num_vertices, dense, bad.as<unsigned int>());
  else
    hipLaunchKernelGGL((scatter_pairs_kernel<int32_t, T>), dim3(grid_for(n)), dim3(kThreads), 0, h.stream,
                       static_cast<int32_t const*>(ids.data), static_cast<T const*>(vals.data), n,
                       ExtLookup{g.ext_lookup, g.ext_min, g.ext_range, g.ext_sorted},
                       (long long)g.num_vertices, dense, bad.as<unsigned int>());
  unsigned int h_bad = 0;
  CGA_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, h.stream));
  CGA_HIP(hipStreamSynchronize(h.stream));
  CGA_EXPECTS(h_bad == 0, CUGRAPH_INVALID_INPUT, std::string(what) + ": vertex ids outside the graph");
}

// ------------------------------------------------------------------ Katz
template <typename T>
CentralityResult* katz_impl(Handle const& h, Graph& g, ArrayView const* betas, double alpha_d,
                            double beta_d, double epsilon_d, size_t max_iterations,
                            bool do_expensive_check)
{
  (void)do_expensive_check;  // no initial guess through the C API (katz.cpp:131: has_initial_guess = false)
  CGA_EXPECTS(alpha_d >= 0.0 && alpha_d <= 1.0, kAlgorithmRejects,
              "Invalid input argument: alpha should be in [0.0, 1.0].");
  CGA_EXPECTS(epsilon_d >= 0.0, kAlgorithmRejects,
              "Invalid input argument: epsilon should be non-negative.");
  T alpha = static_cast<T>(alpha_d), beta = static_cast<T>(beta_d), epsilon = static_cast<T>(epsilon_d);
  int64_t n = g.num_vertices;
  auto res  = std::make_unique<CentralityResult>();
  SessionHolder sh;
  sh.s = spmv_session_create(h, g);
  Ops<T> ops(h.stream, n);
  size_t vb = sizeof(T) * static_cast<size_t>(n > 0 ? n : 1);
  DevBuf cur(vb), old(vb), betas_cg;
  int32_t const* order = spmv_session_order(*sh.s);
  if (betas) {
    // betas[i] belongs to external vertex id vertex_first + i (katz.cpp:101-122)
    CGA_EXPECTS(static_cast<int64_t>(betas->size) == n && betas->type == g.weight_type, CUGRAPH_INVALID_INPUT,
                "betas must hold one value of the graph's weight type per vertex");
    DevBuf ids(static_cast<size_t>(n > 0 ? n : 1) * dtype_size(g.vertex_type)), dense(vb);
    fill_sequence(h.stream, ids.p, g.vertex_type, static_cast<size_t>(n), 0);
    CGA_HIP(hipMemsetAsync(dense.p, 0, vb, h.stream));
    ArrayView iv{ids.p, static_cast<size_t>(n), static_cast<size_t>(n) * dtype_size(g.vertex_type), g.vertex_type};
    dense_from_pairs<T>(h, g, iv, *betas, dense.as<T>(), "betas");
    betas_cg.alloc(vb);
    if (order) ops.gather(dense.as<T>(), order, betas_cg.as<T>());
    else CGA_HIP(hipMemcpyAsync(betas_cg.p, dense.p, vb, hipMemcpyDeviceToDevice, h.stream));
    CGA_HIP(hipStreamSynchronize(h.stream));
  }
  ops.fill(cur.as<T>(), T(0));
  size_t iter = 0;
  if (n > 0) {
    while (true) {
      std::swap(cur, old);
      spmv_apply(*sh.s, old.p, cur.p);
      hipLaunchKernelGGL((katz_update_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, h.stream,
                         cur.as<T>(), betas ? betas_cg.as<T>() : nullptr, n, alpha, beta);
      double diff = ops.reduce(Red::kAbsDiff, cur.as<T>(), old.as<T>());
      ++iter;
      if (static_cast<T>(diff) < epsilon) break;
      CGA_EXPECTS(iter < max_iterations, CUGRAPH_UNKNOWN_ERROR, "Katz Centrality failed to converge.");
    }
    // the C API always normalises (katz.cpp:131: normalize = true)
    double l2 = std::sqrt(ops.reduce(Red::kSumSq, cur.as<T>()));
    CGA_EXPECTS(l2 > 0.0, CUGRAPH_UNKNOWN_ERROR,
                "L2 norm of the computed Katz Centrality values should be positive.");
    ops.divide(cur.as<T>(), static_cast<T>(l2));
  }
  res->values         = values_out<T>(h, g, *sh.s, cur.as<T>());
  res->vertex_ids     = vertex_ids_out(h, g);
  res->num_iterations = iter;
  res->converged      = true;
  return res.release();
}

// ------------------------------------------------------------------ eigenvector centrality
template <typename T>
CentralityResult* eigenvector_impl(Handle const& h, Graph& g, double epsilon_d, size_t max_iterations,
                                   bool do_expensive_check)
{
  (void)do_expensive_check;
  CGA_EXPECTS(epsilon_d >= 0.0, kAlgorithmRejects,
              "Invalid input argument: epsilon should be non-negative.");
  T epsilon = static_cast<T>(epsilon_d);
  int64_t n = g.num_vertices;
  auto res  = std::make_unique<CentralityResult>();
  SessionHolder sh;
  sh.s = spmv_session_create(h, g);
  Ops<T> ops(h.stream, n);
  size_t vb = sizeof(T) * static_cast<size_t>(n > 0 ? n : 1);
  DevBuf cur(vb), old(vb);
  size_t iter = 0;
  if (n > 0) {
    ops.fill(cur.as<T>(), T(1) / static_cast<T>(n));
    while (true) {
      std::swap(cur, old);
      spmv_apply(*sh.s, old.p, cur.p);
      hipLaunchKernelGGL((add_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, h.stream, cur.as<T>(),
                         old.as<T>(), n);
      T hyp = static_cast<T>(std::sqrt(ops.reduce(Red::kSumSq, cur.as<T>())));
      ops.divide(cur.as<T>(), hyp);
      double diff = ops.reduce(Red::kAbsDiff, cur.as<T>(), old.as<T>());
      ++iter;
      if (static_cast<T>(diff) < static_cast<T>(n) * epsilon) break;
      CGA_EXPECTS(iter < max_iterations, CUGRAPH_UNKNOWN_ERROR,
                  "Eigenvector Centrality failed to converge.");
    }
  }
  res->values         = values_out<T>(h, g, *sh.s, cur.as<T>());
  res->vertex_ids     = vertex_ids_out(h, g);
  res->num_iterations = iter;
  res->converged      = true;
  return res.release();
}

// ------------------------------------------------------------------ HITS
// the reversed graph (CSC of the transposed edge set, no weights: hits_impl.cuh sums plain
// values), in the vertex order of `g`; its own plan gives it its own degree-sorted copy
Graph* reversed_unweighted(Handle const& h, Graph const& g)
{
  int64_t n = g.num_vertices, ne = g.num_edges;
  DevBuf rows(static_cast<size_t>(ne > 0 ? ne : 1) * 4);
  expand_offsets(h.stream, g.offsets, g.off64, n, ne, rows.as<int32_t>());
  // g: majors = dst, minors = src.  reversed: majors = src, minors = dst.
  CompressedOut c = compress_coo(h.stream, n, n, ne, g.indices, rows.as<int32_t>(), nullptr, 0);
  auto r                 = std::make_unique<Graph>();
  r->vertex_type         = INT32;
  r->edge_type           = g.edge_type;
  r->weight_type         = g.weight_type;  // value type of the vectors
  r->store_transposed    = true;
  r->num_vertices        = n;
  r->num_global_vertices = n;
  r->num_edges           = ne;
  r->off64               = c.off64;
  r->offsets             = c.offsets.detach();
  r->indices             = static_cast<int32_t*>(c.indices.detach());
  r->degree_sorted       = degrees_descending(h.stream, r->offsets, r->off64, n);
  return r.release();
}

template <typename T>
HitsResult* hits_impl(Handle const& h, Graph& g, double epsilon_d, size_t max_iterations,
                      ArrayView const* guess_v, ArrayView const* guess_x, bool normalize,
                      bool do_expensive_check)
{
  CGA_EXPECTS(epsilon_d >= 0.0, kAlgorithmRejects,
              "Invalid input argument: epsilon should be non-negative.");
  int64_t n = g.num_vertices;
  auto res  = std::make_unique<HitsResult>();
  T epsilon = static_cast<T>(epsilon_d);
  T tol     = static_cast<T>(n) * epsilon;
  res->hub_score_differences = static_cast<double>(std::numeric_limits<T>::max());
  res->number_of_iterations  = max_iterations;
  Ops<T> ops(h.stream, n);
  size_t vb = sizeof(T) * static_cast<size_t>(n > 0 ? n : 1);
  // plain copies of both orientations without weights: hits_impl.cuh ignores edge weights
  std::unique_ptr<Graph> fwd;
  Graph* rev = nullptr;  // cached on the graph
  SessionHolder sa, sb;
  DevBuf hubs(vb), hubs_new(vb), auth(vb), tmp_in(vb), tmp_out(vb);
  if (n > 0) {
    Graph* ga = &g;
    if (g.weights) {  // same structure, weights dropped
      fwd                      = std::make_unique<Graph>();
      fwd->vertex_type         = INT32;
      fwd->edge_type           = g.edge_type;
      fwd->weight_type         = g.weight_type;
      fwd->store_transposed    = true;
      fwd->num_vertices        = n;
      fwd->num_global_vertices = n;
      fwd->num_edges           = g.num_edges;
      fwd->off64               = g.off64;
      fwd->degree_sorted       = g.degree_sorted;
      size_t ob = static_cast<size_t>(n + 1) * (g.off64 ? 8 : 4), ib = static_cast<size_t>(g.num_edges + 4) * 4;
      CGA_HIP(hipMalloc(&fwd->offsets, ob));
      CGA_HIP(hipMalloc(reinterpret_cast<void**>(&fwd->indices), ib));
      CGA_HIP(hipMemsetAsync(fwd->indices, 0, ib, h.stream));
      CGA_HIP(hipMemcpyAsync(fwd->offsets, g.offsets, ob, hipMemcpyDeviceToDevice, h.stream));
      CGA_HIP(hipMemcpyAsync(fwd->indices, g.indices, static_cast<size_t>(g.num_edges) * 4,
                             hipMemcpyDeviceToDevice, h.stream));
      ga = fwd.get();
    }
    if (!g.reversed) g.reversed = reversed_unweighted(h, g);
    rev = g.reversed;
    sa.s = spmv_session_create(h, *ga);
    sb.s = spmv_session_create(h, *rev);
    int32_t const* oa = spmv_session_order(*sa.s);
    int32_t const* ob = spmv_session_order(*sb.s);
    // y (graph order) = A x (graph order), through the session's own vertex order
    auto apply = [&](Session& s, int32_t const* order, T const* x, T* y) {
      if (!order) {
        spmv_apply(s, x, y);
        return;
      }
      ops.gather(x, order, tmp_in.as<T>());
      spmv_apply(s, tmp_in.p, tmp_out.p);
      ops.scatter(tmp_out.as<T>(), order, y);
    };
    if (guess_x) {
      CGA_EXPECTS(guess_v, CUGRAPH_INVALID_INPUT, "initial hubs guess needs vertices and values");
      CGA_HIP(hipMemsetAsync(hubs.p, 0, vb, h.stream));
      dense_from_pairs<T>(h, g, *guess_v, *guess_x, hubs.as<T>(), "initial_hubs_guess");
      if (do_expensive_check)
        CGA_EXPECTS(ops.reduce(Red::kNegCount, hubs.as<T>()) == 0.0, kAlgorithmRejects,
                    "Invalid input argument: initial guess values should be non-negative.");
      double s = ops.reduce(Red::kSum, hubs.as<T>());
      CGA_EXPECTS(s > 0.0, CUGRAPH_UNKNOWN_ERROR, "Norm is required to be a positive value.");
      ops.divide(hubs.as<T>(), static_cast<T>(s));
    } else {
      ops.fill(hubs.as<T>(), T(1) / static_cast<T>(n));
    }
    size_t iter = 0;
    double diff = 0.0;
    while (true) {
      apply(*sa.s, oa, hubs.as<T>(), auth.as<T>());          // authorities = in-sum of hubs
      apply(*sb.s, ob, auth.as<T>(), hubs_new.as<T>());      // hubs = out-sum of authorities
      double mh = ops.reduce(Red::kMax, hubs_new.as<T>());
      CGA_EXPECTS(mh > 0.0, CUGRAPH_UNKNOWN_ERROR, "Norm is required to be a positive value.");
      ops.divide(hubs_new.as<T>(), static_cast<T>(mh));
      double ma = ops.reduce(Red::kMax, auth.as<T>());
      CGA_EXPECTS(ma > 0.0, CUGRAPH_UNKNOWN_ERROR, "Norm is required to be a positive value.");
      ops.divide(auth.as<T>(), static_cast<T>(ma));
      diff = ops.reduce(Red::kAbsDiff, hubs_new.as<T>(), hubs.as<T>());
      std::swap(hubs, hubs_new);
      ++iter;
      if (static_cast<T>(diff) < tol) break;
      CGA_EXPECTS(iter < max_iterations, CUGRAPH_UNKNOWN_ERROR, "HITS failed to converge.");
    }
    if (normalize) {
      double s1 = ops.reduce(Red::kSum, hubs.as<T>());
      CGA_EXPECTS(s1 > 0.0, CUGRAPH_UNKNOWN_ERROR, "Norm is required to be a positive value.");
      ops.divide(hubs.as<T>(), static_cast<T>(s1));
      double s2 = ops.reduce(Red::kSum, auth.as<T>());
      CGA_EXPECTS(s2 > 0.0, CUGRAPH_UNKNOWN_ERROR, "Norm is required to be a positive value.");
      ops.divide(auth.as<T>(), static_cast<T>(s2));
    }
    res->hub_score_differences = static_cast<double>(static_cast<T>(diff));
    res->number_of_iterations  = iter;
  }
  auto copy_out = [&](DevBuf& b) {
    auto out    = std::make_unique<Array>();
    out->type   = g.weight_type;
    out->size   = static_cast<size_t>(n);
    out->nbytes = out->size * sizeof(T);
    if (n > 0) {
      CGA_HIP(hipMalloc(&out->data, out->nbytes));
      CGA_HIP(hipMemcpyAsync(out->data, b.p, out->nbytes, hipMemcpyDeviceToDevice, h.stream));
    }
    return out.release();
  };
  res->hubs        = copy_out(hubs);
  res->authorities = copy_out(auth);
  res->vertex_ids  = vertex_ids_out(h, g);
  CGA_HIP(hipStreamSynchronize(h.stream));
  return res.release();
}

}  // namespace

CentralityResult* run_katz(Handle const& h, Graph& g, ArrayView const* betas, double alpha, double beta,
                           double epsilon, size_t max_iterations, bool do_expensive_check)
{
  if (!g.store_transposed) transpose_graph(h, g);  // katz.cpp:77-83
  return g.weight_type == FLOAT64
           ? katz_impl<double>(h, g, betas, alpha, beta, epsilon, max_iterations, do_expensive_check)
           : katz_impl<float>(h, g, betas, alpha, beta, epsilon, max_iterations, do_expensive_check);
}

CentralityResult* run_eigenvector(Handle const& h, Graph& g, double epsilon, size_t max_iterations,
                                  bool do_expensive_check)
{
  if (!g.store_transposed) transpose_graph(h, g);  // eigenvector_centrality.cpp:68-74
  return g.weight_type == FLOAT64
           ? eigenvector_impl<double>(h, g, epsilon, max_iterations, do_expensive_check)
           : eigenvector_impl<float>(h, g, epsilon, max_iterations, do_expensive_check);
}

HitsResult* run_hits(Handle const& h, Graph& g, double epsilon, size_t max_iterations,
                     ArrayView const* guess_v, ArrayView const* guess_x, bool normalize,
                     bool do_expensive_check)
{
  if (!g.store_transposed) transpose_graph(h, g);  // hits.cpp: transposes like the others
  return g.weight_type == FLOAT64
           ? hits_impl<double>(h, g, epsilon, max_iterations, guess_v, guess_x, normalize, do_expensive_check)
           : hits_impl<float>(h, g, epsilon, max_iterations, guess_v, guess_x, normalize, do_expensive_check);
}

}  // namespace cga
