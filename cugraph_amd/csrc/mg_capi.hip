// Multi-GPU PageRank below the C ABI, under the reference's own names: cugraph_mg_graph_create /
// cugraph_graph_create_mg (cpp/include/cugraph_c/graph.h:242, 299; cpp/src/c_api/graph_mg.cpp) and
// cugraph_pagerank* on the graph they return (cpp/src/c_api/pagerank.cpp:98-249, which shuffles the
// (vertex, value) lists to their owners at :140-144), as cpp/tests/c_api/mg_pagerank_test.c:67-94
// uses them: every rank passes ANY share of the edge list and gets the ranks of the vertices it owns.
//
// One process per GPU.  The library stays free of a communication dependency: the caller hands the
// three collectives it needs to the handle (cugraph_amd_resource_handle_set_comm: all-reduce of
// doubles, variable all-gather and variable all-to-all of bytes, on device memory, ordered on the
// handle's stream) -- RCCL in production, anything in a test.  The layout is the 1D destination
// partition of cugraph_amd/mg.py in its plainest form: equal vertex ranges, every rank's rows in
// descending in-degree order inside its range (ids never leave their owner's range), sources named by
// vertex id, one all-gather of the source vector and one all-reduce of two scalars per iteration.
// The tuned driver (edge-balanced ranges, compact / private slots, the all-to-all exchange) is mg.py.
#include "common.hpp"

#include <algorithm>
#include <vector>

namespace cga {
struct Session;
Session* session_create(Handle const& h, Graph& g, PagerankArgs const& a);
void session_iterate(Session& s, size_t n);
void session_status(Session& s, size_t* iters, bool* stopped, double* delta, double* dangling);
CentralityResult* session_result(Session& s, size_t max_iterations);
void session_exchange_buffers(Session& s, void** x_full, void** scalars);
void session_destroy(Session* s);
bool session_is_f64(Session& s);
void edgelist_source_sums(hipStream_t stream, ArrayView const& src, ArrayView const* weights,
                          int64_t num_vertices, double* sums);
// capi.hip: the local CSC of a rank's destination range (what cugraph_amd_graph_create_local builds)
Graph* make_partition_graph(Handle const& h, cugraph_graph_properties_t const* props, ArrayView const& src,
                            ArrayView const& dst, ArrayView const* weights, int64_t vertex_first,
                            int64_t vertex_last, int64_t num_global);

struct MgInfo {
  int world{1}, rank{0};
  cugraph_data_type_id_t api_vertex_type{INT32};
  std::vector<int64_t> bounds;  // world + 1: rank r owns ids [bounds[r], bounds[r + 1])
  DevBuf order;                 // int32[n_local]: row (new position in the range) -> old id - first
  DevBuf table;                 // int32[V]: old id -> new id
  DevBuf out_sums;              // double[n_local]: out-weight sum of every own vertex, by row
};

void destroy_mg_info(MgInfo* m) { delete m; }

namespace {

constexpr int kThreads = 256;
inline int grid_for(int64_t n)
{
  int64_t b = (n + kThreads - 1) / kThreads;
  return static_cast<int>(b < 1 ? 1 : b > 8192 ? 8192 : b);
}
#define MG_STRIDE(i, n) \
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void add_first_kernel(int32_t const* __restrict__ rank_local, int64_t n, int32_t first, int32_t* __restrict__ out)
{
  MG_STRIDE(i, n) out[i] = first + rank_local[i];
}
__global__ void relabel_kernel(int32_t* __restrict__ ids, int64_t n, int32_t const* __restrict__ table)
{
  MG_STRIDE(i, n) ids[i] = table[ids[i]];
}
__global__ void iota_kernel(int32_t* __restrict__ out, int64_t n, int32_t first)
{
  MG_STRIDE(i, n) out[i] = first + static_cast<int32_t>(i);
}
template <typename T>
__global__ void from_double_kernel(double const* __restrict__ in, int64_t n, T* __restrict__ out)
{
  MG_STRIDE(i, n) out[i] = static_cast<T>(in[i]);
}
template <typename T>
__global__ void to_double_kernel(T const* __restrict__ in, int64_t n, double* __restrict__ out)
{
  MG_STRIDE(i, n) out[i] = static_cast<double>(in[i]);
}
// the pairs of a replicated (vertex, value) list this rank owns, renumbered, values divided by `scale`
template <typename T>
__global__ void own_pairs_kernel(int32_t const* __restrict__ v, double const* __restrict__ x, int64_t n,
                                 int32_t first, int32_t last, int32_t const* __restrict__ table, double scale,
                                 int32_t* __restrict__ v_out, T* __restrict__ x_out, unsigned long long* __restrict__ count)
{
  MG_STRIDE(i, n)
  {
    int32_t const id = v[i];
    if (id >= first && id < last) {
      unsigned long long const k = atomicAdd(count, 1ull);
      v_out[k] = table[id];
      x_out[k] = static_cast<T>(x[i] / scale);
    }
  }
}
template <typename IdT>
__global__ void result_ids_kernel(int32_t const* __restrict__ new_ids, int64_t n, int32_t first,
                                  int32_t const* __restrict__ order, IdT* __restrict__ out)
{
  MG_STRIDE(i, n) out[i] = static_cast<IdT>(first + order[new_ids[i] - first]);
}

// the caller's collectives, with the checks and the degenerate world of one rank in one place
struct Comm {
  Handle const& h;
  explicit Comm(Handle const& handle) : h(handle)
  {
    CGA_EXPECTS(h.comm_size == 1 || (h.has_comm && h.comm.all_reduce_f64 && h.comm.all_gather_v && h.comm.all_to_all_v),
                CUGRAPH_INVALID_HANDLE,
                "a multi-GPU call needs the collectives of its handle: cugraph_amd_resource_handle_set_comm");
  }
  int world() const { return h.comm_size; }
  void all_reduce(double* dev, size_t n, int op) const
  {
    if (world() == 1 || n == 0) return;
    CGA_EXPECTS(h.comm.all_reduce_f64(h.comm.context, dev, n, op, h.stream) == 0, CUGRAPH_UNKNOWN_ERROR,
                "the all_reduce_f64 callback of the handle failed");
  }
  // rank r contributes bytes[r] bytes, stored in rank order; `send` may be this rank's own place in `recv`
  void all_gather_v(void const* send, void* recv, std::vector<size_t> const& bytes) const
  {
    if (world() == 1) {
      if (send != recv && bytes[0]) CGA_HIP(hipMemcpyAsync(recv, send, bytes[0], hipMemcpyDeviceToDevice, h.stream));
      return;
    }
    CGA_EXPECTS(h.comm.all_gather_v(h.comm.context, send, recv, bytes.data(), h.stream) == 0, CUGRAPH_UNKNOWN_ERROR,
                "the all_gather_v callback of the handle failed");
  }
  void all_to_all_v(void const* send, std::vector<size_t> const& send_bytes, void* recv,
                    std::vector<size_t> const& recv_bytes) const
  {
    if (world() == 1) {
      if (send_bytes[0]) CGA_HIP(hipMemcpyAsync(recv, send, send_bytes[0], hipMemcpyDeviceToDevice, h.stream));
      return;
    }
    CGA_EXPECTS(h.comm.all_to_all_v(h.comm.context, send, send_bytes.data(), recv, recv_bytes.data(), h.stream) == 0,
                CUGRAPH_UNKNOWN_ERROR, "the all_to_all_v callback of the handle failed");
  }
  // one double per rank -> the reduced value on the host
  double reduce_scalar(double mine, int op) const
  {
    if (world() == 1) return mine;
    DevBuf d(8);
    CGA_HIP(hipMemcpyAsync(d.p, &mine, 8, hipMemcpyHostToDevice, h.stream));
    all_reduce(d.as<double>(), 1, op);
    double out = 0.0;
    CGA_HIP(hipMemcpyAsync(&out, d.p, 8, hipMemcpyDeviceToHost, h.stream));
    CGA_HIP(hipStreamSynchronize(h.stream));
    return out;
  }
  // `mine` (k int64 per rank) -> all ranks' rows on the host, [world][k]
  std::vector<int64_t> gather_rows(std::vector<int64_t> const& mine) const
  {
    size_t const k = mine.size();
    std::vector<int64_t> all(static_cast<size_t>(world()) * k);
    if (world() == 1) {
      all = mine;
      return all;
    }
    DevBuf d(all.size() * 8);
    CGA_HIP(hipMemcpyAsync(d.as<int64_t>() + static_cast<size_t>(h.rank) * k, mine.data(), k * 8, hipMemcpyHostToDevice, h.stream));
    all_gather_v(d.as<int64_t>() + static_cast<size_t>(h.rank) * k, d.p, std::vector<size_t>(world(), k * 8));
    CGA_HIP(hipMemcpyAsync(all.data(), d.p, all.size() * 8, hipMemcpyDeviceToHost, h.stream));
    CGA_HIP(hipStreamSynchronize(h.stream));
    return all;
  }
};

// the arrays of a list, narrowed to int32 and laid end to end; returns {min, max} of the ids (max < min: none)
std::pair<int64_t, int64_t> concat_ids(hipStream_t stream, std::vector<ArrayView const*> const& parts, DevBuf& out, int64_t total)
{
  out.alloc(static_cast<size_t>(std::max<int64_t>(total, 1)) * 4);
  int64_t lo = INT64_MAX, hi = INT64_MIN, at = 0;
  for (auto const* p : parts) {
    if (!p || p->size == 0) continue;
    int64_t mn      = 0;
    int64_t const mx = narrow_ids(stream, p->data, p->type, p->size, out.as<int32_t>() + at, &mn);
    lo = std::min(lo, mn);
    hi = std::max(hi, mx);
    at += static_cast<int64_t>(p->size);
  }
  return {lo, hi};
}

}  // namespace

// src / dst / weights: `n` arrays each (weights may be nullptr: unweighted), any share of the edge list
Graph* mg_create_graph(Handle const& h, cugraph_graph_properties_t const* props, size_t n,
                       ArrayView const* const* vertices, ArrayView const* const* src, ArrayView const* const* dst,
                       ArrayView const* const* weights, bool drop_self_loops, bool drop_multi_edges)
{
  Comm comm(h);
  hipStream_t const stream = h.stream;
  CGA_EXPECTS(src && dst, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src and dst are required.");
  CGA_EXPECTS(!drop_self_loops && !drop_multi_edges, CUGRAPH_NOT_IMPLEMENTED,
              "drop_self_loops / drop_multi_edges on a partitioned graph");
  std::vector<ArrayView const*> vs, ss, ds, ws;
  int64_t ne = 0, nverts = 0;
  cugraph_data_type_id_t vtype = INT32, wtype = FLOAT32;
  bool have_type = false, have_w = weights != nullptr;
  for (size_t i = 0; i < n; ++i) {
    CGA_EXPECTS(src[i] && dst[i], CUGRAPH_INVALID_INPUT, "Invalid input arguments: src and dst are required.");
    CGA_EXPECTS(src[i]->size == dst[i]->size, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src size != dst size.");
    CGA_EXPECTS(src[i]->type == dst[i]->type, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src type != dst type.");
    CGA_EXPECTS(src[i]->type == INT32 || src[i]->type == INT64, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
                "vertex ids must be INT32 or INT64");
    CGA_EXPECTS(!have_type || src[i]->type == vtype, CUGRAPH_INVALID_INPUT, "Invalid input arguments: all arrays of a list must have one type.");
    vtype     = src[i]->type;
    have_type = true;
    if (have_w && weights[i]) {
      CGA_EXPECTS(weights[i]->size == src[i]->size, CUGRAPH_INVALID_INPUT,
                  "Invalid input arguments: src size != weights size.");
      CGA_EXPECTS(weights[i]->type == FLOAT32 || weights[i]->type == FLOAT64, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
                  "edge weights must be FLOAT32 or FLOAT64");
      CGA_EXPECTS(ws.empty() || weights[i]->type == wtype, CUGRAPH_INVALID_INPUT, "Invalid input arguments: all arrays of a list must have one type.");
      wtype = weights[i]->type;
      ws.push_back(weights[i]);
    } else if (have_w) {  // graph.h:221-228: a rank without edges may pass a NULL weights pointer
      CGA_EXPECTS(src[i]->size == 0, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src size != weights size.");
    }
    ss.push_back(src[i]);
    ds.push_back(dst[i]);
    ne += static_cast<int64_t>(src[i]->size);
    if (vertices && vertices[i]) {
      CGA_EXPECTS(vertices[i]->type == vtype, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src type != vertices type.");
      vs.push_back(vertices[i]);
      nverts += static_cast<int64_t>(vertices[i]->size);
    }
  }
  CGA_EXPECTS(ne < 0x7fffffffLL * 2, CUGRAPH_INVALID_INPUT, "too many edges in one rank's share");
  // Whether the graph is weighted, and in which type, is a property of all ranks together: a rank without
  // edges may have passed no weights at all.  Agreed here, so that every rank makes the same collectives (and
  // a disagreement fails on every rank instead of leaving the others inside one).
  {
    double const mine   = ws.empty() ? 0.0 : wtype == FLOAT64 ? 2.0 : 1.0;
    double const agreed = comm.reduce_scalar(mine, 1);
    double const bad    = comm.reduce_scalar((mine != 0.0 && mine != agreed) || (mine == 0.0 && agreed != 0.0 && ne > 0) ? 1.0 : 0.0, 1);
    CGA_EXPECTS(bad == 0.0, CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: the ranks disagree on the edge weights (present on some, or of two types).");
    have_w = agreed != 0.0;
    wtype  = agreed == 2.0 ? FLOAT64 : FLOAT32;
  }
  size_t const wsz = have_w ? dtype_size(wtype) : 0;

  // ---- this rank's share as int32 ids (+ weights), end to end; the id range over all ranks
  DevBuf s32, d32, v32, w_all;
  auto const rs = concat_ids(stream, ss, s32, ne);
  auto const rd = concat_ids(stream, ds, d32, ne);
  auto const rv = concat_ids(stream, vs, v32, nverts);
  v32.release();
  if (have_w) {
    w_all.alloc(static_cast<size_t>(std::max<int64_t>(ne, 1)) * wsz);
    size_t at = 0;
    for (auto const* w : ws) {
      if (w->size) CGA_HIP(hipMemcpyAsync(static_cast<char*>(w_all.p) + at, w->data, w->size * wsz, hipMemcpyDeviceToDevice, stream));
      at += w->size * wsz;
    }
  }
  int64_t const lo = std::min({rs.first, rd.first, rv.first}), hi = std::max({rs.second, rd.second, rv.second});
  // (the reference renumbers arbitrary ids across ranks; here the ids ARE the vertex numbers, as in its own tests)
  double const g_lo = -comm.reduce_scalar(hi >= lo ? -static_cast<double>(lo) : -4.0e18, 1);
  double const g_hi = comm.reduce_scalar(hi >= lo ? static_cast<double>(hi) : -1.0, 1);
  CGA_EXPECTS(g_hi < 0 || g_lo >= 0, CUGRAPH_INVALID_INPUT, "vertex ids of a multi-GPU graph must be non-negative");
  CGA_EXPECTS(g_hi < 2147483646.0, CUGRAPH_INVALID_INPUT, "vertex ids of a multi-GPU graph must be below 2^31 - 2");
  int64_t const nv = static_cast<int64_t>(g_hi) + 1;

  auto info   = std::make_unique<MgInfo>();
  info->world = comm.world();
  info->rank  = h.rank;
  info->api_vertex_type = vtype;
  int const P = comm.world();
  int64_t const chunk = (nv + P - 1) / P;
  info->bounds.resize(P + 1);
  for (int r = 0; r <= P; ++r) info->bounds[r] = std::min<int64_t>(nv, chunk * r);
  int64_t const first = info->bounds[h.rank], last = info->bounds[h.rank + 1], n_local = last - first;

  // ---- edges to the owner of their destination
  if (P > 1) {
    DevBuf s_g(s32.bytes), d_g(d32.bytes), w_g(have_w ? w_all.bytes : 0);
    std::vector<int64_t> counts(P, 0);
    ArrayView sv{s32.p, static_cast<size_t>(ne), static_cast<size_t>(ne) * 4, INT32};
    ArrayView dv{d32.p, static_cast<size_t>(ne), static_cast<size_t>(ne) * 4, INT32};
    ArrayView wv{w_all.p, static_cast<size_t>(ne), static_cast<size_t>(ne) * wsz, wtype};
    if (ne > 0)
      partition_group_edges(stream, sv, dv, have_w ? &wv : nullptr, info->bounds.data(), P, s_g.p, d_g.p,
                            have_w ? w_g.p : nullptr, counts.data());
    std::vector<int64_t> const all = comm.gather_rows(counts);  // all[r * P + q]: edges rank r has for rank q
    std::vector<size_t> send4(P), recv4(P), sendw(P), recvw(P);
    int64_t ne_in = 0;
    for (int r = 0; r < P; ++r) {
      send4[r] = static_cast<size_t>(counts[r]) * 4;
      recv4[r] = static_cast<size_t>(all[static_cast<size_t>(r) * P + h.rank]) * 4;
      sendw[r] = static_cast<size_t>(counts[r]) * wsz;
      recvw[r] = static_cast<size_t>(all[static_cast<size_t>(r) * P + h.rank]) * wsz;
      ne_in += all[static_cast<size_t>(r) * P + h.rank];
    }
    CGA_EXPECTS(ne_in < 0x7fffffffLL * 2, CUGRAPH_INVALID_INPUT, "too many edges in one rank's partition");
    s32.alloc(static_cast<size_t>(std::max<int64_t>(ne_in, 1)) * 4);
    d32.alloc(static_cast<size_t>(std::max<int64_t>(ne_in, 1)) * 4);
    comm.all_to_all_v(s_g.p, send4, s32.p, recv4);
    comm.all_to_all_v(d_g.p, send4, d32.p, recv4);
    if (have_w) {
      w_all.alloc(static_cast<size_t>(std::max<int64_t>(ne_in, 1)) * wsz);
      comm.all_to_all_v(w_g.p, sendw, w_all.p, recvw);
    }
    CGA_HIP(hipStreamSynchronize(stream));  // the send buffers go out of scope
    ne = ne_in;
  }

  // ---- rows in descending in-degree order inside the own range; the old -> new table of every range
  ArrayView dv{d32.p, static_cast<size_t>(ne), static_cast<size_t>(ne) * 4, INT32};
  info->order.alloc(static_cast<size_t>(std::max<int64_t>(n_local, 1)) * 4);
  DevBuf rank_local(static_cast<size_t>(std::max<int64_t>(n_local, 1)) * 4);
  partition_degree_order(stream, dv, first, n_local, info->order.as<int32_t>(), rank_local.as<int32_t>());
  info->table.alloc(static_cast<size_t>(std::max<int64_t>(nv, 1)) * 4);
  if (n_local > 0)
    hipLaunchKernelGGL(add_first_kernel, dim3(grid_for(n_local)), dim3(kThreads), 0, stream, rank_local.as<int32_t>(),
                       n_local, static_cast<int32_t>(first), info->table.as<int32_t>() + first);
  {
    std::vector<size_t> bytes(P);
    for (int r = 0; r < P; ++r) bytes[r] = static_cast<size_t>(info->bounds[r + 1] - info->bounds[r]) * 4;
    comm.all_gather_v(info->table.as<int32_t>() + first, info->table.p, bytes);
  }
  if (ne > 0) {
    hipLaunchKernelGGL(relabel_kernel, dim3(grid_for(ne)), dim3(kThreads), 0, stream, s32.as<int32_t>(), ne, info->table.as<int32_t>());
    hipLaunchKernelGGL(relabel_kernel, dim3(grid_for(ne)), dim3(kThreads), 0, stream, d32.as<int32_t>(), ne, info->table.as<int32_t>());
  }
  CGA_HIP(hipGetLastError());

  // ---- out-weight sums: every rank over the sources of ITS edges, summed over the ranks, own rows kept
  ArrayView sv{s32.p, static_cast<size_t>(ne), static_cast<size_t>(ne) * 4, INT32};
  ArrayView wv{w_all.p, static_cast<size_t>(ne), static_cast<size_t>(ne) * wsz, wtype};
  {
    DevBuf sums(static_cast<size_t>(std::max<int64_t>(nv, 1)) * 8);
    edgelist_source_sums(stream, sv, have_w ? &wv : nullptr, nv, sums.as<double>());
    comm.all_reduce(sums.as<double>(), static_cast<size_t>(nv), 0);
    info->out_sums.alloc(static_cast<size_t>(std::max<int64_t>(n_local, 1)) * 8);
    if (n_local > 0)  // new ids first .. last - 1 are the rows in order
      CGA_HIP(hipMemcpyAsync(info->out_sums.p, sums.as<double>() + first, static_cast<size_t>(n_local) * 8,
                             hipMemcpyDeviceToDevice, stream));
    CGA_HIP(hipStreamSynchronize(stream));
  }

  Graph* g = make_partition_graph(h, props, sv, dv, have_w ? &wv : nullptr, first, last, nv);
  g->mg    = info.release();
  return g;
}

namespace {

// a (vertex, value) list given in shares -> the pairs this rank owns, renumbered (pagerank.cpp:140-144)
template <typename T>
int64_t own_pairs(Comm const& comm, Graph const& g, ArrayView const* v, ArrayView const* x, bool normalise,
                  DevBuf& v_out, DevBuf& x_out)
{
  MgInfo const& m          = *g.mg;
  Handle const& h          = comm.h;
  hipStream_t const stream = h.stream;
  int64_t const mine = v ? static_cast<int64_t>(v->size) : 0;
  CGA_EXPECTS(!v || (x && x->size == v->size), CUGRAPH_INVALID_INPUT, "vertices and values of a list must have one length");
  std::vector<int64_t> const sizes = comm.gather_rows({mine});
  int64_t total = 0, before = 0;
  for (int r = 0; r < m.world; ++r) {
    if (r < m.rank) before += sizes[r];
    total += sizes[r];
  }
  v_out.alloc(static_cast<size_t>(std::max<int64_t>(total, 1)) * 4);
  x_out.alloc(static_cast<size_t>(std::max<int64_t>(total, 1)) * sizeof(T));
  if (total == 0) return 0;
  DevBuf all_v(static_cast<size_t>(total) * 4), all_x(static_cast<size_t>(total) * 8);
  if (mine > 0) {
    int64_t mn = 0;
    int64_t const mx = narrow_ids(stream, v->data, v->type, v->size, all_v.as<int32_t>() + before, &mn);
    CGA_EXPECTS(mn >= 0 && mx < g.num_global_vertices, CUGRAPH_INVALID_INPUT, "a vertex of a (vertex, value) list is not in the graph");
    if (x->type == FLOAT64)
      hipLaunchKernelGGL((to_double_kernel<double>), dim3(grid_for(mine)), dim3(kThreads), 0, stream,
                         static_cast<double const*>(x->data), mine, all_x.as<double>() + before);
    else
      hipLaunchKernelGGL((to_double_kernel<float>), dim3(grid_for(mine)), dim3(kThreads), 0, stream,
                         static_cast<float const*>(x->data), mine, all_x.as<double>() + before);
  }
  std::vector<size_t> b4(m.world), b8(m.world);
  for (int r = 0; r < m.world; ++r) {
    b4[r] = static_cast<size_t>(sizes[r]) * 4;
    b8[r] = static_cast<size_t>(sizes[r]) * 8;
  }
  comm.all_gather_v(all_v.as<int32_t>() + before, all_v.p, b4);
  comm.all_gather_v(all_x.as<double>() + before, all_x.p, b8);
  double scale = 1.0;
  if (normalise) {  // the library normalises only what it sees (pagerank_impl.cuh:213-226): divide by the global sum here
    std::vector<double> hx(static_cast<size_t>(total));
    CGA_HIP(hipMemcpyAsync(hx.data(), all_x.p, hx.size() * 8, hipMemcpyDeviceToHost, stream));
    CGA_HIP(hipStreamSynchronize(stream));
    scale = 0.0;
    for (double t : hx) scale += t;
    CGA_EXPECTS(scale > 0.0, CUGRAPH_INVALID_INPUT, "Invalid input argument: sum of personalization values should be positive.");
  }
  DevBuf count(8);
  CGA_HIP(hipMemsetAsync(count.p, 0, 8, stream));
  hipLaunchKernelGGL((own_pairs_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0, stream, all_v.as<int32_t>(),
                     all_x.as<double>(), total, static_cast<int32_t>(m.bounds[m.rank]), static_cast<int32_t>(m.bounds[m.rank + 1]),
                     m.table.as<int32_t>(), scale, v_out.as<int32_t>(), x_out.as<T>(), count.as<unsigned long long>());
  unsigned long long kept = 0;
  CGA_HIP(hipMemcpyAsync(&kept, count.p, 8, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  return static_cast<int64_t>(kept);
}

template <typename T>
CentralityResult* mg_pagerank_typed(Handle const& h, Graph& g, PagerankArgs const& a)
{
  Comm comm(h);
  MgInfo const& m          = *g.mg;
  hipStream_t const stream = h.stream;
  cugraph_data_type_id_t const xt = sizeof(T) == 8 ? FLOAT64 : FLOAT32;
  int64_t const first = m.bounds[m.rank], n_local = m.bounds[m.rank + 1] - first;
  CGA_EXPECTS(m.world == comm.world() && m.rank == h.rank, CUGRAPH_INVALID_HANDLE,
              "the handle is not the one (rank, size) this multi-GPU graph was made with");
  CGA_EXPECTS(!a.pre_vertices && !a.pre_sums, CUGRAPH_NOT_IMPLEMENTED,
              "precomputed out-weight sums on a multi-GPU graph (the graph keeps its own)");

  DevBuf ids(static_cast<size_t>(std::max<int64_t>(n_local, 1)) * 4), sums(static_cast<size_t>(std::max<int64_t>(n_local, 1)) * sizeof(T));
  if (n_local > 0) {
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n_local)), dim3(kThreads), 0, stream, ids.as<int32_t>(), n_local,
                       static_cast<int32_t>(first));
    hipLaunchKernelGGL((from_double_kernel<T>), dim3(grid_for(n_local)), dim3(kThreads), 0, stream,
                       m.out_sums.as<double>(), n_local, sums.as<T>());
  }
  ArrayView pre_v{ids.p, static_cast<size_t>(n_local), static_cast<size_t>(n_local) * 4, INT32};
  ArrayView pre_s{sums.p, static_cast<size_t>(n_local), static_cast<size_t>(n_local) * sizeof(T), xt};
  DevBuf gv, gx, pv, px;
  ArrayView guess_v, guess_x, pers_v, pers_x;
  PagerankArgs b = a;
  b.pre_vertices = &pre_v;
  b.pre_sums     = &pre_s;
  // (every rank makes these calls, with an empty share if it has nothing: they are collectives)
  bool const any_guess = comm.reduce_scalar(a.guess_vertices ? 1.0 : 0.0, 1) > 0.0;
  bool const any_pers  = comm.reduce_scalar(a.pers_vertices ? 1.0 : 0.0, 1) > 0.0;
  if (any_guess) {
    int64_t const k = own_pairs<T>(comm, g, a.guess_vertices, a.guess_values, false, gv, gx);
    guess_v = ArrayView{gv.p, static_cast<size_t>(k), static_cast<size_t>(k) * 4, INT32};
    guess_x = ArrayView{gx.p, static_cast<size_t>(k), static_cast<size_t>(k) * sizeof(T), xt};
    b.guess_vertices = &guess_v;
    b.guess_values   = &guess_x;
  }
  if (any_pers) {
    int64_t const k = own_pairs<T>(comm, g, a.pers_vertices, a.pers_values, true, pv, px);
    pers_v = ArrayView{pv.p, static_cast<size_t>(k), static_cast<size_t>(k) * 4, INT32};
    pers_x = ArrayView{px.p, static_cast<size_t>(k), static_cast<size_t>(k) * sizeof(T), xt};
    b.pers_vertices = &pers_v;
    b.pers_values   = &pers_x;
  }

  Session* s = session_create(h, g, b);
  CentralityResult* res = nullptr;
  try {
    std::vector<size_t> bytes(m.world);
    for (int r = 0; r < m.world; ++r) bytes[r] = static_cast<size_t>(m.bounds[r + 1] - m.bounds[r]) * sizeof(T);
    DevBuf tmp(16);
    // after every iteration (and once before the first): this rank's {dangling mass, L1 delta} summed over the
    // ranks into what the next iteration's kernels read (scal[2..3] stay its own, so the sum is idempotent after
    // the stop), and the source vector all-gathered in place (update_edge_src_dst_property.cuh:559-588, reduce_v.cuh:141-144)
    auto exchange = [&] {
      if (m.world == 1) return;
      void *x = nullptr, *sc = nullptr;  // (the source vector is a ping-pong pair: asked for after every iteration)
      session_exchange_buffers(*s, &x, &sc);
      CGA_HIP(hipMemcpyAsync(tmp.p, static_cast<double*>(sc) + 2, 16, hipMemcpyDeviceToDevice, stream));
      comm.all_reduce(tmp.as<double>(), 2, 0);
      CGA_HIP(hipMemcpyAsync(sc, tmp.p, 16, hipMemcpyDeviceToDevice, stream));
      comm.all_gather_v(static_cast<T*>(x) + first, x, bytes);
    };
    exchange();
    size_t const budget = std::max<size_t>(a.max_iterations, 1);
    size_t done = 0;
    while (done < budget) {  // the loop body of the reference runs at least once (pagerank_impl.cuh:235-335)
      size_t const n = std::min<size_t>(4, budget - done);
      for (size_t i = 0; i < n; ++i) {
        session_iterate(*s, 1);
        exchange();
      }
      done += n;
      size_t iters = 0;
      bool stopped = false;
      session_status(*s, &iters, &stopped, nullptr, nullptr);
      if (stopped || g.num_global_vertices == 0) break;
    }
    res = session_result(*s, a.max_iterations);
    // vertices in the caller's ids and type: new id -> old id through the row order
    size_t const k = res->vertex_ids ? res->vertex_ids->size : 0;
    auto out   = std::make_unique<Array>();
    out->size  = k;
    out->type  = m.api_vertex_type;
    out->nbytes = k * dtype_size(m.api_vertex_type);
    if (k) {
      CGA_EXPECTS(res->vertex_ids->type == INT32, CUGRAPH_UNKNOWN_ERROR, "internal: local vertex ids are not INT32");
      CGA_HIP(hipMalloc(&out->data, out->nbytes));
      if (m.api_vertex_type == INT64)
        hipLaunchKernelGGL((result_ids_kernel<int64_t>), dim3(grid_for(k)), dim3(kThreads), 0, stream,
                           static_cast<int32_t const*>(res->vertex_ids->data), static_cast<int64_t>(k),
                           static_cast<int32_t>(first), m.order.as<int32_t>(), static_cast<int64_t*>(out->data));
      else
        hipLaunchKernelGGL((result_ids_kernel<int32_t>), dim3(grid_for(k)), dim3(kThreads), 0, stream,
                           static_cast<int32_t const*>(res->vertex_ids->data), static_cast<int64_t>(k),
                           static_cast<int32_t>(first), m.order.as<int32_t>(), static_cast<int32_t*>(out->data));
      CGA_HIP(hipGetLastError());
      CGA_HIP(hipStreamSynchronize(stream));
    }
    delete res->vertex_ids;
    res->vertex_ids = out.release();
  } catch (...) {
    delete res;
    session_destroy(s);
    throw;
  }
  session_destroy(s);
  return res;
}

}  // namespace

CentralityResult* mg_run_pagerank(Handle const& h, Graph& g, PagerankArgs const& a)
{
  CGA_EXPECTS(g.mg != nullptr, CUGRAPH_INVALID_INPUT, "not a multi-GPU graph");
  if (g.weights && g.weight_type == FLOAT64) return mg_pagerank_typed<double>(h, g, a);
  return mg_pagerank_typed<float>(h, g, a);
}

cugraph_data_type_id_t mg_api_vertex_type(Graph const& g) { return g.mg ? g.mg->api_vertex_type : g.vertex_type; }

}  // namespace cga
