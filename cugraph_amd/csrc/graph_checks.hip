// do_expensive_check of the graph build (reference:
// cpp/src/structure/create_graph_from_edgelist_impl.cuh:72-126 expensive_check_edgelist,
// :243-280 check_symmetric, :282-300 check_no_parallel_edge, called from :895-921).
//
// The reference sorts copies of the edge list; here the checks run on what the build has just
// produced -- a compressed graph whose minors ascend inside every major -- so "parallel edge" is
// "two equal neighbours in a column" and "symmetric" is "(m, n) is stored as often as (n, m)",
// two binary searches per edge.  Failures are reported the way the reference's C API reports a
// cugraph::logic_error from the build: CUGRAPH_UNKNOWN_ERROR with the reference's message
// (cpp/src/c_api/graph_sg.cpp:627-629).
#include "common.hpp"

namespace cga {

namespace {

constexpr int kThreads = 256;

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

// what[0]: ids outside [lo, lo + range); flags[id - lo] = 1 for the others
template <typename IdT>
__global__ void mark_in_range_kernel(IdT const* __restrict__ ids, int64_t n, long long lo, long long range,
                                     unsigned int* __restrict__ flags, unsigned int* __restrict__ what)
{
  GRID_STRIDE(i, n)
  {
    long long k = static_cast<long long>(ids[i]) - lo;
    if (k < 0 || k >= range)
      atomicAdd(&what[0], 1u);
    else
      flags[k] = 1u;
  }
}

__global__ void count_set_kernel(unsigned int const* __restrict__ flags, int64_t n, unsigned long long* out)
{
  unsigned long long c = 0;
  GRID_STRIDE(i, n) c += flags[i] ? 1ull : 0ull;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

template <typename IdT>
__global__ void minmax_ids_kernel(IdT const* __restrict__ in, int64_t n, long long* minmax)
{
  long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
  GRID_STRIDE(i, n)
  {
    long long v = static_cast<long long>(in[i]);
    lo          = v < lo ? v : lo;
    hi          = v > hi ? v : hi;
  }
  for (int o = 32; o > 0; o >>= 1) {
    long long l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&minmax[0], lo);
    atomicMax(&minmax[1], hi);
  }
}

// copies of minor `key` in [beg, end) of a column whose minors ascend
template <typename OffT>
__device__ inline int64_t copies_of(int32_t const* __restrict__ indices, OffT beg, OffT end, int32_t key)
{
  OffT lo = beg, hi = end;  // first position with indices >= key
  while (lo < hi) {
    OffT mid = lo + (hi - lo) / 2;
    if (indices[mid] < key) lo = mid + 1; else hi = mid;
  }
  OffT first = lo;
  hi         = end;  // first position with indices > key
  while (lo < hi) {
    OffT mid = lo + (hi - lo) / 2;
    if (indices[mid] <= key) lo = mid + 1; else hi = mid;
  }
  return static_cast<int64_t>(lo - first);
}

// what[0]: stored edges equal to their predecessor in the same column (parallel edges);
// what[1]: stored edges (m, n) whose reverse (n, m) is stored a different number of times
template <typename OffT>
__global__ void edge_checks_kernel(OffT const* __restrict__ offsets, int32_t const* __restrict__ indices,
                                   int32_t const* __restrict__ majors, int64_t n_edges, int check_parallel,
                                   int check_symmetric, unsigned int* __restrict__ what)
{
  unsigned int par = 0, asym = 0;
  GRID_STRIDE(i, n_edges)
  {
    int32_t const m = majors[i], n = indices[i];
    OffT const beg = offsets[m], end = offsets[m + 1];
    if (check_parallel && static_cast<OffT>(i) > beg && indices[i - 1] == n) ++par;
    if (check_symmetric && m != n) {
      int64_t const here  = copies_of(indices, beg, end, n);
      int64_t const there = copies_of(indices, offsets[n], offsets[n + 1], m);
      if (here != there) ++asym;
    }
  }
  if (par) atomicAdd(&what[0], par);
  if (asym) atomicAdd(&what[1], asym);
}

}  // namespace

// the caller's vertex list (create_graph_from_edgelist_impl.cuh:79-108)
void expensive_check_vertices(Handle const& h, ArrayView const& vertices, bool renumber)
{
  hipStream_t stream = h.stream;
  int64_t const n    = static_cast<int64_t>(vertices.size);
  if (n == 0) return;
  bool const wide = vertices.type == INT64;
  DevBuf mm(2 * sizeof(long long));
  long long init[2] = {0x7fffffffffffffffLL, -0x7fffffffffffffffLL - 1};
  CGA_HIP(hipMemcpyAsync(mm.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
  if (wide)
    hipLaunchKernelGGL((minmax_ids_kernel<int64_t>), dim3(grid_for(n, 1024)), dim3(kThreads), 0, stream,
                       static_cast<int64_t const*>(vertices.data), n, mm.as<long long>());
  else
    hipLaunchKernelGGL((minmax_ids_kernel<int32_t>), dim3(grid_for(n, 1024)), dim3(kThreads), 0, stream,
                       static_cast<int32_t const*>(vertices.data), n, mm.as<long long>());
  long long h_mm[2];
  CGA_HIP(hipMemcpyAsync(h_mm, mm.p, sizeof(h_mm), hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  // (unsigned: 64-bit ids can span more than the signed range; such a list goes on as "wide")
  unsigned long long const span = static_cast<unsigned long long>(h_mm[1]) - static_cast<unsigned long long>(h_mm[0]);
  long long const lo = h_mm[0], range = span < 0x7ffffffeull ? static_cast<long long>(span) + 1 : 0x7fffffffLL;
  // n ids over fewer than n distinct values: duplicates, whatever the values are
  bool duplicates = range < n;
  if (!duplicates && range < 0x7fffffffLL) {
    DevBuf flags(static_cast<size_t>(range) * 4), what(4), count(8);
    CGA_HIP(hipMemsetAsync(flags.p, 0, flags.bytes, stream));
    CGA_HIP(hipMemsetAsync(what.p, 0, 4, stream));
    CGA_HIP(hipMemsetAsync(count.p, 0, 8, stream));
    if (wide)
      hipLaunchKernelGGL((mark_in_range_kernel<int64_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         static_cast<int64_t const*>(vertices.data), n, lo, range, flags.as<unsigned int>(),
                         what.as<unsigned int>());
    else
      hipLaunchKernelGGL((mark_in_range_kernel<int32_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         static_cast<int32_t const*>(vertices.data), n, lo, range, flags.as<unsigned int>(),
                         what.as<unsigned int>());
    hipLaunchKernelGGL(count_set_kernel, dim3(grid_for(range, 1024)), dim3(kThreads), 0, stream,
                       flags.as<unsigned int>(), static_cast<int64_t>(range), count.as<unsigned long long>());
    unsigned long long distinct = 0;
    CGA_HIP(hipMemcpyAsync(&distinct, count.p, 8, hipMemcpyDeviceToHost, stream));
    CGA_HIP(hipStreamSynchronize(stream));
    duplicates = static_cast<int64_t>(distinct) < n;
  }
  // (ids spanning 2^31 or more: a list without renumbering fails the checks below; under
  // renumbering the caller compares the number of distinct ids the renumbering found, make_graph)
  CGA_EXPECTS(!duplicates, CUGRAPH_UNKNOWN_ERROR,
              "Invalid input argument: vertices should not have duplicates.");
  if (!renumber) {
    long long const id_max = wide ? 0x7fffffffffffffffLL : 0x7fffffffLL;
    CGA_EXPECTS(lo >= 0 && h_mm[1] < id_max, CUGRAPH_UNKNOWN_ERROR,
                "Invalid input argument: vertex IDs should be in [0, "
                "std::numeric_limits<vertex_t>::max()) if renumber is false.");
    CGA_EXPECTS(lo == 0 && range == n, CUGRAPH_UNKNOWN_ERROR,
                "Invalid input argument: vertex IDs should be consecutive integers starting from 0 "
                "if renumber is false.");
  }
}

// the built graph against its property flags (create_graph_from_edgelist_impl.cuh:902-920)
void expensive_check_graph(Handle const& h, Graph const& g)
{
  if (!g.is_symmetric && g.is_multigraph) return;
  if (g.num_edges == 0) return;
  CGA_EXPECTS(g.num_vertices == g.num_global_vertices && g.num_source_slots == 0, CUGRAPH_NOT_IMPLEMENTED,
              "do_expensive_check on a partitioned graph");
  hipStream_t stream = h.stream;
  DevBuf majors(static_cast<size_t>(g.num_edges) * 4), what(8);
  expand_offsets(stream, g.offsets, g.off64, g.num_vertices, g.num_edges, majors.as<int32_t>());
  CGA_HIP(hipMemsetAsync(what.p, 0, 8, stream));
  int const par = g.is_multigraph ? 0 : 1, sym = g.is_symmetric ? 1 : 0;
  if (g.off64)
    hipLaunchKernelGGL((edge_checks_kernel<int64_t>), dim3(grid_for(g.num_edges)), dim3(kThreads), 0, stream,
                       static_cast<int64_t const*>(g.offsets), g.indices, majors.as<int32_t>(), g.num_edges,
                       par, sym, what.as<unsigned int>());
  else
    hipLaunchKernelGGL((edge_checks_kernel<int32_t>), dim3(grid_for(g.num_edges)), dim3(kThreads), 0, stream,
                       static_cast<int32_t const*>(g.offsets), g.indices, majors.as<int32_t>(), g.num_edges,
                       par, sym, what.as<unsigned int>());
  unsigned int h_what[2] = {0, 0};
  CGA_HIP(hipMemcpyAsync(h_what, what.p, 8, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  // the reference tests symmetry first (:902-910), then parallel edges (:912-920)
  CGA_EXPECTS(h_what[1] == 0, CUGRAPH_UNKNOWN_ERROR,
              "Invalid input arguments: graph_properties.is_symmetric is true but the input edge list is "
              "not symmetric.");
  CGA_EXPECTS(h_what[0] == 0, CUGRAPH_UNKNOWN_ERROR,
              "Invalid input arguments: graph_properties.is_multigraph is false but the input edge list "
              "has parallel edges.");
}

}  // namespace cga
