// Edge-list hygiene before the graph build: drop_self_loops / drop_multi_edges of
// cugraph_graph_create_sg (reference: cpp/src/c_api/graph_sg.cpp:185-208, which calls
// cugraph::remove_self_loops and cugraph::remove_multi_edges, graph_functions.hpp:1000-1058).
//
//   self-loops : flag (src != dst) -> scan -> order-preserving compaction of src / dst / weights;
//   multi-edges: the surviving list is compressed once with a permutation (the same
//                histogram/scan/scatter/column-sort as the real build), so the copies of an edge
//                sit next to each other, in input order; one copy per run is kept -- the FIRST
//                in input order, or the one of minimum weight (first among equals) when the graph
//                is flagged symmetric, as the reference does to keep symmetry -- and the original
//                list is compacted by the resulting per-input flags.
// The reference keeps "an arbitrary" copy when not symmetric; keeping the first is one such choice
// and makes the result deterministic.
#include "common.hpp"

namespace cga {

void exclusive_scan_u32_to_i64(hipStream_t stream, unsigned int const* counts, int64_t n,
                               int64_t* out);

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

template <typename IdT>
__global__ void flag_not_loop_kernel(IdT const* src, IdT const* dst, int64_t n, unsigned int* keep)
{
  GRID_STRIDE(i, n) keep[i] = src[i] != dst[i] ? 1u : 0u;
}

template <typename T>
__global__ void compact_kernel(T const* in, int64_t n, unsigned int const* keep, int64_t const* pos,
                               T* out)
{
  GRID_STRIDE(i, n)
  {
    if (keep[i]) out[pos[i]] = in[i];
  }
}

// stored position k starts a run of copies when its (row, source) differs from k-1's; the thread
// of a run start walks the run, picks the copy to keep and flags its INPUT position
template <typename WT>
__global__ void pick_copies_kernel(int32_t const* rows, int32_t const* idx, uint32_t const* perm,
                                   int64_t n, WT const* w_in, bool keep_min, unsigned int* keep)
{
  GRID_STRIDE(k, n)
  {
    bool start = k == 0 || rows[k] != rows[k - 1] || idx[k] != idx[k - 1];
    if (!start) continue;
    int64_t best = k;
    if (keep_min && w_in) {
      WT wb = w_in[perm[k]];
      for (int64_t j = k + 1; j < n && rows[j] == rows[k] && idx[j] == idx[k]; ++j) {
        WT wj = w_in[perm[j]];
        if (wj < wb) {
          wb   = wj;
          best = j;
        }
      }
    }
    keep[perm[best]] = 1u;
  }
}

struct Filtered {
  DevBuf src, dst, w;
  int64_t n{0};
};

void compact_lists(hipStream_t stream, ArrayView const& src, ArrayView const& dst, ArrayView const* w,
                   unsigned int const* keep, Filtered& out)
{
  int64_t n = static_cast<int64_t>(src.size);
  DevBuf pos(static_cast<size_t>(n + 1) * sizeof(int64_t));
  exclusive_scan_u32_to_i64(stream, keep, n, pos.as<int64_t>());
  int64_t kept = 0;
  CGA_HIP(hipMemcpyAsync(&kept, pos.as<int64_t>() + n, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  size_t ib = dtype_size(src.type);
  out.n     = kept;
  out.src.alloc(static_cast<size_t>(kept > 0 ? kept : 1) * ib);
  out.dst.alloc(static_cast<size_t>(kept > 0 ? kept : 1) * ib);
  int grid = grid_for(n);
  if (ib == 8) {
    hipLaunchKernelGGL((compact_kernel<int64_t>), dim3(grid), dim3(kThreads), 0, stream,
                       static_cast<int64_t const*>(src.data), n, keep, pos.as<int64_t>(), out.src.as<int64_t>());
    hipLaunchKernelGGL((compact_kernel<int64_t>), dim3(grid), dim3(kThreads), 0, stream,
                       static_cast<int64_t const*>(dst.data), n, keep, pos.as<int64_t>(), out.dst.as<int64_t>());
  } else {
    hipLaunchKernelGGL((compact_kernel<int32_t>), dim3(grid), dim3(kThreads), 0, stream,
                       static_cast<int32_t const*>(src.data), n, keep, pos.as<int64_t>(), out.src.as<int32_t>());
    hipLaunchKernelGGL((compact_kernel<int32_t>), dim3(grid), dim3(kThreads), 0, stream,
                       static_cast<int32_t const*>(dst.data), n, keep, pos.as<int64_t>(), out.dst.as<int32_t>());
  }
  if (w) {
    size_t wb = dtype_size(w->type);
    out.w.alloc(static_cast<size_t>(kept > 0 ? kept : 1) * wb);
    if (wb == 8)
      hipLaunchKernelGGL((compact_kernel<double>), dim3(grid), dim3(kThreads), 0, stream,
                         static_cast<double const*>(w->data), n, keep, pos.as<int64_t>(), out.w.as<double>());
    else
      hipLaunchKernelGGL((compact_kernel<float>), dim3(grid), dim3(kThreads), 0, stream,
                         static_cast<float const*>(w->data), n, keep, pos.as<int64_t>(), out.w.as<float>());
  }
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));  // pos dies here
}

ArrayView view_of(DevBuf const& b, int64_t n, cugraph_data_type_id_t t)
{
  return ArrayView{b.p, static_cast<size_t>(n), static_cast<size_t>(n) * dtype_size(t), t};
}

}  // namespace

// On return src / dst / weights (when given) view the filtered lists, owned by `hold`.
void filter_edgelist(Handle const& h, bool drop_self_loops, bool drop_multi_edges, bool keep_min_weight,
                     bool renumber, ArrayView const* vertices, ArrayView& src, ArrayView& dst,
                     ArrayView* weights, std::vector<DevBuf>& hold)
{
  hipStream_t stream = h.stream;
  auto adopt = [&](Filtered& f) {
    cugraph_data_type_id_t it = src.type;
    hold.push_back(std::move(f.src));
    src = view_of(hold.back(), f.n, it);
    hold.push_back(std::move(f.dst));
    dst = view_of(hold.back(), f.n, it);
    if (weights) {
      cugraph_data_type_id_t wt = weights->type;
      hold.push_back(std::move(f.w));
      *weights = view_of(hold.back(), f.n, wt);
    }
  };
  if (drop_self_loops && src.size > 0) {
    int64_t n = static_cast<int64_t>(src.size);
    DevBuf keep(static_cast<size_t>(n) * 4);
    if (src.type == INT64)
      hipLaunchKernelGGL((flag_not_loop_kernel<int64_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         static_cast<int64_t const*>(src.data), static_cast<int64_t const*>(dst.data), n,
                         keep.as<unsigned int>());
    else
      hipLaunchKernelGGL((flag_not_loop_kernel<int32_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         static_cast<int32_t const*>(src.data), static_cast<int32_t const*>(dst.data), n,
                         keep.as<unsigned int>());
    CGA_HIP(hipGetLastError());
    Filtered f;
    compact_lists(stream, src, dst, weights, keep.as<unsigned int>(), f);
    adopt(f);
  }
  if (drop_multi_edges && src.size > 1) {
    int64_t n = static_cast<int64_t>(src.size);
    CGA_EXPECTS(n <= 0xffffffffLL, CUGRAPH_NOT_IMPLEMENTED, "drop_multi_edges: 2^32 or more edges");
    // dense int32 endpoints of the surviving list (a throw-away renumbering when ids are external)
    DevBuf s32(static_cast<size_t>(n) * 4), d32(static_cast<size_t>(n) * 4);
    int64_t nv = 0;
    if (renumber) {
      Graph tmp;
      tmp.vertex_type = src.type;
      renumber_edgelist(h, tmp, vertices, &src, &dst, true, s32.as<int32_t>(), d32.as<int32_t>());
      nv = tmp.num_vertices;
    } else {
      int64_t smin = 0, dmin = 0;
      int64_t smax = narrow_ids(stream, src.data, src.type, src.size, s32.as<int32_t>(), &smin);
      int64_t dmax = narrow_ids(stream, dst.data, dst.type, dst.size, d32.as<int32_t>(), &dmin);
      CGA_EXPECTS(smin >= 0 && dmin >= 0, CUGRAPH_INVALID_INPUT,
                  "Invalid input arguments: negative vertex ids need renumber=TRUE.");
      nv = 1 + std::max(smax, dmax);
    }
    CGA_EXPECTS(nv < 0x7fffffffLL, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
                "2^31 or more vertices are not supported");
    CompressedOut c = compress_coo_perm(stream, nv, nv, n, d32.as<int32_t>(), s32.as<int32_t>());
    s32.release();
    d32.release();
    DevBuf rows(static_cast<size_t>(n) * 4), keep(static_cast<size_t>(n) * 4);
    expand_offsets(stream, c.offsets.p, c.off64, nv, n, rows.as<int32_t>());
    CGA_HIP(hipMemsetAsync(keep.p, 0, keep.bytes, stream));
    bool f64 = weights && weights->type == FLOAT64;
    if (f64)
      hipLaunchKernelGGL((pick_copies_kernel<double>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         rows.as<int32_t>(), c.indices.as<int32_t>(), c.perm.as<uint32_t>(), n,
                         static_cast<double const*>(weights->data), keep_min_weight, keep.as<unsigned int>());
    else
      hipLaunchKernelGGL((pick_copies_kernel<float>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         rows.as<int32_t>(), c.indices.as<int32_t>(), c.perm.as<uint32_t>(), n,
                         weights ? static_cast<float const*>(weights->data) : nullptr, keep_min_weight,
                         keep.as<unsigned int>());
    CGA_HIP(hipGetLastError());
    Filtered f;
    compact_lists(stream, src, dst, weights, keep.as<unsigned int>(), f);
    adopt(f);
  }
}

}  // namespace cga
