// Internal types shared by the translation units of libcugraph_amd_c.so.
// Nothing here crosses the C ABI: the public view of every struct is an opaque pointer
// (include/cugraph_amd_c.h), as in the reference (cpp/include/cugraph_c/graph.h:25-27).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cugraph_amd_c.h"

namespace cga {

// ---------------------------------------------------------------- errors
struct Error {
  std::string message;
};

struct Failure : std::runtime_error {
  cugraph_error_code_t code;
  Failure(cugraph_error_code_t c, std::string const& m) : std::runtime_error(m), code(c) {}
};

#define CGA_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess) {                                                                   \
      throw cga::Failure(e__ == hipErrorOutOfMemory ? CUGRAPH_ALLOC_ERROR                     \
                                                    : CUGRAPH_UNKNOWN_ERROR,                   \
                         std::string("HIP error: ") + hipGetErrorString(e__) + " at " +        \
                           __FILE__ + ":" + std::to_string(__LINE__));                         \
    }                                                                                          \
  } while (0)

// An argument the ALGORITHM rejects (alpha outside [0, 1], a negative epsilon, personalization
// values that sum to zero, ...): the reference throws these out of cugraph::pagerank & co., and
// run_algorithm turns every such exception into CUGRAPH_UNKNOWN_ERROR (cpp/src/c_api/utils.hpp:53-55;
// pylibcugraph then raises RuntimeError).  CUGRAPH_INVALID_INPUT stays what the C layer's own
// CAPI_EXPECTS checks return (type and size mismatches, pagerank.cpp:268-299).
constexpr cugraph_error_code_t kAlgorithmRejects = CUGRAPH_UNKNOWN_ERROR;

#define CGA_EXPECTS(cond, code, msg)                 \
  do {                                               \
    if (!(cond)) throw cga::Failure((code), (msg)); \
  } while (0)

// ---------------------------------------------------------------- handle / arrays
struct Handle {
  hipStream_t stream{nullptr};
  bool owns_stream{false};
  int device{0};
  int comm_size{1};
  int rank{0};
  // the caller's collectives (cugraph_amd_resource_handle_set_comm): what the C-level multi-GPU path
  // (mg_capi.hip) exchanges through; unset on a single-GPU handle
  cugraph_amd_comm_t comm{};
  bool has_comm{false};
};

inline size_t dtype_size(cugraph_data_type_id_t t)
{
  switch (t) {
    case INT32: return 4;
    case INT64: return 8;
    case FLOAT32: return 4;
    case FLOAT64: return 8;
    case SIZE_T: return 8;
    default: return 0;
  }
}

// non-owning {data, size, nbytes, type} -- reference: cpp/src/c_api/array.hpp:26-45
struct ArrayView {
  void* data{nullptr};
  size_t size{0};
  size_t nbytes{0};
  cugraph_data_type_id_t type{INT32};
};

// owning device buffer -- reference: cpp/src/c_api/array.hpp:47-84
struct Array {
  void* data{nullptr};
  size_t size{0};
  size_t nbytes{0};
  cugraph_data_type_id_t type{INT32};
  ~Array()
  {
    if (data) (void)hipFree(data);
  }
};

struct HostArrayView {
  void* data{nullptr};
  size_t size{0};
  size_t nbytes{0};
  cugraph_data_type_id_t type{INT32};
};

struct HostArray {
  void* data{nullptr};
  size_t size{0};
  size_t nbytes{0};
  cugraph_data_type_id_t type{INT32};
  ~HostArray() { ::operator delete(data); }
};

// RAII device buffer for internal temporaries
struct DevBuf {
  void* p{nullptr};
  size_t bytes{0};
  DevBuf() = default;
  explicit DevBuf(size_t n) { alloc(n); }
  DevBuf(DevBuf const&)            = delete;
  DevBuf& operator=(DevBuf const&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept
  {
    if (this != &o) {
      release();
      p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t n)
  {
    release();
    bytes = n;
    if (n) CGA_HIP(hipMalloc(&p, n));
  }
  void release()
  {
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
  }
  void* detach()
  {
    void* r = p;
    p = nullptr; bytes = 0;
    return r;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// rows of in-degree 1 .. kLowDegreeMax are the thread-per-row bucket of the pull SpMV (pagerank.hip)
// and the classes the renumbering orders by their smallest source (renumber.hip)
#ifndef CGA_LOW_DEGREE_MAX
#define CGA_LOW_DEGREE_MAX 3  // (5 / 7 measured slower at scale 26: iteration 1.96 -> 1.98 / 2.05 ms, profiles/r3_xs_diag.md)
#endif
constexpr int kLowDegreeMax = CGA_LOW_DEGREE_MAX;

// position of `id` in an ascending array of distinct ids, -1 if absent
__host__ __device__ inline int64_t find_sorted(long long const* sorted, int64_t n, long long id)
{
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = lo + (hi - lo) / 2;
    if (sorted[mid] < id) lo = mid + 1; else hi = mid;
  }
  return (lo < n && sorted[lo] == id) ? lo : -1;
}

// external id -> id in a renumbered graph (-1: not a vertex): direct-address table over
// [min, min + range), or a search of the sorted distinct ids when the graph keeps those
struct ExtLookup {
  int32_t const* lookup;
  int64_t min, range;
  long long const* sorted;
  __host__ __device__ inline int64_t find(int64_t id) const
  {
    int64_t k = sorted ? find_sorted(sorted, range, id) : id - min;
    if (k < 0 || k >= range) return -1;
    return lookup[k];
  }
};

// ---------------------------------------------------------------- graph
// Compressed sparse graph.  Majors are dst when store_transposed (CSC, the pull layout PageRank
// wants), src otherwise.  Indices are always int32 internally (V < 2^31); offsets are int32 while
// num_edges < 2^31, else int64 -- the same switch as cpp/src/c_api/graph_sg.cpp:568-572.
struct PullPlan;  // pagerank.hip

struct MgInfo;
void destroy_mg_info(MgInfo* m);

struct Graph {
  cugraph_data_type_id_t vertex_type{INT32};  // type of the ids the caller handed over / gets back
  cugraph_data_type_id_t edge_type{INT32};
  cugraph_data_type_id_t weight_type{FLOAT32};  // FLOAT32 also for unweighted graphs (graph_sg.cpp:247)
  bool store_transposed{false};
  bool is_symmetric{false};
  bool is_multigraph{false};
  bool renumbered{false};
  bool degree_sorted{false};  // majors are in descending-degree order (contiguous degree buckets)

  int64_t num_vertices{0};         // majors held here (SG: all vertices)
  int64_t num_global_vertices{0};  // id space of the minors (== num_vertices on one GPU)
  int64_t vertex_first{0};         // first global id of the local major range (1D partition)
  // compact source space (multi-GPU): minors are SLOTS of the gathered vector, [0, num_source_slots),
  // and x_pos[row] is the slot row's own value goes to (-1: no edge leaves it, nobody gathers it)
  int64_t num_source_slots{0};     // 0: minors are vertex ids in [0, num_global_vertices)
  int32_t* x_pos{nullptr};
  // slots below hot_slots are the first REGION of the gathered vector (all-gathered first, see
  // cugraph_amd_graph_set_hot_slots); 0: one region
  int64_t hot_slots{0};
  int64_t num_edges{0};
  bool off64{false};

  void* offsets{nullptr};     // int32/int64 [num_vertices + 1]
  int32_t* indices{nullptr};  // [num_edges]
  void* weights{nullptr};     // float/double [num_edges] or nullptr
  void* number_map{nullptr};  // internal id -> external id, vertex_type[num_vertices]; nullptr = identity
  // external id -> internal id: ext_lookup[id - ext_min], -1 where the id is not a vertex
  // (set together with number_map: renumber=TRUE graphs and degree-sorted plan copies)
  int32_t* ext_lookup{nullptr};
  int64_t ext_min{0};
  int64_t ext_range{0};
  // ids spanning 2^31 or more (renumber.hip, renumber_sparse_ids): the distinct external ids in
  // ascending order [num_vertices]; then ext_lookup is indexed by the POSITION of an id in it
  long long* ext_sorted{nullptr};

  PullPlan* plan{nullptr};  // cached by the first PageRank call on this graph
  MgInfo* mg{nullptr};      // graphs made by cugraph_mg_graph_create / cugraph_graph_create_mg (mg_capi.hip)
  Graph* reversed{nullptr};  // the other orientation without weights, cached by the first HITS call

  int64_t minor_space() const { return num_source_slots > 0 ? num_source_slots : num_global_vertices; }
  ~Graph();
};

struct CentralityResult {
  Array* vertex_ids{nullptr};
  Array* values{nullptr};
  size_t num_iterations{0};
  bool converged{false};
  ~CentralityResult()
  {
    delete vertex_ids;
    delete values;
  }
};

// reference: cpp/src/c_api/hits.cpp (cugraph_hits_result_t)
struct HitsResult {
  Array* vertex_ids{nullptr};
  Array* hubs{nullptr};
  Array* authorities{nullptr};
  double hub_score_differences{0.0};
  size_t number_of_iterations{0};
  ~HitsResult()
  {
    delete vertex_ids;
    delete hubs;
    delete authorities;
  }
};

// ---------------------------------------------------------------- graph_build.hip
struct CompressedOut {
  DevBuf offsets;  // int32 or int64
  DevBuf indices;  // int32
  DevBuf weights;  // weight type or empty
  DevBuf perm;     // uint32 input position per stored edge (compress_coo_perm only)
  bool off64{false};
};

// COO (majors, minors int32, already validated to lie in [0, n_major) x [0, n_minor)) -> compressed,
// minors ascending inside each major, ties in input order.  major_counts: the number of edges of
// every major, when the caller has it (saves the histogram pass).
CompressedOut compress_coo(hipStream_t stream,
                           int64_t n_major,
                           int64_t n_minor,
                           int64_t n_edges,
                           int32_t const* majors,
                           int32_t const* minors,
                           void const* weights,
                           size_t weight_size,
                           unsigned int const* major_counts = nullptr,
                           bool hubs_first = false);  // majors are in descending degree order

// ties_in_input_order = false: equal minors of a LONG column may come out in any order (saves
// half of the radix passes; the permutation stays valid)
CompressedOut compress_coo_perm(hipStream_t stream, int64_t n_major, int64_t n_minor,
                                int64_t n_edges, int32_t const* majors, int32_t const* minors,
                                bool ties_in_input_order = true);
#ifdef CGA_TIMING_AIDS
void xs_region_check_read(unsigned long long* out);  // pagerank.hip, timing-aid build only
void xs_bounds_check_read(unsigned long long* out);  // pagerank.hip, timing-aid build only
#endif
CompressedOut compress_coo_keys(hipStream_t stream, int64_t n_major, int64_t n_edges, int32_t const* majors,
                                uint32_t const* keys, int sort_from_bit, int sort_to_bit);

// counts[key] += occurrences (counts zeroed by the caller); hubs_first: small keys are the frequent ones
void histogram_u32(hipStream_t stream, int32_t const* keys, int64_t n, unsigned int* counts, bool hubs_first);

// offsets -> one major id per edge (the inverse of the histogram+scan)
void expand_offsets(hipStream_t stream, void const* offsets, bool off64, int64_t n_major,
                    int64_t n_edges, int32_t* majors_out);

// ids of any integer type -> int32, with range check; returns max id seen (or -1 if empty)
int64_t narrow_ids(hipStream_t stream, void const* in, cugraph_data_type_id_t type, size_t n,
                   int32_t* out, int64_t* min_out);

void fill_sequence(hipStream_t stream, void* out, cugraph_data_type_id_t type, size_t n,
                   int64_t first);

void transpose_graph(Handle const& h, Graph& g);

// renumber.hip
void degree_order(hipStream_t stream, unsigned int const* deg, int64_t n, DevBuf& order, DevBuf& rank);
void relabel_ids(hipStream_t stream, int32_t* ids, int64_t n, int32_t const* table);
Graph* degree_sorted_copy(Handle const& h, Graph const& g);
// major_counts_out: if given, receives the number of edges of every major in the NEW numbering
void renumber_edgelist(Handle const& h, Graph& g, ArrayView const* vertices, ArrayView const* src,
                       ArrayView const* dst, bool majors_are_dst, int32_t* src32, int32_t* dst32,
                       DevBuf* major_counts_out = nullptr);

void partition_degree_order(hipStream_t stream, ArrayView const& dst, int64_t first, int64_t count,
                            int32_t* order_out, int32_t* rank_out);
void partition_group_edges(hipStream_t stream, ArrayView const& src, ArrayView const& dst, ArrayView const* wgt,
                           int64_t const* bounds, int n_ranks, void* src_out, void* dst_out, void* wgt_out,
                           int64_t* counts_out);
bool degrees_descending(hipStream_t stream, void const* offsets, bool off64, int64_t n);

// edge_filter.hip: drop_self_loops / drop_multi_edges; the views are re-pointed at filtered copies
void filter_edgelist(Handle const& h, bool drop_self_loops, bool drop_multi_edges, bool keep_min_weight,
                     bool renumber, ArrayView const* vertices, ArrayView& src, ArrayView& dst,
                     ArrayView* weights, std::vector<DevBuf>& hold);

// graph_checks.hip: do_expensive_check of the graph build
void expensive_check_vertices(Handle const& h, ArrayView const& vertices, bool renumber);
void expensive_check_graph(Handle const& h, Graph const& g);

// out[i] = in[i] - shift  /  out[i] = (int64) in[i]
void shift_ids(hipStream_t stream, int32_t const* in, int32_t* out, int64_t n, int64_t shift);
void widen_ids(hipStream_t stream, int32_t const* in, int64_t* out, int64_t n);

// ---------------------------------------------------------------- pagerank.hip
struct PagerankArgs {
  ArrayView const* pre_vertices{nullptr};
  ArrayView const* pre_sums{nullptr};
  ArrayView const* guess_vertices{nullptr};
  ArrayView const* guess_values{nullptr};
  ArrayView const* pers_vertices{nullptr};
  ArrayView const* pers_values{nullptr};
  double alpha{0.85};
  double epsilon{1e-6};
  size_t max_iterations{100};
  bool do_expensive_check{false};
};

CentralityResult* run_pagerank(Handle const& h, Graph& g, PagerankArgs const& a);
// ---------------------------------------------------------------- mg_capi.hip
Graph* mg_create_graph(Handle const& h, cugraph_graph_properties_t const* props, size_t n,
                       ArrayView const* const* vertices, ArrayView const* const* src, ArrayView const* const* dst,
                       ArrayView const* const* weights, bool drop_self_loops, bool drop_multi_edges);
CentralityResult* mg_run_pagerank(Handle const& h, Graph& g, PagerankArgs const& a);
cugraph_data_type_id_t mg_api_vertex_type(Graph const& g);
void destroy_plan(PullPlan* p);

// ---------------------------------------------------------------- centrality.hip
CentralityResult* run_katz(Handle const& h, Graph& g, ArrayView const* betas, double alpha, double beta,
                           double epsilon, size_t max_iterations, bool do_expensive_check);
CentralityResult* run_eigenvector(Handle const& h, Graph& g, double epsilon, size_t max_iterations,
                                  bool do_expensive_check);
HitsResult* run_hits(Handle const& h, Graph& g, double epsilon, size_t max_iterations,
                     ArrayView const* guess_v, ArrayView const* guess_x, bool normalize,
                     bool do_expensive_check);

// ---------------------------------------------------------------- rmat.hip
void rmat_generate(hipStream_t stream, int scale, uint64_t first, uint64_t count, uint64_t seed,
                   uint32_t t_ab, uint32_t t_a, uint32_t t_c, int scramble, int32_t* src,
                   int32_t* dst);
void rmat_weights(hipStream_t stream, uint64_t first, uint64_t count, uint64_t seed, void* w,
                  bool f64);

}  // namespace cga
