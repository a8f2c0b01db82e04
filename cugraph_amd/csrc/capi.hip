// extern "C" surface of libcugraph_amd_c.so (declarations and reference citations:
// include/cugraph_amd_c.h).  Host-only glue: argument checks in the order the reference makes
// them (cpp/src/c_api/graph_sg.cpp:545-593, cpp/src/c_api/pagerank.cpp:268-299), the
// exception -> error-code boundary (cpp/src/c_api/utils.hpp:23-59), and buffer ownership.
#include "common.hpp"

#include <cstring>
#include <memory>

namespace cga {
// pagerank.hip
struct Session;
Session* session_create(Handle const& h, Graph& g, PagerankArgs const& a);
void session_iterate(Session& s, size_t n);
void session_iterate_phase(Session& s, int phase);
int64_t xs_window_slots(int64_t num_source_slots);
void session_timed_iterate(Session& s, size_t n, float* iter_ms, float* kernel_ms);
void session_status(Session& s, size_t* iters, bool* stopped, double* delta, double* dangling);
CentralityResult* session_result(Session& s, size_t max_iterations);
void session_info(Session& s, int64_t info[8]);
void session_exchange_buffers(Session& s, void** x_full, void** scalars);
void noalloc_setup_device(int device);
void edgelist_source_sums_noalloc(Handle const& h, ArrayView const& src, ArrayView const* weights,
                                  int64_t num_vertices, double* sums, long long* num_skipped);
void session_destroy(Session* s);
bool session_is_f64(Session& s);
int64_t session_num_global(Session& s);
void compute_source_sums(hipStream_t stream, Graph const& g, double* sums);
void edgelist_source_sums(hipStream_t stream, ArrayView const& src, ArrayView const* weights,
                          int64_t num_vertices, double* sums);
}  // namespace cga

using namespace cga;

namespace {

template <typename T, typename P>
T* as(P* p)
{
  return reinterpret_cast<T*>(p);
}
template <typename T, typename P>
T const* as(P const* p)
{
  return reinterpret_cast<T const*>(p);
}

cugraph_error_code_t fail(cugraph_error_t** error, cugraph_error_code_t code, std::string const& msg)
{
  if (error) *error = reinterpret_cast<cugraph_error_t*>(new Error{msg});
  return code;
}

// Runs body(); maps exceptions to the reference's error convention.
template <typename F>
cugraph_error_code_t guarded(cugraph_error_t** error, F&& body)
{
  if (error) *error = nullptr;
  try {
    body();
  } catch (Failure const& f) {
    return fail(error, f.code, f.what());
  } catch (std::bad_alloc const&) {
    return fail(error, CUGRAPH_ALLOC_ERROR, "host allocation failed");
  } catch (std::exception const& ex) {
    return fail(error, CUGRAPH_UNKNOWN_ERROR, ex.what());
  }
  return CUGRAPH_SUCCESS;
}

ArrayView* new_view(void* p, size_t n, cugraph_data_type_id_t t)
{
  return new ArrayView{p, n, n * dtype_size(t), t};
}

bool is_float_type(cugraph_data_type_id_t t) { return t == FLOAT32 || t == FLOAT64; }
bool is_int_type(cugraph_data_type_id_t t) { return t == INT32 || t == INT64; }

Graph* make_graph(Handle const& h, cugraph_graph_properties_t const* props, ArrayView const* vertices,
                  ArrayView const* src_in, ArrayView const* dst_in, ArrayView const* weights_in,
                  bool store_transposed, bool renumber, bool drop_self_loops, bool drop_multi_edges,
                  int64_t vertex_first, int64_t vertex_last, int64_t num_global /* <0: single GPU */,
                  int64_t num_source_slots = 0, ArrayView const* x_pos = nullptr,
                  bool do_expensive_check = false)
{
  ArrayView const *src = src_in, *dst = dst_in, *weights = weights_in;
  CGA_EXPECTS(src && dst, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src and dst are required.");
  CGA_EXPECTS(src->size == dst->size, CUGRAPH_INVALID_INPUT,
              "Invalid input arguments: src size != dst size.");
  CGA_EXPECTS(!vertices || src->type == vertices->type, CUGRAPH_INVALID_INPUT,
              "Invalid input arguments: src type != vertices type.");
  CGA_EXPECTS(src->type == dst->type, CUGRAPH_INVALID_INPUT,
              "Invalid input arguments: src type != dst type.");
  CGA_EXPECTS(!weights || weights->size == src->size, CUGRAPH_INVALID_INPUT,
              "Invalid input arguments: src size != weights size.");
  CGA_EXPECTS(is_int_type(src->type), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
              "vertex ids must be INT32 or INT64");
  CGA_EXPECTS(!weights || is_float_type(weights->type), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
              "edge weights must be FLOAT32 or FLOAT64");
  // graph_sg.cpp:185-208: hygiene on the edge list before anything else looks at it
  std::vector<DevBuf> filtered;
  ArrayView fsrc, fdst, fw;
  if (drop_self_loops || drop_multi_edges) {
    CGA_EXPECTS(num_global < 0, CUGRAPH_NOT_IMPLEMENTED,
                "drop_self_loops / drop_multi_edges on a partitioned graph");
    fsrc = *src;
    fdst = *dst;
    if (weights) fw = *weights;
    filter_edgelist(h, drop_self_loops, drop_multi_edges, props && props->is_symmetric, renumber,
                    vertices, fsrc, fdst, weights ? &fw : nullptr, filtered);
    src = &fsrc;
    dst = &fdst;
    if (weights) weights = &fw;
  }

  // create_graph_from_edgelist_impl.cuh:895-900 (graph_checks.hip)
  if (do_expensive_check && vertices) expensive_check_vertices(h, *vertices, renumber);

  hipStream_t stream = h.stream;
  int64_t ne         = static_cast<int64_t>(src->size);
  DevBuf s32(static_cast<size_t>(ne > 0 ? ne : 1) * 4), d32(static_cast<size_t>(ne > 0 ? ne : 1) * 4);
  int64_t smin = 0, dmin = 0, smax = -1, dmax = -1;
  if (!renumber) {
    smax = narrow_ids(stream, src->data, src->type, src->size, s32.as<int32_t>(), &smin);
    dmax = narrow_ids(stream, dst->data, dst->type, dst->size, d32.as<int32_t>(), &dmin);
  }

  auto g              = std::make_unique<Graph>();
  g->vertex_type      = src->type;
  // edge_t := INT64 iff n_edges >= INT32_MAX (graph_sg.cpp:568-572)
  g->edge_type        = static_cast<size_t>(ne) < 0x7fffffffULL ? src->type : INT64;
  g->weight_type      = weights ? weights->type : FLOAT32;  // graph_sg.cpp:574-578
  g->store_transposed = store_transposed;
  g->is_symmetric     = props && props->is_symmetric;
  g->is_multigraph    = props && props->is_multigraph;
  g->num_edges        = ne;

  int64_t n_major, n_minor, major_shift = 0;
  DevBuf major_counts;  // edges per major, when the renumbering has counted them already
  if (num_global >= 0) {  // local part of a 1D destination partition
    CGA_EXPECTS(store_transposed, CUGRAPH_INVALID_INPUT, "a partitioned graph is always stored transposed");
    CGA_EXPECTS(0 <= vertex_first && vertex_first <= vertex_last && vertex_last <= num_global,
                CUGRAPH_INVALID_INPUT, "invalid local vertex range");
    CGA_EXPECTS(ne == 0 || (dmin >= vertex_first && dmax < vertex_last), CUGRAPH_INVALID_INPUT,
                "destination ids outside this rank's vertex range");
    int64_t minor_space = num_source_slots > 0 ? num_source_slots : num_global;
    CGA_EXPECTS(ne == 0 || (smin >= 0 && smax < minor_space), CUGRAPH_INVALID_INPUT,
                num_source_slots > 0 ? "source slots outside [0, num_source_slots)"
                                     : "source ids outside [0, num_global_vertices)");
    g->num_vertices        = vertex_last - vertex_first;
    g->num_global_vertices = num_global;
    g->vertex_first        = vertex_first;
    if (num_source_slots > 0) {
      CGA_EXPECTS(x_pos && x_pos->type == INT32 &&
                    static_cast<int64_t>(x_pos->size) == g->num_vertices,
                  CUGRAPH_INVALID_INPUT, "x_pos must be an INT32 view with one entry per local vertex");
      CGA_EXPECTS(num_source_slots < 0x7fffffffLL, CUGRAPH_INVALID_INPUT, "too many source slots");
      g->num_source_slots = num_source_slots;
      size_t bytes        = static_cast<size_t>(g->num_vertices > 0 ? g->num_vertices : 1) * 4;
      CGA_HIP(hipMalloc(reinterpret_cast<void**>(&g->x_pos), bytes));
      if (g->num_vertices > 0)
        CGA_HIP(hipMemcpyAsync(g->x_pos, x_pos->data, static_cast<size_t>(g->num_vertices) * 4,
                               hipMemcpyDeviceToDevice, stream));
    }
    n_major                = g->num_vertices;
    n_minor                = minor_space;
    major_shift            = vertex_first;
  } else if (renumber) {
    // create_graph_from_edgelist_impl.cuh:925-941: ids become 0..V-1 by descending major degree
    renumber_edgelist(h, *g, vertices, src, dst, store_transposed, s32.as<int32_t>(),
                      d32.as<int32_t>(), &major_counts);
    n_major = n_minor = g->num_vertices;
    // create_graph_from_edgelist_impl.cuh:84-89, for id spans expensive_check_vertices cannot table
    CGA_EXPECTS(!(do_expensive_check && vertices) ||
                  g->num_vertices == static_cast<int64_t>(vertices->size),
                CUGRAPH_UNKNOWN_ERROR, "Invalid input argument: vertices should not have duplicates.");
  } else {
    // create_graph_from_edgelist_impl.cuh:939-949
    int64_t nv = vertices ? static_cast<int64_t>(vertices->size)
                          : 1 + std::max<int64_t>(smax, dmax);
    if (ne == 0 && !vertices) nv = 0;
    CGA_EXPECTS(ne == 0 || (smin >= 0 && dmin >= 0), CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: negative vertex ids need renumber=TRUE.");
    CGA_EXPECTS(ne == 0 || (smax < nv && dmax < nv), CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: vertex ids must lie in [0, number of vertices) when "
                "renumber=FALSE.");
    g->num_vertices = g->num_global_vertices = nv;
    n_major = n_minor = nv;
  }
  CGA_EXPECTS(n_minor < 0x7fffffffLL, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
              "2^31 or more vertices are not supported");

  int32_t* majors = store_transposed ? d32.as<int32_t>() : s32.as<int32_t>();
  int32_t* minors = store_transposed ? s32.as<int32_t>() : d32.as<int32_t>();
  DevBuf shifted;
  if (major_shift != 0 && ne > 0) {
    // local major ids = global id - vertex_first (partitioned build only)
    shifted.alloc(static_cast<size_t>(ne) * 4);
    shift_ids(stream, majors, shifted.as<int32_t>(), ne, major_shift);
    majors = shifted.as<int32_t>();
  }
  CompressedOut out = compress_coo(stream, n_major, n_minor, ne, majors, minors,
                                   weights ? weights->data : nullptr,
                                   weights ? dtype_size(weights->type) : 0,
                                   major_counts.p ? major_counts.as<unsigned int>() : nullptr,
                                   g->degree_sorted);
  g->off64   = out.off64;
  g->offsets = out.offsets.detach();
  g->indices = static_cast<int32_t*>(out.indices.detach());
  g->weights = out.weights.p ? out.weights.detach() : nullptr;
  // majors that already come in descending degree order (renumber=TRUE, or a partition the
  // multi-GPU driver relabelled) run on contiguous degree buckets without a sorted copy
  if (store_transposed && !g->degree_sorted)
    g->degree_sorted = degrees_descending(stream, g->offsets, g->off64, g->num_vertices);
  // create_graph_from_edgelist_impl.cuh:902-920, on the graph just built
  if (do_expensive_check) expensive_check_graph(h, *g);
  return g.release();
}

cugraph_error_code_t check_pair_types(Graph const* g, ArrayView const* v, ArrayView const* x,
                                      char const* vmsg, char const* xmsg, cugraph_error_t** error)
{
  if (v != nullptr) {
    if (mg_api_vertex_type(*g) != v->type) return fail(error, CUGRAPH_INVALID_INPUT, vmsg);
    if (x == nullptr || g->weight_type != x->type) return fail(error, CUGRAPH_INVALID_INPUT, xmsg);
  }
  return CUGRAPH_SUCCESS;
}

cugraph_error_code_t pagerank_entry(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                    const cugraph_type_erased_device_array_view_t* pre_v,
                                    const cugraph_type_erased_device_array_view_t* pre_s,
                                    const cugraph_type_erased_device_array_view_t* guess_v,
                                    const cugraph_type_erased_device_array_view_t* guess_x,
                                    const cugraph_type_erased_device_array_view_t* pers_v,
                                    const cugraph_type_erased_device_array_view_t* pers_x,
                                    double alpha, double epsilon, size_t max_iterations,
                                    bool_t do_expensive_check, bool must_converge,
                                    cugraph_centrality_result_t** result, cugraph_error_t** error)
{
  if (error) *error = nullptr;
  if (result) *result = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  if (!graph || !result) return fail(error, CUGRAPH_INVALID_INPUT, "graph / result pointer is NULL");
  auto* g = as<Graph>(graph);
  cugraph_error_code_t rc;
  rc = check_pair_types(g, as<ArrayView>(pre_v), as<ArrayView>(pre_s),
                        "vertex type of graph and precomputed_vertex_out_weight_vertices must match",
                        "vertex type of graph and precomputed_vertex_out_weight_sums must match", error);
  if (rc != CUGRAPH_SUCCESS) return rc;
  rc = check_pair_types(g, as<ArrayView>(guess_v), as<ArrayView>(guess_x),
                        "vertex type of graph and initial_guess_vertices must match",
                        "vertex type of graph and initial_guess_values must match", error);
  if (rc != CUGRAPH_SUCCESS) return rc;
  rc = check_pair_types(g, as<ArrayView>(pers_v), as<ArrayView>(pers_x),
                        "vertex type of graph and personalization_vector must match",
                        "vertex type of graph and personalization_vector must match", error);
  if (rc != CUGRAPH_SUCCESS) return rc;

  PagerankArgs a;
  a.pre_vertices       = as<ArrayView>(pre_v);
  a.pre_sums           = as<ArrayView>(pre_s);
  a.guess_vertices     = as<ArrayView>(guess_v);
  a.guess_values       = as<ArrayView>(guess_x);
  a.pers_vertices      = as<ArrayView>(pers_v);
  a.pers_values        = as<ArrayView>(pers_x);
  a.alpha              = alpha;
  a.epsilon            = epsilon;
  a.max_iterations     = max_iterations;
  a.do_expensive_check = do_expensive_check != FALSE;
  CentralityResult* res = nullptr;
  // a graph made by cugraph_mg_graph_create / cugraph_graph_create_mg: the partitioned loop (mg_capi.hip)
  rc = guarded(error, [&] { res = g->mg ? mg_run_pagerank(*as<Handle>(handle), *g, a) : run_pagerank(*as<Handle>(handle), *g, a); });
  if (rc != CUGRAPH_SUCCESS) return rc;
  *result = reinterpret_cast<cugraph_centrality_result_t*>(res);
  // pagerank.cpp:315-318: the strict variants report non-convergence but leave *result populated
  if (must_converge && !res->converged)
    return fail(error, CUGRAPH_UNKNOWN_ERROR, "PageRank failed to converge.");
  return CUGRAPH_SUCCESS;
}

}  // namespace

namespace cga {
Graph* make_partition_graph(Handle const& h, cugraph_graph_properties_t const* props, ArrayView const& src,
                            ArrayView const& dst, ArrayView const* weights, int64_t vertex_first,
                            int64_t vertex_last, int64_t num_global)
{
  return make_graph(h, props, nullptr, &src, &dst, weights, true, false, false, false, vertex_first, vertex_last,
                    num_global);
}
}  // namespace cga

extern "C" {

// ------------------------------------------------------------------ errors
const char* cugraph_error_message(const cugraph_error_t* error)
{
  return error ? as<Error>(error)->message.c_str() : nullptr;
}
void cugraph_error_free(cugraph_error_t* error) { delete as<Error>(error); }

// ------------------------------------------------------------------ handle
cugraph_resource_handle_t* cugraph_amd_create_resource_handle(void* hip_stream, int rank, int comm_size)
{
  try {
    auto h       = std::make_unique<Handle>();
    h->stream    = static_cast<hipStream_t>(hip_stream);
    h->rank      = rank;
    h->comm_size = comm_size < 1 ? 1 : comm_size;
    if (hipGetDevice(&h->device) != hipSuccess) return nullptr;
    try {  // the library's init call: address of this device's static scratch (noalloc.hip)
      noalloc_setup_device(h->device);
    } catch (Failure const&) {
      // reported by the entry point that needs it, which tries again
    }
    return reinterpret_cast<cugraph_resource_handle_t*>(h.release());
  } catch (...) {
    return nullptr;
  }
}
cugraph_resource_handle_t* cugraph_create_resource_handle(void* raft_handle)
{
  if (raft_handle != nullptr) return nullptr;  // there is no raft here; see the header
  return cugraph_amd_create_resource_handle(nullptr, 0, 1);
}
int cugraph_resource_handle_get_comm_size(const cugraph_resource_handle_t* handle)
{
  return handle ? as<Handle>(handle)->comm_size : 1;
}
int cugraph_resource_handle_get_rank(const cugraph_resource_handle_t* handle)
{
  return handle ? as<Handle>(handle)->rank : 0;
}
void cugraph_free_resource_handle(cugraph_resource_handle_t* handle) { delete as<Handle>(handle); }

// ------------------------------------------------------------------ device arrays
cugraph_error_code_t cugraph_type_erased_device_array_create(const cugraph_resource_handle_t* handle,
                                                             size_t n_elems,
                                                             cugraph_data_type_id_t dtype,
                                                             cugraph_type_erased_device_array_t** array,
                                                             cugraph_error_t** error)
{
  if (array) *array = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    CGA_EXPECTS(array && dtype_size(dtype) > 0, CUGRAPH_INVALID_INPUT, "bad array arguments");
    auto a    = std::make_unique<Array>();
    a->size   = n_elems;
    a->type   = dtype;
    a->nbytes = n_elems * dtype_size(dtype);
    if (a->nbytes) CGA_HIP(hipMalloc(&a->data, a->nbytes));
    *array = reinterpret_cast<cugraph_type_erased_device_array_t*>(a.release());
  });
}

cugraph_error_code_t cugraph_type_erased_device_array_create_from_view(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* view,
  cugraph_type_erased_device_array_t** array, cugraph_error_t** error)
{
  if (array) *array = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    CGA_EXPECTS(array && view, CUGRAPH_INVALID_INPUT, "bad array arguments");
    auto const* v = as<ArrayView>(view);
    auto a        = std::make_unique<Array>();
    a->size = v->size; a->type = v->type; a->nbytes = v->nbytes;
    hipStream_t st = as<Handle>(handle)->stream;
    if (a->nbytes) {
      CGA_HIP(hipMalloc(&a->data, a->nbytes));
      CGA_HIP(hipMemcpyAsync(a->data, v->data, a->nbytes, hipMemcpyDeviceToDevice, st));
      CGA_HIP(hipStreamSynchronize(st));
    }
    *array = reinterpret_cast<cugraph_type_erased_device_array_t*>(a.release());
  });
}

void cugraph_type_erased_device_array_free(cugraph_type_erased_device_array_t* p) { delete as<Array>(p); }

cugraph_type_erased_device_array_view_t* cugraph_type_erased_device_array_view(
  cugraph_type_erased_device_array_t* array)
{
  if (!array) return nullptr;
  auto* a = as<Array>(array);
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(new_view(a->data, a->size, a->type));
}

cugraph_error_code_t cugraph_type_erased_device_array_view_as_type(
  cugraph_type_erased_device_array_t* array, cugraph_data_type_id_t dtype,
  cugraph_type_erased_device_array_view_t** result_view, cugraph_error_t** error)
{
  if (result_view) *result_view = nullptr;
  return guarded(error, [&] {
    CGA_EXPECTS(array && result_view, CUGRAPH_INVALID_INPUT, "bad array arguments");
    auto* a = as<Array>(array);
    CGA_EXPECTS(dtype_size(dtype) == dtype_size(a->type) && dtype_size(dtype) > 0, CUGRAPH_INVALID_INPUT,
                "Could not treat type erased device array as a different type: sizes differ");
    *result_view = reinterpret_cast<cugraph_type_erased_device_array_view_t*>(new_view(a->data, a->size, dtype));
  });
}

cugraph_type_erased_device_array_view_t* cugraph_type_erased_device_array_view_create(
  void* pointer, size_t n_elems, cugraph_data_type_id_t dtype)
{
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(new_view(pointer, n_elems, dtype));
}
void cugraph_type_erased_device_array_view_free(cugraph_type_erased_device_array_view_t* p) { delete as<ArrayView>(p); }
size_t cugraph_type_erased_device_array_view_size(const cugraph_type_erased_device_array_view_t* p) { return as<ArrayView>(p)->size; }
cugraph_data_type_id_t cugraph_type_erased_device_array_view_type(const cugraph_type_erased_device_array_view_t* p) { return as<ArrayView>(p)->type; }
const void* cugraph_type_erased_device_array_view_pointer(const cugraph_type_erased_device_array_view_t* p) { return as<ArrayView>(p)->data; }

// ------------------------------------------------------------------ host arrays
cugraph_error_code_t cugraph_type_erased_host_array_create(const cugraph_resource_handle_t* handle,
                                                           size_t n_elems, cugraph_data_type_id_t dtype,
                                                           cugraph_type_erased_host_array_t** array,
                                                           cugraph_error_t** error)
{
  if (array) *array = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    CGA_EXPECTS(array && dtype_size(dtype) > 0, CUGRAPH_INVALID_INPUT, "bad array arguments");
    auto a    = std::make_unique<HostArray>();
    a->size   = n_elems;
    a->type   = dtype;
    a->nbytes = n_elems * dtype_size(dtype);
    a->data   = ::operator new(a->nbytes ? a->nbytes : 1);
    *array    = reinterpret_cast<cugraph_type_erased_host_array_t*>(a.release());
  });
}
void cugraph_type_erased_host_array_free(cugraph_type_erased_host_array_t* p) { delete as<HostArray>(p); }
cugraph_type_erased_host_array_view_t* cugraph_type_erased_host_array_view(cugraph_type_erased_host_array_t* array)
{
  if (!array) return nullptr;
  auto* a = as<HostArray>(array);
  return reinterpret_cast<cugraph_type_erased_host_array_view_t*>(
    new HostArrayView{a->data, a->size, a->nbytes, a->type});
}
cugraph_type_erased_host_array_view_t* cugraph_type_erased_host_array_view_create(void* pointer, size_t n_elems, cugraph_data_type_id_t dtype)
{
  return reinterpret_cast<cugraph_type_erased_host_array_view_t*>(
    new HostArrayView{pointer, n_elems, n_elems * dtype_size(dtype), dtype});
}
void cugraph_type_erased_host_array_view_free(cugraph_type_erased_host_array_view_t* p) { delete as<HostArrayView>(p); }
size_t cugraph_type_erased_host_array_size(const cugraph_type_erased_host_array_view_t* p) { return as<HostArrayView>(p)->size; }
cugraph_data_type_id_t cugraph_type_erased_host_array_type(const cugraph_type_erased_host_array_view_t* p) { return as<HostArrayView>(p)->type; }
void* cugraph_type_erased_host_array_pointer(const cugraph_type_erased_host_array_view_t* p) { return as<HostArrayView>(p)->data; }

cugraph_error_code_t cugraph_type_erased_host_array_view_copy(const cugraph_resource_handle_t*,
                                                              cugraph_type_erased_host_array_view_t* dst,
                                                              const cugraph_type_erased_host_array_view_t* src,
                                                              cugraph_error_t** error)
{
  return guarded(error, [&] {
    auto* d = as<HostArrayView>(dst);
    auto const* s = as<HostArrayView>(src);
    CGA_EXPECTS(d && s, CUGRAPH_INVALID_INPUT, "bad array arguments");
    CGA_EXPECTS(d->nbytes == s->nbytes, CUGRAPH_INVALID_INPUT, "Can only copy arrays of the same size");
    CGA_EXPECTS(d->type == s->type, CUGRAPH_INVALID_INPUT, "Can only copy arrays of the same type");
    std::memcpy(d->data, s->data, s->nbytes);
  });
}

// ------------------------------------------------------------------ copies
cugraph_error_code_t cugraph_type_erased_device_array_view_copy_from_host(
  const cugraph_resource_handle_t* handle, cugraph_type_erased_device_array_view_t* dst,
  const byte_t* h_src, cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto* d = as<ArrayView>(dst);
    CGA_EXPECTS(d && (h_src || d->nbytes == 0), CUGRAPH_INVALID_INPUT, "bad array arguments");
    hipStream_t st = as<Handle>(handle)->stream;
    if (d->nbytes) CGA_HIP(hipMemcpyAsync(d->data, h_src, d->nbytes, hipMemcpyHostToDevice, st));
    CGA_HIP(hipStreamSynchronize(st));
  });
}

cugraph_error_code_t cugraph_type_erased_device_array_view_copy_to_host(
  const cugraph_resource_handle_t* handle, byte_t* h_dst,
  const cugraph_type_erased_device_array_view_t* src, cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto const* s = as<ArrayView>(src);
    CGA_EXPECTS(s && (h_dst || s->nbytes == 0), CUGRAPH_INVALID_INPUT, "bad array arguments");
    hipStream_t st = as<Handle>(handle)->stream;
    if (s->nbytes) CGA_HIP(hipMemcpyAsync(h_dst, s->data, s->nbytes, hipMemcpyDeviceToHost, st));
    CGA_HIP(hipStreamSynchronize(st));
  });
}

cugraph_error_code_t cugraph_type_erased_device_array_view_copy(
  const cugraph_resource_handle_t* handle, cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* src, cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto* d = as<ArrayView>(dst);
    auto const* s = as<ArrayView>(src);
    CGA_EXPECTS(d && s, CUGRAPH_INVALID_INPUT, "bad array arguments");
    // cpp/src/c_api/array.cpp: "Can only copy arrays of the same size / type"
    CGA_EXPECTS(d->nbytes == s->nbytes, CUGRAPH_INVALID_INPUT, "Can only copy arrays of the same size");
    CGA_EXPECTS(d->type == s->type, CUGRAPH_INVALID_INPUT, "Can only copy arrays of the same type");
    hipStream_t st = as<Handle>(handle)->stream;
    if (s->nbytes) CGA_HIP(hipMemcpyAsync(d->data, s->data, s->nbytes, hipMemcpyDeviceToDevice, st));
    CGA_HIP(hipStreamSynchronize(st));
  });
}

// ------------------------------------------------------------------ graph
cugraph_error_code_t cugraph_graph_create_sg(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* vertices,
  const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed,
  bool_t renumber, bool_t drop_self_loops, bool_t drop_multi_edges, bool_t do_expensive_check,
  cugraph_graph_t** graph, cugraph_error_t** error)
{
  if (graph) *graph = nullptr;
  if (error) *error = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  if (!graph) return fail(error, CUGRAPH_INVALID_INPUT, "graph pointer is NULL");
  return guarded(error, [&] {
    auto const* s = as<ArrayView>(src);
    CGA_EXPECTS(s != nullptr, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src is NULL.");
    auto const* ei = as<ArrayView>(edge_ids);
    auto const* et = as<ArrayView>(edge_type_ids);
    // graph_sg.cpp:580-593 (edge ids / types are validated, then not used by this path)
    cugraph_data_type_id_t edge_type = s->size < 0x7fffffffULL ? s->type : INT64;
    CGA_EXPECTS(!ei || ei->type == edge_type, CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: Edge id type must match edge type");
    CGA_EXPECTS(!ei || ei->size == s->size, CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: src size != edge id prop size");
    CGA_EXPECTS(!et || et->size == s->size, CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: src size != edge type prop size");
    Graph* g = make_graph(*as<Handle>(handle), properties, as<ArrayView>(vertices), s,
                          as<ArrayView>(dst), as<ArrayView>(weights), store_transposed != FALSE,
                          renumber != FALSE, drop_self_loops != FALSE, drop_multi_edges != FALSE, 0, 0, -1,
                          0, nullptr, do_expensive_check != FALSE);
    *graph = reinterpret_cast<cugraph_graph_t*>(g);
  });
}

cugraph_error_code_t cugraph_sg_graph_create(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed,
  bool_t renumber, bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error)
{
  return cugraph_graph_create_sg(handle, properties, nullptr, src, dst, weights, edge_ids,
                                 edge_type_ids, store_transposed, renumber, FALSE, FALSE,
                                 do_expensive_check, graph, error);
}

cugraph_error_code_t cugraph_graph_create_sg_from_csr(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* offsets,
  const cugraph_type_erased_device_array_view_t* indices,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed,
  bool_t renumber, bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error)
{
  if (graph) *graph = nullptr;
  if (error) *error = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  if (!graph) return fail(error, CUGRAPH_INVALID_INPUT, "graph pointer is NULL");
  return guarded(error, [&] {
    auto const* off = as<ArrayView>(offsets);
    auto const* idx = as<ArrayView>(indices);
    auto const* w   = as<ArrayView>(weights);
    CGA_EXPECTS(off && idx && off->size >= 1, CUGRAPH_INVALID_INPUT, "Invalid input arguments: offsets / indices.");
    CGA_EXPECTS(!w || w->size == idx->size, CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: src size != weights size.");
    CGA_EXPECTS((edge_ids == nullptr) == (edge_type_ids == nullptr), CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: either none or both of edge ids and edge types must be provided.");
    CGA_EXPECTS(is_int_type(off->type) && is_int_type(idx->type), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
                "offsets / indices must be INT32 or INT64");
    // graph_sg.cpp:262-430: the CSR is expanded to an edge list (sources from the offsets) and
    // then goes through the same build as a COO
    Handle const& h    = *as<Handle>(handle);
    int64_t nv         = static_cast<int64_t>(off->size) - 1;
    int64_t ne         = static_cast<int64_t>(idx->size);
    DevBuf srcs(static_cast<size_t>(ne > 0 ? ne : 1) * 4);
    expand_offsets(h.stream, off->data, off->type == INT64, nv, ne, srcs.as<int32_t>());
    DevBuf verts(static_cast<size_t>(nv > 0 ? nv : 1) * dtype_size(idx->type));
    fill_sequence(h.stream, verts.p, idx->type, static_cast<size_t>(nv), 0);
    DevBuf srcs_typed;
    ArrayView sv{srcs.p, static_cast<size_t>(ne), static_cast<size_t>(ne) * 4, INT32};
    if (idx->type == INT64) {
      srcs_typed.alloc(static_cast<size_t>(ne > 0 ? ne : 1) * 8);
      widen_ids(h.stream, srcs.as<int32_t>(), srcs_typed.as<int64_t>(), ne);
      sv = ArrayView{srcs_typed.p, static_cast<size_t>(ne), static_cast<size_t>(ne) * 8, INT64};
    }
    ArrayView vv{verts.p, static_cast<size_t>(nv), static_cast<size_t>(nv) * dtype_size(idx->type), idx->type};
    Graph* g = make_graph(h, properties, &vv, &sv, idx, w, store_transposed != FALSE, renumber != FALSE,
                          false, false, 0, 0, -1, 0, nullptr, do_expensive_check != FALSE);
    *graph = reinterpret_cast<cugraph_graph_t*>(g);
  });
}

cugraph_error_code_t cugraph_sg_graph_create_from_csr(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* offsets,
  const cugraph_type_erased_device_array_view_t* indices,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed,
  bool_t renumber, bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error)
{
  return cugraph_graph_create_sg_from_csr(handle, properties, offsets, indices, weights, edge_ids,
                                          edge_type_ids, store_transposed, renumber,
                                          do_expensive_check, graph, error);
}

void cugraph_graph_free(cugraph_graph_t* graph) { delete as<Graph>(graph); }
void cugraph_sg_graph_free(cugraph_graph_t* graph) { cugraph_graph_free(graph); }
void cugraph_mg_graph_free(cugraph_graph_t* graph) { cugraph_graph_free(graph); }

// ------------------------------------------------------------------ multi-GPU graphs (mg_capi.hip)
cugraph_error_code_t cugraph_amd_resource_handle_set_comm(cugraph_resource_handle_t* handle,
                                                          const cugraph_amd_comm_t* comm, cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto* h = as<Handle>(handle);
    if (!comm) {
      h->has_comm = false;
      h->comm     = cugraph_amd_comm_t{};
      return;
    }
    CGA_EXPECTS(comm->all_reduce_f64 && comm->all_gather_v && comm->all_to_all_v, CUGRAPH_INVALID_INPUT,
                "all three collectives are required");
    h->comm     = *comm;
    h->has_comm = true;
  });
}

cugraph_error_code_t cugraph_graph_create_mg(
  cugraph_resource_handle_t const* handle, cugraph_graph_properties_t const* properties,
  cugraph_type_erased_device_array_view_t const* const* vertices,
  cugraph_type_erased_device_array_view_t const* const* src,
  cugraph_type_erased_device_array_view_t const* const* dst,
  cugraph_type_erased_device_array_view_t const* const* weights,
  cugraph_type_erased_device_array_view_t const* const* edge_ids,
  cugraph_type_erased_device_array_view_t const* const* edge_type_ids, bool_t store_transposed,
  size_t num_arrays, bool_t drop_self_loops, bool_t drop_multi_edges, bool_t do_expensive_check,
  cugraph_graph_t** graph, cugraph_error_t** error)
{
  if (graph) *graph = nullptr;
  if (error) *error = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  if (!graph) return fail(error, CUGRAPH_INVALID_INPUT, "graph pointer is NULL");
  // graph_mg.cpp:374-380
  if ((edge_ids == nullptr) != (edge_type_ids == nullptr))
    return fail(error, CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: either none or both of edge ids and edge types must be provided.");
  if (edge_ids != nullptr)
    return fail(error, CUGRAPH_NOT_IMPLEMENTED, "edge ids / edge types on a multi-GPU graph (PageRank does not read them)");
  (void)store_transposed;     // the partition is stored by destination either way (PageRank's orientation)
  (void)do_expensive_check;
  return guarded(error, [&] {
    Graph* g = mg_create_graph(*as<Handle>(handle), properties, num_arrays,
                               reinterpret_cast<ArrayView const* const*>(vertices),
                               reinterpret_cast<ArrayView const* const*>(src),
                               reinterpret_cast<ArrayView const* const*>(dst),
                               reinterpret_cast<ArrayView const* const*>(weights), drop_self_loops != FALSE,
                               drop_multi_edges != FALSE);
    *graph = reinterpret_cast<cugraph_graph_t*>(g);
  });
}

cugraph_error_code_t cugraph_mg_graph_create(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* src, const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed, size_t num_edges,
  bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error)
{
  (void)num_edges;  // unused in the reference too (graph_mg.cpp, mg_test_utils.cpp:202)
  return cugraph_graph_create_mg(handle, properties, nullptr, &src, &dst, weights ? &weights : nullptr,
                                 edge_ids ? &edge_ids : nullptr, edge_type_ids ? &edge_type_ids : nullptr,
                                 store_transposed, 1, FALSE, FALSE, do_expensive_check, graph, error);
}

cugraph_error_code_t cugraph_amd_graph_create_local(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights, int64_t vertex_first,
  int64_t vertex_last, int64_t num_global_vertices, cugraph_graph_t** graph, cugraph_error_t** error)
{
  if (graph) *graph = nullptr;
  if (error) *error = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  if (!graph) return fail(error, CUGRAPH_INVALID_INPUT, "graph pointer is NULL");
  return guarded(error, [&] {
    CGA_EXPECTS(num_global_vertices >= 0, CUGRAPH_INVALID_INPUT, "num_global_vertices must be >= 0");
    Graph* g = make_graph(*as<Handle>(handle), properties, nullptr, as<ArrayView>(src),
                          as<ArrayView>(dst), as<ArrayView>(weights), true, false, false, false,
                          vertex_first, vertex_last, num_global_vertices);
    *graph = reinterpret_cast<cugraph_graph_t*>(g);
  });
}

cugraph_error_code_t cugraph_amd_graph_create_local_compact(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* src_slots,
  const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights, int64_t vertex_first,
  int64_t vertex_last, int64_t num_global_vertices, int64_t num_source_slots,
  const cugraph_type_erased_device_array_view_t* x_pos, cugraph_graph_t** graph,
  cugraph_error_t** error)
{
  if (graph) *graph = nullptr;
  if (error) *error = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  if (!graph) return fail(error, CUGRAPH_INVALID_INPUT, "graph pointer is NULL");
  return guarded(error, [&] {
    CGA_EXPECTS(num_global_vertices >= 0 && num_source_slots > 0, CUGRAPH_INVALID_INPUT,
                "num_global_vertices must be >= 0 and num_source_slots > 0");
    Graph* g = make_graph(*as<Handle>(handle), properties, nullptr, as<ArrayView>(src_slots),
                          as<ArrayView>(dst), as<ArrayView>(weights), true, false, false, false,
                          vertex_first, vertex_last, num_global_vertices, num_source_slots,
                          as<ArrayView>(x_pos));
    *graph = reinterpret_cast<cugraph_graph_t*>(g);
  });
}

cugraph_error_code_t cugraph_amd_edgelist_source_weight_sums(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* weights,
  cugraph_type_erased_device_array_view_t* sums, cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto const* s = as<ArrayView>(src);
    auto const* w = as<ArrayView>(weights);
    auto* out     = as<ArrayView>(sums);
    CGA_EXPECTS(s && out, CUGRAPH_INVALID_INPUT, "NULL argument");
    CGA_EXPECTS(is_int_type(s->type), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "src must be INT32 or INT64");
    CGA_EXPECTS(!w || (is_float_type(w->type) && w->size == s->size), CUGRAPH_INVALID_INPUT,
                "weights must be FLOAT32/FLOAT64 with one entry per edge");
    CGA_EXPECTS(out->type == FLOAT64, CUGRAPH_INVALID_INPUT, "sums must be FLOAT64");
    edgelist_source_sums(as<Handle>(handle)->stream, *s, w, static_cast<int64_t>(out->size),
                         static_cast<double*>(out->data));
  });
}

cugraph_error_code_t cugraph_amd_edgelist_source_weight_sums_noalloc(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* weights,
  cugraph_type_erased_device_array_view_t* sums, cugraph_type_erased_device_array_view_t* num_skipped,
  cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto const* s = as<ArrayView>(src);
    auto const* w = as<ArrayView>(weights);
    auto* out     = as<ArrayView>(sums);
    auto* skip    = as<ArrayView>(num_skipped);
    CGA_EXPECTS(s && out, CUGRAPH_INVALID_INPUT, "NULL argument");
    CGA_EXPECTS(is_int_type(s->type), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "src must be INT32 or INT64");
    CGA_EXPECTS(!w || (is_float_type(w->type) && w->size == s->size), CUGRAPH_INVALID_INPUT,
                "weights must be FLOAT32/FLOAT64 with one entry per edge");
    CGA_EXPECTS(out->type == FLOAT64, CUGRAPH_INVALID_INPUT, "sums must be FLOAT64");
    CGA_EXPECTS(!skip || (skip->type == INT64 && skip->size >= 1), CUGRAPH_INVALID_INPUT,
                "num_skipped must be an INT64 view of at least one element, or NULL");
    edgelist_source_sums_noalloc(*as<Handle>(handle), *s, w, static_cast<int64_t>(out->size),
                                 static_cast<double*>(out->data),
                                 skip ? static_cast<long long*>(skip->data) : nullptr);
  });
}

cugraph_error_code_t cugraph_amd_partition_degree_order(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* dst,
  int64_t vertex_first, int64_t vertex_last, cugraph_type_erased_device_array_view_t* order,
  cugraph_type_erased_device_array_view_t* rank, cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto const* d = as<ArrayView>(dst);
    auto* o       = as<ArrayView>(order);
    auto* r       = as<ArrayView>(rank);
    CGA_EXPECTS(d && o && r, CUGRAPH_INVALID_INPUT, "NULL argument");
    CGA_EXPECTS(is_int_type(d->type), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "dst must be INT32 or INT64");
    int64_t n = vertex_last - vertex_first;
    CGA_EXPECTS(vertex_first >= 0 && n >= 0 && n < 0x7fffffffLL, CUGRAPH_INVALID_INPUT,
                "invalid local vertex range");
    CGA_EXPECTS(o->type == INT32 && r->type == INT32 && static_cast<int64_t>(o->size) == n &&
                  static_cast<int64_t>(r->size) == n,
                CUGRAPH_INVALID_INPUT, "order / rank must be INT32 views of vertex_last - vertex_first elements");
    partition_degree_order(as<Handle>(handle)->stream, *d, vertex_first, n,
                           static_cast<int32_t*>(o->data), static_cast<int32_t*>(r->data));
  });
}

cugraph_error_code_t cugraph_amd_partition_group_edges(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* dst, const cugraph_type_erased_device_array_view_t* weights,
  const int64_t* bounds, size_t num_ranks, cugraph_type_erased_device_array_view_t* src_out,
  cugraph_type_erased_device_array_view_t* dst_out, cugraph_type_erased_device_array_view_t* weights_out,
  int64_t* counts, cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto const* s = as<ArrayView>(src);
    auto const* d = as<ArrayView>(dst);
    auto const* w = as<ArrayView>(weights);
    auto* so      = as<ArrayView>(src_out);
    auto* dd      = as<ArrayView>(dst_out);
    auto* wo      = as<ArrayView>(weights_out);
    CGA_EXPECTS(s && d && so && dd && bounds && counts, CUGRAPH_INVALID_INPUT, "NULL argument");
    CGA_EXPECTS(is_int_type(d->type) && s->type == d->type && so->type == d->type && dd->type == d->type,
                CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "src / dst and their outputs must share one of INT32, INT64");
    CGA_EXPECTS(s->size == d->size && so->size == d->size && dd->size == d->size, CUGRAPH_INVALID_INPUT,
                "src, dst and their outputs must have one entry per edge");
    CGA_EXPECTS((w == nullptr) == (wo == nullptr), CUGRAPH_INVALID_INPUT, "weights and weights_out go together");
    CGA_EXPECTS(!w || (is_float_type(w->type) && wo->type == w->type && w->size == d->size && wo->size == d->size),
                CUGRAPH_INVALID_INPUT, "weights must be FLOAT32/FLOAT64 with one entry per edge, in and out");
    CGA_EXPECTS(num_ranks >= 1 && num_ranks <= 256, CUGRAPH_INVALID_INPUT, "1 .. 256 ranks");
    for (size_t r = 0; r < num_ranks; ++r)
      CGA_EXPECTS(bounds[r] <= bounds[r + 1], CUGRAPH_INVALID_INPUT, "bounds must ascend");
    partition_group_edges(as<Handle>(handle)->stream, *s, *d, w, bounds, static_cast<int>(num_ranks), so->data,
                          dd->data, wo ? wo->data : nullptr, counts);
  });
}

cugraph_error_code_t cugraph_amd_graph_info(const cugraph_graph_t* graph, int64_t info[8],
                                            cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(graph && info, CUGRAPH_INVALID_INPUT, "graph / info is NULL");
    auto const* g = as<Graph>(graph);
    info[0] = g->num_vertices; info[1] = g->num_edges; info[2] = g->num_global_vertices;
    info[3] = g->vertex_first; info[4] = g->store_transposed; info[5] = g->off64;
    info[6] = g->weights != nullptr; info[7] = g->renumbered;
  });
}

cugraph_error_code_t cugraph_amd_graph_get_compressed(
  cugraph_graph_t* graph, cugraph_type_erased_device_array_view_t** offsets,
  cugraph_type_erased_device_array_view_t** indices,
  cugraph_type_erased_device_array_view_t** weights,
  cugraph_type_erased_device_array_view_t** number_map, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(graph && offsets && indices && weights && number_map, CUGRAPH_INVALID_INPUT, "NULL argument");
    auto* g  = as<Graph>(graph);
    *offsets = reinterpret_cast<cugraph_type_erased_device_array_view_t*>(
      new_view(g->offsets, static_cast<size_t>(g->num_vertices + 1), g->off64 ? INT64 : INT32));
    *indices = reinterpret_cast<cugraph_type_erased_device_array_view_t*>(
      new_view(g->indices, static_cast<size_t>(g->num_edges), INT32));
    *weights = g->weights ? reinterpret_cast<cugraph_type_erased_device_array_view_t*>(new_view(
                              g->weights, static_cast<size_t>(g->num_edges), g->weight_type))
                          : nullptr;
    *number_map = g->number_map ? reinterpret_cast<cugraph_type_erased_device_array_view_t*>(new_view(
                                    g->number_map, static_cast<size_t>(g->num_vertices), g->vertex_type))
                                : nullptr;
  });
}

cugraph_error_code_t cugraph_amd_graph_source_weight_sums(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  cugraph_type_erased_device_array_view_t* sums, cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto* g = as<Graph>(graph);
    auto* s = as<ArrayView>(sums);
    CGA_EXPECTS(g && s, CUGRAPH_INVALID_INPUT, "NULL argument");
    CGA_EXPECTS(g->store_transposed, CUGRAPH_INVALID_INPUT, "graph must be stored transposed");
    CGA_EXPECTS(s->type == FLOAT64 && static_cast<int64_t>(s->size) == g->num_global_vertices,
                CUGRAPH_INVALID_INPUT, "sums must be FLOAT64 with num_global_vertices elements");
    hipStream_t st = as<Handle>(handle)->stream;
    compute_source_sums(st, *g, static_cast<double*>(s->data));
    CGA_HIP(hipStreamSynchronize(st));
  });
}

// ------------------------------------------------------------------ R-MAT
cugraph_error_code_t cugraph_amd_generate_rmat_edgelist(
  const cugraph_resource_handle_t* handle, size_t scale, uint64_t first_edge, uint64_t seed,
  double a, double b, double c, bool_t scramble_vertex_ids,
  cugraph_type_erased_device_array_view_t* src, cugraph_type_erased_device_array_view_t* dst,
  cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto* s = as<ArrayView>(src);
    auto* d = as<ArrayView>(dst);
    CGA_EXPECTS(s && d && s->type == INT32 && d->type == INT32 && s->size == d->size,
                CUGRAPH_INVALID_INPUT, "src/dst must be INT32 views of equal size");
    // generate_rmat_edgelist.cu:48-52
    CGA_EXPECTS(scale >= 1 && scale <= 30, CUGRAPH_INVALID_INPUT,
                "Invalid input argument: scale too large for vertex_t.");
    CGA_EXPECTS(a >= 0.0 && b >= 0.0 && c >= 0.0 && a + b + c <= 1.0, CUGRAPH_INVALID_INPUT,
                "Invalid input argument: a, b, c should be non-negative and a + b + c should not "
                "be larger than 1.0.");
    double ab = a + b;
    auto thr  = [](double p) { return static_cast<uint32_t>(std::floor(p * 16777216.0)); };
    uint32_t t_ab = thr(ab), t_a = thr(ab > 0.0 ? a / ab : 0.0),
             t_c = thr((1.0 - ab) > 0.0 ? c / (1.0 - ab) : 0.0);
    hipStream_t st = as<Handle>(handle)->stream;
    rmat_generate(st, static_cast<int>(scale), first_edge, s->size, seed, t_ab, t_a, t_c,
                  scramble_vertex_ids != FALSE, static_cast<int32_t*>(s->data),
                  static_cast<int32_t*>(d->data));
    CGA_HIP(hipStreamSynchronize(st));
  });
}

cugraph_error_code_t cugraph_amd_generate_rmat_weights(
  const cugraph_resource_handle_t* handle, uint64_t first_edge, uint64_t seed,
  cugraph_type_erased_device_array_view_t* weights, cugraph_error_t** error)
{
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  return guarded(error, [&] {
    auto* w = as<ArrayView>(weights);
    CGA_EXPECTS(w && is_float_type(w->type), CUGRAPH_INVALID_INPUT, "weights must be FLOAT32 or FLOAT64");
    hipStream_t st = as<Handle>(handle)->stream;
    rmat_weights(st, first_edge, w->size, seed, w->data, w->type == FLOAT64);
    CGA_HIP(hipStreamSynchronize(st));
  });
}

// ------------------------------------------------------------------ centrality result
cugraph_type_erased_device_array_view_t* cugraph_centrality_result_get_vertices(cugraph_centrality_result_t* result)
{
  auto* r = as<CentralityResult>(result);
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(
    new_view(r->vertex_ids->data, r->vertex_ids->size, r->vertex_ids->type));
}
cugraph_type_erased_device_array_view_t* cugraph_centrality_result_get_values(cugraph_centrality_result_t* result)
{
  auto* r = as<CentralityResult>(result);
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(
    new_view(r->values->data, r->values->size, r->values->type));
}
size_t cugraph_centrality_result_get_num_iterations(cugraph_centrality_result_t* result) { return as<CentralityResult>(result)->num_iterations; }
bool_t cugraph_centrality_result_converged(cugraph_centrality_result_t* result) { return as<CentralityResult>(result)->converged ? TRUE : FALSE; }
void cugraph_centrality_result_free(cugraph_centrality_result_t* result) { delete as<CentralityResult>(result); }

// ------------------------------------------------------------------ Katz / eigenvector / HITS
cugraph_error_code_t cugraph_katz_centrality(const cugraph_resource_handle_t* handle,
                                             cugraph_graph_t* graph,
                                             const cugraph_type_erased_device_array_view_t* betas,
                                             double alpha, double beta, double epsilon,
                                             size_t max_iterations, bool_t do_expensive_check,
                                             cugraph_centrality_result_t** result,
                                             cugraph_error_t** error)
{
  if (error) *error = nullptr;
  if (result) *result = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  if (!graph || !result) return fail(error, CUGRAPH_INVALID_INPUT, "graph / result pointer is NULL");
  return guarded(error, [&] {
    *result = reinterpret_cast<cugraph_centrality_result_t*>(
      run_katz(*as<Handle>(handle), *as<Graph>(graph), as<ArrayView>(betas), alpha, beta, epsilon,
               max_iterations, do_expensive_check != FALSE));
  });
}

cugraph_error_code_t cugraph_eigenvector_centrality(const cugraph_resource_handle_t* handle,
                                                    cugraph_graph_t* graph, double epsilon,
                                                    size_t max_iterations, bool_t do_expensive_check,
                                                    cugraph_centrality_result_t** result,
                                                    cugraph_error_t** error)
{
  if (error) *error = nullptr;
  if (result) *result = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  if (!graph || !result) return fail(error, CUGRAPH_INVALID_INPUT, "graph / result pointer is NULL");
  return guarded(error, [&] {
    *result = reinterpret_cast<cugraph_centrality_result_t*>(run_eigenvector(
      *as<Handle>(handle), *as<Graph>(graph), epsilon, max_iterations, do_expensive_check != FALSE));
  });
}

cugraph_error_code_t cugraph_hits(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, double epsilon,
  size_t max_iterations, const cugraph_type_erased_device_array_view_t* initial_hubs_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_hubs_guess_values, bool_t normalize,
  bool_t do_expensive_check, cugraph_hits_result_t** result, cugraph_error_t** error)
{
  if (error) *error = nullptr;
  if (result) *result = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  if (!graph || !result) return fail(error, CUGRAPH_INVALID_INPUT, "graph / result pointer is NULL");
  return guarded(error, [&] {
    *result = reinterpret_cast<cugraph_hits_result_t*>(
      run_hits(*as<Handle>(handle), *as<Graph>(graph), epsilon, max_iterations,
               as<ArrayView>(initial_hubs_guess_vertices), as<ArrayView>(initial_hubs_guess_values),
               normalize != FALSE, do_expensive_check != FALSE));
  });
}

cugraph_type_erased_device_array_view_t* cugraph_hits_result_get_vertices(cugraph_hits_result_t* result)
{
  auto* r = as<HitsResult>(result);
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(
    new_view(r->vertex_ids->data, r->vertex_ids->size, r->vertex_ids->type));
}
cugraph_type_erased_device_array_view_t* cugraph_hits_result_get_hubs(cugraph_hits_result_t* result)
{
  auto* r = as<HitsResult>(result);
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(
    new_view(r->hubs->data, r->hubs->size, r->hubs->type));
}
cugraph_type_erased_device_array_view_t* cugraph_hits_result_get_authorities(cugraph_hits_result_t* result)
{
  auto* r = as<HitsResult>(result);
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(
    new_view(r->authorities->data, r->authorities->size, r->authorities->type));
}
double cugraph_hits_result_get_hub_score_differences(cugraph_hits_result_t* result) { return as<HitsResult>(result)->hub_score_differences; }
size_t cugraph_hits_result_get_number_of_iterations(cugraph_hits_result_t* result) { return as<HitsResult>(result)->number_of_iterations; }
void cugraph_hits_result_free(cugraph_hits_result_t* result) { delete as<HitsResult>(result); }

// ------------------------------------------------------------------ PageRank
cugraph_error_code_t cugraph_pagerank(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_vertices,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_sums,
  const cugraph_type_erased_device_array_view_t* initial_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_guess_values, double alpha,
  double epsilon, size_t max_iterations, bool_t do_expensive_check,
  cugraph_centrality_result_t** result, cugraph_error_t** error)
{
  return pagerank_entry(handle, graph, precomputed_vertex_out_weight_vertices,
                        precomputed_vertex_out_weight_sums, initial_guess_vertices,
                        initial_guess_values, nullptr, nullptr, alpha, epsilon, max_iterations,
                        do_expensive_check, true, result, error);
}

cugraph_error_code_t cugraph_pagerank_allow_nonconvergence(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_vertices,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_sums,
  const cugraph_type_erased_device_array_view_t* initial_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_guess_values, double alpha,
  double epsilon, size_t max_iterations, bool_t do_expensive_check,
  cugraph_centrality_result_t** result, cugraph_error_t** error)
{
  return pagerank_entry(handle, graph, precomputed_vertex_out_weight_vertices,
                        precomputed_vertex_out_weight_sums, initial_guess_vertices,
                        initial_guess_values, nullptr, nullptr, alpha, epsilon, max_iterations,
                        do_expensive_check, false, result, error);
}

cugraph_error_code_t cugraph_personalized_pagerank(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_vertices,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_sums,
  const cugraph_type_erased_device_array_view_t* initial_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_guess_values,
  const cugraph_type_erased_device_array_view_t* personalization_vertices,
  const cugraph_type_erased_device_array_view_t* personalization_values, double alpha,
  double epsilon, size_t max_iterations, bool_t do_expensive_check,
  cugraph_centrality_result_t** result, cugraph_error_t** error)
{
  return pagerank_entry(handle, graph, precomputed_vertex_out_weight_vertices,
                        precomputed_vertex_out_weight_sums, initial_guess_vertices,
                        initial_guess_values, personalization_vertices, personalization_values,
                        alpha, epsilon, max_iterations, do_expensive_check, true, result, error);
}

cugraph_error_code_t cugraph_personalized_pagerank_allow_nonconvergence(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_vertices,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_sums,
  const cugraph_type_erased_device_array_view_t* initial_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_guess_values,
  const cugraph_type_erased_device_array_view_t* personalization_vertices,
  const cugraph_type_erased_device_array_view_t* personalization_values, double alpha,
  double epsilon, size_t max_iterations, bool_t do_expensive_check,
  cugraph_centrality_result_t** result, cugraph_error_t** error)
{
  return pagerank_entry(handle, graph, precomputed_vertex_out_weight_vertices,
                        precomputed_vertex_out_weight_sums, initial_guess_vertices,
                        initial_guess_values, personalization_vertices, personalization_values,
                        alpha, epsilon, max_iterations, do_expensive_check, false, result, error);
}

// ------------------------------------------------------------------ session
cugraph_error_code_t cugraph_amd_pagerank_session_create(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* pre_v,
  const cugraph_type_erased_device_array_view_t* pre_s,
  const cugraph_type_erased_device_array_view_t* guess_v,
  const cugraph_type_erased_device_array_view_t* guess_x,
  const cugraph_type_erased_device_array_view_t* pers_v,
  const cugraph_type_erased_device_array_view_t* pers_x, double alpha, double epsilon,
  bool_t do_expensive_check, cugraph_amd_pagerank_session_t** session, cugraph_error_t** error)
{
  if (session) *session = nullptr;
  if (error) *error = nullptr;
  if (!handle) return fail(error, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  if (!graph || !session) return fail(error, CUGRAPH_INVALID_INPUT, "graph / session pointer is NULL");
  auto* g = as<Graph>(graph);
  cugraph_error_code_t rc;
  rc = check_pair_types(g, as<ArrayView>(pre_v), as<ArrayView>(pre_s),
                        "vertex type of graph and precomputed_vertex_out_weight_vertices must match",
                        "vertex type of graph and precomputed_vertex_out_weight_sums must match", error);
  if (rc != CUGRAPH_SUCCESS) return rc;
  rc = check_pair_types(g, as<ArrayView>(guess_v), as<ArrayView>(guess_x),
                        "vertex type of graph and initial_guess_vertices must match",
                        "vertex type of graph and initial_guess_values must match", error);
  if (rc != CUGRAPH_SUCCESS) return rc;
  rc = check_pair_types(g, as<ArrayView>(pers_v), as<ArrayView>(pers_x),
                        "vertex type of graph and personalization_vector must match",
                        "vertex type of graph and personalization_vector must match", error);
  if (rc != CUGRAPH_SUCCESS) return rc;
  return guarded(error, [&] {
    PagerankArgs a;
    a.pre_vertices = as<ArrayView>(pre_v);     a.pre_sums     = as<ArrayView>(pre_s);
    a.guess_vertices = as<ArrayView>(guess_v); a.guess_values = as<ArrayView>(guess_x);
    a.pers_vertices = as<ArrayView>(pers_v);   a.pers_values  = as<ArrayView>(pers_x);
    a.alpha = alpha; a.epsilon = epsilon; a.do_expensive_check = do_expensive_check != FALSE;
    *session = reinterpret_cast<cugraph_amd_pagerank_session_t*>(
      session_create(*as<Handle>(handle), *g, a));
  });
}

cugraph_error_code_t cugraph_amd_pagerank_session_iterate(cugraph_amd_pagerank_session_t* session,
                                                          size_t n, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(session, CUGRAPH_INVALID_INPUT, "session is NULL");
    session_iterate(*as<Session>(session), n);
  });
}

cugraph_error_code_t cugraph_amd_pagerank_session_iterate_phase(cugraph_amd_pagerank_session_t* session,
                                                                int phase, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(session, CUGRAPH_INVALID_INPUT, "session is NULL");
    session_iterate_phase(*as<Session>(session), phase);
  });
}

cugraph_error_code_t cugraph_amd_graph_set_hot_slots(cugraph_graph_t* graph, int64_t hot_slots,
                                                     cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(graph, CUGRAPH_INVALID_INPUT, "graph is NULL");
    Graph* g = as<Graph>(graph);
    CGA_EXPECTS(g->x_pos != nullptr && hot_slots >= 0 && hot_slots <= g->num_source_slots, CUGRAPH_INVALID_INPUT,
                "hot_slots: needs a graph with a compact source space and 0 <= hot_slots <= num_source_slots");
    CGA_EXPECTS(g->plan == nullptr, CUGRAPH_INVALID_INPUT,
                "hot_slots must be set before the first PageRank call on the graph");
    g->hot_slots = hot_slots;
  });
}

int64_t cugraph_amd_xs_window_slots(int64_t num_source_slots) { return xs_window_slots(num_source_slots); }

cugraph_error_code_t cugraph_amd_pagerank_session_timed_iterate(
  cugraph_amd_pagerank_session_t* session, size_t n, float* iteration_ms, float* spmv_kernel_ms,
  cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(session, CUGRAPH_INVALID_INPUT, "session is NULL");
    session_timed_iterate(*as<Session>(session), n, iteration_ms, spmv_kernel_ms);
  });
}

cugraph_error_code_t cugraph_amd_pagerank_session_exchange_buffers(
  cugraph_amd_pagerank_session_t* session, cugraph_type_erased_device_array_view_t** x_full,
  cugraph_type_erased_device_array_view_t** scalars, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(session && x_full && scalars, CUGRAPH_INVALID_INPUT, "NULL argument");
    auto& s = *as<Session>(session);
    void *x = nullptr, *sc = nullptr;
    session_exchange_buffers(s, &x, &sc);
    *x_full  = reinterpret_cast<cugraph_type_erased_device_array_view_t*>(
      new_view(x, static_cast<size_t>(session_num_global(s)), session_is_f64(s) ? FLOAT64 : FLOAT32));
    *scalars = reinterpret_cast<cugraph_type_erased_device_array_view_t*>(new_view(sc, 4, FLOAT64));
  });
}

cugraph_error_code_t cugraph_amd_pagerank_session_status(cugraph_amd_pagerank_session_t* session,
                                                         size_t* num_iterations, bool_t* stopped,
                                                         double* l1_delta, double* dangling_mass,
                                                         cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(session, CUGRAPH_INVALID_INPUT, "session is NULL");
    bool st = false;
    session_status(*as<Session>(session), num_iterations, &st, l1_delta, dangling_mass);
    if (stopped) *stopped = st ? TRUE : FALSE;
  });
}

cugraph_error_code_t cugraph_amd_pagerank_session_result(cugraph_amd_pagerank_session_t* session,
                                                         size_t max_iterations,
                                                         cugraph_centrality_result_t** result,
                                                         cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    CGA_EXPECTS(session && result, CUGRAPH_INVALID_INPUT, "NULL argument");
    *result = reinterpret_cast<cugraph_centrality_result_t*>(
      session_result(*as<Session>(session), max_iterations));
  });
}

cugraph_error_code_t cugraph_amd_pagerank_session_info(cugraph_amd_pagerank_session_t* session,
                                                       int64_t info[8], cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(session && info, CUGRAPH_INVALID_INPUT, "NULL argument");
    session_info(*as<Session>(session), info);
  });
}

void cugraph_amd_pagerank_session_free(cugraph_amd_pagerank_session_t* session)
{
  session_destroy(as<Session>(session));
}

const char* cugraph_amd_version(void) { return "cugraph_amd 0.1 (gfx950)"; }

#ifdef CGA_TIMING_AIDS
// timing-aid build only (make AIDS=1; not in include/cugraph_amd_c.h): out[0] = gathers of region-0
// launches of xs_kernel that read beyond the region, out[1] = region-0 launches checked
int cugraph_amd_aids_region_check(unsigned long long out[2])
{
  try {
    cga::xs_region_check_read(out);
    return 0;
  } catch (...) {
    return 1;
  }
}
// out[0] = accesses of xs_kernel outside its arrays so far (recorded and redirected instead of
// faulting), out[1..5] = kind (1 entry stream, 2 gather, 3 partials), two values, workgroup and
// tile | slice << 32 of the first one
int cugraph_amd_aids_bounds_check(unsigned long long out[8])
{
  try {
    cga::xs_bounds_check_read(out);
    return 0;
  } catch (...) {
    return 1;
  }
}
#endif

}  // extern "C"
