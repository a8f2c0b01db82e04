/*
 * cugraph_amd_c.h -- C ABI of libcugraph_amd_c.so, the MI355X-native drop-in for the PageRank
 * path of libcugraph_c.
 *
 * Part 1 re-declares, with identical names, argument order and ownership rules, the subset of
 * the reference's C API that pylibcugraph binds for `pagerank` / `personalized_pagerank` /
 * `SGGraph` / `ResourceHandle` (each block cites the reference header it replaces; paths are
 * relative to the reference tree).  A pylibcugraph built against these symbols needs no source
 * change for this path -- see INTEGRATION.md.
 *
 * Part 2 (prefix cugraph_amd_) has no counterpart in the reference: R-MAT generation for the
 * benchmark, read-only access to the compressed graph for parity tests, a 1D-partitioned local
 * graph, and a step-wise PageRank session used by the multi-GPU driver and bench.py.
 *
 * Conventions (same as the reference, cpp/src/c_api/utils.hpp:23-59): no exception crosses the
 * ABI; every fallible call returns a cugraph_error_code_t and, on failure, allocates *error
 * (free with cugraph_error_free); *error is NULL on success.  All pointers inside views are
 * device pointers valid on the handle's device.  Calls are synchronous with respect to the
 * handle's HIP stream unless stated otherwise; one handle per host thread / GPU.
 */
#ifndef CUGRAPH_AMD_C_H
#define CUGRAPH_AMD_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ===================================================================== */
/* Part 1 -- the reference's own entry points for this path               */
/* ===================================================================== */

/* ---- cpp/include/cugraph_c/error.h:25-52 ---- */
typedef enum cugraph_error_code_ {
  CUGRAPH_SUCCESS = 0,
  CUGRAPH_UNKNOWN_ERROR,
  CUGRAPH_INVALID_HANDLE,
  CUGRAPH_ALLOC_ERROR,
  CUGRAPH_INVALID_INPUT,
  CUGRAPH_NOT_IMPLEMENTED,
  CUGRAPH_UNSUPPORTED_TYPE_COMBINATION
} cugraph_error_code_t;

typedef struct cugraph_error_ { int32_t align_; } cugraph_error_t;

const char* cugraph_error_message(const cugraph_error_t* error); /* error.h:45 */
void cugraph_error_free(cugraph_error_t* error);                 /* error.h:52 */

/* ---- cpp/include/cugraph_c/resource_handle.h:28-89 ---- */
typedef enum bool_ { FALSE = 0, TRUE = 1 } bool_t;
typedef int8_t byte_t;
typedef enum data_type_id_ { INT32 = 0, INT64, FLOAT32, FLOAT64, SIZE_T, NTYPES } data_type_id_t;
typedef data_type_id_t cugraph_data_type_id_t;

typedef struct cugraph_resource_handle_ { int32_t align_; } cugraph_resource_handle_t;

/* resource_handle.h:58.  The reference takes a raft::handle_t*; here the argument must be NULL
 * (the library makes a handle on the current HIP device and the NULL stream).  Use
 * cugraph_amd_create_resource_handle (part 2) to bind a stream or a partition rank. */
cugraph_resource_handle_t* cugraph_create_resource_handle(void* raft_handle);
int cugraph_resource_handle_get_comm_size(const cugraph_resource_handle_t* handle); /* :70 */
int cugraph_resource_handle_get_rank(const cugraph_resource_handle_t* handle);      /* :82 */
void cugraph_free_resource_handle(cugraph_resource_handle_t* handle);               /* :89 */

/* ---- cpp/include/cugraph_c/array.h:25-331 ---- */
typedef struct { int32_t align_; } cugraph_type_erased_device_array_t;
typedef struct { int32_t align_; } cugraph_type_erased_device_array_view_t;
typedef struct { int32_t align_; } cugraph_type_erased_host_array_t;
typedef struct { int32_t align_; } cugraph_type_erased_host_array_view_t;

cugraph_error_code_t cugraph_type_erased_device_array_create( /* array.h:52 */
  const cugraph_resource_handle_t* handle, size_t n_elems, cugraph_data_type_id_t dtype,
  cugraph_type_erased_device_array_t** array, cugraph_error_t** error);
cugraph_error_code_t cugraph_type_erased_device_array_create_from_view( /* array.h:71 */
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* view,
  cugraph_type_erased_device_array_t** array, cugraph_error_t** error);
void cugraph_type_erased_device_array_free(cugraph_type_erased_device_array_t* p); /* :82 */
cugraph_type_erased_device_array_view_t* cugraph_type_erased_device_array_view(    /* :105 */
  cugraph_type_erased_device_array_t* array);
cugraph_error_code_t cugraph_type_erased_device_array_view_as_type( /* array.h:126 */
  cugraph_type_erased_device_array_t* array, cugraph_data_type_id_t dtype,
  cugraph_type_erased_device_array_view_t** result_view, cugraph_error_t** error);
/* array.h:141 -- views are heap objects the caller frees; they never own the memory */
cugraph_type_erased_device_array_view_t* cugraph_type_erased_device_array_view_create(
  void* pointer, size_t n_elems, cugraph_data_type_id_t dtype);
void cugraph_type_erased_device_array_view_free(cugraph_type_erased_device_array_view_t* p);
size_t cugraph_type_erased_device_array_view_size(const cugraph_type_erased_device_array_view_t* p);
cugraph_data_type_id_t cugraph_type_erased_device_array_view_type(
  const cugraph_type_erased_device_array_view_t* p);
const void* cugraph_type_erased_device_array_view_pointer(
  const cugraph_type_erased_device_array_view_t* p);

cugraph_error_code_t cugraph_type_erased_host_array_create( /* array.h:188 */
  const cugraph_resource_handle_t* handle, size_t n_elems, cugraph_data_type_id_t dtype,
  cugraph_type_erased_host_array_t** array, cugraph_error_t** error);
void cugraph_type_erased_host_array_free(cugraph_type_erased_host_array_t* p);
cugraph_type_erased_host_array_view_t* cugraph_type_erased_host_array_view(
  cugraph_type_erased_host_array_t* array);
cugraph_type_erased_host_array_view_t* cugraph_type_erased_host_array_view_create(
  void* pointer, size_t n_elems, cugraph_data_type_id_t dtype);
void cugraph_type_erased_host_array_view_free(cugraph_type_erased_host_array_view_t* p);
size_t cugraph_type_erased_host_array_size(const cugraph_type_erased_host_array_view_t* p);
cugraph_data_type_id_t cugraph_type_erased_host_array_type(
  const cugraph_type_erased_host_array_view_t* p);
void* cugraph_type_erased_host_array_pointer(const cugraph_type_erased_host_array_view_t* p);
cugraph_error_code_t cugraph_type_erased_host_array_view_copy( /* array.h:279 */
  const cugraph_resource_handle_t* handle, cugraph_type_erased_host_array_view_t* dst,
  const cugraph_type_erased_host_array_view_t* src, cugraph_error_t** error);

cugraph_error_code_t cugraph_type_erased_device_array_view_copy_from_host( /* array.h:295 */
  const cugraph_resource_handle_t* handle, cugraph_type_erased_device_array_view_t* dst,
  const byte_t* h_src, cugraph_error_t** error);
cugraph_error_code_t cugraph_type_erased_device_array_view_copy_to_host( /* array.h:311 */
  const cugraph_resource_handle_t* handle, byte_t* h_dst,
  const cugraph_type_erased_device_array_view_t* src, cugraph_error_t** error);
cugraph_error_code_t cugraph_type_erased_device_array_view_copy( /* array.h:327 */
  const cugraph_resource_handle_t* handle, cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* src, cugraph_error_t** error);

/* ---- cpp/include/cugraph_c/graph.h:25-36, 66-131, 161-213, 321-330 ---- */
typedef struct { int32_t align_; } cugraph_graph_t;
typedef struct {
  bool_t is_symmetric;
  bool_t is_multigraph;
} cugraph_graph_properties_t;

/* graph.h:116 (impl cpp/src/c_api/graph_sg.cpp:508-633).  Inputs are copied; the caller keeps
 * its arrays.  edge_ids / edge_type_ids are accepted for signature compatibility and ignored
 * (PageRank never reads them).  renumber=TRUE relabels vertices by descending in-degree and the
 * result carries the external ids (renumber_edgelist_impl.cuh:238-607); the ids may be any int32 /
 * int64 values.  do_expensive_check verifies what the reference verifies
 * (create_graph_from_edgelist_impl.cuh:72-126, 895-921): no duplicates in `vertices`, consecutive
 * ids from 0 without renumbering, the is_symmetric and is_multigraph flags against the edge list;
 * a failure is CUGRAPH_UNKNOWN_ERROR with the reference's message (graph_sg.cpp:627-629). */
cugraph_error_code_t cugraph_graph_create_sg(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* vertices,
  const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed,
  bool_t renumber, bool_t drop_self_loops, bool_t drop_multi_edges, bool_t do_expensive_check,
  cugraph_graph_t** graph, cugraph_error_t** error);

/* graph.h:66 (deprecated spelling, still what cpp/tests/c_api/test_utils.cpp:101 calls) */
cugraph_error_code_t cugraph_sg_graph_create(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed,
  bool_t renumber, bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error);

/* graph.h:201 / :161 (impl graph_sg.cpp:666-789): offsets are over sources (CSR). */
cugraph_error_code_t cugraph_graph_create_sg_from_csr(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* offsets,
  const cugraph_type_erased_device_array_view_t* indices,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed,
  bool_t renumber, bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error);
cugraph_error_code_t cugraph_sg_graph_create_from_csr(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* offsets,
  const cugraph_type_erased_device_array_view_t* indices,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed,
  bool_t renumber, bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error);

void cugraph_graph_free(cugraph_graph_t* graph);    /* graph.h:321 */
void cugraph_sg_graph_free(cugraph_graph_t* graph); /* graph.h:330 */

/* ---- cpp/include/cugraph_c/centrality_algorithms.h:35-83, 122-307 ---- */
typedef struct { int32_t align_; } cugraph_centrality_result_t;

/* each getter returns a NEW heap view the caller frees (cpp/src/c_api/centrality_result.cpp:21-35) */
cugraph_type_erased_device_array_view_t* cugraph_centrality_result_get_vertices(
  cugraph_centrality_result_t* result);
cugraph_type_erased_device_array_view_t* cugraph_centrality_result_get_values(
  cugraph_centrality_result_t* result);
size_t cugraph_centrality_result_get_num_iterations(cugraph_centrality_result_t* result);
bool_t cugraph_centrality_result_converged(cugraph_centrality_result_t* result);
void cugraph_centrality_result_free(cugraph_centrality_result_t* result);

/* centrality_algorithms.h:122 (impl cpp/src/c_api/pagerank.cpp:254-321): returns
 * CUGRAPH_UNKNOWN_ERROR "PageRank failed to converge." when not converged, with *result
 * still populated.  epsilon is compared with the L1 delta as is (pagerank_impl.cuh:330). */
cugraph_error_code_t cugraph_pagerank(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_vertices,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_sums,
  const cugraph_type_erased_device_array_view_t* initial_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_guess_values, double alpha,
  double epsilon, size_t max_iterations, bool_t do_expensive_check,
  cugraph_centrality_result_t** result, cugraph_error_t** error);

/* centrality_algorithms.h:177 (impl pagerank.cpp:323-383) */
cugraph_error_code_t cugraph_pagerank_allow_nonconvergence(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_vertices,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_sums,
  const cugraph_type_erased_device_array_view_t* initial_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_guess_values, double alpha,
  double epsilon, size_t max_iterations, bool_t do_expensive_check,
  cugraph_centrality_result_t** result, cugraph_error_t** error);

/* centrality_algorithms.h:236 (impl pagerank.cpp:385-471) */
cugraph_error_code_t cugraph_personalized_pagerank(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_vertices,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_sums,
  const cugraph_type_erased_device_array_view_t* initial_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_guess_values,
  const cugraph_type_erased_device_array_view_t* personalization_vertices,
  const cugraph_type_erased_device_array_view_t* personalization_values, double alpha,
  double epsilon, size_t max_iterations, bool_t do_expensive_check,
  cugraph_centrality_result_t** result, cugraph_error_t** error);

/* centrality_algorithms.h:293 (impl pagerank.cpp:473-552) */
cugraph_error_code_t cugraph_personalized_pagerank_allow_nonconvergence(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_vertices,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_sums,
  const cugraph_type_erased_device_array_view_t* initial_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_guess_values,
  const cugraph_type_erased_device_array_view_t* personalization_vertices,
  const cugraph_type_erased_device_array_view_t* personalization_values, double alpha,
  double epsilon, size_t max_iterations, bool_t do_expensive_check,
  cugraph_centrality_result_t** result, cugraph_error_t** error);

/* ---- sibling power iterations on the same pull SpMV ---- */

/* centrality_algorithms.h:327 (impl cpp/src/c_api/eigenvector_centrality.cpp:55-120,
 * cpp/src/centrality/eigenvector_centrality_impl.cuh:46-165): x' = normalise_L2(x + A^T x), stop when
 * sum|x' - x| < V * epsilon; failure to converge is CUGRAPH_UNKNOWN_ERROR without a result. */
cugraph_error_code_t cugraph_eigenvector_centrality(const cugraph_resource_handle_t* handle,
                                                    cugraph_graph_t* graph, double epsilon,
                                                    size_t max_iterations, bool_t do_expensive_check,
                                                    cugraph_centrality_result_t** result,
                                                    cugraph_error_t** error);

/* centrality_algorithms.h:360 (impl cpp/src/c_api/katz.cpp:60-150,
 * cpp/src/centrality/katz_centrality_impl.cuh:43-196): x' = alpha * A^T x + beta (betas[i] for the
 * vertex with external id i when given), from x = 0, stop when sum|x' - x| < epsilon; the result is
 * divided by its L2 norm (the C API always normalises). */
cugraph_error_code_t cugraph_katz_centrality(const cugraph_resource_handle_t* handle,
                                             cugraph_graph_t* graph,
                                             const cugraph_type_erased_device_array_view_t* betas,
                                             double alpha, double beta, double epsilon,
                                             size_t max_iterations, bool_t do_expensive_check,
                                             cugraph_centrality_result_t** result,
                                             cugraph_error_t** error);

/* centrality_algorithms.h:499-597 (impl cpp/src/c_api/hits.cpp, cpp/src/link_analysis/hits_impl.cuh:57-206):
 * authorities = in-sums of hubs, hubs = out-sums of authorities, both divided by their maximum,
 * stop when sum|hubs' - hubs| < V * epsilon; `normalize` divides both by their sums at the end.
 * Edge weights are ignored, as in the reference. */
typedef struct { int32_t align_; } cugraph_hits_result_t;
cugraph_type_erased_device_array_view_t* cugraph_hits_result_get_vertices(cugraph_hits_result_t* result);
cugraph_type_erased_device_array_view_t* cugraph_hits_result_get_hubs(cugraph_hits_result_t* result);
cugraph_type_erased_device_array_view_t* cugraph_hits_result_get_authorities(cugraph_hits_result_t* result);
double cugraph_hits_result_get_hub_score_differences(cugraph_hits_result_t* result);
size_t cugraph_hits_result_get_number_of_iterations(cugraph_hits_result_t* result);
void cugraph_hits_result_free(cugraph_hits_result_t* result);
cugraph_error_code_t cugraph_hits(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, double epsilon,
  size_t max_iterations, const cugraph_type_erased_device_array_view_t* initial_hubs_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_hubs_guess_values, bool_t normalize,
  bool_t do_expensive_check, cugraph_hits_result_t** result, cugraph_error_t** error);

/* ===================================================================== */
/* Part 2 -- extensions (no counterpart in the reference)                 */
/* ===================================================================== */

/* Handle bound to an existing HIP stream (hipStream_t passed as void*, NULL = the NULL stream)
 * and, for the 1D multi-GPU partition, this process's rank / world size (replaces the raft
 * comms handle the reference gets from Dask, python/cugraph/cugraph/dask/comms/comms.py:89-158). */
cugraph_resource_handle_t* cugraph_amd_create_resource_handle(void* hip_stream, int rank,
                                                              int comm_size);


/* ---- multi-GPU graphs under the reference's names ----
 * cugraph_mg_graph_create (cpp/include/cugraph_c/graph.h:242, deprecated there but what
 * cpp/tests/c_api/mg_test_utils.cpp:194 calls) and cugraph_graph_create_mg (graph.h:299): every rank (one
 * process per GPU, handle made by cugraph_amd_create_resource_handle(stream, rank, size)) passes ANY share of
 * the edge list -- an empty one is fine -- and the graph that comes back holds the edges whose DESTINATION the
 * rank owns (1D partition, equal vertex ranges).  cugraph_pagerank*, cugraph_personalized_pagerank* on such a
 * graph run the partitioned power iteration and return the (vertex, rank) pairs of the vertices this rank owns,
 * as cpp/src/c_api/pagerank.cpp does with an MG handle (initial guess and personalization: any share of the
 * list per rank, shuffled to the owners below this boundary, pagerank.cpp:140-144; every rank must make the
 * call).  Differences from the reference, all checked and reported: vertex ids must be 0 .. V-1 (V = largest
 * id + 1 over all ranks; the reference renumbers arbitrary ids), no edge ids / types, no drop_* flags, no
 * precomputed out-weight sums (the graph keeps its own), vertices[] only extends the id range.
 *
 * The library has no communication dependency of its own: the collectives come from the caller, once per
 * handle.  All pointers are DEVICE pointers; a callback must order itself after the work already enqueued on
 * `hip_stream` and leave its result visible to work enqueued on `hip_stream` after it returns (e.g. call RCCL
 * on that stream); it returns 0 on success.  `send` of all_gather_v may be this rank's own place inside
 * `recv`.  op: 0 = sum, 1 = max.  No counterpart in the reference (raft::comms plays this role there). */
typedef struct {
  void* context;
  int (*all_reduce_f64)(void* context, double* buffer, size_t count, int op, void* hip_stream);
  int (*all_gather_v)(void* context, const void* send, void* recv, const size_t* bytes_per_rank, void* hip_stream);
  int (*all_to_all_v)(void* context, const void* send, const size_t* send_bytes_per_rank, void* recv,
                      const size_t* recv_bytes_per_rank, void* hip_stream);
} cugraph_amd_comm_t;

cugraph_error_code_t cugraph_amd_resource_handle_set_comm(cugraph_resource_handle_t* handle,
                                                          const cugraph_amd_comm_t* comm, cugraph_error_t** error);

cugraph_error_code_t cugraph_mg_graph_create(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* src, const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed, size_t num_edges,
  bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error);

cugraph_error_code_t cugraph_graph_create_mg(
  cugraph_resource_handle_t const* handle, cugraph_graph_properties_t const* properties,
  cugraph_type_erased_device_array_view_t const* const* vertices,
  cugraph_type_erased_device_array_view_t const* const* src,
  cugraph_type_erased_device_array_view_t const* const* dst,
  cugraph_type_erased_device_array_view_t const* const* weights,
  cugraph_type_erased_device_array_view_t const* const* edge_ids,
  cugraph_type_erased_device_array_view_t const* const* edge_type_ids, bool_t store_transposed,
  size_t num_arrays, bool_t drop_self_loops, bool_t drop_multi_edges, bool_t do_expensive_check,
  cugraph_graph_t** graph, cugraph_error_t** error);

/* graph.h:337 (deprecated twin of cugraph_graph_free) */
void cugraph_mg_graph_free(cugraph_graph_t* graph);

/* Local part of a 1D destination-partitioned graph: this rank owns destinations
 * [vertex_first, vertex_last) of num_global_vertices and must be given exactly the edges whose
 * dst lies in that range (any src).  Always stored transposed (CSC, global source ids). */
cugraph_error_code_t cugraph_amd_graph_create_local(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights, int64_t vertex_first,
  int64_t vertex_last, int64_t num_global_vertices, cugraph_graph_t** graph,
  cugraph_error_t** error);

/* As cugraph_amd_graph_create_local, with a COMPACT SOURCE SPACE: `src_slots` holds, instead of
 * vertex ids, slots of the vector the ranks all-gather, in [0, num_source_slots); x_pos (INT32,
 * one per local vertex, in the order of the local destination ids) is the slot this rank writes
 * each of its vertices' value to, or -1 for a vertex no edge leaves (nobody ever gathers it).
 * The multi-GPU driver numbers only vertices WITH out-edges (27 M of the 67 M of an R-MAT
 * scale-26 graph), which cuts the per-iteration all-gather to that fraction.  num_source_slots
 * is P equal per-rank chunks; rank r's slots are chunk r. */
cugraph_error_code_t cugraph_amd_graph_create_local_compact(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* src_slots,
  const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights, int64_t vertex_first,
  int64_t vertex_last, int64_t num_global_vertices, int64_t num_source_slots,
  const cugraph_type_erased_device_array_view_t* x_pos, cugraph_graph_t** graph,
  cugraph_error_t** error);

/* sums[v] (FLOAT64, one per vertex id in [0, size of sums)) = sum of the weights (1 per edge when
 * weights is NULL) of the edges of a raw edge list whose source is v.  The multi-GPU driver
 * all-reduces them: out-weight sums, and out-edge counts for the compact source space. */
cugraph_error_code_t cugraph_amd_edgelist_source_weight_sums(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* weights,
  cugraph_type_erased_device_array_view_t* sums, cugraph_error_t** error);

/* The same sums for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.
 * The staging buffer cugraph_amd_edgelist_source_weight_sums allocates per call is a static device
 * array here, so the call allocates nothing, copies nothing to the host and does not synchronise:
 *   - at most 4,194,304 edges per call (CUGRAPH_INVALID_INPUT beyond that);
 *   - ids outside [0, size of sums) cannot be reported as an error without a read-back: they are
 *     skipped, and their number is stored to num_skipped[0] on the device (INT64 view, may be NULL);
 *   - all calls on one GPU share the static array: calls that overlap on one GPU race. */
cugraph_error_code_t cugraph_amd_edgelist_source_weight_sums_noalloc(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* weights,
  cugraph_type_erased_device_array_view_t* sums, cugraph_type_erased_device_array_view_t* num_skipped,
  cugraph_error_t** error);

/* Descending in-degree order of the destinations [vertex_first, vertex_last) a rank owns, from
 * the destination ids (INT32/INT64, global) of the edges routed to it:
 *   order[new] = old - vertex_first,  rank[old - vertex_first] = new   (INT32 views, stable).
 * The multi-GPU driver relabels every partition with it (ids stay inside their owner's range), so
 * that the local SpMV runs on contiguous degree buckets and XS tiles like the single-GPU path
 * (the reference renumbers by degree inside each partition too: renumber_edgelist,
 * cpp/src/structure/renumber_edgelist_impl.cuh, compute_renumber_map). */
cugraph_error_code_t cugraph_amd_partition_degree_order(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* dst,
  int64_t vertex_first, int64_t vertex_last, cugraph_type_erased_device_array_view_t* order,
  cugraph_type_erased_device_array_view_t* rank, cugraph_error_t** error);

/* The send side of the multi-GPU edge shuffle: one rank's share of the edge list grouped by the rank
 * that owns each edge's DESTINATION (rank r owns [bounds[r], bounds[r+1]); bounds: host array of
 * num_ranks + 1 ascending ids), input order kept inside a group; counts[r] (host, num_ranks entries)
 * = edges that go to rank r, i.e. the split sizes of the all-to-all that follows.  src / dst and
 * their outputs: device views of one type (INT32 or INT64) and one length; weights / weights_out
 * both NULL or both FLOAT32 / FLOAT64 views of that length.  Replaces, for this path, the
 * reference's groupby_and_count + shuffle of an edge list over its 2D grid
 * (cpp/src/detail/shuffle_vertex_pairs.cu; cpp/include/cugraph/utilities/shuffle_comm.cuh). */
cugraph_error_code_t cugraph_amd_partition_group_edges(
  const cugraph_resource_handle_t* handle, const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* dst, const cugraph_type_erased_device_array_view_t* weights,
  const int64_t* bounds, size_t num_ranks, cugraph_type_erased_device_array_view_t* src_out,
  cugraph_type_erased_device_array_view_t* dst_out, cugraph_type_erased_device_array_view_t* weights_out,
  int64_t* counts, cugraph_error_t** error);

/* Shape of a graph: info[0]=num_vertices (local majors), [1]=num_edges, [2]=num_global_vertices,
 * [3]=vertex_first, [4]=store_transposed, [5]=offsets are int64, [6]=weighted, [7]=renumbered */
cugraph_error_code_t cugraph_amd_graph_info(const cugraph_graph_t* graph, int64_t info[8],
                                            cugraph_error_t** error);

/* Read-only views of the compressed arrays the graph holds (new heap views, caller frees each;
 * weights view is NULL for an unweighted graph, number_map view NULL when ids are 0..V-1).
 * For parity tests of the COO->CSC build. */
cugraph_error_code_t cugraph_amd_graph_get_compressed(
  cugraph_graph_t* graph, cugraph_type_erased_device_array_view_t** offsets,
  cugraph_type_erased_device_array_view_t** indices,
  cugraph_type_erased_device_array_view_t** weights,
  cugraph_type_erased_device_array_view_t** number_map, cugraph_error_t** error);

/* Sum over this graph's edges of the weight (1 if unweighted) per SOURCE vertex, written to
 * `sums` (FLOAT64, num_global_vertices elements).  The multi-GPU driver all-reduces these and
 * hands each rank its slice as precomputed_vertex_out_weight_sums. */
cugraph_error_code_t cugraph_amd_graph_source_weight_sums(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  cugraph_type_erased_device_array_view_t* sums, cugraph_error_t** error);

/* R-MAT edges [first_edge, first_edge + n) of the stream (seed, a, b, c) into INT32 views src/dst
 * of n elements (quadrant rule of cpp/src/generators/generate_rmat_edgelist.cu:85-102; own
 * Philox4x32-10 stream so that the CPU oracle emits the identical list). */
cugraph_error_code_t cugraph_amd_generate_rmat_edgelist(
  const cugraph_resource_handle_t* handle, size_t scale, uint64_t first_edge, uint64_t seed,
  double a, double b, double c, bool_t scramble_vertex_ids,
  cugraph_type_erased_device_array_view_t* src, cugraph_type_erased_device_array_view_t* dst,
  cugraph_error_t** error);

/* Weights in (0,1] for edges [first_edge, first_edge + n), FLOAT32 or FLOAT64 view. */
cugraph_error_code_t cugraph_amd_generate_rmat_weights(
  const cugraph_resource_handle_t* handle, uint64_t first_edge, uint64_t seed,
  cugraph_type_erased_device_array_view_t* weights, cugraph_error_t** error);

/* ---- step-wise PageRank session ----
 * Same arguments as cugraph_personalized_pagerank_allow_nonconvergence (personalization may be
 * NULL).  create() allocates the rank / source-value buffers and computes out-weight sums;
 * iterate() enqueues up to n iterations on the handle's stream WITHOUT synchronising: once
 * the L1 delta of the previous iteration is below epsilon the remaining enqueued iterations
 * are no-ops on the device.  status() synchronises.  cugraph_pagerank* are thin loops over this. */
typedef struct { int32_t align_; } cugraph_amd_pagerank_session_t;

cugraph_error_code_t cugraph_amd_pagerank_session_create(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_vertices,
  const cugraph_type_erased_device_array_view_t* precomputed_vertex_out_weight_sums,
  const cugraph_type_erased_device_array_view_t* initial_guess_vertices,
  const cugraph_type_erased_device_array_view_t* initial_guess_values,
  const cugraph_type_erased_device_array_view_t* personalization_vertices,
  const cugraph_type_erased_device_array_view_t* personalization_values, double alpha,
  double epsilon, bool_t do_expensive_check, cugraph_amd_pagerank_session_t** session,
  cugraph_error_t** error);

/* Enqueue n iterations (each: SpMV with fused epilogue + scalar finalize). */
cugraph_error_code_t cugraph_amd_pagerank_session_iterate(cugraph_amd_pagerank_session_t* session,
                                                          size_t n, cugraph_error_t** error);

/* Regions of the gathered vector (multi-GPU, no counterpart in the reference).  The driver may
 * lay the slots out as TWO regions, each the concatenation of P equal per-rank sub-chunks: region
 * 0 = slots [0, hot_slots), the hottest sources of every rank, region 1 the rest.  Each region is
 * one all-gather; hot_slots must be a multiple of cugraph_amd_xs_window_slots(num_source_slots)
 * (otherwise it is ignored: one region).  Set it on a graph made by
 * cugraph_amd_graph_create_local_compact before the first session on it.  An iteration can then
 * be enqueued in two steps: phase 0 needs only region 0 (the XS tiles over the hot slots: most of
 * the work), phase 1 the whole vector (the other slots, the low-degree rows, the rank update,
 * the scalars) -- so the all-gather of region 1 can run beside phase 0.  iterate() = phase 0
 * then phase 1.  On a graph without regions phase 0 is empty and phase 1 the whole iteration. */
int64_t cugraph_amd_xs_window_slots(int64_t num_source_slots);
cugraph_error_code_t cugraph_amd_graph_set_hot_slots(cugraph_graph_t* graph, int64_t hot_slots,
                                                     cugraph_error_t** error);
cugraph_error_code_t cugraph_amd_pagerank_session_iterate_phase(cugraph_amd_pagerank_session_t* session,
                                                                int phase, cugraph_error_t** error);

/* As iterate(), but brackets every iteration and its SpMV launch with HIP events on the handle's
 * stream, synchronises, and returns the n per-iteration durations and the durations of the
 * dominant kernel (xs_kernel, or spmv_kernel for graphs without XS tiles) in milliseconds
 * (either output may be NULL).  For bench.py's roofline figure. */
cugraph_error_code_t cugraph_amd_pagerank_session_timed_iterate(
  cugraph_amd_pagerank_session_t* session, size_t n, float* iteration_ms, float* spmv_kernel_ms,
  cugraph_error_t** error);

/* Multi-GPU hooks.  After each iterate(1) the driver must (a) all-gather the source-value
 * vector and (b) all-reduce the two scalars, both in place on the handle's stream order:
 *   *x_full  : FLOAT32/64 view of num_global_vertices elements, the vector the NEXT iteration
 *              gathers from; this rank has just written [vertex_first, vertex_last) of it;
 *   *scalars : FLOAT64 view of 4 elements: [0..1] {dangling mass, L1 delta} as the next iteration's
 *              kernels read them, [2..3] the same two sums as this rank's last executed iteration
 *              produced them.  A multi-GPU driver sums [2..3] over the ranks INTO [0..1]; an
 *              iteration skipped after the stop leaves [2..3] unchanged, so the sum is idempotent.
 * Caller frees both views. */
cugraph_error_code_t cugraph_amd_pagerank_session_exchange_buffers(
  cugraph_amd_pagerank_session_t* session, cugraph_type_erased_device_array_view_t** x_full,
  cugraph_type_erased_device_array_view_t** scalars, cugraph_error_t** error);

/* Synchronise and report: iterations executed, whether the stop criterion has fired, the last
 * L1 delta and dangling mass. */
cugraph_error_code_t cugraph_amd_pagerank_session_status(cugraph_amd_pagerank_session_t* session,
                                                         size_t* num_iterations,
                                                         bool_t* stopped, double* l1_delta,
                                                         double* dangling_mass,
                                                         cugraph_error_t** error);

/* Copy the current ranks out as a centrality result (vertex ids + values); `max_iterations`
 * only decides the converged flag (converged <=> stopped with iterations < max_iterations,
 * pagerank_impl.cuh:337). */
cugraph_error_code_t cugraph_amd_pagerank_session_result(cugraph_amd_pagerank_session_t* session,
                                                         size_t max_iterations,
                                                         cugraph_centrality_result_t** result,
                                                         cugraph_error_t** error);

/* info[0]=algorithmic bytes per iteration, [1]=edges per iteration,
 * [2]=low-degree vertices, [3]=mid-degree vertices, [4]=hub vertices, [5]=hub work items,
 * [6]=workgroups of the spmv_kernel launch, [7]=algorithmic bytes of the dominant kernel alone
 * (xs_kernel when the graph runs on XS tiles, else spmv_kernel) */
cugraph_error_code_t cugraph_amd_pagerank_session_info(cugraph_amd_pagerank_session_t* session,
                                                       int64_t info[8], cugraph_error_t** error);

void cugraph_amd_pagerank_session_free(cugraph_amd_pagerank_session_t* session);

const char* cugraph_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CUGRAPH_AMD_C_H */
