/* A consumer of the C ABI written in C: what a maintainer's plain-C program linking the reference's
 * libcugraph_c would look like against libcugraph_amd_c.so (same names, include/cugraph_amd_c.h).
 * It is shaped after the reference's own boundary test (cpp/tests/c_api/pagerank_test.c:
 * generic_pagerank_test / generic_personalized_pagerank_test and their non-converging twins) but
 * reads its case from a text file, so that tests/test_gpu_c_consumer.py can drive it with every
 * golden case of that file:
 *
 *   pagerank_consumer <case file>
 *
 * case file: "num_vertices num_edges store_transposed alpha epsilon max_iterations allow_nonconvergence
 *             num_personalization", then num_edges lines "src dst weight", then num_personalization
 * lines "vertex value".  Output: one line "vertex value" per vertex, then "converged <0|1>".
 * Exit status 0 when every call succeeded. */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "cugraph_amd_c.h"

#define CHECK(code, err, what)                                                          \
  do {                                                                                  \
    if ((code) != CUGRAPH_SUCCESS) {                                                    \
      fprintf(stderr, "%s failed (%d): %s\n", what, (int)(code),                        \
              (err) ? cugraph_error_message(err) : "no message");                       \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

/* host values -> a device array the caller owns, and a view of it */
static int to_device(const cugraph_resource_handle_t* handle, const void* host, size_t n,
                     cugraph_data_type_id_t type, cugraph_type_erased_device_array_t** array,
                     cugraph_type_erased_device_array_view_t** view)
{
  cugraph_error_t* err = NULL;
  cugraph_error_code_t code = cugraph_type_erased_device_array_create(handle, n, type, array, &err);
  CHECK(code, err, "cugraph_type_erased_device_array_create");
  *view = cugraph_type_erased_device_array_view(*array);
  code  = cugraph_type_erased_device_array_view_copy_from_host(handle, *view, (const byte_t*)host, &err);
  CHECK(code, err, "cugraph_type_erased_device_array_view_copy_from_host");
  return 0;
}

int main(int argc, char** argv)
{
  if (argc != 2 && !(argc == 3 && strcmp(argv[2], "mg") == 0)) {
    fprintf(stderr, "usage: %s <case file> [mg]\n", argv[0]);
    return 2;
  }
  /* "mg": the same case through the multi-GPU entry points at world size 1, as
   * cpp/tests/c_api/mg_pagerank_test.c:67-94 runs it (create_mg_test_graph -> cugraph_mg_graph_create) */
  int const mg = argc == 3;
  FILE* f = fopen(argv[1], "r");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  size_t nv, ne, max_iter, n_pers;
  int transposed, allow;
  double alpha, epsilon;
  if (fscanf(f, "%zu %zu %d %lf %lf %zu %d %zu", &nv, &ne, &transposed, &alpha, &epsilon, &max_iter, &allow,
             &n_pers) != 8)
    return 2;
  int32_t* src   = malloc((ne ? ne : 1) * sizeof(int32_t));
  int32_t* dst   = malloc((ne ? ne : 1) * sizeof(int32_t));
  float* wgt     = malloc((ne ? ne : 1) * sizeof(float));
  int32_t* pers_v = malloc((n_pers ? n_pers : 1) * sizeof(int32_t));
  float* pers_x  = malloc((n_pers ? n_pers : 1) * sizeof(float));
  for (size_t i = 0; i < ne; ++i)
    if (fscanf(f, "%d %d %f", &src[i], &dst[i], &wgt[i]) != 3) return 2;
  for (size_t i = 0; i < n_pers; ++i)
    if (fscanf(f, "%d %f", &pers_v[i], &pers_x[i]) != 2) return 2;
  fclose(f);

  cugraph_error_t* err = NULL;
  cugraph_error_code_t code;
  cugraph_resource_handle_t* handle =
    mg ? cugraph_amd_create_resource_handle(NULL, 0, 1) : cugraph_create_resource_handle(NULL);
  if (!handle) {
    fprintf(stderr, "cugraph_create_resource_handle failed\n");
    return 1;
  }
  cugraph_type_erased_device_array_t *a_src, *a_dst, *a_wgt, *a_pv = NULL, *a_px = NULL;
  cugraph_type_erased_device_array_view_t *v_src, *v_dst, *v_wgt, *v_pv = NULL, *v_px = NULL;
  if (to_device(handle, src, ne, INT32, &a_src, &v_src)) return 1;
  if (to_device(handle, dst, ne, INT32, &a_dst, &v_dst)) return 1;
  if (to_device(handle, wgt, ne, FLOAT32, &a_wgt, &v_wgt)) return 1;
  if (n_pers) {
    if (to_device(handle, pers_v, n_pers, INT32, &a_pv, &v_pv)) return 1;
    if (to_device(handle, pers_x, n_pers, FLOAT32, &a_px, &v_px)) return 1;
  }

  cugraph_graph_properties_t props;
  props.is_symmetric  = FALSE;
  props.is_multigraph = FALSE;
  cugraph_graph_t* graph = NULL;
  /* as create_test_graph of the reference's tests: no vertex list, no renumbering */
  if (mg)
    code = cugraph_mg_graph_create(handle, &props, v_src, v_dst, v_wgt, NULL, NULL, transposed ? TRUE : FALSE, ne,
                                   FALSE, &graph, &err);
  else
    code = cugraph_graph_create_sg(handle, &props, NULL, v_src, v_dst, v_wgt, NULL, NULL,
                                   transposed ? TRUE : FALSE, FALSE, FALSE, FALSE, FALSE, &graph, &err);
  CHECK(code, err, mg ? "cugraph_mg_graph_create" : "cugraph_graph_create_sg");

  cugraph_centrality_result_t* result = NULL;
  if (n_pers && allow)
    code = cugraph_personalized_pagerank_allow_nonconvergence(handle, graph, NULL, NULL, NULL, NULL, v_pv, v_px,
                                                              alpha, epsilon, max_iter, FALSE, &result, &err);
  else if (n_pers)
    code = cugraph_personalized_pagerank(handle, graph, NULL, NULL, NULL, NULL, v_pv, v_px, alpha, epsilon,
                                         max_iter, FALSE, &result, &err);
  else if (allow)
    code = cugraph_pagerank_allow_nonconvergence(handle, graph, NULL, NULL, NULL, NULL, alpha, epsilon, max_iter,
                                                 FALSE, &result, &err);
  else
    code = cugraph_pagerank(handle, graph, NULL, NULL, NULL, NULL, alpha, epsilon, max_iter, FALSE, &result, &err);
  CHECK(code, err, "cugraph_[personalized_]pagerank");

  cugraph_type_erased_device_array_view_t* vertices = cugraph_centrality_result_get_vertices(result);
  cugraph_type_erased_device_array_view_t* values   = cugraph_centrality_result_get_values(result);
  size_t n = cugraph_type_erased_device_array_view_size(vertices);
  if (n != nv || cugraph_type_erased_device_array_view_type(values) != FLOAT32) {
    fprintf(stderr, "unexpected result shape: %zu vertices, value type %d\n", n,
            (int)cugraph_type_erased_device_array_view_type(values));
    return 1;
  }
  int32_t* h_v = malloc(n * sizeof(int32_t));
  float* h_x   = malloc(n * sizeof(float));
  code = cugraph_type_erased_device_array_view_copy_to_host(handle, (byte_t*)h_v, vertices, &err);
  CHECK(code, err, "copy_to_host(vertices)");
  code = cugraph_type_erased_device_array_view_copy_to_host(handle, (byte_t*)h_x, values, &err);
  CHECK(code, err, "copy_to_host(values)");
  for (size_t i = 0; i < n; ++i) printf("%d %.9g\n", h_v[i], h_x[i]);
  printf("converged %d\n", cugraph_centrality_result_converged(result) ? 1 : 0);

  cugraph_type_erased_device_array_view_free(vertices);
  cugraph_type_erased_device_array_view_free(values);
  cugraph_centrality_result_free(result);
  if (mg)
    cugraph_mg_graph_free(graph);
  else
    cugraph_sg_graph_free(graph);
  cugraph_type_erased_device_array_view_free(v_src);
  cugraph_type_erased_device_array_view_free(v_dst);
  cugraph_type_erased_device_array_view_free(v_wgt);
  cugraph_type_erased_device_array_free(a_src);
  cugraph_type_erased_device_array_free(a_dst);
  cugraph_type_erased_device_array_free(a_wgt);
  if (n_pers) {
    cugraph_type_erased_device_array_view_free(v_pv);
    cugraph_type_erased_device_array_view_free(v_px);
    cugraph_type_erased_device_array_free(a_pv);
    cugraph_type_erased_device_array_free(a_px);
  }
  cugraph_free_resource_handle(handle);
  free(src); free(dst); free(wgt); free(pers_v); free(pers_x); free(h_v); free(h_x);
  return 0;
}
