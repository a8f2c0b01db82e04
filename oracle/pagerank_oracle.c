/*
 * oracle/pagerank_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's PageRank hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (cugraph_amd/) never does.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   - CSC layout: lexicographic (dst, src) order of the edge list with
 *     offsets by counting, cpp/src/structure/detail/structure_utils.cuh:144-296
 *     (sort) and :54-88 (offsets).  Ties (equal (src,dst)) keep input order;
 *     the reference leaves that order unspecified.
 *   - out-weight sums: cpp/tests/link_analysis/pagerank_test.cpp:86-93
 *     (CPU reference) == cpp/src/structure/graph_weight_utils_impl.cuh:41-79.
 *   - power iteration: cpp/tests/link_analysis/pagerank_test.cpp:95-129 for the
 *     update rule; iteration counting / convergence flag / "initial guess is
 *     copied verbatim" follow the overload the C API calls,
 *     cpp/src/link_analysis/pagerank_impl.cuh:235-337 and :428-440.
 *   - R-MAT quadrant recursion: cpp/src/generators/generate_rmat_edgelist.cu:85-102
 *     (bit choice rule); the random stream is this project's own Philox4x32-10
 *     keyed by (seed, edge, level pair), identical to the HIP generator.
 *
 * Pinned against: the golden vectors of cpp/tests/c_api/pagerank_test.c:396-555,
 * python/pylibcugraph/pylibcugraph/tests/test_pagerank.py:28-159 and NetworkX
 * (tests/test_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11), counter-based                 */
/* ------------------------------------------------------------------ */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                 uint32_t k0, uint32_t k1, uint32_t out[4])
{
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                   uint32_t* out)
{
  philox4x32_10(c0, c1, c2, c3, k0, k1, out);
}

/* Vertex-id scramble: a fixed bijection on [0, 2^scale).  Same as the device code. */
static inline uint32_t scramble_id(uint32_t v, int scale)
{
  uint32_t mask = (scale >= 32) ? 0xFFFFFFFFu : ((1u << scale) - 1u);
  v = (v * 0x9E3779B1u + 0x7F4A7C15u) & mask;
  /* bit reverse within `scale` bits */
  uint32_t r = 0;
  for (int b = 0; b < scale; ++b) { r |= ((v >> b) & 1u) << (scale - 1 - b); }
  r = (r * 0x85EBCA6Bu + 0xC2B2AE35u) & mask;
  return r;
}

/*
 * Edges [first, first+count) of the R-MAT stream (a,b,c as 24-bit integer thresholds):
 *   src_bit = u0 > t_ab ; dst_bit = u1 > (src_bit ? t_c : t_a)   (generate_rmat_edgelist.cu:90-91)
 */
void oracle_rmat(int scale, uint64_t first, uint64_t count, uint64_t seed, uint32_t t_ab,
                 uint32_t t_a, uint32_t t_c, int scramble, int32_t* src, int32_t* dst)
{
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)count; ++i) {
    uint64_t e = first + (uint64_t)i;
    uint32_t s = 0, d = 0;
    int bit    = scale - 1;
    for (uint32_t j = 0; bit >= 0; ++j) {
      uint32_t o[4];
      philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), j, 0u, k0, k1, o);
      for (int h = 0; h < 2 && bit >= 0; ++h, --bit) {
        uint32_t u0 = o[2 * h] >> 8, u1 = o[2 * h + 1] >> 8;
        uint32_t sb = u0 > t_ab;
        uint32_t db = u1 > (sb ? t_c : t_a);
        s |= sb << bit;
        d |= db << bit;
      }
    }
    if (scramble) {
      s = scramble_id(s, scale);
      d = scramble_id(d, scale);
    }
    src[i] = (int32_t)s;
    dst[i] = (int32_t)d;
  }
}

/* weight of edge e: ((o0 << 21 | o1 >> 11) + 1) * 2^-53  in (0, 1] */
void oracle_rmat_weights(uint64_t first, uint64_t count, uint64_t seed, double* w)
{
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)count; ++i) {
    uint64_t e = first + (uint64_t)i;
    uint32_t o[4];
    philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), 0u, 1u, k0, k1, o);
    uint64_t m = ((uint64_t)o[0] << 21) | (uint64_t)(o[1] >> 11);
    w[i]       = (double)(m + 1) * 0x1.0p-53;
  }
}

/* ------------------------------------------------------------------ */
/* COO -> compressed (major, minor) with stable lexicographic order    */
/* ------------------------------------------------------------------ */
/*
 * Stable LSD counting sort: by minor, then by major.  Result: offsets[nv+1] over majors,
 * minors ascending inside each major segment, ties in input order; perm[k] = input edge
 * that lands at position k (so weights_out[k] = weights_in[perm[k]]).
 * For PageRank (store_transposed) major = dst, minor = src.
 */
int oracle_compress(int64_t nv, int64_t ne, const int32_t* major, const int32_t* minor,
                    int64_t* offsets, int32_t* minors_out, int64_t* perm)
{
  int64_t* cnt = (int64_t*)calloc((size_t)nv + 1, sizeof(int64_t));
  int64_t* tmp = (int64_t*)malloc((size_t)(ne > 0 ? ne : 1) * sizeof(int64_t));
  if (!cnt || !tmp) { free(cnt); free(tmp); return 1; }
  /* pass 1: by minor */
  for (int64_t e = 0; e < ne; ++e) {
    if (minor[e] < 0 || minor[e] >= nv || major[e] < 0 || major[e] >= nv) { free(cnt); free(tmp); return 2; }
    cnt[minor[e] + 1]++;
  }
  for (int64_t v = 0; v < nv; ++v) cnt[v + 1] += cnt[v];
  for (int64_t e = 0; e < ne; ++e) tmp[cnt[minor[e]]++] = e;
  /* pass 2: by major (stable) */
  memset(cnt, 0, ((size_t)nv + 1) * sizeof(int64_t));
  for (int64_t e = 0; e < ne; ++e) cnt[major[e] + 1]++;
  for (int64_t v = 0; v < nv; ++v) cnt[v + 1] += cnt[v];
  memcpy(offsets, cnt, ((size_t)nv + 1) * sizeof(int64_t));
  for (int64_t k = 0; k < ne; ++k) {
    int64_t e  = tmp[k];
    int64_t p  = cnt[major[e]]++;
    perm[p]       = e;
    minors_out[p] = minor[e];
  }
  free(cnt);
  free(tmp);
  return 0;
}

/* ------------------------------------------------------------------ */
/* PageRank                                                            */
/* ------------------------------------------------------------------ */
/*
 * One implementation, two value types.  Scalars that the reference keeps in
 * result_t (dangling_sum, diff_sum) are accumulated in double here: the oracle
 * is the yardstick, and the NetworkX oracle it is pinned to works in double.
 *
 * out_info[0] = iterations executed, out_info[1] = converged (0/1)
 *   (pagerank_impl.cuh:328-337: ++iter; stop if diff < eps, else if iter >= max;
 *    converged = iter < max_iterations).
 * has_nstart: pr holds the caller's initial guess, used verbatim (pagerank_impl.cuh:435-439).
 * out_w_pre : optional precomputed out-weight sums (pagerank_impl.cuh:207-209).
 * pers_*    : optional personalization pairs; values need not sum to one (:316-317).
 */
#define DEFINE_PAGERANK(NAME, T)                                                                  \
  int NAME(int64_t nv, const int64_t* offsets, const int32_t* indices, const T* weights,          \
           const T* out_w_pre, int64_t n_pers, const int32_t* pers_v, const T* pers_val, T* pr,   \
           int has_nstart, double alpha_d, double epsilon_d, int64_t max_iterations,              \
           int64_t* out_info)                                                                     \
  {                                                                                               \
    out_info[0] = 0;                                                                              \
    out_info[1] = 1;                                                                              \
    if (nv == 0) return 0;                                                                        \
    T alpha = (T)alpha_d, epsilon = (T)epsilon_d;                                                 \
    if (!(alpha >= 0 && alpha <= 1) || !(epsilon >= 0)) return 3;                                 \
    if (!has_nstart) {                                                                            \
      for (int64_t i = 0; i < nv; ++i) pr[i] = (T)1.0 / (T)nv;                                    \
    }                                                                                             \
    double pers_sum_d = 0.0;                                                                      \
    for (int64_t i = 0; i < n_pers; ++i) pers_sum_d += (double)pers_val[i];                       \
    T pers_sum = (T)pers_sum_d;                                                                   \
    if (n_pers > 0 && !(pers_sum > 0)) return 4;                                                  \
    T* out_w = (T*)calloc((size_t)nv, sizeof(T));                                                 \
    T* old   = (T*)malloc((size_t)nv * sizeof(T));                                                \
    T* x     = (T*)malloc((size_t)nv * sizeof(T));                                                \
    if (!out_w || !old || !x) { free(out_w); free(old); free(x); return 1; }                      \
    if (out_w_pre) {                                                                              \
      memcpy(out_w, out_w_pre, (size_t)nv * sizeof(T));                                           \
    } else {                                                                                      \
      double* acc = (double*)calloc((size_t)nv, sizeof(double));                                  \
      if (!acc) { free(out_w); free(old); free(x); return 1; }                                    \
      for (int64_t i = 0; i < nv; ++i)                                                            \
        for (int64_t j = offsets[i]; j < offsets[i + 1]; ++j)                                     \
          acc[indices[j]] += weights ? (double)weights[j] : 1.0;                                  \
      for (int64_t i = 0; i < nv; ++i) out_w[i] = (T)acc[i];                                      \
      free(acc);                                                                                  \
    }                                                                                             \
    int64_t iter = 0;                                                                             \
    while (1) {                                                                                   \
      memcpy(old, pr, (size_t)nv * sizeof(T));                                                    \
      double dangling_d = 0.0;                                                                    \
      _Pragma("omp parallel for reduction(+ : dangling_d) schedule(static)")                      \
      for (int64_t i = 0; i < nv; ++i) {                                                          \
        if (out_w[i] == (T)0) {                                                                   \
          dangling_d += (double)old[i];                                                           \
          x[i] = old[i];                                                                          \
        } else {                                                                                  \
          x[i] = old[i] / out_w[i];                                                               \
        }                                                                                         \
      }                                                                                           \
      T dangling = (T)dangling_d;                                                                 \
      T teleport = dangling * alpha + (T)(1.0 - (double)alpha);                                   \
      T base     = n_pers > 0 ? (T)0 : teleport / (T)nv;                                          \
      _Pragma("omp parallel for schedule(dynamic, 4096)")                                         \
      for (int64_t i = 0; i < nv; ++i) {                                                          \
        double acc = 0.0;                                                                         \
        if (weights) {                                                                            \
          for (int64_t j = offsets[i]; j < offsets[i + 1]; ++j)                                   \
            acc += (double)(x[indices[j]] * weights[j] * alpha);                                  \
        } else {                                                                                  \
          for (int64_t j = offsets[i]; j < offsets[i + 1]; ++j)                                   \
            acc += (double)(x[indices[j]] * alpha);                                               \
        }                                                                                         \
        pr[i] = (T)((double)base + acc);                                                          \
      }                                                                                           \
      for (int64_t i = 0; i < n_pers; ++i) pr[pers_v[i]] += teleport * (pers_val[i] / pers_sum);  \
      double diff_d = 0.0;                                                                        \
      _Pragma("omp parallel for reduction(+ : diff_d) schedule(static)")                          \
      for (int64_t i = 0; i < nv; ++i) diff_d += fabs((double)pr[i] - (double)old[i]);            \
      iter++;                                                                                     \
      if (diff_d < (double)epsilon) break;                                                        \
      if (iter >= max_iterations) break;                                                          \
    }                                                                                             \
    out_info[0] = iter;                                                                           \
    out_info[1] = iter < max_iterations ? 1 : 0;                                                  \
    free(out_w);                                                                                  \
    free(old);                                                                                    \
    free(x);                                                                                      \
    return 0;                                                                                     \
  }

DEFINE_PAGERANK(oracle_pagerank_f32, float)
DEFINE_PAGERANK(oracle_pagerank_f64, double)
