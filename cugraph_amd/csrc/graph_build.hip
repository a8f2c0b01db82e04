// COO -> compressed sparse (CSC for the pull layout) on gfx950.
//
// Replaces cugraph::detail::sort_and_compress_edgelist / compute_sparse_offsets
// (reference: cpp/src/structure/detail/structure_utils.cuh:54-88, 144-296).  The reference
// sorts the whole (major, minor) edge list; here the layout is produced by
//   1. degree histogram of the majors          (hist_kernel)
//   2. exclusive scan -> offsets               (scan_* kernels)
//   3. scatter of the minors into their column (scatter_kernel, arrival order arbitrary)
//   4. ordering INSIDE each column by (minor, input position), degree-bucketed:
//        deg <= 16   : 16-lane groups, register bitonic network over DPP/shuffles
//        deg <= 64   : one wave64 per column, same network
//        deg <= 8192 : one workgroup per column, LDS bitonic
//        larger      : one 1024-thread workgroup per column, LSD radix through a scratch copy
// Step 4 makes the result independent of the atomic arrival order of step 3, and equal to a
// stable (major, minor)-lexicographic sort of the input -- the definition of the reference CSC.
// For weighted graphs the sort key carries the input edge position in its low 32 bits, so
// parallel edges keep input order and the weights are gathered once at the end.
//
// All of this is HBM-bound integer work; no MFMA.  Algorithmic bytes:
//   E*(4*4 + 2*b_w) + 3*(V+1)*b_off  (+ 2*E*(4+b_w) for the ordering pass).
#include "common.hpp"

namespace cga {

namespace {

constexpr int kThreads   = 256;
constexpr int kMaxBlocks = 256 * 16;  // 256 CUs x 16 resident 256-thread blocks: grid-stride beyond

inline int grid_for(int64_t work_items, int threads = kThreads)
{
  int64_t b = (work_items + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return static_cast<int>(b);
}

// ------------------------------------------------------------------ id narrowing
template <typename InT>
__global__ void narrow_kernel(InT const* __restrict__ in, size_t n, int32_t* __restrict__ out,
                              long long* __restrict__ minmax /* [0]=min [1]=max */)
{
  long long lo = 0x7fffffffffffffffLL, hi = -1;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    long long v = static_cast<long long>(in[i]);
    out[i]      = static_cast<int32_t>(v);
    lo          = v < lo ? v : lo;
    hi          = v > hi ? v : hi;
  }
  for (int o = 32; o > 0; o >>= 1) {
    long long l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(minmax, lo);
    atomicMax(minmax + 1, hi);
  }
}

__global__ void shift_kernel(int32_t const* in, int32_t* out, int64_t n, int32_t shift)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[i] - shift;
}

__global__ void widen_kernel(int32_t const* in, int64_t* out, int64_t n)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[i];
}

template <typename OutT>
__global__ void sequence_kernel(OutT* out, size_t n, long long first)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = static_cast<OutT>(first + (long long)i);
}

// ------------------------------------------------------------------ 1. histogram
__global__ void hist_kernel(int32_t const* __restrict__ keys, int64_t n,
                            unsigned int* __restrict__ counts)
{
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
    atomicAdd(&counts[keys[i]], 1u);
}

// Long columns (the pull plan's (tile, slice, window) units: each takes thousands to millions of
// keys): global atomics would all land on the same few counters.  Every workgroup takes a
// contiguous span of the keys; when the majors of its span fit a window of kLdsMajors values
// (they do when the input is grouped, as the plan's unit labels in row order are) it counts them
// in LDS and touches each global counter once, else it falls back to global atomics.
constexpr int kLdsMajors = 32768;  // 128 of the 160 KiB of LDS

// smallest major of the workgroup's span and whether all of them fit the LDS window above it
__device__ inline bool span_window(int32_t const* __restrict__ keys, int64_t beg, int64_t end, int32_t* first)
{
  __shared__ int s_min, s_max;
  if (threadIdx.x == 0) { s_min = 0x7fffffff; s_max = -1; }
  __syncthreads();
  int lo = 0x7fffffff, hi = -1;
  for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
    int k = keys[i];
    lo    = k < lo ? k : lo;
    hi    = k > hi ? k : hi;
  }
  for (int o = 32; o > 0; o >>= 1) {
    int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&s_min, lo);
    atomicMax(&s_max, hi);
  }
  __syncthreads();
  *first = s_min;
  return s_max >= s_min && s_max - s_min < kLdsMajors;
}

__global__ __launch_bounds__(1024) void hist_lds_kernel(int32_t const* __restrict__ keys, int64_t n,
                                                        unsigned int* __restrict__ counts)
{
  extern __shared__ unsigned int lds_count[];
  int64_t const span = (n + gridDim.x - 1) / gridDim.x;
  int64_t const beg = blockIdx.x * span, end = beg + span < n ? beg + span : n;
  if (beg >= end) return;
  // keys inside the window [first, first + kLdsMajors) are counted in LDS, the others (none when
  // the span fits the window) with global atomics
  int32_t first;
  (void)span_window(keys, beg, end, &first);
  for (int k = threadIdx.x; k < kLdsMajors; k += blockDim.x) lds_count[k] = 0u;
  __syncthreads();
  for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
    int32_t const key    = keys[i];
    unsigned int const d = static_cast<unsigned int>(key - first);
    if (d < static_cast<unsigned int>(kLdsMajors))
      atomicAdd(&lds_count[d], 1u);
    else
      atomicAdd(&counts[key], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kLdsMajors; k += blockDim.x)
    if (lds_count[k]) atomicAdd(&counts[first + k], lds_count[k]);
}

// ------------------------------------------------------------------ 2. exclusive scan
constexpr int kScanItems = 16;                     // per thread
constexpr int kScanTile  = kThreads * kScanItems;  // 4096 per block

__device__ inline unsigned long long block_reduce_u64(unsigned long long v,
                                                      unsigned long long* lds /* [4] */)
{
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long t = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return t;
}

__global__ void scan_tile_sums_kernel(unsigned int const* __restrict__ counts, int64_t n,
                                      unsigned long long* __restrict__ tile_sums)
{
  __shared__ unsigned long long lds[4];
  int64_t base = (int64_t)blockIdx.x * kScanTile;
  unsigned long long s = 0;
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = base + (int64_t)k * kThreads + threadIdx.x;
    if (i < n) s += counts[i];
  }
  s = block_reduce_u64(s, lds);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = s;
}

// single workgroup: in-place exclusive scan of the tile sums
__global__ void scan_tile_prefix_kernel(unsigned long long* __restrict__ tile_sums, int64_t n_tiles)
{
  __shared__ unsigned long long wave_tot[16];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t base = 0; base < n_tiles; base += blockDim.x) {
    int64_t i            = base + threadIdx.x;
    unsigned long long v = i < n_tiles ? tile_sums[i] : 0ull;
    unsigned long long inc = v;  // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned long long before = carry;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    if (i < n_tiles) tile_sums[i] = before + inc - v;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry = before + inc;
    __syncthreads();
  }
}

template <typename OffT>
__global__ void scan_write_kernel(unsigned int const* __restrict__ counts, int64_t n,
                                  unsigned long long const* __restrict__ tile_prefix,
                                  OffT* __restrict__ offsets /* [n+1] */)
{
  // thread t owns items [t*16, t*16+16) of the tile: sequential local scan + wave/block prefix
  __shared__ unsigned long long wave_tot[4];
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  unsigned int c[kScanItems];
  unsigned long long s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = base + k;
    c[k]      = i < n ? counts[i] : 0u;
    s += c[k];
  }
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long inc = s;
  for (int o = 1; o < 64; o <<= 1) {
    unsigned long long t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  unsigned long long run = tile_prefix[blockIdx.x] + inc - s;
  for (int w = 0; w < wave; ++w) run += wave_tot[w];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = base + k;
    if (i < n) offsets[i] = static_cast<OffT>(run);
    run += c[k];
    if (i == n - 1) offsets[n] = static_cast<OffT>(run);
  }
  if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) offsets[0] = 0;
}

// ------------------------------------------------------------------ 3. scatter
template <typename OffT, typename KeyT>
__global__ void scatter_kernel(int32_t const* __restrict__ majors,
                               int32_t const* __restrict__ minors, int64_t n,
                               OffT const* __restrict__ offsets,
                               unsigned int* __restrict__ cursor, KeyT* __restrict__ keys)
{
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += stride) {
    int32_t m        = majors[e];
    unsigned int pos = atomicAdd(&cursor[m], 1u);
    KeyT k;
    if constexpr (sizeof(KeyT) == 8) {
      k = (static_cast<KeyT>(static_cast<uint32_t>(minors[e])) << 32) |
          static_cast<KeyT>(static_cast<uint32_t>(e));
    } else {
      k = static_cast<KeyT>(minors[e]);
    }
    keys[offsets[m] + pos] = k;
  }
}

// scatter for long columns (see hist_lds_kernel): a workgroup counts its span in LDS, reserves one
// run per major it met with a single global atomic, then hands out the positions of the run with
// LDS atomics.  The positions live in 32-bit LDS words: relative to the first column of the window
// (64-bit offsets: a window whose columns hold 2^32 edges or more goes the global way).
template <typename OffT, typename KeyT>
__global__ __launch_bounds__(1024) void scatter_lds_kernel(int32_t const* __restrict__ majors,
                                                           int32_t const* __restrict__ minors, int64_t n,
                                                           OffT const* __restrict__ offsets, int64_t n_major,
                                                           unsigned int* __restrict__ cursor,
                                                           KeyT* __restrict__ keys)
{
  extern __shared__ unsigned int lds_pos[];
  int64_t const span = (n + gridDim.x - 1) / gridDim.x;
  int64_t const beg = blockIdx.x * span, end = beg + span < n ? beg + span : n;
  if (beg >= end) return;
  auto key_of = [&](int64_t e) {
    if constexpr (sizeof(KeyT) == 8)
      return (static_cast<KeyT>(static_cast<uint32_t>(minors[e])) << 32) | static_cast<KeyT>(static_cast<uint32_t>(e));
    else
      return static_cast<KeyT>(minors[e]);
  };
  // majors inside the window [first, first + kLdsMajors) go through LDS, the others (none when the
  // span fits the window; the cold majority of a degree-sorted graph, whose hubs are the window)
  // through global atomics
  int32_t first;
  (void)span_window(majors, beg, end, &first);
  // columns [first, w_end) are the window; its first edge is the base of the LDS positions
  int64_t const w_end = static_cast<int64_t>(first) + kLdsMajors < n_major ? static_cast<int64_t>(first) + kLdsMajors : n_major;
  int64_t const base  = static_cast<int64_t>(offsets[first]);
  unsigned int const in_lds = static_cast<int64_t>(offsets[w_end]) - base < 0xFFFFFFFFLL ? static_cast<unsigned int>(kLdsMajors) : 0u;
  for (int k = threadIdx.x; k < kLdsMajors; k += blockDim.x) lds_pos[k] = 0u;
  __syncthreads();
  for (int64_t e = beg + threadIdx.x; e < end; e += blockDim.x) {
    unsigned int const d = static_cast<unsigned int>(majors[e] - first);
    if (d < in_lds) atomicAdd(&lds_pos[d], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kLdsMajors; k += blockDim.x) {
    unsigned int const c = lds_pos[k];
    if (c) lds_pos[k] = static_cast<unsigned int>(static_cast<int64_t>(offsets[first + k]) - base) + atomicAdd(&cursor[first + k], c);
  }
  __syncthreads();
  for (int64_t e = beg + threadIdx.x; e < end; e += blockDim.x) {
    int32_t const m      = majors[e];
    unsigned int const d = static_cast<unsigned int>(m - first);
    int64_t const pos    = d < in_lds ? base + static_cast<int64_t>(atomicAdd(&lds_pos[d], 1u))
                                      : static_cast<int64_t>(offsets[m]) + static_cast<int64_t>(atomicAdd(&cursor[m], 1u));
    keys[pos] = key_of(e);
  }
}

// ------------------------------------------------------------------ 4. per-column ordering
template <typename KeyT>
__device__ inline KeyT key_max()
{
  return ~static_cast<KeyT>(0);
}

// G lanes per column (G = 16 or 64); columns with lo < deg <= G are sorted, the rest skipped.
template <typename OffT, typename KeyT, int G>
__global__ void sort_small_kernel(OffT const* __restrict__ offsets, KeyT* __restrict__ keys,
                                  int64_t n_major, int lo)
{
  int lane       = threadIdx.x & (G - 1);
  int64_t group  = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  int64_t groups = ((int64_t)gridDim.x * blockDim.x) / G;
  for (int64_t v = group; v < n_major; v += groups) {
    OffT b      = offsets[v];
    int64_t deg = static_cast<int64_t>(offsets[v + 1] - b);
    if (deg <= lo || deg > G) continue;  // uniform over the G lanes of a group
    KeyT k = lane < deg ? keys[b + lane] : key_max<KeyT>();
#pragma unroll
    for (int size = 2; size <= G; size <<= 1) {
#pragma unroll
      for (int j = size >> 1; j > 0; j >>= 1) {
        KeyT o     = __shfl_xor(k, j, G);
        bool up    = (lane & size) == 0;
        bool lower = (lane & j) == 0;
        KeyT mn = k < o ? k : o, mx = k < o ? o : k;
        k = (lower == up) ? mn : mx;
      }
    }
    if (lane < deg) keys[b + lane] = k;
  }
}

constexpr int kMedCap = 8192;  // keys per workgroup-sorted column (64 KiB of LDS at 8-byte keys)

constexpr int kWaveCap = 1024;  // keys per wave-sorted column (16 registers a lane)

template <typename OffT>
__global__ void classify_columns_kernel(OffT const* __restrict__ offsets, int64_t n_major,
                                        int32_t* __restrict__ med_list,
                                        int32_t* __restrict__ big_list,
                                        int32_t* __restrict__ wave_list,
                                        unsigned int* __restrict__ n_lists /* [3] */)
{
  // one atomic per wave and list: the lanes that append share a reservation
  int const lane = threadIdx.x & 63;
  unsigned long long const below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t const rounds = (n_major + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    int64_t v   = r * stride + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t deg = v < n_major ? static_cast<int64_t>(offsets[v + 1] - offsets[v]) : 0;
    bool const med = deg > kWaveCap && deg <= kMedCap, big = deg > kMedCap, wav = deg > 64 && deg <= kWaveCap;
    unsigned long long const m_med = __ballot(med), m_big = __ballot(big), m_wav = __ballot(wav);
    if (m_wav) {
      unsigned int base = 0;
      if (lane == 0) base = atomicAdd(&n_lists[2], static_cast<unsigned int>(__popcll(m_wav)));
      base = __shfl(base, 0, 64);
      if (wav) wave_list[base + __popcll(m_wav & below)] = static_cast<int32_t>(v);
    }
    if (m_med) {
      unsigned int base = 0;
      if (lane == 0) base = atomicAdd(&n_lists[0], static_cast<unsigned int>(__popcll(m_med)));
      base = __shfl(base, 0, 64);
      if (med) med_list[base + __popcll(m_med & below)] = static_cast<int32_t>(v);
    }
    if (m_big) {
      unsigned int base = 0;
      if (lane == 0) base = atomicAdd(&n_lists[1], static_cast<unsigned int>(__popcll(m_big)));
      base = __shfl(base, 0, 64);
      if (big) big_list[base + __popcll(m_big & below)] = static_cast<int32_t>(v);
    }
  }
}

// One wave per column of 65 .. kWaveCap keys: a bitonic network over R keys a lane (key i of the
// column in register i / 64 of lane i % 64), partners 64 or more apart are registers of the same
// lane, nearer ones come by DPP / shuffle.  No LDS, no barriers: the workgroup version below spends
// its time in 28 barriers for a 128-key column with three of its four waves idle.
template <typename KeyT, int R>
__device__ inline void wave_bitonic(KeyT* __restrict__ col, int deg, int lane)
{
  KeyT k[R];
#pragma unroll
  for (int r = 0; r < R; ++r) k[r] = r * 64 + lane < deg ? col[r * 64 + lane] : key_max<KeyT>();
#pragma unroll
  for (int size = 2; size <= R * 64; size <<= 1) {
#pragma unroll
    for (int j = size >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & (j >> 6)) == 0) {
            int const q   = r | (j >> 6);
            bool const up = ((r * 64) & size) == 0;
            KeyT const a = k[r], b = k[q];
            bool const sw = (a > b) == up;
            k[r]          = sw ? b : a;
            k[q]          = sw ? a : b;
          }
        }
      } else {
        bool const lower = (lane & j) == 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          KeyT const o  = __shfl_xor(k[r], j, 64);
          bool const up = ((r * 64 + lane) & size) == 0;
          KeyT const mn = k[r] < o ? k[r] : o, mx = k[r] < o ? o : k[r];
          k[r]          = (lower == up) ? mn : mx;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r * 64 + lane < deg) col[r * 64 + lane] = k[r];
}

template <typename OffT, typename KeyT>
__global__ __launch_bounds__(256) void sort_wave_kernel(OffT const* __restrict__ offsets, KeyT* __restrict__ keys,
                                                        int32_t const* __restrict__ list,
                                                        unsigned int const* __restrict__ n_list)
{
  int const lane         = threadIdx.x & 63;
  unsigned int const wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  unsigned int const n_cols = *n_list;
  for (unsigned int c = wid; c < n_cols; c += n_waves) {
    int32_t const v = list[c];
    OffT const b    = offsets[v];
    int const deg   = __builtin_amdgcn_readfirstlane(static_cast<int>(offsets[v + 1] - b));
    KeyT* col       = keys + b;
    if (deg <= 128) wave_bitonic<KeyT, 2>(col, deg, lane);
    else if (deg <= 256) wave_bitonic<KeyT, 4>(col, deg, lane);
    else if (deg <= 512) wave_bitonic<KeyT, 8>(col, deg, lane);
    else wave_bitonic<KeyT, 16>(col, deg, lane);
  }
}

template <typename OffT, typename KeyT>
__global__ __launch_bounds__(256) void sort_medium_kernel(OffT const* __restrict__ offsets,
                                                          KeyT* __restrict__ keys,
                                                          int32_t const* __restrict__ list,
                                                          unsigned int const* __restrict__ n_list)
{
  __shared__ KeyT lds[kMedCap];
  unsigned int n_cols = *n_list;
  for (unsigned int c = blockIdx.x; c < n_cols; c += gridDim.x) {
    int32_t v = list[c];
    OffT b    = offsets[v];
    int deg   = static_cast<int>(offsets[v + 1] - b);
    int P     = 128;
    while (P < deg) P <<= 1;
    for (int i = threadIdx.x; i < P; i += blockDim.x) lds[i] = i < deg ? keys[b + i] : key_max<KeyT>();
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
      for (int j = size >> 1; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
          // t-th compare-exchange pair of this stage: i has bit j clear
          int i   = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          int p   = i | j;
          KeyT a = lds[i], d = lds[p];
          bool up = (i & size) == 0;
          if ((a > d) == up) {
            lds[i] = d;
            lds[p] = a;
          }
        }
        __syncthreads();
      }
    }
    for (int i = threadIdx.x; i < deg; i += blockDim.x) keys[b + i] = lds[i];
    __syncthreads();
  }
}

struct RadixPasses {
  int n;
  int shift[8];
};

// One 1024-thread workgroup per column; stable LSD radix, 8 bits per pass, ping-pong between the
// column's slice of `keys` and the same slice of `tmp`.
template <typename OffT, typename KeyT>
__global__ __launch_bounds__(1024) void sort_big_kernel(OffT const* __restrict__ offsets,
                                                        KeyT* __restrict__ keys,
                                                        KeyT* __restrict__ tmp,
                                                        int32_t const* __restrict__ list,
                                                        unsigned int const* __restrict__ n_list,
                                                        RadixPasses passes)
{
  constexpr int R = 3;  // keys per thread and chunk (48 KiB of counters): a chunk is R * 16 runs of 64 keys ("virtual waves")
  __shared__ unsigned int base[256];
  __shared__ unsigned int wave_cnt[16 * R][256];
  __shared__ int one_digit;
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  unsigned int n_cols        = *n_list;
  for (unsigned int c = blockIdx.x; c < n_cols; c += gridDim.x) {
    int32_t v   = list[c];
    OffT b      = offsets[v];
    int64_t deg = static_cast<int64_t>(offsets[v + 1] - b);
    KeyT* in    = keys + b;
    KeyT* out   = tmp + b;
    int moved   = 0;  // passes that moved the keys to the other buffer
    for (int p = 0; p < passes.n; ++p) {
      int sh = passes.shift[p];
      if (threadIdx.x < 256) base[threadIdx.x] = 0;
      if (threadIdx.x == 0) one_digit = 0;
      __syncthreads();
      for (int64_t i = threadIdx.x; i < deg; i += blockDim.x)
        atomicAdd(&base[(unsigned int)(in[i] >> sh) & 255u], 1u);
      __syncthreads();
      // every key has the same digit (bits above what a column's keys differ in): nothing to do
      if (threadIdx.x < 256 && static_cast<int64_t>(base[threadIdx.x]) == deg) one_digit = 1;
      __syncthreads();
      bool const skip = one_digit != 0;
      __syncthreads();  // (the next pass resets the flag)
      if (skip) continue;
      if (wave == 0) {  // exclusive scan of 256 counters by one wave, 4 per lane
        unsigned int c0 = base[lane * 4], c1 = base[lane * 4 + 1], c2 = base[lane * 4 + 2],
                     c3 = base[lane * 4 + 3];
        unsigned int s  = c0 + c1 + c2 + c3, inc = s;
        for (int o = 1; o < 64; o <<= 1) {
          unsigned int t = __shfl_up(inc, o, 64);
          if (lane >= o) inc += t;
        }
        unsigned int ex    = inc - s;
        base[lane * 4]     = ex;
        base[lane * 4 + 1] = ex + c0;
        base[lane * 4 + 2] = ex + c0 + c1;
        base[lane * 4 + 3] = ex + c0 + c1 + c2;
      }
      __syncthreads();
      for (int64_t chunk = 0; chunk < deg; chunk += R * blockDim.x) {
        for (int i = threadIdx.x; i < 16 * R * 256; i += blockDim.x) (&wave_cnt[0][0])[i] = 0;
        __syncthreads();
        KeyT k[R];
        unsigned int d[R], rank[R];
        bool valid[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          int64_t i = chunk + r * (int64_t)blockDim.x + threadIdx.x;
          valid[r]  = i < deg;
          k[r]      = valid[r] ? in[i] : 0;
          d[r]      = (unsigned int)(k[r] >> sh) & 255u;
          // lanes of this run that hold the same digit
          unsigned long long m = __ballot(valid[r]);
#pragma unroll
          for (int bit = 0; bit < 8; ++bit) {
            unsigned long long bal = __ballot(valid[r] && ((d[r] >> bit) & 1u));
            m &= ((d[r] >> bit) & 1u) ? bal : ~bal;
          }
          rank[r] = __popcll(m & lt_mask);
          if (valid[r] && rank[r] == 0) wave_cnt[r * 16 + wave][d[r]] = __popcll(m);
        }
        __syncthreads();
        if (threadIdx.x < 256) {
          unsigned int run = base[threadIdx.x];
          for (int w = 0; w < 16 * R; ++w) {
            unsigned int t            = wave_cnt[w][threadIdx.x];
            wave_cnt[w][threadIdx.x] = run;
            run += t;
          }
          base[threadIdx.x] = run;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (valid[r]) out[wave_cnt[r * 16 + wave][d[r]] + rank[r]] = k[r];
        __syncthreads();
      }
      // the next pass reads what this one wrote (same workgroup, global memory)
      __threadfence_block();
      KeyT* t = in;
      in      = out;
      out     = t;
      ++moved;
    }
    if (moved & 1) {  // result sits in tmp
      for (int64_t i = threadIdx.x; i < deg; i += blockDim.x) out[i] = in[i];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ weighted epilogue
template <typename WT>
__global__ void unpack_weighted_kernel(unsigned long long const* __restrict__ keys, int64_t n,
                                       WT const* __restrict__ w_in, int32_t* __restrict__ indices,
                                       WT* __restrict__ w_out)
{
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned long long k = keys[i];
    indices[i]           = static_cast<int32_t>(k >> 32);
    w_out[i]             = w_in[static_cast<uint32_t>(k)];
  }
}

__global__ void unpack_perm_kernel(unsigned long long const* __restrict__ keys, int64_t n,
                                   int32_t* __restrict__ indices, uint32_t* __restrict__ perm)
{
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned long long k = keys[i];
    indices[i]           = static_cast<int32_t>(k >> 32);
    perm[i]              = static_cast<uint32_t>(k);
  }
}

template <typename OffT>
__global__ void expand_offsets_kernel(OffT const* __restrict__ offsets, int64_t n_major,
                                      int64_t n_edges, int32_t* __restrict__ majors)
{
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_edges; e += stride) {
    // last major m with offsets[m] <= e
    int64_t lo = 0, hi = n_major;  // invariant: offsets[lo] <= e < offsets[hi]
    while (hi - lo > 1) {
      int64_t mid = (lo + hi) >> 1;
      if (static_cast<int64_t>(offsets[mid]) <= e)
        lo = mid;
      else
        hi = mid;
    }
    majors[e] = static_cast<int32_t>(lo);
  }
}

inline int bit_width_u64(uint64_t v)
{
  int b = 0;
  while (v) { ++b; v >>= 1; }
  return b;
}

template <typename OffT, typename KeyT>
void order_columns(hipStream_t stream, OffT const* offsets, KeyT* keys, int64_t n_major,
                   int64_t n_minor, int64_t n_edges, bool ties_in_input_order = true, int sort_from_bit = 0,
                   int sort_to_bit = 64)
{
  if (n_edges < 2) return;
  hipLaunchKernelGGL((sort_small_kernel<OffT, KeyT, 16>), dim3(grid_for(n_major * 16)), dim3(kThreads),
                     0, stream, offsets, keys, n_major, 1);
  hipLaunchKernelGGL((sort_small_kernel<OffT, KeyT, 64>), dim3(grid_for(n_major * 64)), dim3(kThreads),
                     0, stream, offsets, keys, n_major, 16);
  // lists of the longer columns
  int64_t max_long = n_edges / 65 + 1;
  if (max_long > n_major) max_long = n_major;
  DevBuf med(max_long * sizeof(int32_t)), big(max_long * sizeof(int32_t)), wav(max_long * sizeof(int32_t)),
    counts(3 * sizeof(unsigned int));
  CGA_HIP(hipMemsetAsync(counts.p, 0, 3 * sizeof(unsigned int), stream));
  hipLaunchKernelGGL((classify_columns_kernel<OffT>), dim3(grid_for(n_major)), dim3(kThreads), 0,
                     stream, offsets, n_major, med.as<int32_t>(), big.as<int32_t>(), wav.as<int32_t>(),
                     counts.as<unsigned int>());
  unsigned int h_counts[3];
  CGA_HIP(hipMemcpyAsync(h_counts, counts.p, sizeof(h_counts), hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  if (h_counts[2] > 0) {
    int64_t const blocks = (static_cast<int64_t>(h_counts[2]) + 3) / 4;  // 4 waves a workgroup
    hipLaunchKernelGGL((sort_wave_kernel<OffT, KeyT>), dim3(static_cast<unsigned>(std::min<int64_t>(blocks, 1 << 16))),
                       dim3(256), 0, stream, offsets, keys, wav.as<int32_t>(), counts.as<unsigned int>() + 2);
  }
  if (h_counts[0] > 0) {
    int grid = h_counts[0] < 2048u ? (int)h_counts[0] : 2048;
    hipLaunchKernelGGL((sort_medium_kernel<OffT, KeyT>), dim3(grid), dim3(256), 0, stream, offsets,
                       keys, med.as<int32_t>(), counts.as<unsigned int>());
  }
  DevBuf tmp;
  if (h_counts[1] > 0) {
    RadixPasses passes{};
    if constexpr (sizeof(KeyT) == 8) {
      int be = bit_width_u64(static_cast<uint64_t>(n_edges - 1));
      int bv = bit_width_u64(static_cast<uint64_t>(n_minor > 0 ? n_minor - 1 : 0));
      // (the low word is the input position: it only decides the order of equal minors)
      for (int s = 0; s < be && ties_in_input_order; s += 8) passes.shift[passes.n++] = s;
      for (int s = 0; s < bv; s += 8) passes.shift[passes.n++] = 32 + s;
    } else {
      int bv = bit_width_u64(static_cast<uint64_t>(n_minor > 0 ? n_minor - 1 : 0));
      // (sort_from_bit: the low bits of a packed key that need no order, compress_coo_keys)
      for (int s = sort_from_bit; s < bv && s < sort_to_bit; s += 8) passes.shift[passes.n++] = s;
    }
    if (passes.n > 0) {
      tmp.alloc(static_cast<size_t>(n_edges) * sizeof(KeyT));
      int grid = h_counts[1] < 1024u ? (int)h_counts[1] : 1024;
      hipLaunchKernelGGL((sort_big_kernel<OffT, KeyT>), dim3(grid), dim3(1024), 0, stream, offsets,
                         keys, tmp.as<KeyT>(), big.as<int32_t>(), counts.as<unsigned int>() + 1,
                         passes);
    }
  }
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));  // temporaries die here
}

template <typename OffT>
CompressedOut compress_impl(hipStream_t stream, int64_t n_major, int64_t n_minor, int64_t n_edges,
                            int32_t const* majors, int32_t const* minors, void const* weights,
                            size_t weight_size, bool want_perm = false,
                            unsigned int const* major_counts = nullptr, bool ties_in_input_order = true,
                            bool hubs_first = false, int sort_from_bit = 0, int sort_to_bit = 64,
                            bool grouped = false)
{
  CompressedOut out;
  // long columns (512 edges or more on average): LDS-privatised histogram and scatter (hist_lds_kernel)
  bool const no_lds     = std::getenv("CUGRAPH_AMD_BUILD_NO_LDS") != nullptr;
  // (grouped: the caller says that neighbouring edges have neighbouring majors, whatever their number)
  bool const few_majors = n_edges >= (int64_t{1} << 16) && (n_edges / 512 >= n_major || grouped) && !no_lds;
  // majors in descending degree order: the first kLdsMajors of them take a large share of the edges
  bool const lds_scatter = few_majors || (hubs_first && n_edges >= (int64_t{1} << 16) && !no_lds);
  size_t const lds_bytes = static_cast<size_t>(kLdsMajors) * sizeof(unsigned int);
  int const lds_grid     = 1024;
  out.off64 = sizeof(OffT) == 8;
  out.offsets.alloc(static_cast<size_t>(n_major + 1) * sizeof(OffT));
  // +4 elements: the SpMV reads 16-byte aligned vectors and may touch up to 3 elements past the end
  out.indices.alloc(static_cast<size_t>(n_edges + 4) * sizeof(int32_t));
  CGA_HIP(hipMemsetAsync(out.indices.as<char>() + static_cast<size_t>(n_edges) * 4, 0, 16, stream));
  if (weights) {
    out.weights.alloc(static_cast<size_t>(n_edges + 4) * weight_size);
    CGA_HIP(hipMemsetAsync(out.weights.as<char>() + static_cast<size_t>(n_edges) * weight_size, 0,
                           4 * weight_size, stream));
  }

  DevBuf counts(static_cast<size_t>(n_major > 0 ? n_major : 1) * sizeof(unsigned int));
  if (major_counts && n_major > 0) {
    // the caller already counted the edges of every major (the renumbering's degrees)
    CGA_HIP(hipMemcpyAsync(counts.p, major_counts, static_cast<size_t>(n_major) * sizeof(unsigned int),
                           hipMemcpyDeviceToDevice, stream));
  } else {
    CGA_HIP(hipMemsetAsync(counts.p, 0, counts.bytes, stream));
    if (n_edges > 0 && few_majors) {
      CGA_HIP(hipFuncSetAttribute(reinterpret_cast<void const*>(hist_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(kLdsMajors * sizeof(unsigned int))));
      hipLaunchKernelGGL(hist_lds_kernel, dim3(lds_grid), dim3(1024), lds_bytes, stream, majors, n_edges,
                         counts.as<unsigned int>());
    } else if (n_edges > 0) {
      hipLaunchKernelGGL(hist_kernel, dim3(grid_for(n_edges)), dim3(kThreads), 0, stream, majors,
                         n_edges, counts.as<unsigned int>());
    }
  }
  // scatter of (major, minor[, position]) into the columns: keys of type KeyT
  auto scatter = [&](auto* keys) {
    using KeyT = std::remove_pointer_t<decltype(keys)>;
    if (lds_scatter && n_major > 0) {
      auto kern = scatter_lds_kernel<OffT, KeyT>;
      CGA_HIP(hipFuncSetAttribute(reinterpret_cast<void const*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(kLdsMajors * sizeof(unsigned int))));
      hipLaunchKernelGGL(kern, dim3(lds_grid), dim3(1024), lds_bytes, stream, majors, minors, n_edges,
                         out.offsets.template as<OffT>(), n_major, counts.as<unsigned int>(), keys);
      return;
    }
    hipLaunchKernelGGL((scatter_kernel<OffT, KeyT>), dim3(grid_for(n_edges)), dim3(kThreads), 0, stream,
                       majors, minors, n_edges, out.offsets.template as<OffT>(), counts.as<unsigned int>(),
                       keys);
  };

  int64_t n_tiles = (n_major + kScanTile - 1) / kScanTile;
  if (n_tiles < 1) n_tiles = 1;
  DevBuf tile_sums(static_cast<size_t>(n_tiles) * sizeof(unsigned long long));
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)n_tiles), dim3(kThreads), 0, stream,
                     counts.as<unsigned int>(), n_major, tile_sums.as<unsigned long long>());
  hipLaunchKernelGGL(scan_tile_prefix_kernel, dim3(1), dim3(1024), 0, stream,
                     tile_sums.as<unsigned long long>(), n_tiles);
  hipLaunchKernelGGL((scan_write_kernel<OffT>), dim3((unsigned)n_tiles), dim3(kThreads), 0, stream,
                     counts.as<unsigned int>(), n_major, tile_sums.as<unsigned long long>(),
                     out.offsets.as<OffT>());

  if (n_edges > 0) {
    CGA_HIP(hipMemsetAsync(counts.p, 0, counts.bytes, stream));
    if (want_perm) {
      CGA_EXPECTS(n_edges <= 0xffffffffLL, CUGRAPH_NOT_IMPLEMENTED, "2^32 or more edges");
      out.perm.alloc(static_cast<size_t>(n_edges + 4) * sizeof(uint32_t));
      DevBuf keys(static_cast<size_t>(n_edges) * sizeof(unsigned long long));
      scatter(keys.as<unsigned long long>());
      order_columns<OffT, unsigned long long>(stream, out.offsets.as<OffT>(),
                                              keys.as<unsigned long long>(), n_major, n_minor,
                                              n_edges, ties_in_input_order);
      hipLaunchKernelGGL(unpack_perm_kernel, dim3(grid_for(n_edges)), dim3(kThreads), 0, stream,
                         keys.as<unsigned long long>(), n_edges, out.indices.as<int32_t>(),
                         out.perm.as<uint32_t>());
      CGA_HIP(hipGetLastError());
      CGA_HIP(hipStreamSynchronize(stream));
    } else if (weights) {
      CGA_EXPECTS(n_edges <= 0xffffffffLL, CUGRAPH_NOT_IMPLEMENTED,
                  "weighted graphs with 2^32 or more edges per GPU are not supported");
      DevBuf keys(static_cast<size_t>(n_edges) * sizeof(unsigned long long));
      scatter(keys.as<unsigned long long>());
      order_columns<OffT, unsigned long long>(stream, out.offsets.as<OffT>(),
                                              keys.as<unsigned long long>(), n_major, n_minor,
                                              n_edges);
      if (weight_size == 4)
        hipLaunchKernelGGL((unpack_weighted_kernel<float>), dim3(grid_for(n_edges)), dim3(kThreads),
                           0, stream, keys.as<unsigned long long>(), n_edges,
                           static_cast<float const*>(weights), out.indices.as<int32_t>(),
                           out.weights.as<float>());
      else
        hipLaunchKernelGGL((unpack_weighted_kernel<double>), dim3(grid_for(n_edges)),
                           dim3(kThreads), 0, stream, keys.as<unsigned long long>(), n_edges,
                           static_cast<double const*>(weights), out.indices.as<int32_t>(),
                           out.weights.as<double>());
      CGA_HIP(hipGetLastError());
      CGA_HIP(hipStreamSynchronize(stream));
    } else {
      scatter(out.indices.as<uint32_t>());
      order_columns<OffT, uint32_t>(stream, out.offsets.as<OffT>(), out.indices.as<uint32_t>(),
                                    n_major, n_minor, n_edges, true, sort_from_bit, sort_to_bit);
    }
  }
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));
  return out;
}

}  // namespace

// counts[key] += occurrences of key; hubs_first: small keys are the frequent ones (ids of a
// degree-sorted graph), counted through an LDS window
void histogram_u32(hipStream_t stream, int32_t const* keys, int64_t n, unsigned int* counts, bool hubs_first)
{
  if (n == 0) return;
  if (hubs_first && n >= (int64_t{1} << 16) && std::getenv("CUGRAPH_AMD_BUILD_NO_LDS") == nullptr) {
    CGA_HIP(hipFuncSetAttribute(reinterpret_cast<void const*>(hist_lds_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(kLdsMajors * sizeof(unsigned int))));
    hipLaunchKernelGGL(hist_lds_kernel, dim3(1024), dim3(1024), kLdsMajors * sizeof(unsigned int), stream, keys, n,
                       counts);
  } else {
    hipLaunchKernelGGL(hist_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, keys, n, counts);
  }
  CGA_HIP(hipGetLastError());
}

CompressedOut compress_coo(hipStream_t stream, int64_t n_major, int64_t n_minor, int64_t n_edges,
                           int32_t const* majors, int32_t const* minors, void const* weights,
                           size_t weight_size, unsigned int const* major_counts, bool hubs_first)
{
  // same switch as the reference's edge_t choice (cpp/src/c_api/graph_sg.cpp:568-572);
  // CUGRAPH_AMD_FORCE_OFF64=1 takes the 64-bit side on any graph (so that tests reach its kernels)
  if (n_edges < 0x7fffffffLL && std::getenv("CUGRAPH_AMD_FORCE_OFF64") == nullptr)
    return compress_impl<int32_t>(stream, n_major, n_minor, n_edges, majors, minors, weights,
                                  weight_size, false, major_counts, true, hubs_first);
  return compress_impl<int64_t>(stream, n_major, n_minor, n_edges, majors, minors, weights,
                                weight_size, false, major_counts, true, hubs_first);
}

// exclusive scan of 32-bit counts into 64-bit positions, out[n] = total (stream compaction helper)
void exclusive_scan_u32_to_i64(hipStream_t stream, unsigned int const* counts, int64_t n,
                               int64_t* out)
{
  int64_t n_tiles = (n + kScanTile - 1) / kScanTile;
  if (n_tiles < 1) n_tiles = 1;
  DevBuf tile_sums(static_cast<size_t>(n_tiles) * sizeof(unsigned long long));
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)n_tiles), dim3(kThreads), 0, stream,
                     counts, n, tile_sums.as<unsigned long long>());
  hipLaunchKernelGGL(scan_tile_prefix_kernel, dim3(1), dim3(1024), 0, stream,
                     tile_sums.as<unsigned long long>(), n_tiles);
  hipLaunchKernelGGL((scan_write_kernel<int64_t>), dim3((unsigned)n_tiles), dim3(kThreads), 0,
                     stream, counts, n, tile_sums.as<unsigned long long>(), out);
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));  // tile_sums dies here
}

// In-place re-orientation: CSR (majors = sources) <-> CSC (majors = destinations).  The reference
// does the same to the caller's graph when PageRank meets store_transposed == false
// (cpp/src/c_api/pagerank.cpp:105-110, cpp/src/c_api/graph.hpp:57-123).
void transpose_graph(Handle const& h, Graph& g)
{
  CGA_EXPECTS(g.num_global_vertices == g.num_vertices, CUGRAPH_NOT_IMPLEMENTED,
              "transposing a partitioned graph is not supported");
  hipStream_t stream = h.stream;
  DevBuf old_majors(static_cast<size_t>(g.num_edges > 0 ? g.num_edges : 1) * sizeof(int32_t));
  expand_offsets(stream, g.offsets, g.off64, g.num_vertices, g.num_edges, old_majors.as<int32_t>());
  size_t wsize      = g.weights ? dtype_size(g.weight_type) : 0;
  CompressedOut out = compress_coo(stream, g.num_vertices, g.num_vertices, g.num_edges, g.indices,
                                   old_majors.as<int32_t>(), g.weights, wsize);
  (void)hipFree(g.offsets);
  (void)hipFree(g.indices);
  if (g.weights) (void)hipFree(g.weights);
  g.off64   = out.off64;
  g.offsets = out.offsets.detach();
  g.indices = static_cast<int32_t*>(out.indices.detach());
  g.weights = out.weights.p ? out.weights.detach() : nullptr;
  g.store_transposed = !g.store_transposed;
  g.degree_sorted    = false;  // the order was by the OTHER endpoint's degree
  if (g.plan) {
    destroy_plan(g.plan);
    g.plan = nullptr;
  }
  delete g.reversed;
  g.reversed = nullptr;
}

Graph::~Graph()
{
  if (offsets) (void)hipFree(offsets);
  if (indices) (void)hipFree(indices);
  if (weights) (void)hipFree(weights);
  if (number_map) (void)hipFree(number_map);
  if (ext_lookup) (void)hipFree(ext_lookup);
  if (ext_sorted) (void)hipFree(ext_sorted);
  if (x_pos) (void)hipFree(x_pos);
  if (plan) destroy_plan(plan);
  if (mg) destroy_mg_info(mg);
  delete reversed;
}

// as compress_coo, without weights, plus perm[k] = input position of the edge stored at k
CompressedOut compress_coo_perm(hipStream_t stream, int64_t n_major, int64_t n_minor,
                                int64_t n_edges, int32_t const* majors, int32_t const* minors,
                                bool ties_in_input_order)
{
  if (n_edges < 0x7fffffffLL)
    return compress_impl<int32_t>(stream, n_major, n_minor, n_edges, majors, minors, nullptr, 0, true,
                                  nullptr, ties_in_input_order);
  return compress_impl<int64_t>(stream, n_major, n_minor, n_edges, majors, minors, nullptr, 0, true,
                                nullptr, ties_in_input_order);
}

// 32-bit keys grouped by major and, inside a major, ascending in their bits [sort_from_bit,
// sort_to_bit) (the keys of a major agree above; the order of keys that agree in those bits is left
// open): `indices` of the result are the keys.  The majors arrive grouped (see compress_impl).
CompressedOut compress_coo_keys(hipStream_t stream, int64_t n_major, int64_t n_edges, int32_t const* majors,
                                uint32_t const* keys, int sort_from_bit, int sort_to_bit)
{
  int32_t const* as_minors = reinterpret_cast<int32_t const*>(keys);
  int64_t const key_space  = int64_t{1} << 32;
  if (n_edges < 0x7fffffffLL)
    return compress_impl<int32_t>(stream, n_major, key_space, n_edges, majors, as_minors, nullptr, 0, false,
                                  nullptr, true, false, sort_from_bit, sort_to_bit, true);
  return compress_impl<int64_t>(stream, n_major, key_space, n_edges, majors, as_minors, nullptr, 0, false,
                                nullptr, true, false, sort_from_bit, sort_to_bit, true);
}

void expand_offsets(hipStream_t stream, void const* offsets, bool off64, int64_t n_major,
                    int64_t n_edges, int32_t* majors_out)
{
  if (n_edges == 0) return;
  if (off64)
    hipLaunchKernelGGL((expand_offsets_kernel<int64_t>), dim3(grid_for(n_edges)), dim3(kThreads), 0,
                       stream, static_cast<int64_t const*>(offsets), n_major, n_edges, majors_out);
  else
    hipLaunchKernelGGL((expand_offsets_kernel<int32_t>), dim3(grid_for(n_edges)), dim3(kThreads), 0,
                       stream, static_cast<int32_t const*>(offsets), n_major, n_edges, majors_out);
  CGA_HIP(hipGetLastError());
}

int64_t narrow_ids(hipStream_t stream, void const* in, cugraph_data_type_id_t type, size_t n,
                   int32_t* out, int64_t* min_out)
{
  DevBuf mm(2 * sizeof(long long));
  long long init[2] = {0x7fffffffffffffffLL, -1};
  CGA_HIP(hipMemcpyAsync(mm.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
  if (n > 0) {
    if (type == INT32)
      hipLaunchKernelGGL((narrow_kernel<int32_t>), dim3(grid_for((int64_t)n)), dim3(kThreads), 0,
                         stream, static_cast<int32_t const*>(in), n, out, mm.as<long long>());
    else
      hipLaunchKernelGGL((narrow_kernel<int64_t>), dim3(grid_for((int64_t)n)), dim3(kThreads), 0,
                         stream, static_cast<int64_t const*>(in), n, out, mm.as<long long>());
    CGA_HIP(hipGetLastError());
  }
  long long res[2];
  CGA_HIP(hipMemcpyAsync(res, mm.p, sizeof(res), hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  if (min_out) *min_out = res[0];
  return res[1];
}

void shift_ids(hipStream_t stream, int32_t const* in, int32_t* out, int64_t n, int64_t shift)
{
  if (n == 0) return;
  hipLaunchKernelGGL(shift_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, in, out, n,
                     static_cast<int32_t>(shift));
  CGA_HIP(hipGetLastError());
}

void widen_ids(hipStream_t stream, int32_t const* in, int64_t* out, int64_t n)
{
  if (n == 0) return;
  hipLaunchKernelGGL(widen_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, in, out, n);
  CGA_HIP(hipGetLastError());
}

void fill_sequence(hipStream_t stream, void* out, cugraph_data_type_id_t type, size_t n,
                   int64_t first)
{
  if (n == 0) return;
  if (type == INT32)
    hipLaunchKernelGGL((sequence_kernel<int32_t>), dim3(grid_for((int64_t)n)), dim3(kThreads), 0,
                       stream, static_cast<int32_t*>(out), n, (long long)first);
  else
    hipLaunchKernelGGL((sequence_kernel<int64_t>), dim3(grid_for((int64_t)n)), dim3(kThreads), 0,
                       stream, static_cast<int64_t*>(out), n, (long long)first);
  CGA_HIP(hipGetLastError());
}

}  // namespace cga
