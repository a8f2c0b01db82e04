// Vertex renumbering: external ids -> dense internal ids ordered by DESCENDING major degree
// (in-degree for the transposed / pull layout), ties by ascending external id.
//
// Replaces compute_renumber_map / renumber_edgelist (reference:
// cpp/src/structure/renumber_edgelist_impl.cuh:238-607, 1056-1104).  The reference gets its unique
// vertex set from E-sized sorts + merges, its degrees from sort + reduce_by_key, orders by
// sort_by_key(greater) and relabels through a cuco hash map.  Here:
//   - the vertex set comes from a direct-address flag table over [min id, max id] + one scan
//     while the ids span less than 2^31; wider id spaces (hashed 64-bit ids) go through an
//     open-addressing hash set of the distinct ids, which are then sorted and searched;
//   - degrees come from the same histogram kernel as the CSC build;
//   - the order is a stable LSD radix sort, 8 bits per pass (count + scan + scatter), so the
//     result is deterministic;
//   - relabelling is a table lookup.
// Degree-descending order is what gives the SpMV contiguous degree buckets (the reference's
// segment_offsets, renumber_edgelist_impl.cuh:546-603) and packs the hot sources of a
// power-law graph into a few MB of the gathered vector.
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

__global__ void max_u32_kernel(unsigned int const* v, int64_t n, unsigned int* out)
{
  unsigned int m = 0;
  GRID_STRIDE(i, n) m = v[i] > m ? v[i] : m;
  for (int o = 32; o > 0; o >>= 1) {
    unsigned int t = __shfl_xor(m, o, 64);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

__global__ void iota_kernel(int32_t* v, int64_t n) { GRID_STRIDE(i, n) v[i] = (int32_t)i; }

__global__ void fill_u64_kernel(unsigned long long* v, int64_t n, unsigned long long value)
{
  GRID_STRIDE(i, n) v[i] = value;
}

__global__ void gather_u32_kernel(unsigned int const* in, int32_t const* order, int64_t n, unsigned int* out)
{
  GRID_STRIDE(i, n) out[i] = in[order[i]];
}

__global__ void invert_kernel(int32_t const* order, int64_t n, int32_t* rank)
{
  GRID_STRIDE(i, n) rank[order[i]] = (int32_t)i;
}

__global__ void relabel_kernel(int32_t* ids, int64_t n, int32_t const* table)
{
  GRID_STRIDE(i, n) ids[i] = table[ids[i]];
}

template <typename OffT>
__global__ void degree_from_offsets_kernel(OffT const* offsets, int64_t n, unsigned int* deg)
{
  GRID_STRIDE(i, n) deg[i] = (unsigned int)(offsets[i + 1] - offsets[i]);
}

__global__ void hist_u32_kernel(int32_t const* keys, int64_t n, unsigned int* counts)
{
  GRID_STRIDE(i, n) atomicAdd(&counts[keys[i]], 1u);
}

// ---- arbitrary external ids -> dense ids (ascending id order)
template <typename IdT>
__global__ void mark_ids_kernel(IdT const* ids, int64_t n, long long lo, unsigned int* flags)
{
  GRID_STRIDE(i, n) flags[(long long)ids[i] - lo] = 1u;
}

template <typename IdT>
__global__ void to_dense_kernel(IdT const* ids, int64_t n, long long lo, unsigned int const* flags,
                                int64_t const* pos, int32_t* out, unsigned int* n_unknown)
{
  GRID_STRIDE(i, n)
  {
    long long k = (long long)ids[i] - lo;
    if (!flags[k]) atomicAdd(n_unknown, 1u);
    out[i] = (int32_t)pos[k];
  }
}

// the table path of renumber_edgelist: tally[id - lo] = edges whose major is id (low 31 bits),
// top bit: id occurs as a minor
constexpr unsigned int kMinorBit = 0x80000000u;

template <typename IdT>
__global__ void tally_majors_kernel(IdT const* __restrict__ ids, int64_t n, long long lo, unsigned int* tally)
{
  GRID_STRIDE(i, n) atomicAdd(&tally[(long long)ids[i] - lo], 1u);
}

template <typename IdT>
__global__ void tally_minors_kernel(IdT const* __restrict__ ids, int64_t n, long long lo, unsigned int* tally)
{
  GRID_STRIDE(i, n)
  {
    unsigned int* t = &tally[(long long)ids[i] - lo];
    if (!(__atomic_load_n(t, __ATOMIC_RELAXED) & kMinorBit)) atomicOr(t, kMinorBit);
  }
}

__global__ void flags_of_tally_kernel(unsigned int const* __restrict__ tally, int64_t n, unsigned int* flags)
{
  GRID_STRIDE(k, n) flags[k] = tally[k] ? 1u : 0u;
}

__global__ void count_unlisted_kernel(unsigned int const* __restrict__ tally, unsigned int const* __restrict__ flags,
                                      int64_t n, unsigned int* n_unknown)
{
  unsigned int c = 0;
  GRID_STRIDE(k, n) c += (tally[k] && !flags[k]) ? 1u : 0u;
  if (c) atomicAdd(n_unknown, c);
}

__global__ void compact_tally_kernel(unsigned int const* __restrict__ tally, unsigned int const* __restrict__ flags,
                                     int64_t const* __restrict__ pos, int64_t n, unsigned int* __restrict__ deg,
                                     unsigned int* __restrict__ is_minor)
{
  GRID_STRIDE(k, n)
  {
    if (flags[k]) {
      unsigned int const t = tally[k];
      deg[pos[k]]          = t & ~kMinorBit;
      is_minor[pos[k]]     = t >> 31;
    }
  }
}

// out[k] = vertices of degree >= kLowDegreeMax + 1 - k, k = 0 .. kLowDegreeMax
__global__ void count_degree_classes_kernel(unsigned int const* __restrict__ deg, int64_t n, unsigned long long* out)
{
  unsigned long long c[kLowDegreeMax + 1] = {};
  GRID_STRIDE(i, n)
  {
    unsigned int const d = deg[i];
#pragma unroll
    for (int k = 0; k <= kLowDegreeMax; ++k) c[k] += d >= static_cast<unsigned int>(kLowDegreeMax + 1 - k);
  }
  for (int k = 0; k <= kLowDegreeMax; ++k) {
    for (int o = 32; o > 0; o >>= 1) c[k] += __shfl_xor(c[k], o, 64);
    if ((threadIdx.x & 63) == 0 && c[k]) atomicAdd(&out[k], c[k]);
  }
}

// smallest (provisional) new id among the sources of every row of degree 1-3: the rows whose new
// id lies in [first_low, first_low + n_low)
__global__ void low_min_source_kernel(int32_t const* __restrict__ majors, int32_t const* __restrict__ minors,
                                      int64_t n, int64_t first_low, int64_t n_low, unsigned int* __restrict__ minsrc)
{
  GRID_STRIDE(i, n)
  {
    int64_t const row = static_cast<int64_t>(majors[i]) - first_low;
    if (row >= 0 && row < n_low) atomicMin(&minsrc[row], static_cast<unsigned int>(minors[i]));
  }
}

// ids inside [first, first + n_low) -> first + remap[id - first]
__global__ void remap_low_ids_kernel(int32_t* __restrict__ ids, int64_t n, int32_t first, unsigned int n_low,
                                     int32_t const* __restrict__ remap)
{
  GRID_STRIDE(i, n)
  {
    unsigned int const d = static_cast<unsigned int>(ids[i] - first);
    if (d < n_low) ids[i] = first + remap[d];
  }
}

__global__ void offset_i32_kernel(int32_t const* in, int64_t n, int32_t add, int32_t* out)
{
  GRID_STRIDE(i, n) out[i] = in[i] + add;
}

__global__ void flip_keys_kernel(unsigned int* keys, int64_t n, unsigned int top)
{
  GRID_STRIDE(i, n) keys[i] = top - keys[i];
}

__global__ void gather_i32_kernel(int32_t const* in, int32_t const* order, int64_t n, int32_t* out)
{
  GRID_STRIDE(i, n) out[i] = in[order[i]];
}

// endpoint -> new id through the external -> new table (every endpoint is a known vertex by now)
template <typename IdT>
__global__ void lookup_ids_kernel(IdT const* __restrict__ ids, int64_t n, long long lo,
                                  int32_t const* __restrict__ lookup, int32_t* __restrict__ out)
{
  GRID_STRIDE(i, n) out[i] = lookup[(long long)ids[i] - lo];
}

template <typename IdT>
__global__ void number_map_kernel(int64_t range, long long lo, unsigned int const* flags,
                                  int64_t const* pos, int32_t const* rank, IdT* number_map,
                                  int32_t* lookup)
{
  GRID_STRIDE(k, range)
  {
    if (flags[k]) {
      int32_t nw     = rank[pos[k]];
      number_map[nw] = (IdT)(lo + k);
      lookup[k]      = nw;
    } else {
      lookup[k] = -1;
    }
  }
}

// ---- ids spanning 2^31 or more: distinct ids through a hash set, then sorted
constexpr unsigned long long kEmptySlot = 0x8000000000000000ull;  // INT64_MIN: never a legal id here

__device__ inline unsigned long long mix64(unsigned long long x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  return x ^ (x >> 33);
}

template <typename IdT>
__global__ void hash_insert_kernel(IdT const* __restrict__ ids, int64_t n, unsigned long long* table,
                                   unsigned long long mask, unsigned int* saw_sentinel)
{
  GRID_STRIDE(i, n)
  {
    unsigned long long const v = static_cast<unsigned long long>(static_cast<long long>(ids[i]));
    if (v == kEmptySlot) {
      *saw_sentinel = 1u;
      continue;
    }
    unsigned long long h = mix64(v) & mask;
    // the table has twice as many slots as keys (or 2^32, more than twice the 2^31 ids a graph
    // may have): a probe ends, unless there are more distinct ids than any graph here can hold
    unsigned long long probes = 0;
    while (true) {
      unsigned long long const old = atomicCAS(&table[h], kEmptySlot, v);
      if (old == kEmptySlot || old == v) break;
      h = (h + 1) & mask;
      if (++probes > mask) {
        *saw_sentinel = 2u;
        break;
      }
    }
  }
}

__global__ void slot_used_kernel(unsigned long long const* __restrict__ table, int64_t n, unsigned int* flags)
{
  GRID_STRIDE(i, n) flags[i] = table[i] != kEmptySlot ? 1u : 0u;
}

// occupied slots -> keys biased by the smallest id (so that unsigned order = id order)
__global__ void slot_compact_kernel(unsigned long long const* __restrict__ table, int64_t n,
                                    int64_t const* __restrict__ pos, unsigned long long lo,
                                    unsigned long long* __restrict__ keys)
{
  GRID_STRIDE(i, n)
  {
    unsigned long long v = table[i];
    if (v != kEmptySlot) keys[pos[i]] = v - lo;
  }
}

__global__ void bit_flag64_kernel(unsigned long long const* keys, int64_t n, int bit, unsigned int* flags)
{
  GRID_STRIDE(i, n) flags[i] = static_cast<unsigned int>((keys[i] >> bit) & 1ull);
}

// stable split, zeros first (ascending order)
__global__ void bit_split64_kernel(unsigned long long const* keys_in, int64_t n, int bit,
                                   int64_t const* ones_before, unsigned long long* keys_out)
{
  int64_t const total_zeros = n - ones_before[n];
  GRID_STRIDE(i, n)
  {
    unsigned long long k = keys_in[i];
    int64_t ob           = ones_before[i];
    keys_out[((k >> bit) & 1ull) ? total_zeros + ob : i - ob] = k;
  }
}

// sorted biased keys -> the ids themselves (int64)
__global__ void unbias_kernel(unsigned long long const* keys, int64_t n, unsigned long long lo, long long* ids)
{
  GRID_STRIDE(i, n) ids[i] = static_cast<long long>(keys[i] + lo);
}

// endpoint -> its position in the sorted distinct ids
template <typename IdT>
__global__ void to_dense_sorted_kernel(IdT const* __restrict__ ids, int64_t n,
                                       long long const* __restrict__ sorted, int64_t n_sorted,
                                       int32_t* __restrict__ out, unsigned int* n_unknown)
{
  GRID_STRIDE(i, n)
  {
    int64_t k = find_sorted(sorted, n_sorted, static_cast<long long>(ids[i]));
    if (k < 0) {
      atomicAdd(n_unknown, 1u);
      k = 0;
    }
    out[i] = static_cast<int32_t>(k);
  }
}

template <typename IdT>
__global__ void number_map_sorted_kernel(long long const* __restrict__ sorted, int64_t n,
                                         int32_t const* __restrict__ rank, IdT* __restrict__ number_map)
{
  GRID_STRIDE(k, n) number_map[rank[k]] = static_cast<IdT>(sorted[k]);
}

template <typename IdT>
__global__ void minmax_kernel(IdT const* in, int64_t n, long long* minmax)
{
  long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
  GRID_STRIDE(i, n)
  {
    long long v = (long long)in[i];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
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

inline int bit_width_u32(unsigned int v)
{
  int b = 0;
  while (v) { ++b; v >>= 1; }
  return b;
}

}  // namespace

namespace {
__global__ void mark_ids_kernel(int32_t const* __restrict__ ids, int64_t n, unsigned int* __restrict__ flags)
{
  GRID_STRIDE(i, n) { flags[ids[i]] = 1u; }
}

// sort key of a vertex: its major degree; among the vertices of degree 0 the ones that appear as
// a minor (an edge leaves them in a transposed graph) come first
__global__ void degree_keys_kernel(unsigned int const* __restrict__ deg, unsigned int const* __restrict__ is_minor,
                                   int64_t n, unsigned int* __restrict__ keys)
{
  GRID_STRIDE(i, n)
  {
    unsigned int d = deg[i] < 0x7fffffffu ? deg[i] : 0x7ffffffeu;
    keys[i]        = 2u * d + (d > 0 ? 1u : is_minor[i]);
  }
}
}  // namespace

// Order of a degree-sorted graph: descending major degree, ties by ascending id -- except among
// the vertices WITHOUT major edges, where those that still occur as a minor come first.  What is
// left at the very end are the vertices no edge touches; PageRank never has to visit them
// (pagerank.hip: dead rows).
// the same order from "occurs as a minor" flags the caller already has
void degree_order_marked(hipStream_t stream, unsigned int const* deg, unsigned int const* is_minor, int64_t n,
                         DevBuf& order, DevBuf& rank)
{
  size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  DevBuf keys(nn * 4);
  if (n > 0)
    hipLaunchKernelGGL(degree_keys_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, deg, is_minor, n,
                       keys.as<unsigned int>());
  degree_order(stream, keys.as<unsigned int>(), n, order, rank);
}

void degree_order_sources_first(hipStream_t stream, unsigned int const* deg, int32_t const* minors,
                                int64_t n_edges, int64_t n, DevBuf& order, DevBuf& rank)
{
  size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  DevBuf is_minor(nn * 4), keys(nn * 4);
  CGA_HIP(hipMemsetAsync(is_minor.p, 0, is_minor.bytes, stream));
  if (n_edges > 0)
    hipLaunchKernelGGL(mark_ids_kernel, dim3(grid_for(n_edges)), dim3(kThreads), 0, stream, minors, n_edges,
                       is_minor.as<unsigned int>());
  if (n > 0)
    hipLaunchKernelGGL(degree_keys_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, deg,
                       is_minor.as<unsigned int>(), n, keys.as<unsigned int>());
  degree_order(stream, keys.as<unsigned int>(), n, order, rank);
}

// ---- stable LSD radix sort of (key, value) pairs, descending keys, 8 bits per pass over many
// workgroups: per-tile digit counts (digit-major, so one exclusive scan gives every tile's first
// output position of every digit), then a scatter that ranks the keys of a tile in input order.
constexpr int kRsThreads = 256;
constexpr int kRsRounds  = 16;
constexpr int kRsTile    = kRsThreads * kRsRounds;

// lanes of the wave that hold the same digit as this one (valid lanes only)
__device__ inline unsigned long long same_digit_lanes(unsigned int d, bool valid)
{
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int bit = 0; bit < 8; ++bit) {
    unsigned long long const bal = __ballot(valid && ((d >> bit) & 1u));
    m &= ((d >> bit) & 1u) ? bal : ~bal;
  }
  return m;
}

__global__ __launch_bounds__(kRsThreads) void radix_count_kernel(unsigned int const* __restrict__ keys, int64_t n,
                                                                 int shift, int64_t n_tiles,
                                                                 unsigned int* __restrict__ counts)
{
  __shared__ unsigned int cnt[256];
  int const lane = threadIdx.x & 63;
  unsigned long long const below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  cnt[threadIdx.x] = 0u;
  __syncthreads();
  int64_t const base = blockIdx.x * (int64_t)kRsTile;
  for (int r = 0; r < kRsRounds; ++r) {
    int64_t const i      = base + r * kRsThreads + threadIdx.x;
    bool const valid     = i < n;
    unsigned int const d = 255u - ((valid ? keys[i] >> shift : 0u) & 255u);  // descending
    // (degrees are skewed: most keys of a tile share a digit, so one LDS atomic per wave and digit)
    unsigned long long const m = same_digit_lanes(d, valid);
    if (valid && (m & below) == 0ull) atomicAdd(&cnt[d], static_cast<unsigned int>(__popcll(m)));
  }
  __syncthreads();
  counts[threadIdx.x * n_tiles + blockIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(kRsThreads) void radix_scatter_kernel(
  unsigned int const* __restrict__ keys_in, int32_t const* __restrict__ vals_in, int64_t n, int shift,
  int64_t n_tiles, int64_t const* __restrict__ first_pos, unsigned int* __restrict__ keys_out,
  int32_t* __restrict__ vals_out)
{
  __shared__ int64_t run[256];  // next output position of every digit
  __shared__ int64_t wave_pos[kRsThreads / 64][256];
  __shared__ unsigned int wave_cnt[kRsThreads / 64][256];
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long const below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  run[threadIdx.x] = first_pos[threadIdx.x * n_tiles + blockIdx.x];
  int64_t const base = blockIdx.x * (int64_t)kRsTile;
  for (int r = 0; r < kRsRounds; ++r) {
    int64_t const i  = base + r * kRsThreads + threadIdx.x;  // round-major: input order
    bool const valid = i < n;
#pragma unroll
    for (int w = 0; w < kRsThreads / 64; ++w) wave_cnt[w][threadIdx.x] = 0u;
    __syncthreads();
    unsigned int const k = valid ? keys_in[i] : 0u;
    unsigned int const d = 255u - ((k >> shift) & 255u);
    unsigned long long const m = same_digit_lanes(d, valid);
    int const rank = __popcll(m & below);
    if (valid && rank == 0) wave_cnt[wave][d] = static_cast<unsigned int>(__popcll(m));
    __syncthreads();
    {  // thread = digit: where each wave's keys of this digit start
      int64_t p = run[threadIdx.x];
#pragma unroll
      for (int w = 0; w < kRsThreads / 64; ++w) {
        wave_pos[w][threadIdx.x] = p;
        p += wave_cnt[w][threadIdx.x];
      }
      run[threadIdx.x] = p;
    }
    __syncthreads();
    if (valid) {
      int64_t const p = wave_pos[wave][d] + rank;
      keys_out[p]     = k;
      vals_out[p]     = vals_in[i];
    }
  }
}

// order[new] = old, rank[old] = new; stable, descending degree
void degree_order(hipStream_t stream, unsigned int const* deg, int64_t n, DevBuf& order, DevBuf& rank)
{
  size_t nn = static_cast<size_t>(n > 0 ? n : 1);
  order.alloc(nn * sizeof(int32_t));
  rank.alloc(nn * sizeof(int32_t));
  if (n == 0) return;
  int64_t const n_tiles = (n + kRsTile - 1) / kRsTile;
  size_t const n_counts = static_cast<size_t>(n_tiles) * 256;
  DevBuf keys_a(nn * 4), keys_b(nn * 4), vals_b(nn * 4), counts(n_counts * 4), pos((n_counts + 1) * 8), mx(4);
  CGA_HIP(hipMemcpyAsync(keys_a.p, deg, nn * 4, hipMemcpyDeviceToDevice, stream));
  CGA_HIP(hipMemsetAsync(mx.p, 0, 4, stream));
  hipLaunchKernelGGL(max_u32_kernel, dim3(grid_for(n, 1024)), dim3(kThreads), 0, stream, deg, n,
                     mx.as<unsigned int>());
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, order.as<int32_t>(), n);
  unsigned int h_max = 0;
  CGA_HIP(hipMemcpyAsync(&h_max, mx.p, 4, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  unsigned int* kin  = keys_a.as<unsigned int>();
  unsigned int* kout = keys_b.as<unsigned int>();
  int32_t* vin       = order.as<int32_t>();
  int32_t* vout      = vals_b.as<int32_t>();
  int bits           = bit_width_u32(h_max);
  for (int shift = 0; shift < bits; shift += 8) {
    hipLaunchKernelGGL(radix_count_kernel, dim3(static_cast<unsigned>(n_tiles)), dim3(kRsThreads), 0, stream, kin,
                       n, shift, n_tiles, counts.as<unsigned int>());
    exclusive_scan_u32_to_i64(stream, counts.as<unsigned int>(), static_cast<int64_t>(n_counts), pos.as<int64_t>());
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(static_cast<unsigned>(n_tiles)), dim3(kRsThreads), 0, stream,
                       kin, vin, n, shift, n_tiles, pos.as<int64_t>(), kout, vout);
    std::swap(kin, kout);
    std::swap(vin, vout);
  }
  if (vin != order.as<int32_t>())
    CGA_HIP(hipMemcpyAsync(order.p, vin, nn * 4, hipMemcpyDeviceToDevice, stream));
  hipLaunchKernelGGL(invert_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream,
                     order.as<int32_t>(), n, rank.as<int32_t>());
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));
}

void relabel_ids(hipStream_t stream, int32_t* ids, int64_t n, int32_t const* table)
{
  if (n == 0) return;
  hipLaunchKernelGGL(relabel_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, ids, n, table);
  CGA_HIP(hipGetLastError());
}

namespace {
template <typename IdT>
__global__ void local_hist_kernel(IdT const* ids, int64_t n, long long first, long long count,
                                  unsigned int* counts, unsigned int* n_bad)
{
  GRID_STRIDE(i, n)
  {
    long long k = (long long)ids[i] - first;
    if (k < 0 || k >= count)
      atomicAdd(n_bad, 1u);
    else
      atomicAdd(&counts[k], 1u);
  }
}

template <typename OffT>
__global__ void degree_descending_kernel(OffT const* offsets, int64_t n, unsigned int* n_up)
{
  GRID_STRIDE(i, n - 1)
  {
    if (offsets[i + 2] - offsets[i + 1] > offsets[i + 1] - offsets[i]) atomicAdd(n_up, 1u);
  }
}
}  // namespace

// in-degree order of the destinations [first, first + count) a rank owns, from the destination
// ids of its edges: order[new] = old - first, rank[old - first] = new (stable, descending)
void partition_degree_order(hipStream_t stream, ArrayView const& dst, int64_t first, int64_t count,
                            int32_t* order_out, int32_t* rank_out)
{
  size_t nn = static_cast<size_t>(count > 0 ? count : 1);
  DevBuf deg(nn * 4), bad(4);
  CGA_HIP(hipMemsetAsync(deg.p, 0, deg.bytes, stream));
  CGA_HIP(hipMemsetAsync(bad.p, 0, 4, stream));
  int64_t ne = static_cast<int64_t>(dst.size);
  if (ne > 0) {
    if (dst.type == INT64)
      hipLaunchKernelGGL((local_hist_kernel<int64_t>), dim3(grid_for(ne)), dim3(kThreads), 0, stream,
                         static_cast<int64_t const*>(dst.data), ne, first, count,
                         deg.as<unsigned int>(), bad.as<unsigned int>());
    else
      hipLaunchKernelGGL((local_hist_kernel<int32_t>), dim3(grid_for(ne)), dim3(kThreads), 0, stream,
                         static_cast<int32_t const*>(dst.data), ne, first, count,
                         deg.as<unsigned int>(), bad.as<unsigned int>());
    CGA_HIP(hipGetLastError());
  }
  unsigned int h_bad = 0;
  CGA_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  CGA_EXPECTS(h_bad == 0, CUGRAPH_INVALID_INPUT, "destination ids outside this rank's vertex range");
  if (count == 0) return;
  DevBuf order, rank;
  degree_order(stream, deg.as<unsigned int>(), count, order, rank);
  CGA_HIP(hipMemcpyAsync(order_out, order.p, nn * 4, hipMemcpyDeviceToDevice, stream));
  CGA_HIP(hipMemcpyAsync(rank_out, rank.p, nn * 4, hipMemcpyDeviceToDevice, stream));
  CGA_HIP(hipStreamSynchronize(stream));
}

// ---- edges grouped by the rank that owns their destination (the send side of the multi-GPU edge
// shuffle; reference: shuffle_ext_vertex_pairs... / groupby_and_count over the 2D grid,
// cpp/src/detail/shuffle_vertex_pairs.cu).  A stable multisplit by a key of at most 256 values: per
// tile and owner counts -> one scan -> scatter with ballot-matched ranks, the same three steps as a
// pass of degree_order's radix sort, moving the edge arrays themselves.
namespace {
constexpr int kGeMaxRanks = 256;

template <typename IdT>
__device__ inline unsigned int owner_of(IdT d, long long const* s_bounds, int n_ranks)
{
  // last rank whose first destination is <= d (bounds ascend; empty ranges share a bound)
  int lo = 0, hi = n_ranks - 1;
  while (lo < hi) {
    int const mid = (lo + hi + 1) >> 1;
    if (s_bounds[mid] <= static_cast<long long>(d)) lo = mid; else hi = mid - 1;
  }
  return static_cast<unsigned int>(lo);
}

template <typename IdT>
__global__ __launch_bounds__(kRsThreads) void group_count_kernel(IdT const* __restrict__ dst, int64_t n,
                                                                 long long const* __restrict__ bounds, int n_ranks,
                                                                 int64_t n_tiles, unsigned int* __restrict__ counts,
                                                                 unsigned int* __restrict__ n_bad)
{
  __shared__ unsigned int cnt[kGeMaxRanks];
  __shared__ long long s_bounds[kGeMaxRanks + 1];
  int const lane = threadIdx.x & 63;
  unsigned long long const below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  cnt[threadIdx.x] = 0u;
  for (int k = threadIdx.x; k <= n_ranks; k += kRsThreads) s_bounds[k] = bounds[k];
  __syncthreads();
  int64_t const base = blockIdx.x * (int64_t)kRsTile;
  unsigned int bad   = 0u;
  for (int r = 0; r < kRsRounds; ++r) {
    int64_t const i  = base + r * kRsThreads + threadIdx.x;
    bool const valid = i < n;
    IdT const d      = valid ? dst[i] : IdT(0);
    bad += valid && (static_cast<long long>(d) < s_bounds[0] || static_cast<long long>(d) >= s_bounds[n_ranks]);
    unsigned int const o       = owner_of(d, s_bounds, n_ranks);
    unsigned long long const m = same_digit_lanes(o, valid);
    if (valid && (m & below) == 0ull) atomicAdd(&cnt[o], static_cast<unsigned int>(__popcll(m)));
  }
  if (bad) atomicAdd(n_bad, bad);
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < n_ranks) counts[threadIdx.x * n_tiles + blockIdx.x] = cnt[threadIdx.x];
}

template <typename IdT, typename WgtT>
__global__ __launch_bounds__(kRsThreads) void group_scatter_kernel(
  IdT const* __restrict__ src, IdT const* __restrict__ dst, WgtT const* __restrict__ wgt, int64_t n,
  long long const* __restrict__ bounds, int n_ranks, int64_t n_tiles, int64_t const* __restrict__ first_pos,
  IdT* __restrict__ src_out, IdT* __restrict__ dst_out, WgtT* __restrict__ wgt_out)
{
  __shared__ int64_t run[kGeMaxRanks];  // next output position of every owner
  __shared__ int64_t wave_pos[kRsThreads / 64][kGeMaxRanks];
  __shared__ unsigned int wave_cnt[kRsThreads / 64][kGeMaxRanks];
  __shared__ long long s_bounds[kGeMaxRanks + 1];
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long const below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  run[threadIdx.x] = static_cast<int>(threadIdx.x) < n_ranks ? first_pos[threadIdx.x * n_tiles + blockIdx.x] : 0;
  for (int k = threadIdx.x; k <= n_ranks; k += kRsThreads) s_bounds[k] = bounds[k];
  int64_t const base = blockIdx.x * (int64_t)kRsTile;
  for (int r = 0; r < kRsRounds; ++r) {
    int64_t const i  = base + r * kRsThreads + threadIdx.x;  // round-major: input order
    bool const valid = i < n;
#pragma unroll
    for (int w = 0; w < kRsThreads / 64; ++w) wave_cnt[w][threadIdx.x] = 0u;
    __syncthreads();
    IdT const d                = valid ? dst[i] : IdT(0);
    unsigned int const o       = owner_of(d, s_bounds, n_ranks);
    unsigned long long const m = same_digit_lanes(o, valid);
    int const rank             = __popcll(m & below);
    if (valid && rank == 0) wave_cnt[wave][o] = static_cast<unsigned int>(__popcll(m));
    __syncthreads();
    {  // thread = owner: where each wave's edges of this owner start
      int64_t p = run[threadIdx.x];
#pragma unroll
      for (int w = 0; w < kRsThreads / 64; ++w) {
        wave_pos[w][threadIdx.x] = p;
        p += wave_cnt[w][threadIdx.x];
      }
      run[threadIdx.x] = p;
    }
    __syncthreads();
    if (valid) {
      int64_t const p = wave_pos[wave][o] + rank;
      src_out[p]      = src[i];
      dst_out[p]      = d;
      if (wgt) wgt_out[p] = wgt[i];
    }
  }
}

__global__ void owner_totals_kernel(int64_t const* __restrict__ first_pos, int64_t n_tiles, int n_ranks, int64_t n,
                                    long long* __restrict__ totals)
{
  int const o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o < n_ranks) {
    int64_t const beg = first_pos[static_cast<int64_t>(o) * n_tiles];
    int64_t const end = o + 1 < n_ranks ? first_pos[static_cast<int64_t>(o + 1) * n_tiles] : n;
    totals[o]         = end - beg;
  }
}
}  // namespace

// src / dst (same integer type) and optional weights of one rank's share of the edge list ->
// the same edges grouped by the owner of their destination, input order kept inside a group;
// counts_out[r] = edges that go to rank r.  bounds: host, n_ranks + 1 ascending destination bounds.
void partition_group_edges(hipStream_t stream, ArrayView const& src, ArrayView const& dst, ArrayView const* wgt,
                           int64_t const* bounds, int n_ranks, void* src_out, void* dst_out, void* wgt_out,
                           int64_t* counts_out)
{
  CGA_EXPECTS(n_ranks >= 1 && n_ranks <= kGeMaxRanks, CUGRAPH_INVALID_INPUT, "1 .. 256 ranks");
  int64_t const n = static_cast<int64_t>(dst.size);
  for (int r = 0; r < n_ranks; ++r) counts_out[r] = 0;
  if (n == 0) return;
  int64_t const n_tiles = (n + kRsTile - 1) / kRsTile;
  size_t const n_counts = static_cast<size_t>(n_tiles) * n_ranks;
  DevBuf d_bounds(static_cast<size_t>(n_ranks + 1) * 8), counts(n_counts * 4), pos((n_counts + 1) * 8), bad(4),
    totals(static_cast<size_t>(n_ranks) * 8);
  std::vector<long long> h_bounds(bounds, bounds + n_ranks + 1);
  CGA_HIP(hipMemcpyAsync(d_bounds.p, h_bounds.data(), d_bounds.bytes, hipMemcpyHostToDevice, stream));
  CGA_HIP(hipMemsetAsync(bad.p, 0, 4, stream));
  auto run = [&](auto id_tag) {
    using IdT = decltype(id_tag);
    hipLaunchKernelGGL((group_count_kernel<IdT>), dim3(static_cast<unsigned>(n_tiles)), dim3(kRsThreads), 0, stream,
                       static_cast<IdT const*>(dst.data), n, d_bounds.as<long long>(), n_ranks, n_tiles,
                       counts.as<unsigned int>(), bad.as<unsigned int>());
    exclusive_scan_u32_to_i64(stream, counts.as<unsigned int>(), static_cast<int64_t>(n_counts), pos.as<int64_t>());
    auto scatter = [&](auto w_tag) {
      using WgtT = decltype(w_tag);
      hipLaunchKernelGGL((group_scatter_kernel<IdT, WgtT>), dim3(static_cast<unsigned>(n_tiles)), dim3(kRsThreads), 0,
                         stream, static_cast<IdT const*>(src.data), static_cast<IdT const*>(dst.data),
                         wgt ? static_cast<WgtT const*>(wgt->data) : nullptr, n, d_bounds.as<long long>(), n_ranks,
                         n_tiles, pos.as<int64_t>(), static_cast<IdT*>(src_out), static_cast<IdT*>(dst_out),
                         static_cast<WgtT*>(wgt_out));
    };
    if (wgt && wgt->type == FLOAT64)
      scatter(double{});
    else
      scatter(float{});
  };
  if (dst.type == INT64)
    run(int64_t{});
  else
    run(int32_t{});
  hipLaunchKernelGGL(owner_totals_kernel, dim3(1), dim3(kGeMaxRanks), 0, stream, pos.as<int64_t>(), n_tiles, n_ranks, n,
                     totals.as<long long>());
  CGA_HIP(hipGetLastError());
  unsigned int h_bad = 0;
  std::vector<long long> h_tot(n_ranks);
  CGA_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipMemcpyAsync(h_tot.data(), totals.p, totals.bytes, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  CGA_EXPECTS(h_bad == 0, CUGRAPH_INVALID_INPUT, "destination ids outside the bounds of the partition");
  for (int r = 0; r < n_ranks; ++r) counts_out[r] = h_tot[r];
}

// true when the majors of a compressed graph are in non-increasing degree order
bool degrees_descending(hipStream_t stream, void const* offsets, bool off64, int64_t n)
{
  if (n < 2) return true;
  DevBuf up(4);
  CGA_HIP(hipMemsetAsync(up.p, 0, 4, stream));
  if (off64)
    hipLaunchKernelGGL((degree_descending_kernel<int64_t>), dim3(grid_for(n, 2048)), dim3(kThreads), 0,
                       stream, static_cast<int64_t const*>(offsets), n, up.as<unsigned int>());
  else
    hipLaunchKernelGGL((degree_descending_kernel<int32_t>), dim3(grid_for(n, 2048)), dim3(kThreads), 0,
                       stream, static_cast<int32_t const*>(offsets), n, up.as<unsigned int>());
  CGA_HIP(hipGetLastError());
  unsigned int h_up = 1;
  CGA_HIP(hipMemcpyAsync(&h_up, up.p, 4, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  return h_up == 0;
}

// Locality of the low-degree rows: a pull iteration gives the rows of degree 1 .. kLowDegreeMax one
// thread each, and consecutive threads gather from wherever their rows' sources are.  The order
// among rows of EQUAL degree is free (the reference's is whatever its sort leaves), so inside each
// of those classes the rows are put in the order of their smallest source: neighbouring threads
// then read neighbouring slots of the gathered vector (scale 26: 0.31 -> 0.22 ms for those rows).
// Works on the relabelled endpoints (a row's class follows from its new id) and updates them, the
// order and the rank in place; true when anything moved.  deg: degrees by the labels `order` maps to.
// CUGRAPH_AMD_RENUMBER_LOCALITY=0 / 1: never / also on small graphs (tests); by default from 4 M
// edges.  (Ids that go through the hash-set path keep the plain order.)
bool refine_low_degree_order(hipStream_t stream, unsigned int const* deg, int64_t nv, int64_t ne,
                             int32_t* majors, int32_t* minors, DevBuf& order, DevBuf& rank)
{
  char const* loc_env = std::getenv("CUGRAPH_AMD_RENUMBER_LOCALITY");
  bool const locality = loc_env ? loc_env[0] == '1' : ne >= (int64_t{1} << 22);
  if (!locality || nv <= 0 || ne <= 0) return false;
  DevBuf n_ge((kLowDegreeMax + 1) * sizeof(unsigned long long));
  CGA_HIP(hipMemsetAsync(n_ge.p, 0, n_ge.bytes, stream));
  hipLaunchKernelGGL(count_degree_classes_kernel, dim3(grid_for(nv, 1024)), dim3(kThreads), 0, stream, deg, nv,
                     n_ge.as<unsigned long long>());
  unsigned long long h_ge[kLowDegreeMax + 1];  // vertices of degree >= kLowDegreeMax + 1, ..., >= 2, >= 1
  CGA_HIP(hipMemcpyAsync(h_ge, n_ge.p, sizeof(h_ge), hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  int64_t const first = static_cast<int64_t>(h_ge[0]), n_low = static_cast<int64_t>(h_ge[kLowDegreeMax]) - first;
  if (n_low < 2) return false;
  DevBuf minsrc(static_cast<size_t>(n_low) * 4), remap(static_cast<size_t>(n_low) * 4);
  CGA_HIP(hipMemsetAsync(minsrc.p, 0xff, minsrc.bytes, stream));
  hipLaunchKernelGGL(low_min_source_kernel, dim3(grid_for(ne)), dim3(kThreads), 0, stream, majors, minors, ne,
                     first, n_low, minsrc.as<unsigned int>());
  // descending sort of (nv - 1 - smallest source) = ascending smallest source, class by class
  hipLaunchKernelGGL(flip_keys_kernel, dim3(grid_for(n_low)), dim3(kThreads), 0, stream,
                     minsrc.as<unsigned int>(), n_low, static_cast<unsigned int>(nv - 1));
  DevBuf old_order(static_cast<size_t>(n_low) * 4);
  CGA_HIP(hipMemcpyAsync(old_order.p, order.as<int32_t>() + first, static_cast<size_t>(n_low) * 4,
                         hipMemcpyDeviceToDevice, stream));
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n_low)), dim3(kThreads), 0, stream, remap.as<int32_t>(), n_low);
  for (int c = 0; c < kLowDegreeMax; ++c) {  // degree kLowDegreeMax, ..., 2, 1
    int64_t const a = static_cast<int64_t>(h_ge[c]) - first, b = static_cast<int64_t>(h_ge[c + 1]) - first;
    if (b - a < 2) continue;
    DevBuf sub_order, sub_rank;
    degree_order(stream, minsrc.as<unsigned int>() + a, b - a, sub_order, sub_rank);
    hipLaunchKernelGGL(gather_i32_kernel, dim3(grid_for(b - a)), dim3(kThreads), 0, stream,
                       old_order.as<int32_t>() + a, sub_order.as<int32_t>(), b - a,
                       order.as<int32_t>() + first + a);
    hipLaunchKernelGGL(offset_i32_kernel, dim3(grid_for(b - a)), dim3(kThreads), 0, stream,
                       sub_rank.as<int32_t>(), b - a, static_cast<int32_t>(a), remap.as<int32_t>() + a);
    CGA_HIP(hipStreamSynchronize(stream));  // sub_order / sub_rank die here
  }
  hipLaunchKernelGGL(remap_low_ids_kernel, dim3(grid_for(ne)), dim3(kThreads), 0, stream, minors, ne,
                     static_cast<int32_t>(first), static_cast<unsigned int>(n_low), remap.as<int32_t>());
  hipLaunchKernelGGL(remap_low_ids_kernel, dim3(grid_for(ne)), dim3(kThreads), 0, stream, majors, ne,
                     static_cast<int32_t>(first), static_cast<unsigned int>(n_low), remap.as<int32_t>());
  hipLaunchKernelGGL(invert_kernel, dim3(grid_for(nv)), dim3(kThreads), 0, stream, order.as<int32_t>(), nv,
                     rank.as<int32_t>());
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));  // minsrc / remap / old_order die here
  return true;
}

// A degree-sorted copy of a compressed graph whose ids are 0..V-1 (the pull plan of graphs built
// with renumber=FALSE).  number_map (int32) and ext_lookup describe the permutation.
Graph* degree_sorted_copy(Handle const& h, Graph const& g)
{
  hipStream_t stream = h.stream;
  int64_t n = g.num_vertices, ne = g.num_edges;
  CGA_EXPECTS(g.num_global_vertices == g.num_vertices, CUGRAPH_NOT_IMPLEMENTED,
              "degree-sorted copies of partitioned graphs are not supported");
  DevBuf deg(static_cast<size_t>(n > 0 ? n : 1) * 4);
  if (n > 0) {
    if (g.off64)
      hipLaunchKernelGGL((degree_from_offsets_kernel<int64_t>), dim3(grid_for(n)), dim3(kThreads), 0,
                         stream, static_cast<int64_t const*>(g.offsets), n, deg.as<unsigned int>());
    else
      hipLaunchKernelGGL((degree_from_offsets_kernel<int32_t>), dim3(grid_for(n)), dim3(kThreads), 0,
                         stream, static_cast<int32_t const*>(g.offsets), n, deg.as<unsigned int>());
  }
  DevBuf order, rank;
  degree_order_sources_first(stream, deg.as<unsigned int>(), g.indices, ne, n, order, rank);
  DevBuf majors(static_cast<size_t>(ne > 0 ? ne : 1) * 4), minors(static_cast<size_t>(ne > 0 ? ne : 1) * 4);
  expand_offsets(stream, g.offsets, g.off64, n, ne, majors.as<int32_t>());
  if (ne > 0)
    CGA_HIP(hipMemcpyAsync(minors.p, g.indices, static_cast<size_t>(ne) * 4, hipMemcpyDeviceToDevice, stream));
  relabel_ids(stream, majors.as<int32_t>(), ne, rank.as<int32_t>());
  relabel_ids(stream, minors.as<int32_t>(), ne, rank.as<int32_t>());
  if (g.store_transposed)
    refine_low_degree_order(stream, deg.as<unsigned int>(), n, ne, majors.as<int32_t>(), minors.as<int32_t>(), order,
                            rank);
  DevBuf counts(static_cast<size_t>(n > 0 ? n : 1) * 4);
  if (n > 0)
    hipLaunchKernelGGL(gather_u32_kernel, dim3(grid_for(n)), dim3(kThreads), 0, stream, deg.as<unsigned int>(),
                       order.as<int32_t>(), n, counts.as<unsigned int>());
  deg.release();
  CompressedOut out = compress_coo(stream, n, n, ne, majors.as<int32_t>(), minors.as<int32_t>(),
                                   g.weights, g.weights ? dtype_size(g.weight_type) : 0,
                                   counts.as<unsigned int>(), true);
  auto ig                 = std::make_unique<Graph>();
  ig->vertex_type         = INT32;
  ig->edge_type           = g.edge_type;
  ig->weight_type         = g.weight_type;
  ig->store_transposed    = g.store_transposed;
  ig->is_symmetric        = g.is_symmetric;
  ig->is_multigraph       = g.is_multigraph;
  ig->renumbered          = true;
  ig->degree_sorted       = true;
  ig->num_vertices        = n;
  ig->num_global_vertices = n;
  ig->num_edges           = ne;
  ig->off64               = out.off64;
  ig->offsets             = out.offsets.detach();
  ig->indices             = static_cast<int32_t*>(out.indices.detach());
  ig->weights             = out.weights.p ? out.weights.detach() : nullptr;
  ig->number_map          = order.detach();
  ig->ext_lookup          = static_cast<int32_t*>(rank.detach());
  ig->ext_min             = 0;
  ig->ext_range           = n;
  return ig.release();
}

// renumber=TRUE: arbitrary external ids (vertex_type) -> degree-sorted dense ids.
// On return src32/dst32 hold the relabelled endpoints, g.number_map / ext_lookup are set and
// g.num_vertices is the number of distinct vertices.
namespace {
// renumber_edgelist for ids that span 2^31 or more (no direct-address table): the distinct ids are
// collected in a hash set, sorted, and every endpoint finds its dense (ascending-id) label by
// binary search -- the same labels the table path gives, so everything after it is shared.
void renumber_sparse_ids(Handle const& h, Graph& g, ArrayView const* vertices, ArrayView const* src,
                         ArrayView const* dst, bool majors_are_dst, int32_t* src32, int32_t* dst32,
                         long long lo, unsigned long long span, DevBuf* major_counts_out)
{
  hipStream_t stream = h.stream;
  int64_t const ne   = static_cast<int64_t>(src->size);
  bool const wide    = src->type == INT64;
  int64_t const n_keys = vertices ? static_cast<int64_t>(vertices->size) : 2 * ne;
  unsigned long long slots = 1024;
  while (slots < 2ull * static_cast<unsigned long long>(n_keys) && slots < (1ull << 32)) slots <<= 1;
  DevBuf table(slots * 8), saw(4);
  CGA_HIP(hipMemsetAsync(saw.p, 0, 4, stream));
  {
    // every slot = kEmptySlot: only the top byte of each 8 is set
    DevBuf& t = table;
    CGA_HIP(hipMemsetAsync(t.p, 0, t.bytes, stream));
    hipLaunchKernelGGL(fill_u64_kernel, dim3(grid_for(static_cast<int64_t>(slots))), dim3(kThreads), 0, stream,
                       t.as<unsigned long long>(), static_cast<int64_t>(slots), kEmptySlot);
  }
  auto insert = [&](ArrayView const* a) {
    if (!a || a->size == 0) return;
    int64_t n = static_cast<int64_t>(a->size);
    if (wide)
      hipLaunchKernelGGL((hash_insert_kernel<int64_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         static_cast<int64_t const*>(a->data), n, table.as<unsigned long long>(), slots - 1,
                         saw.as<unsigned int>());
    else
      hipLaunchKernelGGL((hash_insert_kernel<int32_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         static_cast<int32_t const*>(a->data), n, table.as<unsigned long long>(), slots - 1,
                         saw.as<unsigned int>());
  };
  if (vertices) {
    insert(vertices);
  } else {
    insert(src);
    insert(dst);
  }
  DevBuf used(slots * 4), upos((slots + 1) * 8);
  hipLaunchKernelGGL(slot_used_kernel, dim3(grid_for(static_cast<int64_t>(slots))), dim3(kThreads), 0, stream,
                     table.as<unsigned long long>(), static_cast<int64_t>(slots), used.as<unsigned int>());
  exclusive_scan_u32_to_i64(stream, used.as<unsigned int>(), static_cast<int64_t>(slots), upos.as<int64_t>());
  int64_t nv             = 0;
  unsigned int h_saw     = 0;
  CGA_HIP(hipMemcpyAsync(&nv, upos.as<int64_t>() + slots, 8, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipMemcpyAsync(&h_saw, saw.p, 4, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  CGA_EXPECTS(h_saw != 2, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "2^31 or more vertices are not supported");
  CGA_EXPECTS(h_saw == 0, CUGRAPH_NOT_IMPLEMENTED, "renumber=TRUE: the smallest 64-bit integer as a vertex id");
  CGA_EXPECTS(nv < 0x7fffffffLL, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "2^31 or more vertices are not supported");
  size_t const nn = static_cast<size_t>(nv > 0 ? nv : 1);
  DevBuf keys_a(nn * 8), keys_b(nn * 8);
  hipLaunchKernelGGL(slot_compact_kernel, dim3(grid_for(static_cast<int64_t>(slots))), dim3(kThreads), 0, stream,
                     table.as<unsigned long long>(), static_cast<int64_t>(slots), upos.as<int64_t>(),
                     static_cast<unsigned long long>(lo), keys_a.as<unsigned long long>());
  table.release();
  used.release();
  upos.release();
  // ---- ascending order: stable one-bit splits over the bits the span has
  {
    DevBuf flags(nn * 4), pos((nn + 1) * 8);
    unsigned long long* kin  = keys_a.as<unsigned long long>();
    unsigned long long* kout = keys_b.as<unsigned long long>();
    int bits = 0;
    while (bits < 64 && (span >> bits) != 0) ++bits;
    for (int b = 0; b < bits && nv > 1; ++b) {
      hipLaunchKernelGGL(bit_flag64_kernel, dim3(grid_for(nv)), dim3(kThreads), 0, stream, kin, nv, b,
                         flags.as<unsigned int>());
      exclusive_scan_u32_to_i64(stream, flags.as<unsigned int>(), nv, pos.as<int64_t>());
      hipLaunchKernelGGL(bit_split64_kernel, dim3(grid_for(nv)), dim3(kThreads), 0, stream, kin, nv, b,
                         pos.as<int64_t>(), kout);
      std::swap(kin, kout);
    }
    if (kin != keys_a.as<unsigned long long>())
      CGA_HIP(hipMemcpyAsync(keys_a.p, kin, nn * 8, hipMemcpyDeviceToDevice, stream));
  }
  DevBuf sorted(nn * 8);
  if (nv > 0)
    hipLaunchKernelGGL(unbias_kernel, dim3(grid_for(nv)), dim3(kThreads), 0, stream,
                       keys_a.as<unsigned long long>(), nv, static_cast<unsigned long long>(lo),
                       sorted.as<long long>());
  keys_a.release();
  keys_b.release();
  // ---- endpoints -> dense (ascending-id) labels
  DevBuf unknown(4);
  CGA_HIP(hipMemsetAsync(unknown.p, 0, 4, stream));
  auto densify = [&](ArrayView const* a, int32_t* out) {
    if (a->size == 0) return;
    int64_t n = static_cast<int64_t>(a->size);
    if (wide)
      hipLaunchKernelGGL((to_dense_sorted_kernel<int64_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         static_cast<int64_t const*>(a->data), n, sorted.as<long long>(), nv, out,
                         unknown.as<unsigned int>());
    else
      hipLaunchKernelGGL((to_dense_sorted_kernel<int32_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         static_cast<int32_t const*>(a->data), n, sorted.as<long long>(), nv, out,
                         unknown.as<unsigned int>());
  };
  densify(src, src32);
  densify(dst, dst32);
  unsigned int h_unknown = 0;
  CGA_HIP(hipMemcpyAsync(&h_unknown, unknown.p, 4, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  CGA_EXPECTS(h_unknown == 0, CUGRAPH_INVALID_INPUT,
              "Invalid input arguments: edge endpoints missing from the vertex list.");
  // ---- major degrees, order, relabel (as in renumber_edgelist)
  DevBuf deg(nn * 4);
  CGA_HIP(hipMemsetAsync(deg.p, 0, deg.bytes, stream));
  if (ne > 0)
    hipLaunchKernelGGL(hist_u32_kernel, dim3(grid_for(ne)), dim3(kThreads), 0, stream,
                       majors_are_dst ? dst32 : src32, ne, deg.as<unsigned int>());
  DevBuf order, rank;
  degree_order_sources_first(stream, deg.as<unsigned int>(), majors_are_dst ? src32 : dst32, ne, nv, order, rank);
  relabel_ids(stream, src32, ne, rank.as<int32_t>());
  relabel_ids(stream, dst32, ne, rank.as<int32_t>());
  if (major_counts_out && nv > 0) {  // the degrees in the new order: the build's histogram, for free
    major_counts_out->alloc(static_cast<size_t>(nv) * 4);
    hipLaunchKernelGGL(gather_u32_kernel, dim3(grid_for(nv)), dim3(kThreads), 0, stream, deg.as<unsigned int>(),
                       order.as<int32_t>(), nv, major_counts_out->as<unsigned int>());
  }
  // ---- number_map (new -> external); external -> new = search of `sorted`, then rank
  DevBuf nmap(nn * dtype_size(g.vertex_type));
  if (nv > 0) {
    if (wide)
      hipLaunchKernelGGL((number_map_sorted_kernel<int64_t>), dim3(grid_for(nv)), dim3(kThreads), 0, stream,
                         sorted.as<long long>(), nv, rank.as<int32_t>(), nmap.as<int64_t>());
    else
      hipLaunchKernelGGL((number_map_sorted_kernel<int32_t>), dim3(grid_for(nv)), dim3(kThreads), 0, stream,
                         sorted.as<long long>(), nv, rank.as<int32_t>(), nmap.as<int32_t>());
  }
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));
  g.num_vertices = g.num_global_vertices = nv;
  g.renumbered    = true;
  g.degree_sorted = true;
  g.number_map    = nmap.detach();
  g.ext_lookup    = static_cast<int32_t*>(rank.detach());
  g.ext_sorted    = static_cast<long long*>(sorted.detach());
  g.ext_min       = 0;
  g.ext_range     = nv;
}
}  // namespace

void renumber_edgelist(Handle const& h, Graph& g, ArrayView const* vertices, ArrayView const* src,
                       ArrayView const* dst, bool majors_are_dst, int32_t* src32, int32_t* dst32,
                       DevBuf* major_counts_out)
{
  hipStream_t stream = h.stream;
  int64_t ne         = static_cast<int64_t>(src->size);
  bool wide          = src->type == INT64;
  // ---- id range
  DevBuf mm(2 * sizeof(long long));
  long long init[2] = {0x7fffffffffffffffLL, -0x7fffffffffffffffLL - 1};
  CGA_HIP(hipMemcpyAsync(mm.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
  auto scan_minmax = [&](ArrayView const* a) {
    if (!a || a->size == 0) return;
    int64_t n = static_cast<int64_t>(a->size);
    if (wide)
      hipLaunchKernelGGL((minmax_kernel<int64_t>), dim3(grid_for(n, 1024)), dim3(kThreads), 0, stream,
                         static_cast<int64_t const*>(a->data), n, mm.as<long long>());
    else
      hipLaunchKernelGGL((minmax_kernel<int32_t>), dim3(grid_for(n, 1024)), dim3(kThreads), 0, stream,
                         static_cast<int32_t const*>(a->data), n, mm.as<long long>());
  };
  scan_minmax(vertices);
  scan_minmax(src);
  scan_minmax(dst);
  long long h_mm[2];
  CGA_HIP(hipMemcpyAsync(h_mm, mm.p, sizeof(h_mm), hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  bool empty = h_mm[1] < h_mm[0];
  long long lo = empty ? 0 : h_mm[0];
  // (unsigned: the span of 64-bit ids can exceed the signed range)
  unsigned long long const span = empty ? 0ull
                                        : static_cast<unsigned long long>(h_mm[1]) -
                                            static_cast<unsigned long long>(h_mm[0]);
  bool const sparse = span >= 0x7ffffffeull || (!empty && std::getenv("CUGRAPH_AMD_RENUMBER_SPARSE") != nullptr);
  if (sparse) {
    renumber_sparse_ids(h, g, vertices, src, dst, majors_are_dst, src32, dst32, lo, span, major_counts_out);
    return;
  }
  long long range = empty ? 0 : static_cast<long long>(span) + 1;
  size_t rr = static_cast<size_t>(range > 0 ? range : 1);
  // ---- one table over [lo, lo + range): edges per major and "occurs as a minor" (top bit) of
  // every external id, counted on the ids as they came in
  DevBuf tally(rr * 4), flags(rr * 4), pos((rr + 1) * 8);
  CGA_HIP(hipMemsetAsync(tally.p, 0, rr * 4, stream));
  ArrayView const* major_ids = majors_are_dst ? dst : src;
  ArrayView const* minor_ids = majors_are_dst ? src : dst;
  if (ne > 0) {
    if (wide) {
      hipLaunchKernelGGL((tally_majors_kernel<int64_t>), dim3(grid_for(ne)), dim3(kThreads), 0, stream,
                         static_cast<int64_t const*>(major_ids->data), ne, lo, tally.as<unsigned int>());
      hipLaunchKernelGGL((tally_minors_kernel<int64_t>), dim3(grid_for(ne)), dim3(kThreads), 0, stream,
                         static_cast<int64_t const*>(minor_ids->data), ne, lo, tally.as<unsigned int>());
    } else {
      hipLaunchKernelGGL((tally_majors_kernel<int32_t>), dim3(grid_for(ne)), dim3(kThreads), 0, stream,
                         static_cast<int32_t const*>(major_ids->data), ne, lo, tally.as<unsigned int>());
      hipLaunchKernelGGL((tally_minors_kernel<int32_t>), dim3(grid_for(ne)), dim3(kThreads), 0, stream,
                         static_cast<int32_t const*>(minor_ids->data), ne, lo, tally.as<unsigned int>());
    }
  }
  // ---- vertex set: the given list, or every id an edge mentions
  DevBuf unknown(4);
  CGA_HIP(hipMemsetAsync(unknown.p, 0, 4, stream));
  if (vertices) {
    CGA_HIP(hipMemsetAsync(flags.p, 0, rr * 4, stream));
    if (vertices->size > 0) {
      int64_t n = static_cast<int64_t>(vertices->size);
      if (wide)
        hipLaunchKernelGGL((mark_ids_kernel<int64_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                           static_cast<int64_t const*>(vertices->data), n, lo, flags.as<unsigned int>());
      else
        hipLaunchKernelGGL((mark_ids_kernel<int32_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                           static_cast<int32_t const*>(vertices->data), n, lo, flags.as<unsigned int>());
    }
    if (range > 0)  // ids the edges mention but the list does not hold
      hipLaunchKernelGGL(count_unlisted_kernel, dim3(grid_for(range, 1024)), dim3(kThreads), 0, stream,
                         tally.as<unsigned int>(), flags.as<unsigned int>(), static_cast<int64_t>(range),
                         unknown.as<unsigned int>());
  } else if (range > 0) {
    hipLaunchKernelGGL(flags_of_tally_kernel, dim3(grid_for(range)), dim3(kThreads), 0, stream,
                       tally.as<unsigned int>(), static_cast<int64_t>(range), flags.as<unsigned int>());
  }
  exclusive_scan_u32_to_i64(stream, flags.as<unsigned int>(), range, pos.as<int64_t>());
  int64_t nv             = 0;
  unsigned int h_unknown = 0;
  CGA_HIP(hipMemcpyAsync(&nv, pos.as<int64_t>() + range, 8, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipMemcpyAsync(&h_unknown, unknown.p, 4, hipMemcpyDeviceToHost, stream));
  CGA_HIP(hipStreamSynchronize(stream));
  CGA_EXPECTS(h_unknown == 0, CUGRAPH_INVALID_INPUT,
              "Invalid input arguments: edge endpoints missing from the vertex list.");
  // ---- degrees and minor marks of the dense (ascending-id) labels, order, rank
  size_t const nn = static_cast<size_t>(nv > 0 ? nv : 1);
  DevBuf deg(nn * 4), is_minor(nn * 4);
  if (range > 0)
    hipLaunchKernelGGL(compact_tally_kernel, dim3(grid_for(range)), dim3(kThreads), 0, stream,
                       tally.as<unsigned int>(), flags.as<unsigned int>(), pos.as<int64_t>(),
                       static_cast<int64_t>(range), deg.as<unsigned int>(), is_minor.as<unsigned int>());
  DevBuf order, rank;
  degree_order_marked(stream, deg.as<unsigned int>(), is_minor.as<unsigned int>(), nv, order, rank);
  tally.release();
  // ---- number_map (new -> external) and the external -> new lookup; endpoints through the lookup
  DevBuf nmap(nn * dtype_size(g.vertex_type));
  DevBuf lookup(rr * 4);
  if (range > 0) {
    if (wide)
      hipLaunchKernelGGL((number_map_kernel<int64_t>), dim3(grid_for(range)), dim3(kThreads), 0, stream,
                         range, lo, flags.as<unsigned int>(), pos.as<int64_t>(), rank.as<int32_t>(),
                         nmap.as<int64_t>(), lookup.as<int32_t>());
    else
      hipLaunchKernelGGL((number_map_kernel<int32_t>), dim3(grid_for(range)), dim3(kThreads), 0, stream,
                         range, lo, flags.as<unsigned int>(), pos.as<int64_t>(), rank.as<int32_t>(),
                         nmap.as<int32_t>(), lookup.as<int32_t>());
  }
  auto relabel = [&](ArrayView const* a, int32_t* out) {
    if (a->size == 0) return;
    int64_t n = static_cast<int64_t>(a->size);
    if (wide)
      hipLaunchKernelGGL((lookup_ids_kernel<int64_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         static_cast<int64_t const*>(a->data), n, lo, lookup.as<int32_t>(), out);
    else
      hipLaunchKernelGGL((lookup_ids_kernel<int32_t>), dim3(grid_for(n)), dim3(kThreads), 0, stream,
                         static_cast<int32_t const*>(a->data), n, lo, lookup.as<int32_t>(), out);
  };
  relabel(src, src32);
  relabel(dst, dst32);
  // ---- locality of the low-degree rows (refine_low_degree_order)
  if (majors_are_dst && refine_low_degree_order(stream, deg.as<unsigned int>(), nv, ne, dst32, src32, order, rank)) {
    if (wide)
      hipLaunchKernelGGL((number_map_kernel<int64_t>), dim3(grid_for(range)), dim3(kThreads), 0, stream,
                         range, lo, flags.as<unsigned int>(), pos.as<int64_t>(), rank.as<int32_t>(),
                         nmap.as<int64_t>(), lookup.as<int32_t>());
    else
      hipLaunchKernelGGL((number_map_kernel<int32_t>), dim3(grid_for(range)), dim3(kThreads), 0, stream,
                         range, lo, flags.as<unsigned int>(), pos.as<int64_t>(), rank.as<int32_t>(),
                         nmap.as<int32_t>(), lookup.as<int32_t>());
    CGA_HIP(hipGetLastError());
  }
  if (major_counts_out && nv > 0) {  // the degrees in the new order: the build's histogram, for free
    major_counts_out->alloc(static_cast<size_t>(nv) * 4);
    hipLaunchKernelGGL(gather_u32_kernel, dim3(grid_for(nv)), dim3(kThreads), 0, stream, deg.as<unsigned int>(),
                       order.as<int32_t>(), nv, major_counts_out->as<unsigned int>());
  }
  CGA_HIP(hipGetLastError());
  CGA_HIP(hipStreamSynchronize(stream));
  g.num_vertices = g.num_global_vertices = nv;
  g.renumbered   = true;
  g.degree_sorted = true;
  g.number_map   = nmap.detach();
  g.ext_lookup   = static_cast<int32_t*>(lookup.detach());
  g.ext_min      = lo;
  g.ext_range    = range;
}

}  // namespace cga
