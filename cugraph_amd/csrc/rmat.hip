// R-MAT edge generator for the benchmark inputs.
//
// Quadrant rule per bit level as in the reference generator
// (cpp/src/generators/generate_rmat_edgelist.cu:85-102):
//     src_bit = r0 > a+b ;  dst_bit = r1 > (src_bit ? c/(1-a-b) : a/(a+b))
// The random stream is this project's own: Philox4x32-10 with key = seed and counter =
// (edge index, level pair, stream id), two bit levels per Philox call, uniform 24-bit integers
// compared against integer thresholds -- so the CPU oracle (oracle/pagerank_oracle.c) emits the
// identical edge list bit for bit.
#include "common.hpp"

namespace cga {

namespace {

__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1, uint32_t (&out)[4])
{
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
    uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = static_cast<uint32_t>(p1);
    uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = static_cast<uint32_t>(p0);
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ inline uint32_t scramble_id(uint32_t v, int scale)
{
  uint32_t mask = scale >= 32 ? 0xFFFFFFFFu : ((1u << scale) - 1u);
  v             = (v * 0x9E3779B1u + 0x7F4A7C15u) & mask;
  uint32_t r    = __brev(v) >> (32 - scale);
  return (r * 0x85EBCA6Bu + 0xC2B2AE35u) & mask;
}

__global__ void rmat_kernel(int scale, uint64_t first, uint64_t count, uint32_t k0, uint32_t k1,
                            uint32_t t_ab, uint32_t t_a, uint32_t t_c, int scramble,
                            int32_t* __restrict__ src, int32_t* __restrict__ dst)
{
  uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < count;
       i += stride) {
    uint64_t e = first + i;
    uint32_t s = 0, d = 0;
    int bit = scale - 1;
    for (uint32_t j = 0; bit >= 0; ++j) {
      uint32_t o[4];
      philox4x32_10(static_cast<uint32_t>(e), static_cast<uint32_t>(e >> 32), j, 0u, k0, k1, o);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (bit >= 0) {
          uint32_t u0 = o[2 * h] >> 8, u1 = o[2 * h + 1] >> 8;
          uint32_t sb = u0 > t_ab ? 1u : 0u;
          uint32_t db = u1 > (sb ? t_c : t_a) ? 1u : 0u;
          s |= sb << bit;
          d |= db << bit;
          --bit;
        }
      }
    }
    if (scramble) {
      s = scramble_id(s, scale);
      d = scramble_id(d, scale);
    }
    src[i] = static_cast<int32_t>(s);
    dst[i] = static_cast<int32_t>(d);
  }
}

template <typename WT>
__global__ void rmat_weights_kernel(uint64_t first, uint64_t count, uint32_t k0, uint32_t k1,
                                    WT* __restrict__ w)
{
  uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < count;
       i += stride) {
    uint64_t e = first + i;
    uint32_t o[4];
    philox4x32_10(static_cast<uint32_t>(e), static_cast<uint32_t>(e >> 32), 0u, 1u, k0, k1, o);
    uint64_t m = (static_cast<uint64_t>(o[0]) << 21) | static_cast<uint64_t>(o[1] >> 11);
    w[i]       = static_cast<WT>(static_cast<double>(m + 1) * 0x1.0p-53);
  }
}

}  // namespace

void rmat_generate(hipStream_t stream, int scale, uint64_t first, uint64_t count, uint64_t seed,
                   uint32_t t_ab, uint32_t t_a, uint32_t t_c, int scramble, int32_t* src,
                   int32_t* dst)
{
  if (count == 0) return;
  uint64_t blocks = (count + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(rmat_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, scale,
                     first, count, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32),
                     t_ab, t_a, t_c, scramble, src, dst);
  CGA_HIP(hipGetLastError());
}

void rmat_weights(hipStream_t stream, uint64_t first, uint64_t count, uint64_t seed, void* w,
                  bool f64)
{
  if (count == 0) return;
  uint64_t blocks = (count + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (f64)
    hipLaunchKernelGGL((rmat_weights_kernel<double>), dim3(static_cast<unsigned>(blocks)),
                       dim3(256), 0, stream, first, count, static_cast<uint32_t>(seed),
                       static_cast<uint32_t>(seed >> 32), static_cast<double*>(w));
  else
    hipLaunchKernelGGL((rmat_weights_kernel<float>), dim3(static_cast<unsigned>(blocks)), dim3(256),
                       0, stream, first, count, static_cast<uint32_t>(seed),
                       static_cast<uint32_t>(seed >> 32), static_cast<float*>(w));
  CGA_HIP(hipGetLastError());
}

}  // namespace cga
