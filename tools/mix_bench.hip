// Microbenchmark: a 4 GiB entry stream (nt, chunked like xs_kernel) with a fraction of the entries also
// doing a random 4-byte gather into a table far larger than the L2s.  Is time = stream + misses, or the max?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void fill(uint32_t* p, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = hash32((uint32_t)i * 2654435761u + 99u);
}
// SEL: an entry gathers when (v & 255) < SEL   (SEL/256 of the entries)
template <int THREADS, int EPL, bool STREAM>
__global__ __launch_bounds__(THREADS) void mix(const uint32_t* __restrict__ p, long n, const float* __restrict__ table,
                                               uint32_t tmask, uint32_t sel, float* out)
{
  long per = n / gridDim.x;
  const uint32_t* b = p + per * blockIdx.x;
  float acc = 0.f;
  for (long pb = 0; pb + THREADS * EPL <= per; pb += THREADS * EPL) {
    uint32_t v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      if (STREAM) v[e] = __builtin_nontemporal_load(b + pb + e * THREADS + threadIdx.x);
      else v[e] = hash32((uint32_t)(pb + e * THREADS + threadIdx.x) + blockIdx.x * 7919u);
    }
    float x[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) x[e] = (v[e] & 255u) < sel ? table[(v[e] >> 8) & tmask] : 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc += x[e];
  }
  if (acc == 123.456f) out[0] = acc;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 3;
}
int main() {
  const long n = 1L << 30;
  uint32_t* p; float* table; float* out;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&table, 1L << 28)); CK(hipMalloc(&out, 4));
  CK(hipMemset(table, 0, 1L << 28));
  fill<<<4096, 256>>>(p, n); CK(hipDeviceSynchronize());
  uint32_t tmask = (1u << 24) - 1;  // 16 M floats = 64 MiB ... the hash has 24 bits after >> 8
  printf("%-10s %12s %12s %12s\n", "gather %", "stream+g ms", "g only ms", "Mgathers");
  for (uint32_t sel : {0u, 4u, 8u, 18u, 32u, 64u, 128u, 256u}) {
    float a = timeit([&] { mix<256, 8, true><<<1024, 256>>>(p, n, table, tmask, sel, out); });
    float b = timeit([&] { mix<256, 8, false><<<1024, 256>>>(p, n, table, tmask, sel, out); });
    printf("%-10.2f %12.3f %12.3f %12.1f\n", sel / 2.56, a, b, n * (sel / 256.0) / 1e6);
  }
  return 0;
}
