// Microbenchmark: LDS atomic adds to random words of a 128 KiB array, one 1024-thread workgroup per CU,
// by operand type.  Settles whether fp32 row accumulators (twice the rows per XS tile) can pay on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// SPREAD: 0 = random words of the whole array, 1 = lane l of a wave hits word (base + l): conflict-free
template <typename T, int SPREAD>
__global__ __launch_bounds__(1024) void k(long per_thread, T* out)
{
  extern __shared__ double lds_raw[];
  T* acc = reinterpret_cast<T*>(lds_raw);
  constexpr uint32_t words = (128u << 10) / sizeof(T);
  for (uint32_t i = threadIdx.x; i < words; i += 1024) acc[i] = T(0);
  __syncthreads();
  uint32_t s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u;
  for (long i = 0; i < per_thread; i += 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      uint32_t h = hash32(SPREAD ? (blockIdx.x * 16u + (threadIdx.x >> 6)) * 977u + (uint32_t)(i + e) : s + (uint32_t)(i + e));
      uint32_t w = SPREAD ? ((h & ~63u) + (threadIdx.x & 63u)) & (words - 1) : h & (words - 1);
      atomicAdd(&acc[w], T(1));
    }
  }
  __syncthreads();
  T sum = T(0);
  for (uint32_t i = threadIdx.x; i < words; i += 1024) sum += acc[i];
  if (sum == T(3)) out[0] = sum;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 3;
}
template <typename T, int SPREAD> void run(char const* name, void* out) {
  long per = 4096; int blocks = 256;
  CK(hipFuncSetAttribute(reinterpret_cast<void const*>(k<T, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10));
  float ms = timeit([&] { k<T, SPREAD><<<blocks, 1024, 128 << 10>>>(per, static_cast<T*>(out)); });
  double n = (double)per * blocks * 1024;
  printf("%-28s %8.3f ms  %7.1f G adds/s  %5.1f cycles per wave-instruction per CU at 2.4 GHz\n", name, ms, n / ms / 1e6,
         ms * 1e-3 * 2.4e9 / (per * 16.0));
}
int main() {
  void* out; CK(hipMalloc(&out, 64));
  run<float, 0>("f32 random", out);
  run<double, 0>("f64 random", out);
  run<unsigned int, 0>("u32 random", out);
  run<unsigned long long, 0>("u64 random", out);
  run<float, 1>("f32 lane-consecutive", out);
  run<double, 1>("f64 lane-consecutive", out);
  run<unsigned int, 1>("u32 lane-consecutive", out);
  run<unsigned long long, 1>("u64 lane-consecutive", out);
  return 0;
}
