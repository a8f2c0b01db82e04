// Microbenchmark: do two random gathers that fall into the two 64-B halves of one 128-B line cost one
// fabric request or two?  (table far larger than the L2s; indices hashed in registers)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// MODE 0: one gather; 1: + same 64-B half (offset 4 floats); 2: + other half of the 128-B line (xor 16 floats);
// 3: + the neighbouring 128-B line (xor 32 floats); 4: + an unrelated random line
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ table, uint32_t tmask, long per_thread, float* out)
{
  float acc = 0.f;
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  for (long i = 0; i < per_thread; i += 4) {
    uint32_t a[4], b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = hash32(s + (uint32_t)(i + e)) & tmask;
      b[e] = MODE == 1 ? (a[e] ^ 4u) : MODE == 2 ? (a[e] ^ 16u) : MODE == 3 ? (a[e] ^ 32u) : (hash32(a[e]) & tmask);
    }
    float x[4], y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[e] = table[a[e]]; y[e] = MODE ? table[b[e]] : 0.f; }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += x[e] + y[e];
  }
  if (acc == 123.456f) out[0] = acc;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 3;
}
int main() {
  float* table; float* out;
  CK(hipMalloc(&table, 1L << 30)); CK(hipMalloc(&out, 4)); CK(hipMemset(table, 0, 1L << 30));
  uint32_t tmask = (1u << 28) - 1;  // 1 GiB of floats
  long per = 64; int blocks = 4096;  // 67 M primary gathers
  double n = (double)per * blocks * 256;
  printf("one gather                       %8.3f ms\n", timeit([&] { k<0><<<blocks, 256>>>(table, tmask, per, out); }));
  printf("+ same 64-B half                 %8.3f ms\n", timeit([&] { k<1><<<blocks, 256>>>(table, tmask, per, out); }));
  printf("+ other half of the 128-B line   %8.3f ms\n", timeit([&] { k<2><<<blocks, 256>>>(table, tmask, per, out); }));
  printf("+ neighbouring 128-B line        %8.3f ms\n", timeit([&] { k<3><<<blocks, 256>>>(table, tmask, per, out); }));
  printf("+ unrelated line                 %8.3f ms   (%.0f M primary gathers)\n", timeit([&] { k<4><<<blocks, 256>>>(table, tmask, per, out); }), n / 1e6);
  return 0;
}
