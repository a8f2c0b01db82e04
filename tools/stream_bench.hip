// Microbenchmark: read bandwidth of a 4 GiB stream under the access shapes xs_kernel could use.
//   hipcc -O3 --offload-arch=gfx950 tools/stream_bench.hip -o tools/stream_bench && tools/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v4i_t __attribute__((ext_vector_type(4)));

// each workgroup owns a contiguous chunk; pass = THREADS*EPL dwords, entry e of lane tid = pb + e*THREADS + tid
template <int THREADS, int EPL, bool NT>
__global__ __launch_bounds__(THREADS) void chunk4(const uint32_t* __restrict__ p, long n, uint32_t* out)
{
  long per = n / gridDim.x;
  const uint32_t* b = p + per * blockIdx.x;
  uint32_t acc = 0;
  for (long pb = 0; pb + THREADS * EPL <= per; pb += THREADS * EPL) {
    uint32_t v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e)
      v[e] = NT ? __builtin_nontemporal_load(b + pb + e * THREADS + threadIdx.x) : b[pb + e * THREADS + threadIdx.x];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc ^= v[e];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// same but 16 bytes per lane per load
template <int THREADS, int EPL, bool NT>
__global__ __launch_bounds__(THREADS) void chunk16(const v4i_t* __restrict__ p, long n4, uint32_t* out)
{
  long per = n4 / gridDim.x;
  const v4i_t* b = p + per * blockIdx.x;
  uint32_t acc = 0;
  for (long pb = 0; pb + THREADS * EPL <= per; pb += THREADS * EPL) {
    v4i_t v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e)
      v[e] = NT ? __builtin_nontemporal_load(b + pb + e * THREADS + threadIdx.x) : b[pb + e * THREADS + threadIdx.x];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc ^= v[e].x ^ v[e].y ^ v[e].z ^ v[e].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// grid-stride interleaved 16 B per lane
template <int EPL, bool NT>
__global__ __launch_bounds__(256) void stride16(const v4i_t* __restrict__ p, long n4, uint32_t* out)
{
  uint32_t acc = 0;
  long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i + (EPL - 1) * stride < n4; i += EPL * stride) {
    v4i_t v[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) v[e] = NT ? __builtin_nontemporal_load(p + i + e * stride) : p[i + e * stride];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc ^= v[e].x ^ v[e].y ^ v[e].z ^ v[e].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
void timeit(const char* name, long bytes, F f)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("%-44s %8.3f ms %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
}

int main()
{
  const long n = 1L << 30;  // dwords: 4 GiB
  uint32_t* p; uint32_t* out;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemset(p, 1, n * 4));
  long bytes = n * 4;
  timeit("chunk4  256thr epl8  1024 blk", bytes, [&] { chunk4<256, 8, false><<<1024, 256>>>(p, n, out); });
  timeit("chunk4  256thr epl8  1024 blk NT", bytes, [&] { chunk4<256, 8, true><<<1024, 256>>>(p, n, out); });
  timeit("chunk4  256thr epl16 1024 blk NT", bytes, [&] { chunk4<256, 16, true><<<1024, 256>>>(p, n, out); });
  timeit("chunk4  256thr epl8  2048 blk NT", bytes, [&] { chunk4<256, 8, true><<<2048, 256>>>(p, n, out); });
  timeit("chunk4  512thr epl8  1024 blk NT", bytes, [&] { chunk4<512, 8, true><<<1024, 512>>>(p, n, out); });
  timeit("chunk4  1024thr epl8 512 blk NT", bytes, [&] { chunk4<1024, 8, true><<<512, 1024>>>(p, n, out); });
  timeit("chunk16 256thr epl2  1024 blk NT", bytes, [&] { chunk16<256, 2, true><<<1024, 256>>>((v4i_t*)p, n / 4, out); });
  timeit("chunk16 256thr epl4  1024 blk NT", bytes, [&] { chunk16<256, 4, true><<<1024, 256>>>((v4i_t*)p, n / 4, out); });
  timeit("chunk16 256thr epl4  1024 blk", bytes, [&] { chunk16<256, 4, false><<<1024, 256>>>((v4i_t*)p, n / 4, out); });
  timeit("chunk16 256thr epl8  1024 blk NT", bytes, [&] { chunk16<256, 8, true><<<1024, 256>>>((v4i_t*)p, n / 4, out); });
  timeit("chunk16 256thr epl4  2048 blk NT", bytes, [&] { chunk16<256, 4, true><<<2048, 256>>>((v4i_t*)p, n / 4, out); });
  timeit("stride16 epl2 2048 blk", bytes, [&] { stride16<2, false><<<2048, 256>>>((v4i_t*)p, n / 4, out); });
  timeit("stride16 epl4 2048 blk", bytes, [&] { stride16<4, false><<<2048, 256>>>((v4i_t*)p, n / 4, out); });
  timeit("stride16 epl4 2048 blk NT", bytes, [&] { stride16<4, true><<<2048, 256>>>((v4i_t*)p, n / 4, out); });
  timeit("stride16 epl4 8192 blk", bytes, [&] { stride16<4, false><<<8192, 256>>>((v4i_t*)p, n / 4, out); });
  timeit("stride16 epl8 1024 blk", bytes, [&] { stride16<8, false><<<1024, 256>>>((v4i_t*)p, n / 4, out); });
  return 0;
}
