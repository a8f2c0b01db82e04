// Microbenchmark: random 4-byte gathers per second from tables of different sizes (L1 / L2 /
// Infinity Cache / HBM resident), plus LDS gathers.  Calibrates the SpMV design.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o tools/gather_bench && tools/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline uint32_t hash32(uint32_t x)
{
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// idx stream read coalesced as int4, 4 gathers per load, U loads in flight
template <int U>
__global__ __launch_bounds__(256) void gather_kernel(const int4* __restrict__ idx, long n4,
                                                     const float* __restrict__ table, float* out)
{
  float acc = 0.f;
  long stride = (long)gridDim.x * blockDim.x;
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    int4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = idx[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += table[v[u].x] + table[v[u].y] + table[v[u].z] + table[v[u].w];
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int U>
__global__ __launch_bounds__(1024) void lds_gather_kernel(const int4* __restrict__ idx, long n4,
                                                          const float* __restrict__ table, int tsize, float* out)
{
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < tsize; i += blockDim.x) lds[i] = table[i];
  __syncthreads();
  float acc = 0.f;
  long stride = (long)gridDim.x * blockDim.x;
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    int4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = idx[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += lds[v[u].x] + lds[v[u].y] + lds[v[u].z] + lds[v[u].w];
  }
  if (acc == 123.456f) out[0] = acc;
}

__global__ void fill_idx(int* idx, long n, uint32_t mask, int sorted_runs)
{
  long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t h = hash32((uint32_t)i * 2654435761u + 12345u);
    idx[i] = (int)(h & mask);
  }
}

int main()
{
  const long n = 1L << 28;  // 268M gathers, 1 GiB of indices
  int* idx; float* table; float* out;
  CK(hipMalloc(&idx, n * 4));
  CK(hipMalloc(&table, 1L << 31));
  CK(hipMalloc(&out, 4));
  CK(hipMemset(table, 0, 1L << 31));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-28s %10s %12s %12s\n", "table", "ms", "Ggather/s", "idx GB/s");
  for (int bits : {12, 14, 16, 18, 19, 20, 22, 24, 26, 28}) {
    uint32_t mask = (1u << bits) - 1;
    fill_idx<<<4096, 256>>>(idx, n, mask, 0);
    CK(hipDeviceSynchronize());
    for (int blocks : {2048, 8192}) {
      gather_kernel<2><<<blocks, 256>>>((const int4*)idx, n / 4, table, out);
      CK(hipEventRecord(e0));
      for (int r = 0; r < 3; ++r) gather_kernel<2><<<blocks, 256>>>((const int4*)idx, n / 4, table, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      char name[64]; snprintf(name, 64, "global %8.0f KiB, %d blk", (1u << bits) * 4 / 1024.0, blocks);
      printf("%-28s %10.3f %12.1f %12.1f\n", name, ms, n / ms / 1e6, n * 4.0 / ms / 1e6);
    }
  }
  for (int bits : {12, 14, 15}) {
    uint32_t mask = (1u << bits) - 1;
    fill_idx<<<4096, 256>>>(idx, n, mask, 0);
    CK(hipDeviceSynchronize());
    int tsize = 1 << bits;
    CK(hipFuncSetAttribute((const void*)lds_gather_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_gather_kernel<2><<<256, 1024, tsize * 4>>>((const int4*)idx, n / 4, table, tsize, out);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) lds_gather_kernel<2><<<256, 1024, tsize * 4>>>((const int4*)idx, n / 4, table, tsize, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    char name[64]; snprintf(name, 64, "LDS %8.0f KiB", tsize * 4 / 1024.0);
    printf("%-28s %10.3f %12.1f %12.1f\n", name, ms, n / ms / 1e6, n * 4.0 / ms / 1e6);
  }
  return 0;
}
