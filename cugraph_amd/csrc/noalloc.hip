// cugraph_amd_edgelist_source_weight_sums_noalloc: the per-source weight sums of a raw edge list
// (cugraph_amd_edgelist_source_weight_sums, capi.hip / pagerank.hip: edgelist_source_sums) for
// callers that cannot allocate device memory -- code under HIP graph capture, tight serving loops,
// harnesses that forbid hipMalloc.  No counterpart in the reference.
//
// The existing entry point stages the source ids, narrowed to int32 and range-checked, in a buffer
// it allocates per call and reads the id range back to the host.  Here the staging buffer (and a
// small header of counters) is a file-scope __device__ array of a fixed size, so a call allocates
// nothing, copies nothing to the host and never synchronises: ids outside [0, size of sums) cannot
// be reported as an error any more, they are skipped and counted on the device instead.
//
// Every GPU holds its own copy of the array at its own address until the process ends; all calls
// on one GPU share it, so calls that overlap on one GPU (two streams, two threads) race.
#include "common.hpp"

#include <mutex>

namespace cga {

namespace {

constexpr size_t kNoAllocMaxEdges = size_t(1) << 22;  // edges per call: 16 MiB of staged ids
constexpr size_t kHeaderBytes     = 256;              // {ids skipped}, padded to the staging area's alignment
constexpr int kMaxDevices         = 64;
constexpr int kThreads            = 256;

// In host code this name is only a stand-in for the symbol: the memory is reached through the
// per-device address hipGetSymbolAddress returns (scratch_of), never through the name.
static __device__ __align__(256) unsigned char scratch[kHeaderBytes + kNoAllocMaxEdges * sizeof(int32_t)];

std::mutex g_mutex;
void* g_scratch_ptr[kMaxDevices] = {};

inline int grid_for(int64_t items)
{
  int64_t b = (items + kThreads - 1) / kThreads;
  return static_cast<int>(b < 1 ? 1 : b > 4096 ? 4096 : b);
}

// ids -> int32 in the staging area, -1 for an id outside [0, num_vertices); those are counted
template <typename IdT>
__global__ __launch_bounds__(kThreads) void stage_ids_kernel(IdT const* __restrict__ ids, int64_t n,
                                                             int64_t num_vertices,
                                                             int32_t* __restrict__ staged,
                                                             unsigned long long* __restrict__ skipped)
{
  unsigned long long bad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t const v = static_cast<int64_t>(ids[i]);
    bool const ok   = v >= 0 && v < num_vertices;
    staged[i]       = ok ? static_cast<int32_t>(v) : -1;
    bad += ok ? 0u : 1u;
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(skipped, bad);
}

template <typename WT>
__global__ __launch_bounds__(kThreads) void sums_from_staged_kernel(int32_t const* __restrict__ staged,
                                                                    WT const* __restrict__ weights,
                                                                    int64_t n, double* __restrict__ sums)
{
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int32_t const s = staged[i];
    if (s >= 0) atomicAdd(&sums[s], weights ? static_cast<double>(weights[i]) : 1.0);
  }
}

// header and sums to zero.  A kernel, not hipMemsetAsync: memset nodes of a captured HIP graph were seen
// to write stale argument bytes (the address of a later kernel's argument) into the header from the second
// replay on (ROCm 7.0 runtime, tests/test_zz_no_alloc.py: the capture test), kernel nodes replay exactly.
__global__ __launch_bounds__(kThreads) void clear_kernel(unsigned long long* __restrict__ header,
                                                         double* __restrict__ sums, int64_t n)
{
  if (blockIdx.x == 0 && threadIdx.x < kHeaderBytes / sizeof(unsigned long long)) header[threadIdx.x] = 0ull;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    sums[i] = 0.0;
}

__global__ void publish_skipped_kernel(unsigned long long const* __restrict__ skipped, long long* __restrict__ out)
{
  out[0] = static_cast<long long>(skipped[0]);
}

}  // namespace

// Once per device, with that device current (called by the library's init call, handle creation,
// and again by the entry point should a handle predate it): the address of THIS device's copy.
void noalloc_setup_device(int device)
{
  CGA_EXPECTS(device >= 0 && device < kMaxDevices, CUGRAPH_INVALID_INPUT,
              "no-alloc scratch: device index out of range");
  std::lock_guard<std::mutex> lock(g_mutex);
  if (g_scratch_ptr[device]) return;
  int current = -1;
  CGA_HIP(hipGetDevice(&current));
  CGA_EXPECTS(current == device, CUGRAPH_INVALID_INPUT,
              "no-alloc scratch: the handle's device must be the current device");
  void* ptr = nullptr;
  CGA_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  CGA_EXPECTS(ptr != nullptr, CUGRAPH_UNKNOWN_ERROR, "no-alloc scratch: no device address for the scratch array");
  g_scratch_ptr[device] = ptr;
}

void edgelist_source_sums_noalloc(Handle const& h, ArrayView const& src, ArrayView const* weights,
                                  int64_t num_vertices, double* sums, long long* num_skipped)
{
  size_t const ne = src.size;
  CGA_EXPECTS(ne <= kNoAllocMaxEdges, CUGRAPH_INVALID_INPUT,
              "cugraph_amd_edgelist_source_weight_sums_noalloc: at most 4194304 edges per call (the static "
              "staging area); split the edge list or use cugraph_amd_edgelist_source_weight_sums");
  CGA_EXPECTS(num_vertices <= 0x7fffffffLL, CUGRAPH_INVALID_INPUT, "sums: at most 2^31 - 1 vertices");
  noalloc_setup_device(h.device);
  auto* base     = static_cast<unsigned char*>(g_scratch_ptr[h.device]);
  auto* skipped  = reinterpret_cast<unsigned long long*>(base);
  auto* staged   = reinterpret_cast<int32_t*>(base + kHeaderBytes);
  hipStream_t st = h.stream;
  // the array keeps the last call's contents: the counter must start at zero (the staged ids are
  // all rewritten before they are read)
  hipLaunchKernelGGL(clear_kernel, dim3(grid_for(num_vertices)), dim3(kThreads), 0, st, skipped, sums,
                     num_vertices > 0 ? num_vertices : int64_t{0});
  if (ne > 0) {
    int const grid = grid_for(static_cast<int64_t>(ne));
    if (src.type == INT32)
      hipLaunchKernelGGL((stage_ids_kernel<int32_t>), dim3(grid), dim3(kThreads), 0, st,
                         static_cast<int32_t const*>(src.data), static_cast<int64_t>(ne), num_vertices, staged, skipped);
    else
      hipLaunchKernelGGL((stage_ids_kernel<int64_t>), dim3(grid), dim3(kThreads), 0, st,
                         static_cast<int64_t const*>(src.data), static_cast<int64_t>(ne), num_vertices, staged, skipped);
    if (weights && weights->type == FLOAT64)
      hipLaunchKernelGGL((sums_from_staged_kernel<double>), dim3(grid), dim3(kThreads), 0, st, staged,
                         static_cast<double const*>(weights->data), static_cast<int64_t>(ne), sums);
    else
      hipLaunchKernelGGL((sums_from_staged_kernel<float>), dim3(grid), dim3(kThreads), 0, st, staged,
                         weights ? static_cast<float const*>(weights->data) : nullptr, static_cast<int64_t>(ne), sums);
  }
  if (num_skipped) hipLaunchKernelGGL(publish_skipped_kernel, dim3(1), dim3(1), 0, st, skipped, num_skipped);
  CGA_HIP(hipGetLastError());
}

}  // namespace cga
