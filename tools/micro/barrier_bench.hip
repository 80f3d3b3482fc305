// Micro-benchmark (GPU box): cost of a device-wide barrier inside one kernel vs the launch boundary between two
// dependent kernels -- the number that decides whether a persistent Lanczos kernel can beat two launches per step.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/barrier_bench.hip -o gpurun_out/barrier_bench && gpurun_out/barrier_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned nblk, unsigned& phase) {
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = 1;
    __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope by default in HIP? use builtin below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned target = (phase + 1) * nblk;
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1 << 22)) { ok = 0; break; }   // never hang: give up and report
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  ++phase;
  __syncthreads();
  return ok != 0;
}

__global__ __launch_bounds__(256) void k_persist(unsigned* ctr, int iters, double* data, int n, int* fail) {
  unsigned phase = 0;
  const int nblk = gridDim.x;
  for (int it = 0; it < iters; ++it) {
    // a little work: each block touches its slice (so that the fences have something to order)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += nblk * 256) data[i] += 1.0;
    if (!grid_barrier(ctr, nblk, phase)) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
  }
}
__global__ __launch_bounds__(256) void k_step(double* data, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) data[i] += 1.0;
}

int main() {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
  printf("device %s, %d CUs\n", pr.name, pr.multiProcessorCount);
  unsigned* ctr; int* fail; double* data;
  const int n = 1 << 20;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&data, n * 8));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int work : {0, 1 << 16, 1 << 20}) {
    for (int G : {64, 256, 512, 1024, 2048}) {
      int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_persist, 256, 0));
      if (G > occ * pr.multiProcessorCount) { printf("G=%d exceeds co-residency (%d per CU)\n", G, occ); continue; }
      CK(hipMemsetAsync(ctr, 0, 4, st)); CK(hipMemsetAsync(fail, 0, 4, st)); CK(hipMemsetAsync(data, 0, n * 8, st));
      hipLaunchKernelGGL(k_persist, dim3(G), dim3(256), 0, st, ctr, 10, data, work, fail);  // warm
      CK(hipMemsetAsync(ctr, 0, 4, st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(k_persist, dim3(G), dim3(256), 0, st, ctr, iters, data, work, fail);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      int hf = 0; CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
      printf("persistent: work %7d doubles, G=%4d blocks: %.2f us per barrier-iteration (fail=%d)\n", work, G, 1e3 * ms / iters, hf);
    }
    for (int G : {256, 1024}) {
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, st, data, work);
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_step, dim3(G), dim3(256), 0, st, data, work);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("launches:   work %7d doubles, G=%4d blocks: %.2f us per dependent launch\n", work, G, 1e3 * ms / iters);
    }
  }
  return 0;
}
