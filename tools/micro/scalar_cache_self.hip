// Same stream, same kernel launched again and again: every block reads its slot's value with a block-uniform plain load (scalar
// loads, one per wave) and, at its end, lane 0 stores the next value with a vector store -- the way fk_check kept `nextm[slot]`.
// Do all four waves of a block always see the value the previous launch stored?  Variants: an event record between the launches
// (as the driver has), a second stream running traffic beside it.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/scalar_cache_self.hip -o /tmp/scs && /tmp/scs [launches] [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_self(int* __restrict__ state, int version, int spin, unsigned long long* __restrict__ err) {
  int v;
  if (MODE == 0) v = state[blockIdx.x];
  else v = __hip_atomic_load(&state[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((threadIdx.x & 63) == 0 && v != version) {
    atomicAdd(&err[0], 1ull);
    if (v == version - 1) atomicAdd(&err[1], 1ull);
  }
  long long t0 = wall_clock64();
  const int mine = spin + 37 * (blockIdx.x & 7);
  while (wall_clock64() - t0 < mine) {}
  __syncthreads();
  if (threadIdx.x == 0) state[blockIdx.x] = version + 1;
}
__global__ void k_stream(const double* __restrict__ a, double* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] + 1.0;
}
int main(int argc, char** argv) {
  const int V = argc > 1 ? atoi(argv[1]) : 200000, NB = argc > 2 ? atoi(argv[2]) : 64;
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  int* state; CK(hipMalloc(&state, NB * 4));
  unsigned long long* err; CK(hipMalloc(&err, 16));
  double *a, *b; const size_t n = 1 << 22; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMemset(a, 0, n * 8));
  hipEvent_t ev[16];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int variant = 0; variant < 6; ++variant) {
    const int mode = variant & 1, with_event = (variant >> 1) == 1 || (variant >> 1) == 2, with_traffic = (variant >> 1) == 2;
    CK(hipMemset(state, 0, NB * 4));
    CK(hipMemset(err, 0, 16));
    CK(hipDeviceSynchronize());
    for (int v = 0; v < V; ++v) {
      if (mode == 0) hipLaunchKernelGGL((k_self<0>), dim3(NB), dim3(256), 0, s, state, v, (v % 5 == 0) ? 2000 : 100, err);
      else hipLaunchKernelGGL((k_self<1>), dim3(NB), dim3(256), 0, s, state, v, (v % 5 == 0) ? 2000 : 100, err);
      if (with_event) CK(hipEventRecord(ev[v & 15], s));
      if (with_traffic && (v & 3) == 0) hipLaunchKernelGGL(k_stream, dim3(512), dim3(256), 0, s2, (const double*)a, b, n);
      if ((v & 1023) == 1023) CK(hipStreamSynchronize(s));
    }
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, err, 16, hipMemcpyDeviceToHost));
    printf("%-18s %-22s %-22s: %d launches x %d blocks x 4 waves: %llu waves read another value (%llu: the value of one launch ago)\n",
           mode == 0 ? "plain (scalar) load" : "agent-scope load", with_event ? "event between launches" : "launches back to back",
           with_traffic ? "traffic on 2nd stream" : "alone", V, NB, h[0], h[1]);
  }
  return 0;
}
