// Does a block-uniform plain load of a record that another stream's kernel rewrote (ring of buffers, event-ordered) ever return
// the record of RING versions ago to SOME waves of a block?  ("A race the tests did not see": fk_check read its
// pool entry that way.)   hipcc --offload-arch=gfx950 -O3 tools/micro/scalar_cache_ring.hip -o /tmp/scr && /tmp/scr [versions] [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
struct __attribute__((aligned(16))) Rec { int v[12]; };   // 48 bytes, like PoolSeg
constexpr int RING = 4;

__global__ void k_write(Rec* __restrict__ t, int nb, int version) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  Rec r;
  for (int k = 0; k < 12; ++k) r.v[k] = version;
  t[b] = r;
}
// mode 0: plain block-uniform load (the compiler makes it scalar loads, one set per wave); mode 1: agent-scope loads
template <int MODE>
__global__ __launch_bounds__(256) void k_check(const Rec* __restrict__ t, int version, int spin, unsigned long long* __restrict__ err) {
  Rec r;
  if (MODE == 0) {
    r = t[blockIdx.x];
  } else {
    const int* p = reinterpret_cast<const int*>(&t[blockIdx.x]);
    for (int k = 0; k < 12; ++k) r.v[k] = __hip_atomic_load(const_cast<int*>(p + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  int bad = 0, older = 0;
  for (int k = 0; k < 12; ++k) {
    bad |= (r.v[k] != version);
    older |= (r.v[k] == version - RING);
  }
  if ((threadIdx.x & 63) == 0 && bad) {
    atomicAdd(&err[0], 1ull);                 // waves that saw a wrong record
    if (older) atomicAdd(&err[1], 1ull);      // ... of exactly RING versions ago
  }
  // stay resident for a while, as the convergence check does (blocks of one launch start at different times)
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}
__global__ void k_stream(const double* __restrict__ a, double* __restrict__ b, size_t n) {  // background traffic, as the steps make
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] + 1.0;
}

int main(int argc, char** argv) {
  const int V = argc > 1 ? atoi(argv[1]) : 100000, NB = argc > 2 ? atoi(argv[2]) : 64;
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  Rec* t[RING];
  for (int r = 0; r < RING; ++r) { CK(hipMalloc(&t[r], sizeof(Rec) * NB)); CK(hipMemset(t[r], 0, sizeof(Rec) * NB)); }
  unsigned long long* err; CK(hipMalloc(&err, 16)); CK(hipMemset(err, 0, 16));
  double *a, *b; const size_t n = 1 << 22; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMemset(a, 0, n * 8));
  std::vector<hipEvent_t> ew(RING), ec(RING);
  for (int r = 0; r < RING; ++r) { CK(hipEventCreateWithFlags(&ew[r], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ec[r], hipEventDisableTiming)); }
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(err, 0, 16));
    for (int v = 1; v <= V; ++v) {
      const int r = v % RING;
      if (v > RING) CK(hipEventSynchronize(ec[r]));           // the check that read this buffer last has finished
      hipLaunchKernelGGL(k_write, dim3((NB + 63) / 64), dim3(64), 0, sa, t[r], NB, v);
      CK(hipEventRecord(ew[r], sa));
      if (v % 3 == 0) hipLaunchKernelGGL(k_stream, dim3(512), dim3(256), 0, sa, (const double*)a, b, n);   // traffic beside the checks
      CK(hipStreamWaitEvent(sb, ew[r], 0));
      if (mode == 0) hipLaunchKernelGGL((k_check<0>), dim3(NB), dim3(256), 0, sb, (const Rec*)t[r], v, (v % 7 == 0) ? 3000 : 200, err);
      else hipLaunchKernelGGL((k_check<1>), dim3(NB), dim3(256), 0, sb, (const Rec*)t[r], v, (v % 7 == 0) ? 3000 : 200, err);
      CK(hipEventRecord(ec[r], sb));
    }
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, err, 16, hipMemcpyDeviceToHost));
    printf("%s: %d versions x %d blocks x 4 waves: %llu waves saw a wrong record (%llu of them the record of %d versions ago)\n",
           mode == 0 ? "plain block-uniform load (scalar cache)" : "agent-scope loads", V, NB, h[0], h[1], RING);
  }
  return 0;
}
