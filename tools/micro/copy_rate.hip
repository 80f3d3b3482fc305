// Plain device-to-device copy rate with 16-byte accesses, a few shapes (which one ai_bench_copy uses is decided here).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/copy_rate.hip -o /tmp/copy_rate && /tmp/copy_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
// contiguous chunk per block (each block streams its own range)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_chunk(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n16) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
  size_t i = b0 + threadIdx.x;
  for (; i + (U - 1) * 256 < b1; i += U * 256) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * 256]) : src[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * 256]); else dst[i + u * 256] = v[u];
    }
  }
  for (; i < b1; i += 256) dst[i] = src[i];
}
template <typename F>
double run(F launch, size_t bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipEventRecord(e0, 0);
  for (int r = 0; r < 10; ++r) launch();
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return 10 * 2.0 * bytes / (ms * 1e-3) / 1e9;
}
int main() {
  const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
  v4f *s, *d;
  hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes);
  for (int g : {2048, 4096, 8192, 16384, 65536}) {
    printf("grid %6d  stride U4 %.0f  U8 %.0f  U4nt %.0f | chunk U4 %.0f  U8 %.0f  U8nt %.0f GB/s\n", g,
           run([&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
           run([&] { hipLaunchKernelGGL((k_copy<8, false>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
           run([&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
           run([&] { hipLaunchKernelGGL((k_copy_chunk<4, false>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
           run([&] { hipLaunchKernelGGL((k_copy_chunk<8, false>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes),
           run([&] { hipLaunchKernelGGL((k_copy_chunk<8, true>), dim3(g), dim3(256), 0, 0, s, d, n16); }, bytes));
  }
  hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
  printf("hipMemcpyAsync d2d %.0f GB/s\n", run([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, bytes));
  return 0;
}
