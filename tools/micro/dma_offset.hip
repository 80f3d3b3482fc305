// Where does the immediate offset of an LDS-DMA load go?  One wave copies 1 KB with `global_load_lds_dwordx4 ... offset:1024`
// (M0 = 0) and dumps its LDS: the source bytes 1024..2047 land at LDS 1024.. (offset added to both addresses) or at LDS 0..
// (global address only).   hipcc --offload-arch=gfx950 -O2 tools/micro/dma_offset.hip -o /tmp/dma_offset && /tmp/dma_offset
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const int* src, int* out) {
  __shared__ int lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -1;
  __syncthreads();
  const int* p = src + 4 * threadIdx.x;
  unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\ts_waitcnt vmcnt(0)\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(p), "s"(base) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
  int *src, *out, h[2048];
  hipMalloc(&src, 4096 * 4); hipMalloc(&out, 2048 * 4);
  for (int i = 0; i < 2048; ++i) h[i] = i;
  int hs[4096]; for (int i = 0; i < 4096; ++i) hs[i] = i;
  hipMemcpy(src, hs, sizeof(hs), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("lds[0]=%d lds[255]=%d lds[256]=%d lds[511]=%d lds[512]=%d\n", h[0], h[255], h[256], h[511], h[512]);
  printf(h[256] == 256 ? "offset applies to BOTH the global and the LDS address\n" : (h[0] == 256 ? "offset applies to the GLOBAL address only\n" : "unexpected\n"));
  return 0;
}
