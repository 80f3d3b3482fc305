// Exclusive prefix sum over int32 (row counts -> row pointers, flags -> ranks).
// Two launches for up to 8.4 M elements: tile scans + tile totals, then every block adds the sum of the totals in front of it
// (which it forms itself: at most 4096 values).  Until round 4 the totals were scanned by a recursive call -- six launches per
// scan with the two one-word blits of its single-tile case, 47 scans per 200k-point chunk.  Longer inputs recurse as before.
#include "ai_common.h"

#define SCAN_ITEMS 8
#define SCAN_TILE (AI_BLOCK * SCAN_ITEMS)  // 2048 elements per block

// Block-wide exclusive scan of one tile; writes tile total to sums[blockIdx.x].
__global__ __launch_bounds__(AI_BLOCK) void k_scan_tiles(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                         int64_t n, int32_t* __restrict__ sums) {
  __shared__ int32_t wsum[AI_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int32_t tsum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    tsum += v[i];
  }
  // inclusive scan of tsum across the wave
  int32_t inc = tsum;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int32_t woff = 0, total = 0;
#pragma unroll
  for (int i = 0; i < AI_BLOCK / 64; ++i) {
    if (i < w) woff += wsum[i];
    total += wsum[i];
  }
  int32_t run = woff + inc - tsum;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(AI_BLOCK) void k_scan_add(int32_t* __restrict__ out, int64_t n,
                                                       const int32_t* __restrict__ offs) {
  const int32_t off = offs[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) out[base + i] += off;
  // the element one past the end carries the grand total
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = offs[gridDim.x];
}

// every block adds the sum of the tile totals in front of its tile; the element one past the end gets the grand total
#define SCAN_MAX_DIRECT_TILES 4096
__global__ __launch_bounds__(AI_BLOCK) void k_scan_add_direct(int32_t* __restrict__ out, int64_t n, const int32_t* __restrict__ sums,
                                                              int32_t* __restrict__ total_out) {
  __shared__ int32_t wsum[AI_BLOCK / 64];
  int32_t acc = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += AI_BLOCK) acc += sums[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  int32_t off = 0;
#pragma unroll
  for (int i = 0; i < AI_BLOCK / 64; ++i) off += wsum[i];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  if (off != 0) {
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
      if (base + i < n) out[base + i] += off;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    out[n] = off + sums[blockIdx.x];
    if (total_out) __hip_atomic_store(total_out, off + sums[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (pinned host memory)
  }
}

static int64_t tiles_of(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

size_t ai_scan_tmp_elems(int64_t n) {
  // sums arrays for each recursion level, each with one extra slot for its total
  size_t tot = 0;
  int64_t m = n;
  do {
    m = tiles_of(m < 1 ? 1 : m);
    tot += (size_t)m + 1;
  } while (m > 1);
  return tot + 8;
}

int ai_exclusive_scan_i32(hipStream_t stream, const int32_t* in, int32_t* out, int64_t n, int32_t* tmp, int32_t* total_out) {
  if (n <= 0) {
    AI_HIP(hipMemsetAsync(out, 0, sizeof(int32_t), stream));
    if (total_out) AI_HIP(hipMemcpyAsync(total_out, out, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    return AI_OK;
  }
  const int64_t nt = tiles_of(n);
  int32_t* sums = tmp;  // nt + 1 entries
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)nt), dim3(AI_BLOCK), 0, stream, in, out, n, sums);
  AI_KERNEL_CHECK();
  if (nt <= SCAN_MAX_DIRECT_TILES) {
    hipLaunchKernelGGL(k_scan_add_direct, dim3((unsigned)nt), dim3(AI_BLOCK), 0, stream, out, n, (const int32_t*)sums, total_out);
    AI_KERNEL_CHECK();
    return AI_OK;
  }
  // sums[0..nt] <- exclusive scan of tile totals (sums[nt] = grand total)
  if (nt == 1) {
    // single tile: the total is sums[0] and the only offset is zero
    AI_HIP(hipMemcpyAsync(sums + 1, sums, sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    AI_HIP(hipMemsetAsync(sums, 0, sizeof(int32_t), stream));
  } else {
    AI_TRY(ai_exclusive_scan_i32(stream, sums, sums, nt, tmp + nt + 1, nullptr));
  }
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nt), dim3(AI_BLOCK), 0, stream, out, n, sums);
  AI_KERNEL_CHECK();
  if (total_out) AI_HIP(hipMemcpyAsync(total_out, out + n, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  return AI_OK;
}
