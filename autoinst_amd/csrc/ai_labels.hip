// Label bookkeeping after the NCuts hot path ("next" row 4): the per-point
// parts of the scorer and of the chunk merge, as sorts / histograms on the device.  The O(#instances)
// arithmetic on top of them (IoU tables, greedy matching, AP integration) stays on the host.
//
//   ai_label_pairs      -- the contingency table every score is made of: sorted distinct (a_i, b_i)
//                          pairs with their counts.  pipeline/metrics/metrics_class.py:302-309
//                          (filter_labels: np.unique + np.where per label), :60-114 / :181-235
//                          (np.intersect1d / np.union1d of index lists per (pred, gt) pair),
//                          pipeline/metrics/modified_LSTQ.py:34-60 (np.unique of pred + gt * 2^32).
//   ai_merge_associate  -- pipeline/utils/point_cloud/point_cloud_utils.py:397-463: crop of the merged map
//                          to a cube around the new chunk, per-instance bounding boxes, number of chunk
//                          points of instance id2 inside the box of instance id1, and the reference's
//                          "union" = number of distinct scalar coordinate values of both instances
//                          (np.unique of the concatenated (K, 3) arrays WITHOUT axis flattens them).
//   ai_unique_points    -- open3d's PointCloud.remove_duplicated_points() at :489: keep the first
//                          point of every exact coordinate triple, order preserved.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "ai_common.h"

namespace {

template <typename T>
int to_device(const T* src, size_t count, int mem_kind, DevBuf<T>& own, const T** dev, hipStream_t st) {
  if (mem_kind == AI_MEM_DEVICE) {
    *dev = src;
    return AI_OK;
  }
  AI_TRY(own.alloc(count));
  AI_HIP(hipMemcpyAsync(own.p, src, count * sizeof(T), hipMemcpyHostToDevice, st));
  *dev = own.p;
  return AI_OK;
}

inline unsigned grid_for(int64_t n) { return (unsigned)((n + AI_BLOCK - 1) / AI_BLOCK); }

// float64 -> uint64 with the same order (and -0.0 == +0.0, as np.unique compares values)
__device__ __forceinline__ uint64_t ordered_bits(double v) {
  if (v == 0.0) v = 0.0;
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_ordered_bits(uint64_t k) {
  const uint64_t b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// ------------------------------------------------------------------ label pairs
__global__ __launch_bounds__(AI_BLOCK) void kl_pair_keys(const int32_t* __restrict__ a, const int32_t* __restrict__ b, int64_t n,
                                                         uint64_t* __restrict__ key) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  key[i] = ((uint64_t)((uint32_t)a[i] ^ 0x80000000u) << 32) | (uint64_t)((uint32_t)b[i] ^ 0x80000000u);
}

template <typename K>
__global__ __launch_bounds__(AI_BLOCK) void kl_heads(const K* __restrict__ key, int64_t n, int32_t* __restrict__ head) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}

// one thread per sorted element: a run's head writes the run's key and start
__global__ __launch_bounds__(AI_BLOCK) void kl_pair_emit(const uint64_t* __restrict__ key, const int32_t* __restrict__ pos, int64_t n,
                                                         int32_t* __restrict__ out_a, int32_t* __restrict__ out_b,
                                                         int32_t* __restrict__ start) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  if (pos[i + 1] == pos[i]) return;
  const int64_t r = pos[i];
  const uint64_t k = key[i];
  out_a[r] = (int32_t)((uint32_t)(k >> 32) ^ 0x80000000u);
  out_b[r] = (int32_t)((uint32_t)k ^ 0x80000000u);
  start[r] = (int32_t)i;
}

// ------------------------------------------------------------------ merge association
struct Cube {
  double lo[3], hi[3];
};

// crop flags (:405-417, open3d's crop keeps min <= p <= max) for map points that belong to an instance
__global__ __launch_bounds__(AI_BLOCK) void km_crop(const double* __restrict__ xyz, const int32_t* __restrict__ inst, int64_t n, Cube c,
                                                    int32_t n_inst, int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  const double x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
  const int32_t id = inst[i];
  flag[i] = (id > 0 && id < n_inst && x >= c.lo[0] && x <= c.hi[0] && y >= c.lo[1] && y <= c.hi[1] && z >= c.lo[2] && z <= c.hi[2]) ? 1 : 0;
}

__global__ __launch_bounds__(AI_BLOCK) void km_inst_flag(const int32_t* __restrict__ inst, int64_t n, int32_t n_inst,
                                                         int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t id = inst[i];
  flag[i] = (id > 0 && id < n_inst) ? 1 : 0;
}

// selected points -> three (value, tag) entries each; side 0 additionally feeds its instance's box
// (min / max of order-preserving integers: exact and order-independent) and point count
__global__ __launch_bounds__(AI_BLOCK) void km_scalars(const double* __restrict__ xyz, const int32_t* __restrict__ inst,
                                                       const int32_t* __restrict__ pos, int64_t n, int64_t base, uint32_t side,
                                                       uint64_t* __restrict__ val, uint32_t* __restrict__ tag,
                                                       unsigned long long* __restrict__ box, int32_t* __restrict__ npts) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  if (pos[i + 1] == pos[i]) return;
  const int64_t o = (base + pos[i]) * 3;
  const uint32_t id = (uint32_t)inst[i];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const uint64_t k = ordered_bits(xyz[i * 3 + a]);
    val[o + a] = k;
    tag[o + a] = (side << 31) | id;
    if (side == 0) {
      atomicMin(&box[(size_t)id * 6 + a], (unsigned long long)k);
      atomicMax(&box[(size_t)id * 6 + 3 + a], (unsigned long long)k);
    }
  }
  atomicAdd(&npts[id], 1);
}

__global__ __launch_bounds__(AI_BLOCK) void km_box_init(unsigned long long* __restrict__ box, int32_t n_inst) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n_inst * 6) return;
  box[i] = (i % 6 < 3) ? ~0ull : 0ull;
}

__global__ __launch_bounds__(AI_BLOCK) void km_val_tag_heads(const uint64_t* __restrict__ val, const uint32_t* __restrict__ tag, int64_t n,
                                                             int32_t* __restrict__ head) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  head[i] = (i == 0 || val[i] != val[i - 1] || tag[i] != tag[i - 1]) ? 1 : 0;
}

// distinct (value, instance) entries, compacted; every entry counts one distinct scalar of its instance
__global__ __launch_bounds__(AI_BLOCK) void km_distinct(const uint64_t* __restrict__ val, const uint32_t* __restrict__ tag,
                                                        const int32_t* __restrict__ pos, int64_t n, uint64_t* __restrict__ dval,
                                                        uint32_t* __restrict__ dtag, int32_t* __restrict__ nscal1,
                                                        int32_t* __restrict__ nscal2) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  if (pos[i + 1] == pos[i]) return;
  const uint32_t t = tag[i];
  dval[pos[i]] = val[i];
  dtag[pos[i]] = t;
  atomicAdd((t >> 31) ? &nscal2[t & 0x7fffffffu] : &nscal1[t], 1);
}

// a value's entries are sorted side 0 first; every side-0 entry walks to the value's side-1 entries
__global__ __launch_bounds__(AI_BLOCK) void km_common(const uint64_t* __restrict__ dval, const uint32_t* __restrict__ dtag, int64_t nd,
                                                      int32_t n2, int32_t* __restrict__ common) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= nd) return;
  const uint32_t t = dtag[i];
  if (t >> 31) return;
  const uint64_t v = dval[i];
  for (int64_t j = i + 1; j < nd && dval[j] == v; ++j) {
    const uint32_t u = dtag[j];
    if (u >> 31) atomicAdd(&common[(size_t)t * n2 + (u & 0x7fffffffu)], 1);
  }
}

// :451-456  intersection = #points of instance id2 with min_bound <= p <= max_bound of instance id1
#define KM_BOX_TILE 256
__global__ __launch_bounds__(AI_BLOCK) void km_inside(const double* __restrict__ xyz, const int32_t* __restrict__ inst, int64_t n,
                                                      int32_t n1, int32_t n2, const unsigned long long* __restrict__ box,
                                                      int32_t* __restrict__ inter) {
  __shared__ double sb[KM_BOX_TILE][6];
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  const int32_t id2 = (i < n) ? inst[i] : 0;
  const bool live = id2 > 0 && id2 < n2;
  double x = 0.0, y = 0.0, z = 0.0;
  if (live) {
    x = xyz[i * 3];
    y = xyz[i * 3 + 1];
    z = xyz[i * 3 + 2];
  }
  for (int32_t b0 = 1; b0 < n1; b0 += KM_BOX_TILE) {
    const int32_t nb = min(KM_BOX_TILE, n1 - b0);
    __syncthreads();
    for (int q = threadIdx.x; q < nb * 6; q += AI_BLOCK) {
      const unsigned long long k = box[(size_t)b0 * 6 + q];
      // an instance without cropped points keeps (max, min) = (+inf-ish, -inf-ish) reversed: never inside
      sb[q / 6][q % 6] = from_ordered_bits(k);
    }
    __syncthreads();
    if (!live) continue;
    for (int32_t b = 0; b < nb; ++b)
      if (x >= sb[b][0] && x <= sb[b][3] && y >= sb[b][1] && y <= sb[b][4] && z >= sb[b][2] && z <= sb[b][5])
        atomicAdd(&inter[(size_t)(b0 + b) * n2 + id2], 1);
  }
}

// ------------------------------------------------------------------ unique points
// open3d keys an unordered_map on the Eigen vector: equality by value and std::hash<double>, which maps
// -0.0 and +0.0 to the same bucket; every other pair of distinct bit patterns is a distinct point
__device__ __forceinline__ uint64_t value_bits(double v) {
  if (v == 0.0) v = 0.0;
  return (uint64_t)__double_as_longlong(v);
}

__global__ __launch_bounds__(AI_BLOCK) void ku_axis_keys(const double* __restrict__ xyz, const int32_t* __restrict__ order, int64_t n,
                                                         int axis, uint64_t* __restrict__ key) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int64_t p = order ? order[i] : i;
  key[i] = value_bits(xyz[p * 3 + axis]);
}

__global__ __launch_bounds__(AI_BLOCK) void ku_iota(int32_t* __restrict__ a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i < n) a[i] = (int32_t)i;
}

__global__ __launch_bounds__(AI_BLOCK) void ku_first_flags(const double* __restrict__ xyz, const int32_t* __restrict__ order, int64_t n,
                                                           int32_t* __restrict__ keep) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int64_t p = order[i];
  bool first = i == 0;
  if (!first) {
    const int64_t q = order[i - 1];
#pragma unroll
    for (int a = 0; a < 3; ++a) first |= value_bits(xyz[p * 3 + a]) != value_bits(xyz[q * 3 + a]);
  }
  keep[p] = first ? 1 : 0;  // stable sorts: the first of a run is the smallest original index
}

__global__ __launch_bounds__(AI_BLOCK) void ku_compact(const int32_t* __restrict__ pos, int64_t n, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  if (pos[i + 1] != pos[i]) out[pos[i]] = (int32_t)i;
}

template <typename K, typename V>
int sort_pairs(hipStream_t st, K* kin, K* kout, V* vin, V* vout, int64_t n, int bits) {
  size_t tmp_bytes = 0;
  AI_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0, bits, st));
  DevBuf<uint8_t> tmp;
  AI_TRY(tmp.alloc(tmp_bytes));
  AI_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0, bits, st));
  return AI_OK;
}

int scan_flags(hipStream_t st, int32_t* flag_to_pos, int64_t n, DevBuf<int32_t>& tmp) {
  AI_TRY(tmp.alloc(ai_scan_tmp_elems(n)));
  return ai_exclusive_scan_i32(st, flag_to_pos, flag_to_pos, n, tmp.p);
}

}  // namespace

extern "C" int ai_label_pairs(ai_ctx* ctx, const int32_t* a, const int32_t* b, int64_t n, int mem_kind, int64_t cap, int32_t* pair_a,
                              int32_t* pair_b, int64_t* pair_count, int64_t* n_pairs) {
  if (!ctx || !a || !b || !n_pairs || n < 0 || n >= ((int64_t)1 << 31) - 1 || cap < 0 || (cap > 0 && (!pair_a || !pair_b || !pair_count))) {
    ai_set_error("ai_label_pairs: bad argument");
    return AI_ERR_BAD_ARG;
  }
  *n_pairs = 0;
  if (n == 0) return AI_OK;
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);
  hipStream_t st = ctx->stream;
  DevBuf<int32_t> own_a, own_b, pos, scan_tmp, oa, ob, ostart;
  DevBuf<uint64_t> key, skey;
  DevBuf<uint8_t> tmp;
  const int32_t *da, *db;
  AI_TRY(to_device(a, (size_t)n, mem_kind, own_a, &da, st));
  AI_TRY(to_device(b, (size_t)n, mem_kind, own_b, &db, st));
  AI_TRY(key.alloc(n));
  AI_TRY(skey.alloc(n));
  AI_TRY(pos.alloc(n + 1));
  hipLaunchKernelGGL(kl_pair_keys, dim3(grid_for(n)), dim3(AI_BLOCK), 0, st, da, db, n, key.p);
  AI_KERNEL_CHECK();
  size_t tmp_bytes = 0;
  AI_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, key.p, skey.p, (size_t)n, 0, 64, st));
  AI_TRY(tmp.alloc(tmp_bytes));
  AI_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, key.p, skey.p, (size_t)n, 0, 64, st));
  hipLaunchKernelGGL(kl_heads<uint64_t>, dim3(grid_for(n)), dim3(AI_BLOCK), 0, st, (const uint64_t*)skey.p, n, pos.p);
  AI_KERNEL_CHECK();
  AI_TRY(scan_flags(st, pos.p, n, scan_tmp));
  int32_t total = 0;
  AI_HIP(hipMemcpyAsync(&total, pos.p + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  *n_pairs = total;
  const int64_t m = std::min<int64_t>(total, cap);
  if (m == 0) return AI_OK;
  AI_TRY(oa.alloc(total));
  AI_TRY(ob.alloc(total));
  AI_TRY(ostart.alloc(total));
  hipLaunchKernelGGL(kl_pair_emit, dim3(grid_for(n)), dim3(AI_BLOCK), 0, st, (const uint64_t*)skey.p, (const int32_t*)pos.p, n, oa.p, ob.p,
                     ostart.p);
  AI_KERNEL_CHECK();
  // run r covers sorted positions [start[r], start[r + 1]); the last run ends at n
  std::vector<int32_t> hs((size_t)m + 1);
  const int64_t ncopy = std::min<int64_t>(m + 1, total);
  AI_HIP(hipMemcpyAsync(pair_a, oa.p, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(pair_b, ob.p, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(hs.data(), ostart.p, (size_t)ncopy * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  if (ncopy == m) hs[m] = (int32_t)n;
  for (int64_t r = 0; r < m; ++r) pair_count[r] = (int64_t)hs[r + 1] - hs[r];
  return AI_OK;
}

extern "C" int ai_merge_associate(ai_ctx* ctx, const double* map_xyz, const int32_t* map_inst, int64_t n_map, const double* chunk_xyz,
                                  const int32_t* chunk_inst, int64_t n_chunk, const double* center, double side_length, int32_t n_inst1,
                                  int32_t n_inst2, int mem_kind, int32_t* inter, int32_t* common, int32_t* n_scalars1,
                                  int32_t* n_scalars2, int32_t* n_points1) {
  if (!ctx || !map_xyz || !map_inst || !chunk_xyz || !chunk_inst || !center || !inter || !common || !n_scalars1 || !n_scalars2 ||
      !n_points1 || n_map <= 0 || n_chunk <= 0 || n_inst1 < 1 || n_inst2 < 1 || !(side_length > 0.0) || n_map >= ((int64_t)1 << 29) ||
      n_chunk >= ((int64_t)1 << 29) || (int64_t)n_inst1 * n_inst2 >= ((int64_t)1 << 28)) {
    ai_set_error("ai_merge_associate: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);
  hipStream_t st = ctx->stream;
  DevBuf<double> own_m, own_c;
  DevBuf<int32_t> own_mi, own_ci, pos1, pos2, scan_tmp, d_inter, d_common, d_ns1, d_ns2, d_np1, d_np2, head;
  DevBuf<unsigned long long> box;
  const double *dm, *dc;
  const int32_t *dmi, *dci;
  AI_TRY(to_device(map_xyz, (size_t)n_map * 3, mem_kind, own_m, &dm, st));
  AI_TRY(to_device(map_inst, (size_t)n_map, mem_kind, own_mi, &dmi, st));
  AI_TRY(to_device(chunk_xyz, (size_t)n_chunk * 3, mem_kind, own_c, &dc, st));
  AI_TRY(to_device(chunk_inst, (size_t)n_chunk, mem_kind, own_ci, &dci, st));
  Cube cube;
  for (int a = 0; a < 3; ++a) {  // :405-413  min_bound = center - side / 2, max_bound = center + side / 2
    const double half = side_length / 2.0;
    cube.lo[a] = center[a] - half;
    cube.hi[a] = center[a] + half;
  }
  const size_t n12 = (size_t)n_inst1 * n_inst2;
  AI_TRY(pos1.alloc(n_map + 1));
  AI_TRY(pos2.alloc(n_chunk + 1));
  AI_TRY(d_inter.alloc(n12));
  AI_TRY(d_common.alloc(n12));
  AI_TRY(d_ns1.alloc(n_inst1));
  AI_TRY(d_ns2.alloc(n_inst2));
  AI_TRY(d_np1.alloc(n_inst1));
  AI_TRY(d_np2.alloc(n_inst2));
  AI_TRY(box.alloc((size_t)n_inst1 * 6));
  AI_HIP(hipMemsetAsync(d_inter.p, 0, n12 * sizeof(int32_t), st));
  AI_HIP(hipMemsetAsync(d_common.p, 0, n12 * sizeof(int32_t), st));
  AI_HIP(hipMemsetAsync(d_ns1.p, 0, (size_t)n_inst1 * sizeof(int32_t), st));
  AI_HIP(hipMemsetAsync(d_ns2.p, 0, (size_t)n_inst2 * sizeof(int32_t), st));
  AI_HIP(hipMemsetAsync(d_np1.p, 0, (size_t)n_inst1 * sizeof(int32_t), st));
  AI_HIP(hipMemsetAsync(d_np2.p, 0, (size_t)n_inst2 * sizeof(int32_t), st));
  hipLaunchKernelGGL(km_box_init, dim3(grid_for((int64_t)n_inst1 * 6)), dim3(AI_BLOCK), 0, st, box.p, n_inst1);
  hipLaunchKernelGGL(km_crop, dim3(grid_for(n_map)), dim3(AI_BLOCK), 0, st, dm, dmi, n_map, cube, n_inst1, pos1.p);
  hipLaunchKernelGGL(km_inst_flag, dim3(grid_for(n_chunk)), dim3(AI_BLOCK), 0, st, dci, n_chunk, n_inst2, pos2.p);
  AI_KERNEL_CHECK();
  AI_TRY(scan_flags(st, pos1.p, n_map, scan_tmp));
  DevBuf<int32_t> scan_tmp2;
  AI_TRY(scan_flags(st, pos2.p, n_chunk, scan_tmp2));
  int32_t sel[2] = {0, 0};
  AI_HIP(hipMemcpyAsync(&sel[0], pos1.p + n_map, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(&sel[1], pos2.p + n_chunk, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  const int64_t ns = 3 * ((int64_t)sel[0] + sel[1]);
  if (ns > 0) {
    DevBuf<uint64_t> val, val2, dval;
    DevBuf<uint32_t> tag, tag2, dtag;
    AI_TRY(val.alloc(ns));
    AI_TRY(val2.alloc(ns));
    AI_TRY(tag.alloc(ns));
    AI_TRY(tag2.alloc(ns));
    hipLaunchKernelGGL(km_scalars, dim3(grid_for(n_map)), dim3(AI_BLOCK), 0, st, dm, dmi, (const int32_t*)pos1.p, n_map, (int64_t)0, 0u, val.p,
                       tag.p, box.p, d_np1.p);
    hipLaunchKernelGGL(km_scalars, dim3(grid_for(n_chunk)), dim3(AI_BLOCK), 0, st, dc, dci, (const int32_t*)pos2.p, n_chunk, (int64_t)sel[0], 1u,
                       val.p, tag.p, box.p, d_np2.p);
    AI_KERNEL_CHECK();
    // order by (value, side, instance): stable sort by the tag, then by the value
    AI_TRY(sort_pairs(st, tag.p, tag2.p, val.p, val2.p, ns, 32));
    AI_TRY(sort_pairs(st, val2.p, val.p, tag2.p, tag.p, ns, 64));
    AI_TRY(head.alloc(ns + 1));
    hipLaunchKernelGGL(km_val_tag_heads, dim3(grid_for(ns)), dim3(AI_BLOCK), 0, st, (const uint64_t*)val.p, (const uint32_t*)tag.p, ns, head.p);
    AI_KERNEL_CHECK();
    DevBuf<int32_t> scan_tmp3;
    AI_TRY(scan_flags(st, head.p, ns, scan_tmp3));
    int32_t nd = 0;
    AI_HIP(hipMemcpyAsync(&nd, head.p + ns, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    AI_HIP(hipStreamSynchronize(st));
    AI_TRY(dval.alloc(nd));
    AI_TRY(dtag.alloc(nd));
    hipLaunchKernelGGL(km_distinct, dim3(grid_for(ns)), dim3(AI_BLOCK), 0, st, (const uint64_t*)val.p, (const uint32_t*)tag.p,
                       (const int32_t*)head.p, ns, dval.p, dtag.p, d_ns1.p, d_ns2.p);
    hipLaunchKernelGGL(km_common, dim3(grid_for(nd)), dim3(AI_BLOCK), 0, st, (const uint64_t*)dval.p, (const uint32_t*)dtag.p, (int64_t)nd,
                       n_inst2, d_common.p);
    hipLaunchKernelGGL(km_inside, dim3(grid_for(n_chunk)), dim3(AI_BLOCK), 0, st, dc, dci, n_chunk, n_inst1, n_inst2,
                       (const unsigned long long*)box.p, d_inter.p);
    AI_KERNEL_CHECK();
  }
  AI_HIP(hipMemcpyAsync(inter, d_inter.p, n12 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(common, d_common.p, n12 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(n_scalars1, d_ns1.p, (size_t)n_inst1 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(n_scalars2, d_ns2.p, (size_t)n_inst2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(n_points1, d_np1.p, (size_t)n_inst1 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  return AI_OK;
}

extern "C" int ai_unique_points(ai_ctx* ctx, const double* xyz, int64_t n, int mem_kind, int32_t* keep_index, int64_t* n_keep) {
  if (!ctx || !xyz || !keep_index || !n_keep || n < 0 || n >= ((int64_t)1 << 30)) {
    ai_set_error("ai_unique_points: bad argument");
    return AI_ERR_BAD_ARG;
  }
  *n_keep = 0;
  if (n == 0) return AI_OK;
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);
  hipStream_t st = ctx->stream;
  DevBuf<double> own;
  DevBuf<uint64_t> key, skey;
  DevBuf<int32_t> oa, ob, keep, scan_tmp, out;
  const double* dx;
  AI_TRY(to_device(xyz, (size_t)n * 3, mem_kind, own, &dx, st));
  AI_TRY(key.alloc(n));
  AI_TRY(skey.alloc(n));
  AI_TRY(oa.alloc(n));
  AI_TRY(ob.alloc(n));
  AI_TRY(keep.alloc(n + 1));
  hipLaunchKernelGGL(ku_iota, dim3(grid_for(n)), dim3(AI_BLOCK), 0, st, oa.p, n);
  AI_KERNEL_CHECK();
  int32_t *cur = oa.p, *nxt = ob.p;
  for (int axis = 2; axis >= 0; --axis) {  // least-significant key first; each pass is stable
    hipLaunchKernelGGL(ku_axis_keys, dim3(grid_for(n)), dim3(AI_BLOCK), 0, st, dx, (const int32_t*)cur, n, axis, key.p);
    AI_KERNEL_CHECK();
    AI_TRY(sort_pairs(st, key.p, skey.p, cur, nxt, n, 64));
    std::swap(cur, nxt);
  }
  hipLaunchKernelGGL(ku_first_flags, dim3(grid_for(n)), dim3(AI_BLOCK), 0, st, dx, (const int32_t*)cur, n, keep.p);
  AI_KERNEL_CHECK();
  AI_TRY(scan_flags(st, keep.p, n, scan_tmp));
  int32_t total = 0;
  AI_HIP(hipMemcpyAsync(&total, keep.p + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  AI_TRY(out.alloc(total));
  hipLaunchKernelGGL(ku_compact, dim3(grid_for(n)), dim3(AI_BLOCK), 0, st, (const int32_t*)keep.p, n, out.p);
  AI_KERNEL_CHECK();
  if (mem_kind == AI_MEM_DEVICE)
    AI_HIP(hipMemcpyAsync(keep_index, out.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  else
    AI_HIP(hipMemcpyAsync(keep_index, out.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  *n_keep = total;
  return AI_OK;
}
