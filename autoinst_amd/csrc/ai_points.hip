// The two point-cloud steps on either side of the NCuts hot path ("next" rows 1-2),
// on the same uniform-cell-list machinery as the affinity build:
//
//   ai_radius_mean_pool  -- pipeline/utils/point_cloud/chunk_generation.py:243-256: for every major-voxel
//                           point, the mean of the TARL features of all scan points within
//                           MAJOR_VOXEL_SIZE / 2 (zero row if none)  -> the (N, 96) float64 matrix that
//                           ncuts_utils.py:136-142 feeds to the TARL factor;
//   ai_nn1_project       -- pipeline/utils/point_cloud/point_cloud_utils.py:144-174
//                           (kDTree_1NN_feature_reprojection): every fine point takes the label of its
//                           nearest major-voxel point (optionally only within max_radius).
//
// The reference does both with a Python loop over points around an open3d KD-tree query.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "ai_common.h"

namespace {

struct PGrid {
  double minx, miny, minz, inv_cell;
  int nx, ny, nz;
};

__device__ __forceinline__ void pcell_of(const PGrid& g, double x, double y, double z, int& cx, int& cy, int& cz) {
  cx = min(max((int)floor((x - g.minx) * g.inv_cell), 0), g.nx - 1);
  cy = min(max((int)floor((y - g.miny) * g.inv_cell), 0), g.ny - 1);
  cz = min(max((int)floor((z - g.minz) * g.inv_cell), 0), g.nz - 1);
}

__global__ __launch_bounds__(AI_BLOCK) void kp_bounds(const double* __restrict__ xyz, int64_t n, double* __restrict__ part) {
  __shared__ double sm[6][AI_BLOCK / 64];
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * AI_BLOCK)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double v = xyz[i * 3 + a];
      mn[a] = fmin(mn[a], v);
      mx[a] = fmax(mx[a], v);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = fmin(mn[a], __shfl_xor(mn[a], o, 64));
      mx[a] = fmax(mx[a], __shfl_xor(mx[a], o, 64));
    }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      sm[a][w] = mn[a];
      sm[3 + a][w] = mx[a];
    }
  __syncthreads();
  if (threadIdx.x < 6) {
    double r = sm[threadIdx.x][0];
    for (int i = 1; i < AI_BLOCK / 64; ++i) r = (threadIdx.x < 3) ? fmin(r, sm[threadIdx.x][i]) : fmax(r, sm[threadIdx.x][i]);
    part[blockIdx.x * 6 + threadIdx.x] = r;
  }
}

// linear cell id as the sort key (the grid is small enough for 32 bits), value = point index
__global__ __launch_bounds__(AI_BLOCK) void kp_keys(const double* __restrict__ xyz, int64_t n, PGrid g, uint32_t* __restrict__ key,
                                                    int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  pcell_of(g, xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], cx, cy, cz);
  key[i] = (uint32_t)((cz * g.ny + cy) * g.nx + cx);
  idx[i] = (int32_t)i;
}

__global__ __launch_bounds__(AI_BLOCK) void kp_gather(const double* __restrict__ xyz, const int32_t* __restrict__ order,
                                                      const uint32_t* __restrict__ skey, int64_t n, double* __restrict__ X,
                                                      double* __restrict__ Y, double* __restrict__ Z, int32_t* __restrict__ cstart,
                                                      int32_t* __restrict__ cend) {
  const int64_t p = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (p >= n) return;
  const int64_t o = order[p];
  X[p] = xyz[o * 3];
  Y[p] = xyz[o * 3 + 1];
  Z[p] = xyz[o * 3 + 2];
  const uint32_t c = skey[p];
  if (p == 0 || skey[p - 1] != c) cstart[c] = (int32_t)p;
  if (p == n - 1 || skey[p + 1] != c) cend[c] = (int32_t)(p + 1);
}

// 16 lanes per query point: mean of the float32 feature rows of all source points with
// distance < radius (open3d's radius search is strict), accumulated in float64 like np.mean over
// the float64 copy the reference holds.
__global__ __launch_bounds__(AI_BLOCK) void kp_radius_mean(const double* __restrict__ q, int64_t nq, PGrid g, double radius,
                                                           const double* __restrict__ X, const double* __restrict__ Y,
                                                           const double* __restrict__ Z, const int32_t* __restrict__ order,
                                                           const int32_t* __restrict__ cstart, const int32_t* __restrict__ cend,
                                                           const float* __restrict__ feat, int32_t dim, double* __restrict__ out,
                                                           int32_t* __restrict__ count) {
  const int64_t gid = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  const int64_t i = gid >> 4;
  const int t = (int)(gid & 15);
  if (i >= nq) return;
  const double x = q[i * 3], y = q[i * 3 + 1], z = q[i * 3 + 2];
  int cx, cy, cz;
  // a query may lie outside the sources' bounding box: clamp, the +-1 ring still covers radius <= cell
  pcell_of(g, x, y, z, cx, cy, cz);
  constexpr int MAXK = 24;  // dim <= 384
  double acc[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; ++k) acc[k] = 0.0;
  int cnt = 0;
  const double r2 = radius * radius;
  for (int dz = -1; dz <= 1; ++dz) {
    const int zz = cz + dz;
    if (zz < 0 || zz >= g.nz) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = cy + dy;
      if (yy < 0 || yy >= g.ny) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = cx + dx;
        if (xx < 0 || xx >= g.nx) continue;
        const int32_t cc = (zz * g.ny + yy) * g.nx + xx;
        const int32_t s = cstart[cc];
        if (s < 0) continue;
        const int32_t e = cend[cc];
        for (int32_t p = s; p < e; ++p) {
          const double ddx = x - X[p], ddy = y - Y[p], ddz = z - Z[p];
          const double d2 = ddx * ddx + ddy * ddy + ddz * ddz;
          if (d2 < r2) {
            const float* f = feat + (int64_t)order[p] * dim;
#pragma unroll
            for (int k = 0; k < MAXK; ++k) {
              const int c = t + 16 * k;
              if (c < dim) acc[k] += (double)f[c];
            }
            ++cnt;
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    const int c = t + 16 * k;
    if (c < dim) out[i * dim + c] = cnt ? acc[k] / (double)cnt : 0.0;  // np.mean: sum, then one division
  }
  if (t == 0 && count) count[i] = cnt;
}

// One thread per fine point: exact nearest source point by growing rings of cells; a ring r can
// only hold points at distance > (r - 1) * cell, so the search stops once best <= r * cell.
__global__ __launch_bounds__(AI_BLOCK) void kp_nn1(const double* __restrict__ q, int64_t nq, PGrid g, double cell,
                                                   const double* __restrict__ X, const double* __restrict__ Y,
                                                   const double* __restrict__ Z, const int32_t* __restrict__ order,
                                                   const int32_t* __restrict__ cstart, const int32_t* __restrict__ cend,
                                                   int32_t* __restrict__ nn_idx, double* __restrict__ nn_dist) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= nq) return;
  const double x = q[i * 3], y = q[i * 3 + 1], z = q[i * 3 + 2];
  int cx, cy, cz;
  pcell_of(g, x, y, z, cx, cy, cz);
  // distance from the query to its (clamped) home cell: queries outside the box start further out
  const double ox = fmax(fmax(g.minx - x, x - (g.minx + g.nx * cell)), 0.0);
  const double oy = fmax(fmax(g.miny - y, y - (g.miny + g.ny * cell)), 0.0);
  const double oz = fmax(fmax(g.minz - z, z - (g.minz + g.nz * cell)), 0.0);
  const double outside = sqrt(ox * ox + oy * oy + oz * oz);
  double best = 1e300;
  int32_t bi = -1;
  const int rmax = max(g.nx, max(g.ny, g.nz));
  for (int r = 0; r <= rmax; ++r) {
    for (int dz = -r; dz <= r; ++dz) {
      const int zz = cz + dz;
      if (zz < 0 || zz >= g.nz) continue;
      for (int dy = -r; dy <= r; ++dy) {
        const int yy = cy + dy;
        if (yy < 0 || yy >= g.ny) continue;
        for (int dx = -r; dx <= r; ++dx) {
          if (max(abs(dx), max(abs(dy), abs(dz))) != r) continue;  // only the shell of ring r
          const int xx = cx + dx;
          if (xx < 0 || xx >= g.nx) continue;
          const int32_t cc = (zz * g.ny + yy) * g.nx + xx;
          const int32_t s = cstart[cc];
          if (s < 0) continue;
          const int32_t e = cend[cc];
          for (int32_t p = s; p < e; ++p) {
            const double ddx = x - X[p], ddy = y - Y[p], ddz = z - Z[p];
            const double d2 = ddx * ddx + ddy * ddy + ddz * ddz;
            // ties: the smaller source index wins, so the answer does not depend on the cell order
            if (d2 < best || (d2 == best && order[p] < bi)) {
              best = d2;
              bi = order[p];
            }
          }
        }
      }
    }
    // every unvisited point sits in ring >= r + 1, i.e. at least r * cell away from the home cell
    if (bi >= 0 && sqrt(best) <= (double)r * cell - outside) break;
  }
  nn_idx[i] = bi;
  if (nn_dist) nn_dist[i] = sqrt(best);
}

struct Cells {
  PGrid g;
  double cell;
  DevBuf<int32_t> order, cstart, cend;
  DevBuf<double> X, Y, Z;
};

int build_cells(ai_ctx* ctx, const double* d_xyz, int64_t n, double cell, Cells& C, const char* who) {
  hipStream_t st = ctx->stream;
  const int nb = 256;
  DevBuf<double> part;
  AI_TRY(part.alloc((size_t)nb * 6));
  hipLaunchKernelGGL(kp_bounds, dim3(nb), dim3(AI_BLOCK), 0, st, d_xyz, n, part.p);
  AI_KERNEL_CHECK();
  std::vector<double> hp((size_t)nb * 6);
  AI_HIP(hipMemcpyAsync(hp.data(), part.p, hp.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (int b = 0; b < nb; ++b)
    for (int a = 0; a < 3; ++a) {
      mn[a] = std::min(mn[a], hp[b * 6 + a]);
      mx[a] = std::max(mx[a], hp[b * 6 + 3 + a]);
    }
  for (int a = 0; a < 3; ++a)
    if (!(mn[a] <= mx[a]) || !(mx[a] - mn[a] < 1e15)) {
      ai_set_error("%s: coordinates are not finite", who);
      return AI_ERR_BAD_ARG;
    }
  // grow the cell until the dense table fits (a coarser grid only means more candidates per query)
  for (;;) {
    const double ex = (mx[0] - mn[0]) / cell, ey = (mx[1] - mn[1]) / cell, ez = (mx[2] - mn[2]) / cell;
    if ((floor(ex) + 1) * (floor(ey) + 1) * (floor(ez) + 1) <= (double)((int64_t)1 << 27)) break;
    cell *= 1.5;
  }
  C.cell = cell;
  C.g.minx = mn[0];
  C.g.miny = mn[1];
  C.g.minz = mn[2];
  C.g.inv_cell = 1.0 / cell;
  C.g.nx = (int)floor((mx[0] - mn[0]) / cell) + 1;
  C.g.ny = (int)floor((mx[1] - mn[1]) / cell) + 1;
  C.g.nz = (int)floor((mx[2] - mn[2]) / cell) + 1;
  const int64_t ncell = (int64_t)C.g.nx * C.g.ny * C.g.nz;
  const unsigned gb = (unsigned)((n + AI_BLOCK - 1) / AI_BLOCK);
  DevBuf<uint32_t> key, skey;
  DevBuf<int32_t> idx;
  AI_TRY(key.alloc(n));
  AI_TRY(skey.alloc(n));
  AI_TRY(idx.alloc(n));
  AI_TRY(C.order.alloc(n));
  AI_TRY(C.cstart.alloc(ncell));
  AI_TRY(C.cend.alloc(ncell));
  AI_TRY(C.X.alloc(n));
  AI_TRY(C.Y.alloc(n));
  AI_TRY(C.Z.alloc(n));
  hipLaunchKernelGGL(kp_keys, dim3(gb), dim3(AI_BLOCK), 0, st, d_xyz, n, C.g, key.p, idx.p);
  AI_KERNEL_CHECK();
  int bits = 1;
  while (((int64_t)1 << bits) < ncell) ++bits;
  size_t tmp_bytes = 0;
  AI_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key.p, skey.p, idx.p, C.order.p, (size_t)n, 0, bits, st));
  DevBuf<uint8_t> tmp;
  AI_TRY(tmp.alloc(tmp_bytes));
  AI_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, key.p, skey.p, idx.p, C.order.p, (size_t)n, 0, bits, st));
  AI_HIP(hipMemsetAsync(C.cstart.p, 0xff, (size_t)ncell * sizeof(int32_t), st));
  AI_HIP(hipMemsetAsync(C.cend.p, 0, (size_t)ncell * sizeof(int32_t), st));
  hipLaunchKernelGGL(kp_gather, dim3(gb), dim3(AI_BLOCK), 0, st, d_xyz, (const int32_t*)C.order.p, (const uint32_t*)skey.p, n, C.X.p, C.Y.p,
                     C.Z.p, C.cstart.p, C.cend.p);
  AI_KERNEL_CHECK();
  AI_HIP(hipStreamSynchronize(st));  // key / skey / idx / tmp go out of scope
  return AI_OK;
}

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

}  // namespace

extern "C" int ai_radius_mean_pool(ai_ctx* ctx, const double* query_xyz, int64_t nq, const double* src_xyz, int64_t ns,
                                   const float* src_feat, int32_t dim, double radius, int mem_kind, double* out, int32_t* count_out) {
  if (!ctx || !query_xyz || !src_xyz || !src_feat || !out || nq <= 0 || ns <= 0 || dim <= 0 || dim > 384 || !(radius > 0.0) ||
      nq >= ((int64_t)1 << 30) || ns >= ((int64_t)1 << 30)) {
    ai_set_error("ai_radius_mean_pool: bad argument (null pointer, empty input, dim > 384 or radius <= 0)");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);
  hipStream_t st = ctx->stream;
  DevBuf<double> own_q, own_s, d_out;
  DevBuf<float> own_f;
  DevBuf<int32_t> d_cnt;
  const double *dq, *ds;
  const float* df;
  AI_TRY(to_device(query_xyz, (size_t)nq * 3, mem_kind, own_q, &dq, st));
  AI_TRY(to_device(src_xyz, (size_t)ns * 3, mem_kind, own_s, &ds, st));
  AI_TRY(to_device(src_feat, (size_t)ns * dim, mem_kind, own_f, &df, st));
  Cells C;
  AI_TRY(build_cells(ctx, ds, ns, radius * (1.0 + 1e-9), C, "ai_radius_mean_pool"));
  double* o = out;
  int32_t* c = count_out;
  if (mem_kind != AI_MEM_DEVICE) {
    AI_TRY(d_out.alloc((size_t)nq * dim));
    AI_TRY(d_cnt.alloc(nq));
    o = d_out.p;
    c = d_cnt.p;
  }
  const unsigned gq = (unsigned)((nq * 16 + AI_BLOCK - 1) / AI_BLOCK);
  hipLaunchKernelGGL(kp_radius_mean, dim3(gq), dim3(AI_BLOCK), 0, st, dq, nq, C.g, radius, (const double*)C.X.p, (const double*)C.Y.p,
                     (const double*)C.Z.p, (const int32_t*)C.order.p, (const int32_t*)C.cstart.p, (const int32_t*)C.cend.p, df, dim, o, c);
  AI_KERNEL_CHECK();
  if (mem_kind != AI_MEM_DEVICE) {
    AI_HIP(hipMemcpyAsync(out, o, (size_t)nq * dim * sizeof(double), hipMemcpyDeviceToHost, st));
    if (count_out) AI_HIP(hipMemcpyAsync(count_out, c, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  }
  AI_HIP(hipStreamSynchronize(st));
  return AI_OK;
}

extern "C" int ai_nn1_project(ai_ctx* ctx, const double* to_xyz, int64_t nt, const double* from_xyz, int64_t nf, int mem_kind,
                              int32_t* nn_index, double* nn_dist) {
  if (!ctx || !to_xyz || !from_xyz || !nn_index || nt <= 0 || nf <= 0 || nt >= ((int64_t)1 << 30) || nf >= ((int64_t)1 << 30)) {
    ai_set_error("ai_nn1_project: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);
  hipStream_t st = ctx->stream;
  DevBuf<double> own_t, own_f, d_dist;
  DevBuf<int32_t> d_idx;
  const double *dt, *dfm;
  AI_TRY(to_device(to_xyz, (size_t)nt * 3, mem_kind, own_t, &dt, st));
  AI_TRY(to_device(from_xyz, (size_t)nf * 3, mem_kind, own_f, &dfm, st));
  // 0.5 m cells: a couple of 0.35 m major-voxel points per cell on a surface-like cloud
  Cells C;
  AI_TRY(build_cells(ctx, dfm, nf, 0.5, C, "ai_nn1_project"));
  int32_t* oi = nn_index;
  double* od = nn_dist;
  if (mem_kind != AI_MEM_DEVICE) {
    AI_TRY(d_idx.alloc(nt));
    AI_TRY(d_dist.alloc(nt));
    oi = d_idx.p;
    od = d_dist.p;
  }
  const unsigned gq = (unsigned)((nt + AI_BLOCK - 1) / AI_BLOCK);
  hipLaunchKernelGGL(kp_nn1, dim3(gq), dim3(AI_BLOCK), 0, st, dt, nt, C.g, C.cell, (const double*)C.X.p, (const double*)C.Y.p,
                     (const double*)C.Z.p, (const int32_t*)C.order.p, (const int32_t*)C.cstart.p, (const int32_t*)C.cend.p, oi, od);
  AI_KERNEL_CHECK();
  if (mem_kind != AI_MEM_DEVICE) {
    AI_HIP(hipMemcpyAsync(nn_index, oi, (size_t)nt * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (nn_dist) AI_HIP(hipMemcpyAsync(nn_dist, od, (size_t)nt * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  AI_HIP(hipStreamSynchronize(st));
  return AI_OK;
}
