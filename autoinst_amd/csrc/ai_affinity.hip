// Affinity build on a uniform cell list (cell = radius): radius graph + spatial / TARL / DINO
// kernels -> CSR on the device.
//
// Replaces pipeline/ncuts/ncuts_utils.py:60-67,112-156,159,167 (see include/autoinst_hip.h):
//   A_ij = 1[d_ij <= r] * exp(-theta t_ij) * exp(-alpha d_ij) * exp(-gamma g_ij),
//   t_ij = 0 when either TARL row is all-zero, A_ii = 1.
// The reference forms five dense N x N matrices; here points are Morton-sorted by cell, each
// point scans its 27 neighbouring cells (count pass, scan, fill pass), and a wave-per-row
// kernel computes the feature distances only for pairs inside the radius (2*E*F flops
// instead of 2*N^2*F).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "ai_common.h"

namespace {

struct Grid {
  double minx, miny, minz, inv_cell;
  int nx, ny, nz;
};

__device__ __forceinline__ uint32_t part1by2(uint32_t x) {
  x &= 0x3ffu;
  x = (x ^ (x << 16)) & 0xff0000ffu;
  x = (x ^ (x << 8)) & 0x0300f00fu;
  x = (x ^ (x << 4)) & 0x030c30c3u;
  x = (x ^ (x << 2)) & 0x09249249u;
  return x;
}

__device__ __forceinline__ void cell_of(const Grid& g, double x, double y, double z, int& cx, int& cy, int& cz) {
  cx = (int)floor((x - g.minx) * g.inv_cell);
  cy = (int)floor((y - g.miny) * g.inv_cell);
  cz = (int)floor((z - g.minz) * g.inv_cell);
  cx = min(max(cx, 0), g.nx - 1);
  cy = min(max(cy, 0), g.ny - 1);
  cz = min(max(cz, 0), g.nz - 1);
}

// per-block min / max of the coordinates -> part[block][6]
__global__ __launch_bounds__(AI_BLOCK) void k_bounds(const double* __restrict__ xyz, int64_t n, double* __restrict__ part) {
  __shared__ double sm[6][AI_BLOCK / 64];
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * AI_BLOCK) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double v = xyz[i * 3 + a];
      mn[a] = fmin(mn[a], v);
      mx[a] = fmax(mx[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = fmin(mn[a], __shfl_xor(mn[a], o, 64));
      mx[a] = fmax(mx[a], __shfl_xor(mx[a], o, 64));
    }
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      sm[a][w] = mn[a];
      sm[3 + a][w] = mx[a];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double r = sm[threadIdx.x][0];
    for (int i = 1; i < AI_BLOCK / 64; ++i) r = (threadIdx.x < 3) ? fmin(r, sm[threadIdx.x][i]) : fmax(r, sm[threadIdx.x][i]);
    part[blockIdx.x * 6 + threadIdx.x] = r;
  }
}

__global__ __launch_bounds__(AI_BLOCK) void k_cell_keys(const double* __restrict__ xyz, int64_t n, Grid g,
                                                        uint32_t* __restrict__ key, int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  cell_of(g, xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], cx, cy, cz);
  key[i] = part1by2((uint32_t)cx) | (part1by2((uint32_t)cy) << 1) | (part1by2((uint32_t)cz) << 2);
  idx[i] = (int32_t)i;
}

// sorted SoA coordinates + linear cell id per sorted point
__global__ __launch_bounds__(AI_BLOCK) void k_gather_sorted(const double* __restrict__ xyz, const int32_t* __restrict__ orig,
                                                            int64_t n, Grid g, double* __restrict__ X, double* __restrict__ Y,
                                                            double* __restrict__ Z, int32_t* __restrict__ cellid) {
  const int64_t p = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (p >= n) return;
  const int64_t o = orig[p];
  const double x = xyz[o * 3], y = xyz[o * 3 + 1], z = xyz[o * 3 + 2];
  X[p] = x;
  Y[p] = y;
  Z[p] = z;
  int cx, cy, cz;
  cell_of(g, x, y, z, cx, cy, cz);
  cellid[p] = (cz * g.ny + cy) * g.nx + cx;
}

__global__ __launch_bounds__(AI_BLOCK) void k_cell_ranges(const int32_t* __restrict__ cellid, int64_t n,
                                                          int32_t* __restrict__ cstart, int32_t* __restrict__ cend) {
  const int64_t p = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (p >= n) return;
  const int32_t c = cellid[p];
  if (p == 0 || cellid[p - 1] != c) cstart[c] = (int32_t)p;
  if (p == n - 1 || cellid[p + 1] != c) cend[c] = (int32_t)(p + 1);
}

// cdist's euclidean kernel: sqrt((dx*dx + dy*dy) + dz*dz), every product and sum rounded on
// its own (no fused multiply-add), so the radius test sees the reference's distance bit for bit
// (ncuts_utils.py:60-61).
__device__ __forceinline__ double dist3(double ax, double ay, double az, double bx, double by, double bz) {
  const double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by), dz = __dsub_rn(az, bz);
  const double s = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
  return __dsqrt_rn(s);
}

// AI_NB_LANES lanes per (sorted) point walk the 27 neighbouring cells: the candidates of a cell are taken AI_NB_LANES at a
// time (consecutive sorted points: one coalesced read of X / Y / Z each), the neighbours of a round are compacted in
// candidate order with a ballot, so a row's entries keep the order cell by cell, point by point.  FILL = false counts
// the neighbours within the radius; FILL = true writes their ids and distances.  (One thread per point walking the ~115
// candidates serially took 0.16 + 0.24 ms per 200k-point chunk.)
#define AI_NB_LANES 8
template <bool FILL>
__global__ __launch_bounds__(AI_BLOCK) void k_neighbours(const double* __restrict__ X, const double* __restrict__ Y,
                                                         const double* __restrict__ Z, const int32_t* __restrict__ cellid,
                                                         const int32_t* __restrict__ cstart, const int32_t* __restrict__ cend,
                                                         int64_t n, Grid g, double radius, int32_t* __restrict__ cnt,
                                                         const int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                         double* __restrict__ dist, int32_t* __restrict__ stash_col,
                                                         double* __restrict__ stash_dist, int stash_cap,
                                                         const int32_t* __restrict__ cnt_in) {
  // FILL = false with a stash: the counting pass also keeps the first stash_cap hits of every row (ids + distances, row p at
  // p * stash_cap), so that filling the CSR is a copy (k_nb_unstash) instead of a second walk; FILL = true with cnt_in: only
  // the rows with more than stash_cap neighbours are walked again.
  const int64_t gid = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  const int64_t p = gid / AI_NB_LANES;
  const int l = (int)(gid % AI_NB_LANES);
  const int sub = (threadIdx.x & 63) / AI_NB_LANES;  // which group of the wave
  const bool live = p < n && !(FILL && cnt_in != nullptr && cnt_in[p < n ? p : 0] <= stash_cap);
  // every lane of the wave runs the same loops (ballots need the whole wave); a dead group simply finds nothing
  const int64_t pp = live ? p : n - 1;
  const double x = X[pp], y = Y[pp], z = Z[pp];
  const int32_t c = cellid[pp];
  const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
  int32_t k = 0;
  const int32_t base = (FILL && live) ? rowptr[p] : 0;
  for (int dz = -1; dz <= 1; ++dz) {
    const int zz = cz + dz;
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = cy + dy;
      // the x-neighbours are taken as two ranges of the sorted order instead of three cells: the points are sorted by
      // Morton code with x in the lowest bit, so the cells (2 i, 2 i + 1) of one (y, z) row have consecutive codes and
      // their points are contiguous -- (cx - 1, cx) | cx + 1 for odd cx, cx - 1 | (cx, cx + 1) for even cx: the same
      // candidates in the same order, a third fewer dependent look-ups and fuller rounds (0.29 -> 0.27 ms for the two
      // passes over a 200k-point chunk; issuing all 18 look-ups before the walk costs registers and is slower: 0.30 ms)
      for (int half = 0; half < 2; ++half) {
        const int xa = (cx & 1) ? (half == 0 ? cx - 1 : cx + 1) : (half == 0 ? cx - 1 : cx);
        const int xb = (cx & 1) ? (half == 0 ? cx : cx + 1) : (half == 0 ? cx - 1 : cx + 1);
        const bool row_in = live && zz >= 0 && zz < g.nz && yy >= 0 && yy < g.ny;
        const bool ina = row_in && xa >= 0 && xa < g.nx, inb = row_in && xb != xa && xb >= 0 && xb < g.nx;
        const int32_t rowc = row_in ? (zz * g.ny + yy) * g.nx : 0;
        const int32_t sa = ina ? cstart[rowc + xa] : -1, sb = inb ? cstart[rowc + xb] : -1;
        const int32_t ea = (ina && sa >= 0) ? cend[rowc + xa] : 0, eb = (inb && sb >= 0) ? cend[rowc + xb] : 0;
        const int32_t s = (sa >= 0) ? sa : (sb >= 0 ? sb : 0);
        const int32_t e = (sb >= 0) ? eb : ea;
        // rounds of AI_NB_LANES candidates; groups of one wave may need different numbers of rounds
        int rounds = (e - s + AI_NB_LANES - 1) / AI_NB_LANES;
        int wr = rounds;
#pragma unroll
        for (int o = 32; o >= AI_NB_LANES; o >>= 1) wr = max(wr, __shfl_xor(wr, o, 64));
        for (int it = 0; it < wr; ++it) {
          const int32_t q = s + it * AI_NB_LANES + l;
          bool hit = false;
          double d = 0.0;
          if (it < rounds && q < e) {
            d = dist3(x, y, z, X[q], Y[q], Z[q]);
            hit = d <= radius;
          }
          const unsigned long long bal = __ballot(hit);
          const unsigned int gbits = (unsigned int)((bal >> (sub * AI_NB_LANES)) & ((1u << AI_NB_LANES) - 1u));
          if (FILL && hit) {
            const int before = __popc(gbits & ((1u << l) - 1u));
            col[base + k + before] = q;
            dist[base + k + before] = d;
          }
          if (!FILL && hit && stash_col != nullptr) {
            const int at = k + __popc(gbits & ((1u << l) - 1u));
            if (at < stash_cap) {
              stash_col[p * stash_cap + at] = q;
              stash_dist[p * stash_cap + at] = d;
            }
          }
          k += __popc(gbits);
        }
      }
    }
  }
  if (!FILL && live && l == 0) cnt[p] = k;
}

// total of the per-row neighbour counts in 64 bits (the row pointers are int32: a graph with 2^31 or more entries is refused)
__global__ __launch_bounds__(AI_BLOCK) void k_count_total(const int32_t* __restrict__ cnt, int64_t n, unsigned long long* __restrict__ total) {
  unsigned long long a = 0;
  for (int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * AI_BLOCK) a += (unsigned long long)cnt[i];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0 && a) atomicAdd(total, a);
}

// the stashed hits of the counting pass into their CSR rows (rows with more than `cap` neighbours are left to a second walk)
__global__ __launch_bounds__(AI_BLOCK) void k_nb_unstash(const int32_t* __restrict__ cnt, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ stash_col, const double* __restrict__ stash_dist,
                                                         int cap, int64_t n, int32_t* __restrict__ col, double* __restrict__ dist) {
  const int64_t gid = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  const int64_t p = gid / AI_NB_LANES;
  const int l = (int)(gid % AI_NB_LANES);
  if (p >= n) return;
  const int c = cnt[p];
  if (c > cap) return;
  const int32_t base = rowptr[p];
  for (int k = l; k < c; k += AI_NB_LANES) {
    col[base + k] = stash_col[p * cap + k];
    dist[base + k] = stash_dist[p * cap + k];
  }
}

// rows with more than `cap` neighbours (they need the second walk)
__global__ __launch_bounds__(AI_BLOCK) void k_count_over(const int32_t* __restrict__ cnt, int64_t n, int cap, unsigned long long* __restrict__ over) {
  unsigned long long a = 0;
  for (int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * AI_BLOCK) a += (cnt[i] > cap) ? 1ull : 0ull;
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0 && a) atomicAdd(over, a);
}

// all-zero feature row = "no TARL feature for this point" (ncuts_utils.py:143)
__global__ __launch_bounds__(AI_BLOCK) void k_zero_rows(const double* __restrict__ f, int32_t dim, const int32_t* __restrict__ orig,
                                                        int64_t n, uint8_t* __restrict__ flag) {
  const int64_t gid = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  const int64_t p = gid >> 4;
  const int t = (int)(gid & 15);
  if (p >= n) return;
  const double* r = f + (int64_t)orig[p] * dim;
  int any = 0;
  for (int k = t; k < dim; k += 16) any |= (r[k] != 0.0);
  any |= __shfl_xor(any, 8, 16);
  any |= __shfl_xor(any, 4, 16);
  any |= __shfl_xor(any, 2, 16);
  any |= __shfl_xor(any, 1, 16);
  if (t == 0) flag[p] = any ? 0 : 1;
}

// 16 lanes share one edge: squared Euclidean distance of two feature rows.
__device__ __forceinline__ double sqdist16(const double* __restrict__ a, const double* __restrict__ b, int dim, int t) {
  double s = 0.0;
  for (int k = t; k < dim; k += 16) {
    const double d = a[k] - b[k];
    s = fma(d, d, s);
  }
  return ai_group16_sum(s);
}
// the same with this row's own features already in registers (K16 = dim / 16 values per lane)
template <int K16>
__device__ __forceinline__ double sqdist16_reg(const double (&fi)[K16 > 0 ? K16 : 1], const double* __restrict__ b, int t) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < K16; ++k) {
    const double d = fi[k] - b[t + 16 * k];
    s = fma(d, d, s);
  }
  return ai_group16_sum(s);
}

// One wave per row, four edges in flight (16 lanes each) for the feature distances.  On entry val[] holds the spatial
// distance d_ij; on exit the affinity.  Factors are multiplied in the reference's order
// (tarl * spatial * dino, ncuts_utils.py:151-156).  TK / DK = feature width / 16 when the row's own
// features are kept in registers for the whole row (96-d TARL: 6, 384-d DINO: 24), 0 = generic
// width read from memory per edge.  Lane t always owns dimensions t, t + 16, ... in both forms, so
// the summation order (and therefore every bit of the result) is the same.
template <int TK, int DK>
__global__ __launch_bounds__(AI_BLOCK) void k_weights(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                      double* __restrict__ val, const int32_t* __restrict__ orig, int64_t n,
                                                      const double* __restrict__ tarl, int32_t tdim,
                                                      const uint8_t* __restrict__ notarl, const double* __restrict__ dino,
                                                      int32_t ddim, double alpha, double theta, double gamma,
                                                      const int32_t* __restrict__ sam, int32_t nviews, double beta, int extra) {
  // workgroups are dealt round-robin to the 8 XCDs: remap so that each XCD walks one contiguous eighth of
  // the Morton-ordered rows and the neighbours' feature rows are re-used from ITS 4 MB L2 (the 154 MB
  // feature matrix itself only fits the Infinity Cache)
  const int64_t row = (int64_t)ai_xcd_task(blockIdx.x, gridDim.x) * (AI_BLOCK / 64) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const int grp = lane >> 4, t = lane & 15;
  const int64_t oi = orig ? orig[row] : row;  // a graph uploaded with ai_csr_from_host has no permutation of its own
  const bool use_t = (theta != 0.0) && tarl != nullptr;
  const bool use_d = (gamma != 0.0) && dino != nullptr;
  const bool nti = use_t ? (notarl[row] != 0) : false;
  double fti[TK > 0 ? TK : 1], fdi[DK > 0 ? DK : 1];
  if (TK > 0 && use_t) {
#pragma unroll
    for (int k = 0; k < TK; ++k) fti[k] = tarl[oi * tdim + t + 16 * k];
  }
  if (DK > 0 && use_d) {
#pragma unroll
    for (int k = 0; k < DK; ++k) fdi[k] = dino[oi * ddim + t + 16 * k];
  }
  const int32_t e0 = rowptr[row], e1 = rowptr[row + 1];
  // 64 edges per round: sixteen passes of four edges (16 lanes each) leave the squared feature distance
  // of edge eb + 4 q + grp in every lane of group grp; lane (grp, t) keeps the one of pass q = t, so that
  // afterwards EVERY lane owns one edge and the sqrt / exp epilogue (the bulk of the instructions) runs
  // with all 64 lanes busy instead of one lane per edge
  for (int32_t eb = e0; eb < e1; eb += 64) {
    double my_t2 = 0.0, my_g2 = 0.0;
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
      if (eb + 4 * q >= e1) break;  // wave-uniform
      const int32_t e = eb + 4 * q + grp;
      const bool act = e < e1;
      const int32_t j = act ? col[e] : (int32_t)row;
      const int64_t oj = orig ? orig[j] : j;
      double t2 = 0.0, g2 = 0.0;
      if (use_t) {
        const bool skip = nti || (notarl[j] != 0);
        t2 = (TK > 0) ? sqdist16_reg<TK>(fti, tarl + oj * tdim, t) : sqdist16(tarl + oi * tdim, tarl + oj * tdim, tdim, t);
        if (skip) t2 = 0.0;
      }
      if (use_d) g2 = (DK > 0) ? sqdist16_reg<DK>(fdi, dino + oj * ddim, t) : sqdist16(dino + oi * ddim, dino + oj * ddim, ddim, t);
      if (t == q) {
        my_t2 = t2;
        my_g2 = g2;
      }
    }
    const int32_t e = eb + 4 * t + grp;
    if (e < e1) {
      const double d = val[e];
      // extra = 1: val[] already holds the affinity and one more camera's SAM / DINO factors multiply it
      double w = extra ? d : 1.0;
      if (use_t) w = exp(-theta * sqrt(my_t2));
      if (!extra && alpha != 0.0) w = w * exp(-alpha * d);
      if (sam != nullptr && beta != 0.0) {
        // SAM factor (utils/image/image_utils.py:64-89): fraction of the views in which both points carry an
        // id (!= -1) and the ids differ
        const int64_t oj = orig ? orig[col[e]] : col[e];
        int co = 0, diff = 0;
        for (int v = 0; v < nviews; ++v) {
          const int32_t a = sam[oi * nviews + v], b = sam[oj * nviews + v];
          const bool both = (a != -1) && (b != -1);
          co += both ? 1 : 0;
          diff += (both && a != b) ? 1 : 0;
        }
        const double frac = co ? (double)diff / (double)co : 0.0;
        w = w * exp(-beta * frac);
      }
      if (use_d) w = w * exp(-gamma * sqrt(my_g2));
      val[e] = w;
    }
  }
}


// ---- LDS-tiled form of the feature factors (the default when there is no SAM factor and the widths are
// multiples of 16).  The wave-per-row kernel above gathers E x F x 8 bytes of neighbour feature rows through
// L2 (5.6 GB for a 200k-point chunk with 96-d features: 32x the algorithmic bytes).  Here a block owns
// 16 Morton-consecutive rows; their ~600 entries touch only ~200 DISTINCT columns (consecutive rows share their 27
// neighbour cells), so the block collects that set in an LDS hash table, stages the feature rows of the set ONCE per
// 16-dimension slab, and every entry then reads its neighbour's slab from LDS.
#define AW_SLAB 16     // dimensions per slab
#define AW_SLAB_LOG 4

template <int AW_TABLE, int AW_TSHIFT>
__device__ __forceinline__ int aw_find(const int32_t* keys, int32_t c) {
  unsigned h = ((unsigned)c * 2654435761u) >> AW_TSHIFT;  // log2(AW_TABLE) bits
  for (int i = 0; i < AW_TABLE; ++i) {
    const int32_t k = keys[h];
    if (k == c) return (int)h;
    if (k < 0) return -1;
    h = (h + 1) & (AW_TABLE - 1);
  }
  return -1;
}

__device__ __forceinline__ double aw_weight(double dist, double t2, double g2, bool use_t, bool use_d, double alpha, double theta,
                                            double gamma) {
  // factors in the reference's order: tarl * spatial * dino (ncuts_utils.py:151-156; no SAM factor on this path)
  double w = 1.0;
  if (use_t) w = exp(-theta * sqrt(t2));
  if (alpha != 0.0) w = w * exp(-alpha * dist);
  if (use_d) w = w * exp(-gamma * sqrt(g2));
  return w;
}

// Lanes over dimensions: 8 lanes share one entry and each owns two dimensions of the slab, so a 16-lane row group reads
// two staged rows as two contiguous 128-byte segments (one ds_read_b128 per lane; the two segments of a 16-lane clock group
// share their 32 banks when the two staged positions have the same parity: 2-way conflicts half of the time, 35 % of the LDS
// cycles by SQ_LDS_BANK_CONFLICT, profiles/r02_pmc_affinity.txt), the row's own values sit
// in two registers, and the per-entry partial sums stay in registers across the slabs (ACC accumulators = 2 * ACC entry
// slots per row per round; a slot without an entry points at the row itself and adds zeros).  (Round 2's first tiled
// form gave every lane its own entries and walked a slab's 16 dimensions sequentially: 64 lanes then read 64 different
// staged rows at the same dimension -- 16 bank pairs, 2-4-way conflicts -- and the row's own slab was re-read from LDS by
// all 16 lanes; its hash build counted through one contended LDS atomic and probed every entry twice.  0.50 ms / 2.16 ms
// per 200k chunk (96-d / 96-d + 384-d) against 0.42 / 1.52 ms for this form; DESIGN section 5 has the phase timings.)
// Slabs go straight from global memory into LDS (global_load_lds_dwordx4: no staging registers, which is what lets three
// blocks share a CU).  Summation order of a squared distance: lane m of the 8 sums dimensions
// 16 s + 2 m, 16 s + 2 m + 1 over the slabs s = 0, 1, ... with one fused multiply-add each; the 8 partials are added as the
// balanced tree ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)) (three DPP steps; every node is commutative, so all lanes
// hold the same bits).  aw_sqdist_tree is the same order for one thread: (i, j) and (j, i) agree bit for bit whichever
// path computes them.
__device__ __forceinline__ double aw_sqdist_tree(const double* __restrict__ a, const double* __restrict__ b, int dim) {
  double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int s = 0; s < dim; s += AW_SLAB) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const double d0 = a[s + 2 * m] - b[s + 2 * m];
      p[m] = fma(d0, d0, p[m]);
      const double d1 = a[s + 2 * m + 1] - b[s + 2 * m + 1];
      p[m] = fma(d1, d1, p[m]);
    }
  }
  return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}
__device__ __forceinline__ double aw_group8_sum(double v) {
  v += ai_dpp<0xB1>(v);   // pairs
  v += ai_dpp<0x4E>(v);   // quads
  v += ai_dpp<0x141>(v);  // the two quads of a half row
  return v;
}

#ifndef AW_GRP
#define AW_GRP 2      // entry pairs whose LDS reads are in flight together (k_weights_lanes; 1 / 2 / 4: 437 / 425 / 453 us at 96-d, 1499 / 1409 / 1416 us at 96-d + 384-d)
#endif
#ifndef AW_PACK
#define AW_PACK 2     // staged offsets per register
#endif
#ifdef AW_PHASES
__device__ unsigned long long g_aw_phase[8];
#define AW_STAMP(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_aw_phase[k], t_ - aw_t); aw_t = t_; } } while (0)
#else
#define AW_STAMP(k) do { } while (0)
#endif
#define AW_ECAP 2048   // entries of a 16-row tile whose hash slot is remembered (more: fallback); twice that for 32 rows
// NT = threads per block = 16 x the rows of a tile.  256: 16-row tiles, ~200 distinct columns, 32 KB slabs, three blocks per CU.
// 512 (round 5): 32-row tiles -- consecutive Morton rows share their neighbour cells, so 32 rows touch ~300 distinct columns, not
// 2 x 200: a quarter fewer staged bytes per row, and the staging wait is what the kernel's time is -- 48 KB slabs,
// two blocks per CU (the same 96 KB of slabs in flight per CU), 128 registers per lane instead of 168.
template <int NT, int AW_MAXD, int ACC>
__global__ __launch_bounds__(NT, (NT == 256 ? 3 : 4)) void k_weights_lanes(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                               double* __restrict__ val, const int32_t* __restrict__ orig, int64_t n,
                                                               const double* __restrict__ tarl, int32_t tdim,
                                                               const uint8_t* __restrict__ notarl, const double* __restrict__ dino,
                                                               int32_t ddim, double alpha, double theta, double gamma) {
  constexpr int AW_ROWS = NT / 16;                    // 16 lanes per row
  constexpr int AW_STAGE = (AW_MAXD * AW_SLAB) / NT;  // staged values per thread per slab
  constexpr int EPT = ACC / 8;                              // finished entries per lane (epilogue)
  constexpr int AW_TABLE = 2 * NT;                          // hash slots
  constexpr int AW_TSHIFT = (NT == 256) ? 23 : 22;          // 32 - log2(AW_TABLE)
  static_assert(AW_MAXD < AW_TABLE && ACC % 8 == 0 && AW_SLAB == 16 && (NT == 256 || NT == 512), "tile shape");
  // the hash table is dead once every entry knows its staged position: it shares its LDS with the slab buffer
  __shared__ __attribute__((aligned(16))) double xs[AW_MAXD * AW_SLAB];
  int32_t* const keys = reinterpret_cast<int32_t*>(xs);
  uint16_t* const cidx = reinterpret_cast<uint16_t*>(keys + AW_TABLE);
  static_assert(sizeof(double) * AW_MAXD * AW_SLAB >= AW_TABLE * 6, "the slab buffer holds the hash table");
  constexpr int ECAP = AW_ECAP * (NT / 256);
  __shared__ uint16_t eslot[ECAP];  // hash slot, then staged position, of every entry of the tile
  __shared__ int32_t corig[AW_MAXD];
  __shared__ int32_t srow[AW_ROWS + 1];
  __shared__ int32_t wcnt[NT / 64];
  __shared__ int32_t s_over;
#ifdef AW_PHASES
  unsigned long long aw_t = wall_clock64();
#endif
  const int nblk = gridDim.x;
  const int64_t r0 = (int64_t)ai_xcd_task(blockIdx.x, nblk) * AW_ROWS;
  if (r0 >= n) return;
  const int nrows = (int)min((int64_t)AW_ROWS, n - r0);
  const int tid = threadIdx.x;
  const bool use_t = (theta != 0.0) && tarl != nullptr;
  const bool use_d = (gamma != 0.0) && dino != nullptr;
  keys[tid] = -1;
  keys[tid + NT] = -1;
  if (tid <= nrows) srow[tid] = rowptr[r0 + tid];
  if (tid == 0) s_over = 0;
  __syncthreads();
  const int32_t e0 = srow[0], e1 = srow[nrows];
  // ---- distinct columns of the tile (every row is its own neighbour, so the rows themselves are in the set): open
  // addressing in LDS; the slot an entry ended in is remembered, so that it is probed once
  if (e1 - e0 > ECAP) {
    if (tid == 0) s_over = 1;
  } else {
    for (int32_t e = e0 + tid; e < e1; e += NT) {
      const int32_t c = col[e];
      unsigned h = ((unsigned)c * 2654435761u) >> AW_TSHIFT;
      bool done = false;
      for (int i = 0; i < AW_TABLE && !done; ++i) {
        const int32_t old = atomicCAS(&keys[h], -1, c);
        if (old == -1 || old == c) {
          done = true;
        } else {
          h = (h + 1) & (AW_TABLE - 1);
        }
      }
      if (!done) s_over = 1;
      eslot[e - e0] = (uint16_t)h;
    }
  }
  __syncthreads();
  AW_STAMP(0);  // hash build
  // ---- number the occupied slots (ballot + prefix over the four waves: no counter to contend for)
  int nd = 0;
  {
    const int32_t k0 = keys[2 * tid], k1 = keys[2 * tid + 1];
    const int mine = (k0 >= 0) + (k1 >= 0);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) wcnt[tid >> 6] = incl;
    __syncthreads();
    int pos = incl - mine;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
      if (w < (tid >> 6)) pos += wcnt[w];
      nd += wcnt[w];
    }
    if (nd <= AW_MAXD && !s_over) {
      if (k0 >= 0) {
        cidx[2 * tid] = (uint16_t)pos;
        corig[pos++] = orig[k0];
      }
      if (k1 >= 0) {
        cidx[2 * tid + 1] = (uint16_t)pos;
        corig[pos] = orig[k1];
      }
    }
  }
  if (nd > AW_MAXD || s_over) {
    // ---- fallback (dense clouds: more than AW_MAXD distinct neighbours): one thread per entry, rows from global memory
    for (int32_t e = e0 + tid; e < e1; e += NT) {
      int lo = 0, hi = nrows - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (srow[mid] <= e) lo = mid; else hi = mid - 1;
      }
      const int64_t i = r0 + lo, j = col[e];
      const int64_t oi = orig[i], oj = orig[j];
      double t2 = 0.0, g2 = 0.0;
      if (use_t && !(notarl[i] || notarl[j])) t2 = aw_sqdist_tree(tarl + oi * tdim, tarl + oj * tdim, tdim);
      if (use_d) g2 = aw_sqdist_tree(dino + oi * ddim, dino + oj * ddim, ddim);
      val[e] = aw_weight(val[e], t2, g2, use_t, use_d, alpha, theta, gamma);
    }
    return;
  }
  __syncthreads();
  for (int i = tid; i < e1 - e0; i += NT) eslot[i] = cidx[eslot[i]];  // hash slot -> staged position
  __syncthreads();
  AW_STAMP(1);  // numbering, slot -> position
  const int l = tid & 15, g = tid >> 4;  // lane in the row group, row of the tile
  const int h = l >> 3, m = l & 7;       // entry slot of the pair in flight, lane of the 8 that share an entry
  const bool rlive = g < nrows;
  const int32_t p0 = rlive ? srow[g] : 0, p1 = rlive ? srow[g + 1] : 0;
  int maxlen = 0;
  for (int r = 0; r < nrows; ++r) maxlen = max(maxlen, srow[r + 1] - srow[r]);  // block-uniform
  const uint32_t ci = rlive ? (uint32_t)cidx[aw_find<AW_TABLE, AW_TSHIFT>(keys, (int32_t)(r0 + g))] : 0u;
  const bool nti = (use_t && rlive) ? (notarl[r0 + g] != 0) : false;
  for (int base = 0; base < maxlen; base += 2 * ACC) {
    const int rounds = min(ACC, (maxlen - base + 1) >> 1);  // block-uniform number of entry pairs in this round
    // staged position of entry slot 2 * it + h of this row, two per register; a slot without an entry points at the row
    // itself (distance 0: adds nothing)
    // (kept as the BYTE offset of this lane's two dimensions in the slab buffer)
    // (two 16-bit offsets per register: the 16 registers this saves are what the grouped reads below need).  The 32 reads of
    // `eslot` are unconditional (a slot without an entry reads position 0 and discards it), so that they are issued together.
    static_assert(AW_MAXD * AW_SLAB * 8 <= 65536, "a staged byte offset fits 16 bits");
    uint32_t cj2[ACC / AW_PACK];
    {
      uint32_t sl_[ACC];
#pragma unroll
      for (int it = 0; it < ACC; ++it) {
        const int32_t e = p0 + base + 2 * it + h;
        sl_[it] = eslot[(it < rounds && e < p1) ? e - e0 : 0];
      }
#pragma unroll
      for (int it = 0; it < ACC; ++it) {
        const int32_t e = p0 + base + 2 * it + h;
        const uint32_t c = (it < rounds && e < p1) ? sl_[it] : ci;
        const uint32_t off = c * (AW_SLAB * 8) + 16 * m;
        if (AW_PACK == 1) cj2[it] = off;
        else if ((it & 1) == 0) cj2[it >> 1] = off;
        else cj2[it >> 1] |= off << 16;
      }
    }
    AW_STAMP(2);  // row set-up, offsets of the round
    double acc[ACC];
    // one feature block: squared distances of the round's entries, finished sums handed to lane (it & 7) of the 8
    auto run = [&](const double* __restrict__ f, int64_t dim, double (&fin)[EPT]) {
      // one slab of the tile's distinct rows straight into LDS (global_load_lds_dwordx4: 16 bytes per lane, a wave's 64
      // lanes fill 1 KB = 8 staged rows; no staging registers): piece p = (row p >> 3, dimensions 2 (p & 7), +1)
      // where this lane's pieces start in slab 0, in 16-byte units (n * dim < 2^33, checked by the host), formed once per
      // feature block: read from `corig` inside `stage`, each of the 8 DMAs of a slab waited for its own LDS round trip
      // (the compiler cannot move a read of LDS across a DMA that writes LDS)
      uint32_t roff[AW_STAGE / 2];
#pragma unroll
      for (int i = 0; i < AW_STAGE / 2; ++i) {
        const int p = tid + NT * i;
        const int c = min(p >> 3, nd - 1);
        roff[i] = (uint32_t)(((uint64_t)(uint32_t)corig[c] * (uint64_t)dim) >> 1) + (uint32_t)(p & 7);
      }
      auto stage = [&](int sl) {
        const char* fs = reinterpret_cast<const char*>(f + sl * AW_SLAB);
#pragma unroll
        for (int i = 0; i < AW_STAGE / 2; ++i) {
          const char* src = fs + ((uint64_t)roff[i] << 4);
          double* dst = xs + 2 * ((tid & ~63) + NT * i);  // wave-uniform; lane l lands at dst + 2 l
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
      };
#pragma unroll
      for (int it = 0; it < ACC; ++it) acc[it] = 0.0;
      const int nslab = (int)(dim >> AW_SLAB_LOG);
      for (int sl = 0; sl < nslab; ++sl) {
        __syncthreads();  // the previous slab (or the hash table) has been consumed
        stage(sl);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (rlive) {
          const double2 fi = *reinterpret_cast<const double2*>(&xs[ci * AW_SLAB + 2 * m]);
          // entry pairs in groups of AW_GRP: the group's LDS reads are issued together and their latencies overlap.  (With
          // the block-uniform guard `it < rounds` around every single pair -- rounds 1-3 -- the compiler emitted, per pair,
          // read / s_waitcnt lgkmcnt(0) / four DP instructions / scalar branch: one LDS round trip of ~130 cycles exposed per
          // pair, 25 pairs x 6 slabs per wave.)  A slot past the last pair points at the row itself: d = +0, fma(0, 0, acc)
          // = acc exactly, so the pairs a group adds beyond `rounds` change no bit.
#pragma unroll
          for (int g0 = 0; g0 < ACC; g0 += AW_GRP) {
            if (g0 < rounds) {
              double2 b[AW_GRP];
#pragma unroll
              for (int u = 0; u < AW_GRP; ++u)
                b[u] = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(xs) +
                                                         (AW_PACK == 1 ? cj2[g0 + u] : ((g0 + u) & 1) ? (cj2[(g0 + u) >> 1] >> 16) : (cj2[(g0 + u) >> 1] & 0xFFFFu)));
#pragma unroll
              for (int u = 0; u < AW_GRP; ++u) {
                const double d0 = fi.x - b[u].x;
                acc[g0 + u] = fma(d0, d0, acc[g0 + u]);
                const double d1 = fi.y - b[u].y;
                acc[g0 + u] = fma(d1, d1, acc[g0 + u]);
              }
            }
          }
        }
      }
      AW_STAMP(3);  // slabs
#pragma unroll
      for (int it = 0; it < ACC; ++it) {
        const double tot = aw_group8_sum(acc[it]);
        if ((it & 7) == m) fin[it >> 3] = tot;
      }
    };
    double t2[EPT], g2[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) t2[k] = g2[k] = 0.0;
    if (use_t) run(tarl, tdim, t2);
    if (use_d) run(dino, ddim, g2);
    AW_STAMP(4);  // reductions over the 8 lanes
    // lane m of entry slot h finishes the entries 2 * (8 k + m) + h of the round
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int32_t e = p0 + base + 2 * (8 * k + m) + h;
      if (e < p1) {
        double t = t2[k];
        if (use_t && (nti || notarl[col[e]] != 0)) t = 0.0;
        val[e] = aw_weight(val[e], t, g2[k], use_t, use_d, alpha, theta, gamma);
      }
    }
    AW_STAMP(5);  // weights
  }
}

}  // namespace

// ----------------------------------------------------------------------------- host side
static int upload_if_host(const double* src, size_t count, int mem_kind, DevBuf<double>& own, const double** dev,
                          hipStream_t stream) {
  if (src == nullptr) {
    *dev = nullptr;
    return AI_OK;
  }
  if (mem_kind == AI_MEM_DEVICE) {
    *dev = src;
    return AI_OK;
  }
  AI_TRY(own.alloc(count));
  AI_HIP(hipMemcpyAsync(own.p, src, count * sizeof(double), hipMemcpyHostToDevice, stream));
  *dev = own.p;
  return AI_OK;
}

#ifdef AW_PHASES
extern "C" int ai_debug_aw_phases(unsigned long long* out, int reset) {
  AI_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_aw_phase), sizeof(unsigned long long) * 8));
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    AI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_aw_phase), z, sizeof(z)));
  }
  return AI_OK;
}
#endif

extern "C" int ai_affinity_build(ai_ctx* ctx, const double* xyz, int64_t n, const double* tarl, int32_t tarl_dim,
                                 const double* dino, int32_t dino_dim, double alpha, double theta, double gamma,
                                 double radius, int mem_kind, ai_csr** out) {
  return ai_affinity_build_sam(ctx, xyz, n, tarl, tarl_dim, dino, dino_dim, nullptr, 0, alpha, 0.0, gamma, theta, radius, mem_kind, out);
}

extern "C" int ai_affinity_build_sam(ai_ctx* ctx, const double* xyz, int64_t n, const double* tarl, int32_t tarl_dim,
                                     const double* dino, int32_t dino_dim, const int32_t* sam, int32_t sam_views, double alpha,
                                     double beta, double gamma, double theta, double radius, int mem_kind, ai_csr** out) {
  if (!ctx || !xyz || !out || n <= 0 || !(radius > 0.0)) {
    ai_set_error("ai_affinity_build: bad argument (ctx/xyz/out null, n <= 0 or radius <= 0)");
    return AI_ERR_BAD_ARG;
  }
  if (beta != 0.0 && (sam == nullptr || sam_views <= 0)) {
    // ncuts_utils.py:116-117 raises ValueError("The length should be longer than 0!")
    ai_set_error("ai_affinity_build: beta != 0 needs SAM ids (the reference raises ValueError here)");
    return AI_ERR_BAD_ARG;
  }
  if (n >= (int64_t)1 << 30) {
    ai_set_error("ai_affinity_build: n = %lld exceeds the int32 row-id range of this build", (long long)n);
    return AI_ERR_BAD_ARG;
  }
  if (gamma != 0.0 && (dino == nullptr || dino_dim <= 0)) {
    // ncuts_utils.py:126-127 raises ValueError("The length should be longer than 0!")
    ai_set_error("ai_affinity_build: gamma != 0 needs DINO features (the reference raises ValueError here)");
    return AI_ERR_BAD_ARG;
  }
  if (theta != 0.0 && (tarl == nullptr || tarl_dim <= 0)) {
    ai_set_error("ai_affinity_build: theta != 0 needs TARL features");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  *out = nullptr;
  ArenaScope arena_scope(&ctx->arena);  // scratch comes from the context's arena; the graph itself is hipMalloc'd

  DevBuf<double> own_xyz, own_tarl, own_dino;
  const double *d_xyz, *d_tarl, *d_dino;
  AI_TRY(upload_if_host(xyz, (size_t)n * 3, mem_kind, own_xyz, &d_xyz, st));
  AI_TRY(upload_if_host(theta != 0.0 ? tarl : nullptr, (size_t)n * (size_t)(tarl_dim > 0 ? tarl_dim : 0), mem_kind, own_tarl, &d_tarl, st));
  AI_TRY(upload_if_host(gamma != 0.0 ? dino : nullptr, (size_t)n * (size_t)(dino_dim > 0 ? dino_dim : 0), mem_kind, own_dino, &d_dino, st));
  DevBuf<int32_t> own_sam;
  const int32_t* d_sam = nullptr;
  if (beta != 0.0) {
    if (mem_kind == AI_MEM_DEVICE) {
      d_sam = sam;
    } else {
      AI_TRY(own_sam.alloc((size_t)n * sam_views));
      AI_HIP(hipMemcpyAsync(own_sam.p, sam, (size_t)n * sam_views * sizeof(int32_t), hipMemcpyHostToDevice, st));
      d_sam = own_sam.p;
    }
  }

  // bounds
  const int nb = 256;
  DevBuf<double> part;
  AI_TRY(part.alloc((size_t)nb * 6));
  hipLaunchKernelGGL(k_bounds, dim3(nb), dim3(AI_BLOCK), 0, st, d_xyz, n, part.p);
  AI_KERNEL_CHECK();
  std::vector<double> hpart((size_t)nb * 6);
  AI_HIP(hipMemcpyAsync(hpart.data(), part.p, hpart.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (int b = 0; b < nb; ++b)
    for (int a = 0; a < 3; ++a) {
      if (hpart[b * 6 + a] < mn[a]) mn[a] = hpart[b * 6 + a];
      if (hpart[b * 6 + 3 + a] > mx[a]) mx[a] = hpart[b * 6 + 3 + a];
    }
  for (int a = 0; a < 3; ++a)
    if (!(mn[a] <= mx[a]) || !(mx[a] - mn[a] < 1e15)) {
      ai_set_error("ai_affinity_build: coordinates are not finite");
      return AI_ERR_BAD_ARG;
    }
  Grid g;
  const double cell = radius * (1.0 + 1e-9);  // a hair wider than r: neighbours are always within +-1 cell
  g.minx = mn[0];
  g.miny = mn[1];
  g.minz = mn[2];
  g.inv_cell = 1.0 / cell;
  const double ex = (mx[0] - mn[0]) / cell, ey = (mx[1] - mn[1]) / cell, ez = (mx[2] - mn[2]) / cell;
  if (ex >= 1023.0 || ey >= 1023.0 || ez >= 1023.0) {
    ai_set_error("ai_affinity_build: extent / radius = (%.0f, %.0f, %.0f) exceeds 1023 cells per axis", ex, ey, ez);
    return AI_ERR_BAD_ARG;
  }
  g.nx = (int)floor(ex) + 1;
  g.ny = (int)floor(ey) + 1;
  g.nz = (int)floor(ez) + 1;
  const int64_t ncell = (int64_t)g.nx * g.ny * g.nz;
  if (ncell > ((int64_t)1 << 28)) {
    ai_set_error("ai_affinity_build: %lld grid cells exceed the dense cell-table limit", (long long)ncell);
    return AI_ERR_BAD_ARG;
  }

  const unsigned gb = (unsigned)((n + AI_BLOCK - 1) / AI_BLOCK);
  DevBuf<uint32_t> key, key2;
  DevBuf<int32_t> idx, cellid, cstart, cend, cnt, scantmp;
  DevBuf<double> X, Y, Z;
  DevBuf<uint8_t> notarl;
  AI_TRY(key.alloc(n));
  AI_TRY(key2.alloc(n));
  AI_TRY(idx.alloc(n));
  hipLaunchKernelGGL(k_cell_keys, dim3(gb), dim3(AI_BLOCK), 0, st, d_xyz, n, g, key.p, idx.p);
  AI_KERNEL_CHECK();

  // the graph object (rows in Morton-cell order, orig[] maps back)
  ai_csr* A = new ai_csr();
  A->n = n;
  A->nnz = 0;
  A->rowptr = nullptr;
  A->col = nullptr;
  A->val = nullptr;
  A->orig = nullptr;
  A->device = ctx->device;
  ai_register_graph(ctx, A);
  auto fail = [&](int s) {
    ai_csr_free(ctx, A);
    return s;
  };
#define AI_TRYF(expr)                \
  do {                               \
    int _s = (expr);                 \
    if (_s != AI_OK) return fail(_s); \
  } while (0)
#define AI_HIPF(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ai_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return fail((_e == hipErrorOutOfMemory) ? AI_ERR_OOM : AI_ERR_HIP);                \
    }                                                                                     \
  } while (0)

  AI_HIPF(ctx->graphs.alloc((void**)&A->orig, (size_t)n * sizeof(int32_t)));
  AI_HIPF(ctx->graphs.alloc((void**)&A->rowptr, (size_t)(n + 1) * sizeof(int32_t)));
  {
    size_t tmp_bytes = 0;
    AI_HIPF(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key.p, key2.p, idx.p, A->orig, (size_t)n, 0, 30, st));
    DevBuf<uint8_t> tmp;
    AI_TRYF(tmp.alloc(tmp_bytes));
    AI_HIPF(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, key.p, key2.p, idx.p, A->orig, (size_t)n, 0, 30, st));
    // tmp comes from the call's arena (released when the call returns, after the final synchronisation): no wait here
    if (!ai_current_arena()) AI_HIPF(hipStreamSynchronize(st));
  }
  AI_TRYF(X.alloc(n));
  AI_TRYF(Y.alloc(n));
  AI_TRYF(Z.alloc(n));
  AI_TRYF(cellid.alloc(n));
  AI_TRYF(cstart.alloc(ncell));
  AI_TRYF(cend.alloc(ncell));
  AI_TRYF(cnt.alloc(n + 1));
  AI_TRYF(scantmp.alloc(ai_scan_tmp_elems(n)));
  hipLaunchKernelGGL(k_gather_sorted, dim3(gb), dim3(AI_BLOCK), 0, st, d_xyz, A->orig, n, g, X.p, Y.p, Z.p, cellid.p);
  AI_HIPF(hipGetLastError());
  AI_HIPF(hipMemsetAsync(cstart.p, 0xff, (size_t)ncell * sizeof(int32_t), st));
  AI_HIPF(hipMemsetAsync(cend.p, 0, (size_t)ncell * sizeof(int32_t), st));
  hipLaunchKernelGGL(k_cell_ranges, dim3(gb), dim3(AI_BLOCK), 0, st, cellid.p, n, cstart.p, cend.p);
  AI_HIPF(hipGetLastError());
  const unsigned gnb = (unsigned)((n * AI_NB_LANES + AI_BLOCK - 1) / AI_BLOCK);
  // the counting walk keeps its hits (up to AI_NB_STASH per row), so that filling the CSR is a copy
  constexpr int AI_NB_STASH = 128;
  DevBuf<int32_t> stash_col;
  DevBuf<double> stash_dist;
  AI_TRYF(stash_col.alloc((size_t)n * AI_NB_STASH));
  AI_TRYF(stash_dist.alloc((size_t)n * AI_NB_STASH));
  hipLaunchKernelGGL(k_neighbours<false>, dim3(gnb), dim3(AI_BLOCK), 0, st, X.p, Y.p, Z.p, cellid.p, cstart.p, cend.p, n, g,
                     radius, cnt.p, (const int32_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, stash_col.p, stash_dist.p, AI_NB_STASH,
                     (const int32_t*)nullptr);
  AI_HIPF(hipGetLastError());
  DevBuf<unsigned long long> total;
  AI_TRYF(total.alloc(2));
  AI_HIPF(hipMemsetAsync(total.p, 0, 2 * sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_count_total, dim3(256), dim3(AI_BLOCK), 0, st, (const int32_t*)cnt.p, n, total.p);
  AI_HIPF(hipGetLastError());
  hipLaunchKernelGGL(k_count_over, dim3(256), dim3(AI_BLOCK), 0, st, (const int32_t*)cnt.p, n, AI_NB_STASH, total.p + 1);
  AI_HIPF(hipGetLastError());
  AI_TRYF(ai_exclusive_scan_i32(st, cnt.p, A->rowptr, n, scantmp.p));
  int32_t nnz32 = 0;
  unsigned long long tot2[2] = {0, 0};
  AI_HIPF(hipMemcpyAsync(&nnz32, A->rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIPF(hipMemcpyAsync(tot2, total.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  AI_HIPF(hipStreamSynchronize(st));
  const unsigned long long nnz64 = tot2[0], rows_over = tot2[1];
  if (nnz64 >= (1ull << 31)) {
    ai_set_error("ai_affinity_build: the radius graph has %llu entries; this build indexes entries with int32 (< 2^31)", nnz64);
    return fail(AI_ERR_BAD_ARG);
  }
  if (nnz32 < n) {
    ai_set_error("ai_affinity_build: internal error, nnz = %d < n (every point is its own neighbour)", nnz32);
    return fail(AI_ERR_INTERNAL);
  }
  A->nnz = nnz32;
  AI_HIPF(ctx->graphs.alloc((void**)&A->col, (size_t)A->nnz * sizeof(int32_t)));
  AI_HIPF(ctx->graphs.alloc((void**)&A->val, (size_t)A->nnz * sizeof(double)));
  hipLaunchKernelGGL(k_nb_unstash, dim3(gnb), dim3(AI_BLOCK), 0, st, (const int32_t*)cnt.p, (const int32_t*)A->rowptr, (const int32_t*)stash_col.p,
                     (const double*)stash_dist.p, AI_NB_STASH, n, A->col, A->val);
  AI_HIPF(hipGetLastError());
  if (rows_over > 0) {  // dense clouds: the rows that did not fit the stash are walked a second time
    hipLaunchKernelGGL(k_neighbours<true>, dim3(gnb), dim3(AI_BLOCK), 0, st, X.p, Y.p, Z.p, cellid.p, cstart.p, cend.p, n, g,
                       radius, (int32_t*)nullptr, (const int32_t*)A->rowptr, A->col, A->val, (int32_t*)nullptr, (double*)nullptr,
                       AI_NB_STASH, (const int32_t*)cnt.p);
    AI_HIPF(hipGetLastError());
  }
  if (d_tarl) {
    AI_TRYF(notarl.alloc(n));
    const unsigned gz = (unsigned)((n * 16 + AI_BLOCK - 1) / AI_BLOCK);
    hipLaunchKernelGGL(k_zero_rows, dim3(gz), dim3(AI_BLOCK), 0, st, d_tarl, tarl_dim, A->orig, n, notarl.p);
    AI_HIPF(hipGetLastError());
  }
  {
    const unsigned gw = (unsigned)((n + (AI_BLOCK / 64) - 1) / (AI_BLOCK / 64));
    const bool has_t = d_tarl != nullptr, has_d = d_dino != nullptr;
    static const int force_rowwise = getenv("AI_WEIGHTS_ROWWISE") ? atoi(getenv("AI_WEIGHTS_ROWWISE")) : 0;
    const bool tiled = !force_rowwise && (has_t || has_d) && d_sam == nullptr && (!has_t || tarl_dim % AW_SLAB == 0) && (!has_d || dino_dim % AW_SLAB == 0) &&
                       (uint64_t)n * (uint64_t)std::max(has_t ? tarl_dim : 0, has_d ? dino_dim : 0) < ((uint64_t)1 << 33);  // staging offsets: 32 bits of 16-byte units
    if (tiled) {
      // LDS-tiled: every distinct neighbour's feature row is read once per 16-row tile
      // 32-row tiles (512 threads) by default; AI_WEIGHTS_TILE=16 selects the 16-row form of rounds 2-4 (same bits: an entry's sum
      // does not depend on the tile it is computed in)
      static const int tile16 = getenv("AI_WEIGHTS_TILE") ? atoi(getenv("AI_WEIGHTS_TILE")) != 32 : 1;
      if (tile16)
        hipLaunchKernelGGL((k_weights_lanes<256, 256, 32>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, (const int32_t*)A->rowptr,
                           (const int32_t*)A->col, A->val, (const int32_t*)A->orig, n, d_tarl, tarl_dim, (const uint8_t*)notarl.p, d_dino, dino_dim,
                           alpha, theta, gamma);
      else
        hipLaunchKernelGGL((k_weights_lanes<512, 384, 32>), dim3((unsigned)((n + 31) / 32)), dim3(512), 0, st, (const int32_t*)A->rowptr,
                           (const int32_t*)A->col, A->val, (const int32_t*)A->orig, n, d_tarl, tarl_dim, (const uint8_t*)notarl.p, d_dino, dino_dim,
                           alpha, theta, gamma);
    } else {
    // wave per row (SAM factor, widths that are not multiples of 16): the row's own features stay in registers
    // when the width is the reference's (96-d TARL, 384-d DINO);
    // an absent factor takes the width-0 instantiation so that it costs no registers
    const bool t96 = d_tarl != nullptr && tarl_dim == 96, d384 = d_dino != nullptr && dino_dim == 384;
#define AI_LAUNCH_W(TK, DK)                                                                                                          \
  hipLaunchKernelGGL((k_weights<TK, DK>), dim3(gw), dim3(AI_BLOCK), 0, st, (const int32_t*)A->rowptr, (const int32_t*)A->col, A->val, \
                     (const int32_t*)A->orig, n, d_tarl, tarl_dim, (const uint8_t*)notarl.p, d_dino, dino_dim, alpha, theta, gamma, d_sam, \
                     sam_views, beta, 0)
    if (t96 && d384)
      AI_LAUNCH_W(6, 24);
    else if (t96)
      AI_LAUNCH_W(6, 0);
    else if (d384)
      AI_LAUNCH_W(0, 24);
    else
      AI_LAUNCH_W(0, 0);
#undef AI_LAUNCH_W
    }
    AI_HIPF(hipGetLastError());
  }
  AI_HIPF(hipStreamSynchronize(st));
  *out = A;
  return AI_OK;
#undef AI_TRYF
#undef AI_HIPF
}

// One more camera's factors on an existing graph (ncuts_utils.py:118-123 and :128-133 loop over the cameras):
// every stored value is multiplied by exp(-beta * sam fraction) * exp(-gamma * ||dino_i - dino_j||).
extern "C" int ai_affinity_apply_camera(ai_ctx* ctx, ai_csr* csr, const double* dino, int32_t dino_dim, const int32_t* sam,
                                        int32_t sam_views, double beta, double gamma, int mem_kind) {
  if (!ctx || !csr || (gamma != 0.0 && (!dino || dino_dim <= 0)) || (beta != 0.0 && (!sam || sam_views <= 0))) {
    ai_set_error("ai_affinity_apply_camera: bad argument (a non-zero weight needs its feature matrix)");
    return AI_ERR_BAD_ARG;
  }
  if (gamma == 0.0 && beta == 0.0) return AI_OK;
  AI_CHECK_GRAPH(csr, "ai_affinity_apply_camera");
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);
  hipStream_t st = ctx->stream;
  const int64_t n = csr->n;
  DevBuf<double> own_dino;
  DevBuf<int32_t> own_sam;
  const double* d_dino = nullptr;
  const int32_t* d_sam = nullptr;
  AI_TRY(upload_if_host(gamma != 0.0 ? dino : nullptr, (size_t)n * (size_t)(dino_dim > 0 ? dino_dim : 0), mem_kind, own_dino, &d_dino, st));
  if (beta != 0.0) {
    if (mem_kind == AI_MEM_DEVICE) {
      d_sam = sam;
    } else {
      AI_TRY(own_sam.alloc((size_t)n * sam_views));
      AI_HIP(hipMemcpyAsync(own_sam.p, sam, (size_t)n * sam_views * sizeof(int32_t), hipMemcpyHostToDevice, st));
      d_sam = own_sam.p;
    }
  }
  const unsigned gw = (unsigned)((n + (AI_BLOCK / 64) - 1) / (AI_BLOCK / 64));
  if (d_dino && dino_dim == 384)
    hipLaunchKernelGGL((k_weights<0, 24>), dim3(gw), dim3(AI_BLOCK), 0, st, (const int32_t*)csr->rowptr, (const int32_t*)csr->col, csr->val,
                       (const int32_t*)csr->orig, n, (const double*)nullptr, 0, (const uint8_t*)nullptr, d_dino, dino_dim, 0.0, 0.0, gamma, d_sam,
                       sam_views, beta, 1);
  else
    hipLaunchKernelGGL((k_weights<0, 0>), dim3(gw), dim3(AI_BLOCK), 0, st, (const int32_t*)csr->rowptr, (const int32_t*)csr->col, csr->val,
                       (const int32_t*)csr->orig, n, (const double*)nullptr, 0, (const uint8_t*)nullptr, d_dino, dino_dim, 0.0, 0.0, gamma, d_sam,
                       sam_views, beta, 1);
  AI_HIP(hipGetLastError());
  AI_HIP(hipStreamSynchronize(st));
  return AI_OK;
}
