// Recursive normalized cut on the device, all segments of one recursion depth in lock step.
//
// Replaces pipeline/ncuts/normalized_cut.py:1-63.  What the reference does per segment
//   W = w + I; d = colsum(W); L = D^-1/2 (D - W) D^-1/2; eigsh(L, 2, sigma=1e-10);
//   ev = eigenvector of the 2nd-smallest eigenvalue; 10-threshold sweep; recurse if mcut < T
// is kept exactly (same thresholds, strict >, first strictly-smaller cost, mask side first,
// split_lim gate on the ORIGINAL point count); what changes is how ev is found and that every
// segment of a depth is processed together:
//   * rows of a segment are contiguous ("compact order"); after a split the rows are stably
//     partitioned (mask side first) and the CSR is rebuilt without the cut edges, so the
//     left-to-right order of leaf segments is the reference's emission order;
//   * a DISCONNECTED segment (union-find over the CSR) gets an explicit null-space vector of L,
//     z = D^1/2 (1_A / vol_A - 1_B / vol_B): SciPy returns an arbitrary null-space vector there;
//   * a CONNECTED segment is solved by Lanczos on M = D^-1/2 W D^-1/2 = I - L without
//     re-orthogonalisation, every Lanczos vector kept in HBM, the known top eigenvector
//     u1 = D^1/2 1 / sqrt(vol) projected out of each new vector; the top Ritz pair of T_m is the
//     pair of L's 2nd-smallest eigenvalue.  All segments step together: one fused SpMV launch
//     per step for the whole frontier, per-segment alpha / beta by two-stage fixed-order sums
//     (no float atomics: results are reproducible run to run).
// tests/gpu_model.py is the NumPy model of this algorithm.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>

#include "ai_common.h"

#define AI_TASK_ROWS 128  // rows per block ("task"); a task never straddles two segments
#define AI_SLAB_VECS 32   // Lanczos vectors per HBM slab
#define AI_SWEEP_VALS 40  // per-task sweep partials: cut[10], assocA[10], assocB[10], cntA[10]

namespace {

// per-segment scalars that the kernels read (structure of arrays on the device)
struct SegDev {
  const int32_t* start;    // [S+1] first compact row of each segment
  const int32_t* task0;    // [S+1] first task of each segment
  int32_t* mode;           // [S] 0 = Lanczos, 1 = null-space vector, 2 = idle
  int32_t* frozen;         // [S] Lanczos finished for this segment
  int32_t* m;              // [S] size of T at freeze
  double* g;               // [S] u1 . R_j
  double* b;               // [S] norm of the projected R_j
  double* rb;              // [S] 1 / b
  double* gp;              // [S] previous step's g
  double* rbp;             // [S] previous step's 1 / b
  double* alpha;           // [S] alpha_j of the running step
  double* theta;           // [S] top Ritz value at the last check
  double* resid;           // [S] Ritz residual estimate at the last check
  double* vol;             // [S] sum of degrees
};

// ----------------------------------------------------------------------------- degrees, scaling
// deg_i = 1 + sum_j w_ij (W = w + I, normalized_cut.py:38,42); s_i = 1 / sqrt(deg_i) (:43)
__global__ __launch_bounds__(AI_BLOCK) void k_degree(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                     const int32_t* __restrict__ task_hi, const int32_t* __restrict__ rowptr,
                                                     const double* __restrict__ wraw, double* __restrict__ deg,
                                                     double* __restrict__ sinv, double* __restrict__ pvol) {
  __shared__ double sm[AI_BLOCK / 64];
  const int t = blockIdx.x;
  const int lo = task_lo[t], hi = task_hi[t];
  const int l = threadIdx.x & (AI_LPR - 1), r = threadIdx.x / AI_LPR;
  double acc = 0.0;
  for (int row = lo + r; row < hi; row += AI_BLOCK / AI_LPR) {
    const int p0 = rowptr[row], p1 = rowptr[row + 1];
    double s = 0.0;
    for (int p = p0 + l; p < p1; p += AI_LPR) s += wraw[p];
    s = ai_group16_sum(s);
    if (l == 0) {
      const double d = s + 1.0;
      deg[row] = d;
      sinv[row] = 1.0 / sqrt(d);
      acc += d;
    }
  }
  const double tot = ai_block_sum(acc, sm);
  if (threadIdx.x == 0) pvol[t] = tot;
}

// one block per segment: out[s] = sum of part[task0[s] .. task0[s+1]) in a fixed order
__global__ __launch_bounds__(AI_BLOCK) void k_seg_sum(const int32_t* __restrict__ task0, const double* __restrict__ part,
                                                      double* __restrict__ out) {
  __shared__ double sm[AI_BLOCK / 64];
  const int s = blockIdx.x;
  double a = 0.0;
  for (int t = task0[s] + threadIdx.x; t < task0[s + 1]; t += AI_BLOCK) a += part[t];
  const double tot = ai_block_sum(a, sm);
  if (threadIdx.x == 0) out[s] = tot;
}

// wm_ij = (s_i * w_ij) * s_j  (row scaling then column scaling, like D2 * (D - W) * D2, :47);
// sinv2_i = s_i * s_i is the "+ I" term of W; u1_i = sqrt(deg_i / vol_seg)
__global__ __launch_bounds__(AI_BLOCK) void k_scale(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                    const int32_t* __restrict__ task_hi, const int32_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ col, const double* __restrict__ wraw,
                                                    const double* __restrict__ deg, const double* __restrict__ sinv,
                                                    const double* __restrict__ vol, double* __restrict__ wm,
                                                    double* __restrict__ sinv2, double* __restrict__ u1) {
  const int t = blockIdx.x;
  const int lo = task_lo[t], hi = task_hi[t];
  const double v = vol[task_seg[t]];
  const int l = threadIdx.x & (AI_LPR - 1), r = threadIdx.x / AI_LPR;
  for (int row = lo + r; row < hi; row += AI_BLOCK / AI_LPR) {
    const int p0 = rowptr[row], p1 = rowptr[row + 1];
    const double si = sinv[row];
    for (int p = p0 + l; p < p1; p += AI_LPR) wm[p] = (si * wraw[p]) * sinv[col[p]];
    if (l == 0) {
      sinv2[row] = si * si;
      u1[row] = sqrt(deg[row] / v);
    }
  }
}

// ----------------------------------------------------------------------------- connected components
// Union-find with the smaller root as representative, so a component's label is its first row
// and labels do not depend on scheduling.  parent[] is read and written with agent-scope
// relaxed atomics: a CU's L1 is not refreshed by other CUs' stores inside one launch.
__device__ __forceinline__ int32_t uf_load(int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t uf_find(int32_t* parent, int32_t x) {
  int32_t p = uf_load(&parent[x]);
  while (p != x) {
    const int32_t gp = uf_load(&parent[p]);
    if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // path halving
    x = p;
    p = gp;
  }
  return x;
}
__device__ __forceinline__ void uf_unite(int32_t* parent, int32_t a, int32_t b) {
  for (int guard = 0; guard < (1 << 24); ++guard) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) {
      const int32_t t = a;
      a = b;
      b = t;
    }
    // hook the larger root under the smaller one
    const int32_t old = atomicCAS(&parent[a], a, b);
    if (old == a) return;
    a = old;
  }
}

__global__ __launch_bounds__(AI_BLOCK) void k_cc_init(int32_t* __restrict__ parent, int32_t n) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i < n) parent[i] = i;
}

__global__ __launch_bounds__(AI_BLOCK) void k_cc_hook(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                      int32_t n, int32_t* parent) {
  const int64_t gid = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  const int32_t row = (int32_t)(gid / AI_LPR);
  const int l = (int)(gid & (AI_LPR - 1));
  if (row >= n) return;
  const int p0 = rowptr[row], p1 = rowptr[row + 1];
  for (int p = p0 + l; p < p1; p += AI_LPR) {
    const int32_t c = col[p];
    if (c < row) uf_unite(parent, row, c);
  }
}

__global__ __launch_bounds__(AI_BLOCK) void k_cc_compress(int32_t* parent, int32_t n) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  int32_t x = i;
  // chains are static in this launch (no hooking), any value read is an ancestor
  for (;;) {
    const int32_t p = uf_load(&parent[x]);
    if (p == x) break;
    x = p;
  }
  __hip_atomic_store(&parent[i], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// roots per segment (integer atomics: order independent)
__global__ __launch_bounds__(AI_BLOCK) void k_cc_count(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                       const int32_t* __restrict__ task_hi, const int32_t* __restrict__ parent,
                                                       int32_t* __restrict__ ncomp) {
  const int t = blockIdx.x;
  int c = 0;
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK) c += (parent[row] == row);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&ncomp[task_seg[t]], c);
}

// ----------------------------------------------------------------------------- null-space vector
__global__ __launch_bounds__(AI_BLOCK) void k_null_rootcount(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                             const int32_t* __restrict__ task_hi, const int32_t* __restrict__ mode,
                                                             const int32_t* __restrict__ parent, int32_t* __restrict__ rcnt) {
  const int t = blockIdx.x;
  if (mode[task_seg[t]] != 1) return;
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK) atomicAdd(&rcnt[parent[row]], 1);
}

// rc[row] = size of the component if row is its root, else 0 (scanned to rank components by first row)
__global__ __launch_bounds__(AI_BLOCK) void k_null_rootvals(const int32_t* __restrict__ parent, const int32_t* __restrict__ rcnt,
                                                            int32_t n, int32_t* __restrict__ rc) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i < n) rc[i] = (parent[i] == i) ? rcnt[i] : 0;
}

// side A = first component + every later component that still ends within the first half of
// the segment's rows; partial volumes of both sides
__global__ __launch_bounds__(AI_BLOCK) void k_null_side(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                        const int32_t* __restrict__ task_hi, const int32_t* __restrict__ mode,
                                                        const int32_t* __restrict__ seg_start, const int32_t* __restrict__ parent,
                                                        const int32_t* __restrict__ rcnt, const int32_t* __restrict__ ex,
                                                        const double* __restrict__ deg, uint8_t* __restrict__ side,
                                                        double* __restrict__ pvolA, double* __restrict__ pvolB) {
  __shared__ double sm[AI_BLOCK / 64];
  const int t = blockIdx.x;
  const int s = task_seg[t];
  if (mode[s] != 1) {
    if (threadIdx.x == 0) {
      pvolA[t] = 0.0;
      pvolB[t] = 0.0;
    }
    return;
  }
  const int s0 = seg_start[s], ns = seg_start[s + 1] - s0;
  const int exs = ex[s0];
  double va = 0.0, vb = 0.0;
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK) {
    const int r = parent[row];
    const int cb = ex[r] - exs;
    const bool inA = (r == s0) || (2 * (int64_t)(cb + rcnt[r]) <= (int64_t)ns);
    side[row] = inA ? 1 : 0;
    if (inA) va += deg[row]; else vb += deg[row];
  }
  const double ta = ai_block_sum(va, sm);
  const double tb = ai_block_sum(vb, sm);
  if (threadIdx.x == 0) {
    pvolA[t] = ta;
    pvolB[t] = tb;
  }
}

__global__ __launch_bounds__(AI_BLOCK) void k_null_vec(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                       const int32_t* __restrict__ task_hi, const int32_t* __restrict__ mode,
                                                       const double* __restrict__ volA, const double* __restrict__ volB,
                                                       const double* __restrict__ deg, const uint8_t* __restrict__ side,
                                                       double* __restrict__ ev) {
  const int t = blockIdx.x;
  const int s = task_seg[t];
  if (mode[s] != 1) return;
  const double ia = 1.0 / volA[s], ib = 1.0 / volB[s];
  const double rn = 1.0 / sqrt(ia + ib);  // ||D^1/2 (1_A/volA - 1_B/volB)||^2 = 1/volA + 1/volB
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK)
    ev[row] = sqrt(deg[row]) * (side[row] ? ia : -ib) * rn;
}

// ----------------------------------------------------------------------------- Lanczos
// R_0 = hash(original id); partials of (R.R, u1.R)
__global__ __launch_bounds__(AI_BLOCK) void k_lz_init(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                      const int32_t* __restrict__ task_hi, const int32_t* __restrict__ mode,
                                                      const int32_t* __restrict__ orig, const double* __restrict__ u1,
                                                      double* __restrict__ R0, double2* __restrict__ pB) {
  __shared__ double sm[AI_BLOCK / 64];
  const int t = blockIdx.x;
  if (mode[task_seg[t]] != 0) return;
  double nn = 0.0, gg = 0.0;
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK) {
    const double r = ai_hash_unit((uint32_t)orig[row]);
    R0[row] = r;
    nn = fma(r, r, nn);
    gg = fma(u1[row], r, gg);
  }
  const double tn = ai_block_sum(nn, sm);
  const double tg = ai_block_sum(gg, sm);
  if (threadIdx.x == 0) pB[t] = make_double2(tn, tg);
}

// THE hot kernel: y = M v_j for every running segment, v_j = (R_j - g u1) / b kept implicit:
//   y_i = ((sum_k wm_ik R_j[k] + sinv2_i R_j[i]) - g u1_i) / b        (M u1 = u1)
// 16 lanes per row, coalesced 64-B / 128-B runs of col / wm, gathers of R_j served by L2 / MALL;
// per-task partial of alpha = v_j . y.
__global__ __launch_bounds__(AI_BLOCK) void k_lz_spmv(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                      const int32_t* __restrict__ task_hi, int ntask,
                                                      const int32_t* __restrict__ mode, const int32_t* __restrict__ frozen,
                                                      const double* __restrict__ seg_g, const double* __restrict__ seg_rb,
                                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                      const double* __restrict__ wm, const double* __restrict__ sinv2,
                                                      const double* __restrict__ u1, const double* __restrict__ Rj,
                                                      double* __restrict__ Y, double* __restrict__ pA) {
  __shared__ double sm[AI_BLOCK / 64];
  const int t = ai_xcd_task(blockIdx.x, ntask);
  const int s = task_seg[t];
  if (mode[s] != 0 || frozen[s]) return;
  const double g = seg_g[s], rb = seg_rb[s];
  const int lo = task_lo[t], hi = task_hi[t];
  const int l = threadIdx.x & (AI_LPR - 1), r = threadIdx.x / AI_LPR;
  double acc = 0.0;
  for (int row = lo + r; row < hi; row += AI_BLOCK / AI_LPR) {
    const int p0 = rowptr[row], p1 = rowptr[row + 1];
    double sum = 0.0;
    for (int p = p0 + l; p < p1; p += AI_LPR) sum = fma(wm[p], Rj[col[p]], sum);
    sum = ai_group16_sum(sum);
    if (l == 0) {
      const double ri = Rj[row], ui = u1[row];
      const double y = (fma(sinv2[row], ri, sum) - g * ui) * rb;
      const double v = (ri - g * ui) * rb;
      Y[row] = y;
      acc = fma(v, y, acc);
    }
  }
  const double tot = ai_block_sum(acc, sm);
  if (threadIdx.x == 0) pA[t] = tot;
}

// alpha_j per running segment (one block per segment)
__global__ __launch_bounds__(AI_BLOCK) void k_lz_alpha(SegDev sd, const double* __restrict__ pA, double* __restrict__ alpha_hist,
                                                       int mcap, int j) {
  __shared__ double sm[AI_BLOCK / 64];
  const int s = blockIdx.x;
  if (sd.mode[s] != 0 || sd.frozen[s]) return;
  double a = 0.0;
  for (int t = sd.task0[s] + threadIdx.x; t < sd.task0[s + 1]; t += AI_BLOCK) a += pA[t];
  const double tot = ai_block_sum(a, sm);
  if (threadIdx.x == 0) {
    sd.alpha[s] = tot;
    alpha_hist[(size_t)s * mcap + j] = tot;
  }
}

// R_{j+1} = y - alpha v_j - b v_{j-1}; partials of (R.R, u1.R) of the new vector
__global__ __launch_bounds__(AI_BLOCK) void k_lz_update(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                        const int32_t* __restrict__ task_hi, SegDev sd,
                                                        const double* __restrict__ u1, const double* __restrict__ Y,
                                                        const double* __restrict__ Rj, const double* __restrict__ Rjm1,
                                                        double* __restrict__ Rnext, double2* __restrict__ pB) {
  __shared__ double sm[AI_BLOCK / 64];
  const int t = blockIdx.x;
  const int s = task_seg[t];
  if (sd.mode[s] != 0 || sd.frozen[s]) return;
  const double g = sd.g[s], rb = sd.rb[s], b = sd.b[s], gp = sd.gp[s], rbp = sd.rbp[s], al = sd.alpha[s];
  double nn = 0.0, gg = 0.0;
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK) {
    const double ui = u1[row];
    const double v = (Rj[row] - g * ui) * rb;
    const double vm = (rbp != 0.0) ? (Rjm1[row] - gp * ui) * rbp : 0.0;
    const double r = Y[row] - al * v - b * vm;
    Rnext[row] = r;
    nn = fma(r, r, nn);
    gg = fma(ui, r, gg);
  }
  const double tn = ai_block_sum(nn, sm);
  const double tg = ai_block_sum(gg, sm);
  if (threadIdx.x == 0) pB[t] = make_double2(tn, tg);
}

// Number of eigenvalues of T_m (diag a[0..m), off-diag b[1..m)) that are < x, by sign changes of
// the leading principal minors p_i = det(T_i - x I), rescaled by powers of two.
__device__ __forceinline__ int sturm_lt(const double* __restrict__ a, const double* __restrict__ b, int m, double x) {
  double pm = 1.0, p = a[0] - x;
  int cnt = (p < 0.0) ? 1 : 0;
  for (int i = 1; i < m; ++i) {
    const double bb = b[i];
    double pn = (a[i] - x) * p - (bb * bb) * pm;
    if (pn == 0.0) pn = (p > 0.0) ? -1e-300 : 1e-300;  // a zero takes the sign opposite to its predecessor
    cnt += ((pn < 0.0) != (p < 0.0)) ? 1 : 0;
    pm = p;
    p = pn;
    const double ap = fabs(p);
    if (ap > 1e100 || ap < 1e-100) {
      const double sc = (ap > 1e100) ? 0x1p-400 : 0x1p400;
      p *= sc;
      pm *= sc;
    }
  }
  return cnt;
}

// One wave per running segment, after step j (m = j + 1 rows of T):
//   * b_{m}, g_{m} of the new vector from the task partials; shift the running scalars;
//   * if `check`: top eigenvalue of T_m by 64-way multisection, |s_m| by the backward
//     recurrence (the growing, hence stable, direction), residual = b_m |s_m|; freeze the segment
//     when residual <= tol, when T has reached the segment's dimension, or on breakdown.
__global__ __launch_bounds__(64) void k_lz_norm_check(SegDev sd, const double2* __restrict__ pB, double* __restrict__ alpha_hist,
                                                      double* __restrict__ b_hist, double* __restrict__ g_hist, int mcap,
                                                      int m /* = j + 1; 0 right after init */, int check, double tol, int max_iter,
                                                      int32_t* __restrict__ n_running, const int32_t* __restrict__ rowptr,
                                                      unsigned long long* __restrict__ work /* [rows, nnz] of the SpMV launches */) {
  const int s = blockIdx.x;
  if (sd.mode[s] != 0 || sd.frozen[s]) return;
  const int lane = threadIdx.x;
  double nn = 0.0, gg = 0.0;
  for (int t = sd.task0[s] + lane; t < sd.task0[s + 1]; t += 64) {
    const double2 v = pB[t];
    nn += v.x;
    gg += v.y;
  }
  nn = ai_wave_sum(nn);
  gg = ai_wave_sum(gg);
  const double nrm2 = nn - gg * gg;  // ||R - g u1||^2, u1 has unit norm
  const double bnew = sqrt(fmax(nrm2, 0.0));
  double* bh = b_hist + (size_t)s * (mcap + 1);
  double* gh = g_hist + (size_t)s * (mcap + 1);
  const double* ah = alpha_hist + (size_t)s * mcap;
  if (lane == 0) {
    bh[m] = bnew;
    gh[m] = gg;
    sd.gp[s] = sd.g[s];
    sd.rbp[s] = (m == 0) ? 0.0 : sd.rb[s];
    sd.g[s] = gg;
    sd.b[s] = bnew;
    sd.rb[s] = (bnew > 0.0) ? 1.0 / bnew : 0.0;
  }
  const int ns = sd.start[s + 1] - sd.start[s];
  const unsigned long long nnz_s = (unsigned long long)(rowptr[sd.start[s + 1]] - rowptr[sd.start[s]]);
  if (m == 0) {
    if (lane == 0) {
      atomicAdd(n_running, 1);
      atomicAdd(&work[0], (unsigned long long)ns);
      atomicAdd(&work[1], nnz_s);
    }
    return;
  }
  const int cap = min(ns - 1, max_iter);
  const bool breakdown = !(bnew > 1e-14);
  const bool last = (m >= cap) || (m >= mcap);
  if (!(check || breakdown || last)) {
    if (lane == 0) {
      atomicAdd(n_running, 1);
      atomicAdd(&work[0], (unsigned long long)ns);
      atomicAdd(&work[1], nnz_s);
    }
    return;
  }
  // ---- top eigenvalue of T_m
  double lo = -1e300, hi = -1e300;
  for (int i = lane; i < m; i += 64) {
    const double bl = (i > 0) ? bh[i] : 0.0, br = (i + 1 < m) ? bh[i + 1] : 0.0;
    lo = fmax(lo, ah[i]);
    hi = fmax(hi, ah[i] + fabs(bl) + fabs(br));
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmax(lo, __shfl_xor(lo, o, 64));
    hi = fmax(hi, __shfl_xor(hi, o, 64));
  }
  // lambda_max >= max diagonal; widen both ends a little so that count(lo) < m <= count(hi)
  for (int round = 0; round < 12; ++round) {
    const double w = (hi - lo) * (1.0 / 65.0);
    const double x = lo + (lane + 1) * w;
    const int c = sturm_lt(ah, bh, m, x);
    const unsigned long long above = __ballot(c == m);  // lanes whose x exceeds every eigenvalue
    if (above == 0ull) {
      lo = lo + 64.0 * w;
    } else {
      const int l0 = __ffsll((long long)above) - 1;
      hi = lo + (l0 + 1) * w;
      lo = lo + l0 * w;
    }
    if (hi - lo <= 4.4e-16 * fmax(fabs(hi), 1e-300)) break;
  }
  const double theta = 0.5 * (lo + hi);
  // ---- |s_m| / ||s|| by the recurrence from the bottom row upwards (s_m = 1)
  if (lane == 0) {
    // s_m = 1 at the start; `scale` follows the rescalings so that |s_m| = scale at the end
    double sk1 = 0.0, sk = 1.0, sumsq = 1.0, scale = 1.0;  // s_{k+1}, s_k
    for (int k = m - 1; k >= 1; --k) {
      const double bu = (k + 1 < m) ? bh[k + 1] : 0.0;
      const double sm1 = ((theta - ah[k]) * sk - bu * sk1) / bh[k];
      sk1 = sk;
      sk = sm1;
      sumsq += sk * sk;
      if (sumsq > 1e200) {  // only ratios matter
        sk *= 1e-100;
        sk1 *= 1e-100;
        sumsq *= 1e-200;
        scale *= 1e-100;
      }
    }
    const double resid = bnew * scale / sqrt(sumsq);
    const bool conv = (resid <= tol);
    sd.theta[s] = theta;
    sd.resid[s] = resid;
    if (conv || breakdown || last) {
      sd.frozen[s] = 1;
      sd.m[s] = m;
    } else {
      atomicAdd(n_running, 1);
      atomicAdd(&work[0], (unsigned long long)ns);
      atomicAdd(&work[1], nnz_s);
    }
  }
}

// ev = sum_j coef_j R_j + cu u1 for the rows of frozen Lanczos segments (one slab of vectors per launch)
__global__ __launch_bounds__(AI_BLOCK) void k_ritz(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                   const int32_t* __restrict__ task_hi, const int32_t* __restrict__ mode,
                                                   const int32_t* __restrict__ seg_m, const double* __restrict__ coef, int mcap,
                                                   const double* __restrict__ cu, const double* __restrict__ u1,
                                                   const double* __restrict__ slab, size_t stride, int j0, int nvec, int first,
                                                   double* __restrict__ ev) {
  const int t = blockIdx.x;
  const int s = task_seg[t];
  if (mode[s] != 0) return;
  const int m = seg_m[s];
  const int jn = min(nvec, m - j0);
  if (jn <= 0 && !first) return;
  const double* cs = coef + (size_t)s * mcap + j0;
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK) {
    double acc = first ? cu[s] * u1[row] : ev[row];
    for (int j = 0; j < jn; ++j) acc = fma(cs[j], slab[(size_t)j * stride + row], acc);
    ev[row] = acc;
  }
}

// ----------------------------------------------------------------------------- threshold sweep
struct MinMaxPart {
  double mn, mx, sumsq, amax;
  int32_t amax_id;   // original id of the entry of largest magnitude (smallest id on ties)
  int32_t amax_neg;  // that entry is negative
};

__global__ __launch_bounds__(AI_BLOCK) void k_minmax(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                     const int32_t* __restrict__ task_hi, const int32_t* __restrict__ mode,
                                                     const double* __restrict__ ev, const int32_t* __restrict__ orig,
                                                     MinMaxPart* __restrict__ part) {
  __shared__ MinMaxPart sm[AI_BLOCK / 64];
  const int t = blockIdx.x;
  if (mode[task_seg[t]] == 2) return;
  double mn = 1e300, mx = -1e300, ss = 0.0, am = -1.0;
  int32_t aid = 0x7fffffff, aneg = 0;
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK) {
    const double e = ev[row];
    const double a = fabs(e);
    const int32_t id = orig[row];
    mn = fmin(mn, e);
    mx = fmax(mx, e);
    ss = fma(e, e, ss);
    if (a > am || (a == am && id < aid)) {
      am = a;
      aid = id;
      aneg = e < 0.0;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    mn = fmin(mn, __shfl_xor(mn, o, 64));
    mx = fmax(mx, __shfl_xor(mx, o, 64));
    ss += __shfl_xor(ss, o, 64);
    const double oa = __shfl_xor(am, o, 64);
    const int32_t oi = __shfl_xor(aid, o, 64), on = __shfl_xor(aneg, o, 64);
    if (oa > am || (oa == am && oi < aid)) {
      am = oa;
      aid = oi;
      aneg = on;
    }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[w].mn = mn;
    sm[w].mx = mx;
    sm[w].sumsq = ss;
    sm[w].amax = am;
    sm[w].amax_id = aid;
    sm[w].amax_neg = aneg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    MinMaxPart r = sm[0];
    for (int i = 1; i < AI_BLOCK / 64; ++i) {
      r.mn = fmin(r.mn, sm[i].mn);
      r.mx = fmax(r.mx, sm[i].mx);
      r.sumsq += sm[i].sumsq;
      if (sm[i].amax > r.amax || (sm[i].amax == r.amax && sm[i].amax_id < r.amax_id)) {
        r.amax = sm[i].amax;
        r.amax_id = sm[i].amax_id;
        r.amax_neg = sm[i].amax_neg;
      }
    }
    part[t] = r;
  }
}

// Per segment: unit norm + sign convention folded into one scale; np.allclose(mn, mx) test
// (normalized_cut.py:22); thresholds t_k = k * step + mn, step = (mx - mn) / 10, exactly as
// np.linspace(mn, mx, 10, endpoint=False) computes them (:28).
__global__ void k_minmax_final(const int32_t* __restrict__ task0, const int32_t* __restrict__ mode,
                               const MinMaxPart* __restrict__ part, int S, int raw, double* __restrict__ scale,
                               int32_t* __restrict__ nosplit, double* __restrict__ thr) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  if (mode[s] == 2) {
    nosplit[s] = 1;
    scale[s] = 1.0;
    return;
  }
  MinMaxPart r = part[task0[s]];
  for (int t = task0[s] + 1; t < task0[s + 1]; ++t) {
    const MinMaxPart q = part[t];
    r.mn = fmin(r.mn, q.mn);
    r.mx = fmax(r.mx, q.mx);
    r.sumsq += q.sumsq;
    if (q.amax > r.amax || (q.amax == r.amax && q.amax_id < r.amax_id)) {
      r.amax = q.amax;
      r.amax_id = q.amax_id;
      r.amax_neg = q.amax_neg;
    }
  }
  double sc = 1.0;
  if (!raw) {
    const double nrm = sqrt(r.sumsq);
    sc = (nrm > 0.0) ? 1.0 / nrm : 1.0;
    if (r.amax_neg) sc = -sc;
  }
  const double mn = (sc > 0.0) ? r.mn * sc : r.mx * sc;
  const double mx = (sc > 0.0) ? r.mx * sc : r.mn * sc;
  scale[s] = sc;
  nosplit[s] = (fabs(mn - mx) <= 1e-8 + 1e-5 * fabs(mx)) ? 1 : 0;
  const double step = __ddiv_rn(__dsub_rn(mx, mn), 10.0);
  for (int k = 0; k < AI_NUM_CUTS; ++k) thr[s * AI_NUM_CUTS + k] = __dadd_rn(__dmul_rn((double)k, step), mn);
}

// bin_i = number of thresholds strictly below ev_i: mask_k(i) = (ev_i > t_k) = (k < bin_i)
__global__ __launch_bounds__(AI_BLOCK) void k_bin(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                  const int32_t* __restrict__ task_hi, const double* __restrict__ scale,
                                                  const double* __restrict__ thr, const double* __restrict__ ev,
                                                  uint8_t* __restrict__ bin) {
  const int t = blockIdx.x;
  const int s = task_seg[t];
  const double sc = scale[s];
  double th[AI_NUM_CUTS];
#pragma unroll
  for (int k = 0; k < AI_NUM_CUTS; ++k) th[k] = thr[s * AI_NUM_CUTS + k];
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK) {
    const double e = ev[row] * sc;
    int b = 0;
#pragma unroll
    for (int k = 0; k < AI_NUM_CUTS; ++k) b += (e > th[k]) ? 1 : 0;
    bin[row] = (uint8_t)b;
  }
}

// All 10 cut costs in one pass over the edges (normalized_cut.py:4-11 for each threshold):
//   cut_k    = sum over stored (i, j) with i in A_k, j in B_k of w_ij   (= (sum W - W_AA - W_BB) / 2)
//   assocA_k = sum_{i in A_k} deg_i, assocB_k = sum_{i in B_k} deg_i   (deg of W = w + I)
// An entry (i, j) with bin_j < bin_i is cut for every k in [bin_j, bin_i).
__global__ __launch_bounds__(AI_BLOCK) void k_sweep(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                    const int32_t* __restrict__ task_hi, const int32_t* __restrict__ nosplit,
                                                    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    const double* __restrict__ wraw, const double* __restrict__ deg,
                                                    const uint8_t* __restrict__ bin, double* __restrict__ part) {
  __shared__ double sm[AI_BLOCK / 64];
  const int t = blockIdx.x;
  if (nosplit[task_seg[t]]) return;
  const int l = threadIdx.x & (AI_LPR - 1), r = threadIdx.x / AI_LPR;
  double cut[AI_NUM_CUTS], aa[AI_NUM_CUTS], ab[AI_NUM_CUTS], ca[AI_NUM_CUTS];
#pragma unroll
  for (int k = 0; k < AI_NUM_CUTS; ++k) cut[k] = aa[k] = ab[k] = ca[k] = 0.0;
  for (int row = task_lo[t] + r; row < task_hi[t]; row += AI_BLOCK / AI_LPR) {
    const int bi = bin[row];
    const int p0 = rowptr[row], p1 = rowptr[row + 1];
    for (int p = p0 + l; p < p1; p += AI_LPR) {
      const int bj = bin[col[p]];
      const double w = wraw[p];
#pragma unroll
      for (int k = 0; k < AI_NUM_CUTS; ++k) cut[k] += (k >= bj && k < bi) ? w : 0.0;
    }
    if (l == 0) {
      const double d = deg[row];
#pragma unroll
      for (int k = 0; k < AI_NUM_CUTS; ++k) {
        const bool inA = k < bi;
        aa[k] += inA ? d : 0.0;
        ab[k] += inA ? 0.0 : d;
        ca[k] += inA ? 1.0 : 0.0;
      }
    }
  }
  double* out = part + (size_t)t * AI_SWEEP_VALS;
#pragma unroll
  for (int k = 0; k < AI_NUM_CUTS; ++k) {
    const double c = ai_block_sum(cut[k], sm);
    const double a = ai_block_sum(aa[k], sm);
    const double b = ai_block_sum(ab[k], sm);
    const double n = ai_block_sum(ca[k], sm);
    if (threadIdx.x == 0) {
      out[k] = c;
      out[AI_NUM_CUTS + k] = a;
      out[2 * AI_NUM_CUTS + k] = b;
      out[3 * AI_NUM_CUTS + k] = n;
    }
  }
}

// Per segment: ncut_k = cut_k / assocA_k + cut_k / assocB_k; first strictly smaller cost wins
// (normalized_cut.py:29-32); split iff mcut < T (:56).
__global__ __launch_bounds__(64) void k_sweep_final(const int32_t* __restrict__ task0, const int32_t* __restrict__ nosplit,
                                                    const double* __restrict__ part, double T, double* __restrict__ costs,
                                                    int32_t* __restrict__ kstar, int32_t* __restrict__ split,
                                                    int32_t* __restrict__ ntrue, double* __restrict__ mcut_out) {
  const int s = blockIdx.x;
  const int lane = threadIdx.x;
  if (nosplit[s]) {
    if (lane == 0) {
      split[s] = 0;
      kstar[s] = 0;
      ntrue[s] = 0;
      mcut_out[s] = INFINITY;
    }
    if (lane < AI_NUM_CUTS) costs[s * AI_NUM_CUTS + lane] = NAN;
    return;
  }
  // lanes 0..39 each own one of the 40 partial columns
  double acc = 0.0;
  if (lane < AI_SWEEP_VALS)
    for (int t = task0[s]; t < task0[s + 1]; ++t) acc += part[(size_t)t * AI_SWEEP_VALS + lane];
  const double cutv = __shfl(acc, lane % AI_NUM_CUTS, 64);
  const double av = __shfl(acc, AI_NUM_CUTS + lane % AI_NUM_CUTS, 64);
  const double bv = __shfl(acc, 2 * AI_NUM_CUTS + lane % AI_NUM_CUTS, 64);
  const double nv = __shfl(acc, 3 * AI_NUM_CUTS + lane % AI_NUM_CUTS, 64);
  const double cost = __dadd_rn(__ddiv_rn(cutv, av), __ddiv_rn(cutv, bv));
  if (lane < AI_NUM_CUTS) costs[s * AI_NUM_CUTS + lane] = cost;
  // every lane walks the 10 costs (shuffles need the whole wave), lane 0 stores the decision
  double best = INFINITY;
  int kb = 0;
  double nb = 0.0;
  for (int k = 0; k < AI_NUM_CUTS; ++k) {
    const double c = __shfl(cost, k, 64);
    const double n = __shfl(nv, k, 64);
    if (c < best) {
      best = c;
      kb = k;
      nb = n;
    }
  }
  if (lane == 0) {
    kstar[s] = kb;
    mcut_out[s] = best;
    split[s] = (best < T) ? 1 : 0;
    ntrue[s] = (int32_t)nb;
  }
}

// ----------------------------------------------------------------------------- partition + rebuild
__global__ __launch_bounds__(AI_BLOCK) void k_split_flags(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                          const int32_t* __restrict__ task_hi, const int32_t* __restrict__ split,
                                                          const int32_t* __restrict__ kstar, const uint8_t* __restrict__ bin,
                                                          int32_t* __restrict__ flag) {
  const int t = blockIdx.x;
  const int s = task_seg[t];
  const int sp = split[s], ks = kstar[s];
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK) flag[row] = (sp && (int)bin[row] > ks) ? 1 : 0;
}

// Stable partition inside each parent (mask side first, normalized_cut.py:57-59).  Writes the
// caller-order id of every row to its position in the final ordering and the row's index in the
// next level's compact order (-1: the row's segment is finished).
__global__ __launch_bounds__(AI_BLOCK) void k_partition(const int32_t* __restrict__ task_seg, const int32_t* __restrict__ task_lo,
                                                        const int32_t* __restrict__ task_hi, const int32_t* __restrict__ seg_start,
                                                        const int32_t* __restrict__ seg_gstart, const int32_t* __restrict__ split,
                                                        const int32_t* __restrict__ ntrue,
                                                        const int32_t* __restrict__ childA, const int32_t* __restrict__ childB,
                                                        const int32_t* __restrict__ flag, const int32_t* __restrict__ fscan,
                                                        const int32_t* __restrict__ orig, int32_t* __restrict__ final_order,
                                                        int32_t* __restrict__ map, int32_t* __restrict__ orig_next) {
  const int t = blockIdx.x;
  const int s = task_seg[t];
  const int s0 = seg_start[s], g0 = seg_gstart[s], nt = split[s] ? ntrue[s] : 0;
  const int cA = childA[s], cB = childB[s];
  const int f0 = fscan[s0];
  for (int row = task_lo[t] + threadIdx.x; row < task_hi[t]; row += AI_BLOCK) {
    const int f = flag[row];
    const int rt = fscan[row] - f0;            // mask-side rows before this one
    const int rf = (row - s0) - rt;            // other-side rows before this one
    const int newpos = f ? rt : nt + rf;
    const int32_t id = orig[row];
    final_order[g0 + newpos] = id;
    int32_t dst = -1;
    if (f) {
      if (cA >= 0) dst = cA + rt;
    } else {
      if (cB >= 0) dst = cB + rf;
    }
    map[row] = dst;
    if (dst >= 0) orig_next[dst] = id;
  }
}

// kept entries of a surviving row: both ends on the same side of the cut
__global__ __launch_bounds__(AI_BLOCK) void k_rebuild_count(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const int32_t* __restrict__ flag, const int32_t* __restrict__ map,
                                                            int32_t n, int32_t* __restrict__ newcnt) {
  const int64_t gid = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  const int32_t row = (int32_t)(gid / AI_LPR);
  const int l = (int)(gid & (AI_LPR - 1));
  if (row >= n) return;
  const int32_t dst = map[row];
  if (dst < 0) return;
  const int f = flag[row];
  int c = 0;
  for (int p = rowptr[row] + l; p < rowptr[row + 1]; p += AI_LPR) c += (flag[col[p]] == f) ? 1 : 0;
  c += __shfl_xor(c, 8, 16);
  c += __shfl_xor(c, 4, 16);
  c += __shfl_xor(c, 2, 16);
  c += __shfl_xor(c, 1, 16);
  if (l == 0) newcnt[dst] = c;
}

__global__ __launch_bounds__(AI_BLOCK) void k_rebuild_fill(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const double* __restrict__ wraw, const int32_t* __restrict__ flag,
                                                           const int32_t* __restrict__ map, int32_t n,
                                                           const int32_t* __restrict__ new_rowptr, int32_t* __restrict__ new_col,
                                                           double* __restrict__ new_w) {
  const int64_t gid = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  const int32_t row = (int32_t)(gid / AI_LPR);
  const int l = (int)(gid & (AI_LPR - 1));
  const int grp = (threadIdx.x & 63) / AI_LPR;
  const bool live = row < n;
  const int32_t dst = live ? map[row] : -1;
  const int f = (live && dst >= 0) ? flag[row] : -1;
  const int p0 = (dst >= 0) ? rowptr[row] : 0, p1 = (dst >= 0) ? rowptr[row + 1] : 0;
  int out = (dst >= 0) ? new_rowptr[dst] : 0;
  // all 64 lanes run the same number of rounds so that __ballot sees the whole wave
  int rounds = (p1 - p0 + AI_LPR - 1) / AI_LPR;
  for (int o = 32; o >= AI_LPR; o >>= 1) rounds = max(rounds, __shfl_xor(rounds, o, 64));
  for (int it = 0; it < rounds; ++it) {
    const int p = p0 + it * AI_LPR + l;
    int c = -1;
    bool keep = false;
    if (p < p1) {
      c = col[p];
      keep = (flag[c] == f);
    }
    const unsigned long long bal = __ballot(keep);
    const unsigned int gbits = (unsigned int)((bal >> (grp * AI_LPR)) & 0xffffull);
    const int before = __popc(gbits & ((1u << l) - 1u));
    if (keep) {
      new_col[out + before] = map[c];
      new_w[out + before] = wraw[p];
    }
    out += __popc(gbits);
  }
}

// y = L_sym x = x - M x  (test hook, whole graph as one segment)
__global__ __launch_bounds__(AI_BLOCK) void k_lsym_apply(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const double* __restrict__ wm, const double* __restrict__ sinv2,
                                                         const double* __restrict__ x, int32_t n, double* __restrict__ y) {
  const int64_t gid = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  const int32_t row = (int32_t)(gid / AI_LPR);
  const int l = (int)(gid & (AI_LPR - 1));
  if (row >= n) return;
  double sum = 0.0;
  for (int p = rowptr[row] + l; p < rowptr[row + 1]; p += AI_LPR) sum = fma(wm[p], x[col[p]], sum);
  sum = ai_group16_sum(sum);
  if (l == 0) y[row] = x[row] - fma(sinv2[row], x[row], sum);
}

__global__ __launch_bounds__(AI_BLOCK) void k_iota(int32_t* __restrict__ a, int32_t n) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i < n) a[i] = i;
}
__global__ __launch_bounds__(AI_BLOCK) void k_scatter_d(const double* __restrict__ src, const int32_t* __restrict__ orig,
                                                        int32_t n, double scale, double* __restrict__ dst) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i < n) dst[orig[i]] = src[i] * scale;
}
__global__ __launch_bounds__(AI_BLOCK) void k_gather_d(const double* __restrict__ src, const int32_t* __restrict__ orig,
                                                       int32_t n, double* __restrict__ dst) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i < n) dst[i] = src[orig[i]];
}
__global__ __launch_bounds__(AI_BLOCK) void k_scatter_mask(const uint8_t* __restrict__ bin, const int32_t* __restrict__ orig,
                                                           int32_t n, int kstar, uint8_t* __restrict__ dst) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i < n) dst[orig[i]] = ((int)bin[i] > kstar) ? 1 : 0;
}

// ----------------------------------------------------------------------------- host: tridiagonal eigenvector
// Eigenvector of the largest eigenvalue of T (diag a[0..m), off-diag b[1..m)): bisection for
// the eigenvalue, then inverse iteration with a pivoted tridiagonal solve (as LAPACK dstein).
static int sturm_lt_host(const double* a, const double* b, int m, double x) {
  int cnt = 0;
  double q = a[0] - x;
  if (q < 0) ++cnt;
  for (int i = 1; i < m; ++i) {
    if (fabs(q) < 1e-300) q = (q < 0) ? -1e-300 : 1e-300;
    q = a[i] - x - b[i] * b[i] / q;
    if (q < 0) ++cnt;
  }
  return cnt;
}

static void tridiag_top(const double* a, const double* b, int m, double* theta_out, std::vector<double>& s) {
  s.assign(m, 0.0);
  if (m == 1) {
    *theta_out = a[0];
    s[0] = 1.0;
    return;
  }
  double lo = -1e300, hi = -1e300, nrm = 0.0;
  for (int i = 0; i < m; ++i) {
    const double bl = i > 0 ? fabs(b[i]) : 0.0, br = i + 1 < m ? fabs(b[i + 1]) : 0.0;
    lo = std::max(lo, a[i]);
    hi = std::max(hi, a[i] + bl + br);
    nrm = std::max(nrm, fabs(a[i]) + bl + br);
  }
  lo -= 1e-14 * std::max(fabs(lo), 1.0);
  hi += 1e-14 * std::max(fabs(hi), 1.0);
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid <= lo || mid >= hi) break;
    if (sturm_lt_host(a, b, m, mid) == m) hi = mid; else lo = mid;
  }
  const double theta = 0.5 * (lo + hi);
  *theta_out = theta;
  // LU of (T - theta I) with partial pivoting (rows i, i+1): dl, d, du, du2
  std::vector<double> d(m), du(m, 0.0), du2(m, 0.0), dl(m, 0.0);
  std::vector<int> piv(m, 0);
  const double tiny = 2.3e-16 * std::max(nrm, 1e-300);
  for (int i = 0; i < m; ++i) d[i] = a[i] - theta;
  for (int i = 0; i + 1 < m; ++i) {
    du[i] = b[i + 1];
    dl[i] = b[i + 1];
  }
  for (int i = 0; i + 1 < m; ++i) {
    if (fabs(d[i]) >= fabs(dl[i])) {
      if (fabs(d[i]) < tiny) d[i] = tiny;
      const double f = dl[i] / d[i];
      dl[i] = f;
      d[i + 1] -= f * du[i];
      du2[i] = 0.0;
      piv[i] = 0;
    } else {
      const double f = d[i] / dl[i];
      d[i] = dl[i];
      dl[i] = f;
      const double t = du[i];
      du[i] = d[i + 1];
      d[i + 1] = t - f * du[i];
      if (i + 2 < m) {
        du2[i] = du[i + 1];
        du[i + 1] = -f * du[i + 1];
      }
      piv[i] = 1;
    }
  }
  if (fabs(d[m - 1]) < tiny) d[m - 1] = tiny;
  std::vector<double> x(m);
  for (int i = 0; i < m; ++i) x[i] = 1.0 + 0.001 * ((i * 2654435761u) % 1000) / 1000.0;  // fixed, generic start
  for (int iter = 0; iter < 4; ++iter) {
    // forward: L y = P x
    for (int i = 0; i + 1 < m; ++i) {
      if (piv[i]) std::swap(x[i], x[i + 1]);
      x[i + 1] -= dl[i] * x[i];
    }
    // backward: U z = y
    x[m - 1] /= d[m - 1];
    if (m >= 2) x[m - 2] = (x[m - 2] - du[m - 2] * x[m - 1]) / d[m - 2];
    for (int i = m - 3; i >= 0; --i) x[i] = (x[i] - du[i] * x[i + 1] - du2[i] * x[i + 2]) / d[i];
    double n2 = 0.0;
    for (int i = 0; i < m; ++i) n2 += x[i] * x[i];
    const double rn = 1.0 / sqrt(n2);
    for (int i = 0; i < m; ++i) x[i] *= rn;
  }
  s = x;
}

// ----------------------------------------------------------------------------- host: driver
struct SegHost {
  int start, n, gstart;
  int mode;  // 0 Lanczos, 1 null vector
};

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

class Solver {
 public:
  Solver(ai_ctx* c, const ai_csr* a) : ctx(c), A(a), st(c->stream) {}

  ai_ctx* ctx;
  const ai_csr* A;
  hipStream_t st;
  ai_ncut_opts opt{1e-10, 4000, 16, 0};
  bool time_spmv = false;           // opts.reserved bit 0: HIP events around every SpMV launch
  std::vector<hipEvent_t> evpool;   // 2 per launch of the current level
  DevBuf<unsigned long long> work;  // [rows, nnz] processed by the SpMV kernel
  ai_ncut_stats stats{};

  // active set
  int na = 0;                 // active rows
  std::vector<SegHost> segs;  // active segments (host copy)
  int S() const { return (int)segs.size(); }
  const int32_t *rowptr = nullptr, *col = nullptr, *orig = nullptr;
  const double* wraw = nullptr;
  DevBuf<int32_t> b_rowptr[2], b_col[2], b_orig[2];
  DevBuf<double> b_wraw[2];
  DevBuf<int32_t> orig_id;  // identity when the graph has no permutation of its own
  int pp = 0;               // ping-pong index of the NEXT level's buffers

  // per-row work arrays
  DevBuf<double> deg, sinv, sinv2, u1, wm, ev, Y;
  DevBuf<int32_t> parent, rcnt, rc, ex, flag, fscan, map, newcnt, scantmp, final_order;
  DevBuf<uint8_t> side, bin;
  // tasks
  std::vector<int32_t> h_task_seg, h_task_lo, h_task_hi, h_task0, h_seg_start;
  DevBuf<int32_t> task_seg, task_lo, task_hi, task0, seg_start;
  int ntask = 0;
  DevBuf<double> pvol, pA, pvolA, pvolB, psweep;
  DevBuf<double2> pB;
  DevBuf<MinMaxPart> pmm;
  // per-segment device arrays
  DevBuf<int32_t> s_mode, s_frozen, s_m, s_ncomp, s_nosplit, s_kstar, s_split, s_ntrue, s_gstart, s_childA, s_childB, n_running;
  DevBuf<double> s_g, s_b, s_rb, s_gp, s_rbp, s_alpha, s_theta, s_resid, s_vol, s_volA, s_volB, s_scale, s_thr, s_costs, s_mcut, s_cu;
  // Lanczos history + vectors
  DevBuf<double> alpha_hist, b_hist, g_hist, coef;
  int mcap = 0;
  std::vector<double*> slabs;
  size_t slab_stride = 0;
  ~Solver() {
    for (double* p : slabs) (void)hipFree(p);
    for (hipEvent_t e : evpool) (void)hipEventDestroy(e);
  }

  SegDev segdev() {
    SegDev d;
    d.start = seg_start.p;
    d.task0 = task0.p;
    d.mode = s_mode.p;
    d.frozen = s_frozen.p;
    d.m = s_m.p;
    d.g = s_g.p;
    d.b = s_b.p;
    d.rb = s_rb.p;
    d.gp = s_gp.p;
    d.rbp = s_rbp.p;
    d.alpha = s_alpha.p;
    d.theta = s_theta.p;
    d.resid = s_resid.p;
    d.vol = s_vol.p;
    return d;
  }

  int alloc_rows() {
    const size_t n = (size_t)A->n, e = (size_t)A->nnz;
    AI_TRY(deg.alloc(n));
    AI_TRY(sinv.alloc(n));
    AI_TRY(sinv2.alloc(n));
    AI_TRY(u1.alloc(n));
    AI_TRY(wm.alloc(e));
    AI_TRY(ev.alloc(n));
    AI_TRY(Y.alloc(n));
    AI_TRY(parent.alloc(n));
    AI_TRY(rcnt.alloc(n));
    AI_TRY(rc.alloc(n + 1));
    AI_TRY(ex.alloc(n + 1));
    AI_TRY(flag.alloc(n + 1));
    AI_TRY(fscan.alloc(n + 1));
    AI_TRY(map.alloc(n));
    AI_TRY(newcnt.alloc(n + 1));
    AI_TRY(scantmp.alloc(ai_scan_tmp_elems((int64_t)n)));
    AI_TRY(final_order.alloc(n));
    AI_TRY(side.alloc(n));
    AI_TRY(bin.alloc(n));
    AI_TRY(n_running.alloc(1));
    AI_TRY(work.alloc(2));
    AI_HIP(hipMemsetAsync(work.p, 0, 2 * sizeof(unsigned long long), st));
    return AI_OK;
  }

  int alloc_segs(int S_) {
    const size_t s = (size_t)S_ + 1;
    AI_TRY(s_mode.ensure(s));
    AI_TRY(s_frozen.ensure(s));
    AI_TRY(s_m.ensure(s));
    AI_TRY(s_ncomp.ensure(s));
    AI_TRY(s_nosplit.ensure(s));
    AI_TRY(s_kstar.ensure(s));
    AI_TRY(s_split.ensure(s));
    AI_TRY(s_ntrue.ensure(s));
    AI_TRY(s_gstart.ensure(s));
    AI_TRY(s_childA.ensure(s));
    AI_TRY(s_childB.ensure(s));
    AI_TRY(s_g.ensure(s));
    AI_TRY(s_b.ensure(s));
    AI_TRY(s_rb.ensure(s));
    AI_TRY(s_gp.ensure(s));
    AI_TRY(s_rbp.ensure(s));
    AI_TRY(s_alpha.ensure(s));
    AI_TRY(s_theta.ensure(s));
    AI_TRY(s_resid.ensure(s));
    AI_TRY(s_vol.ensure(s));
    AI_TRY(s_volA.ensure(s));
    AI_TRY(s_volB.ensure(s));
    AI_TRY(s_scale.ensure(s));
    AI_TRY(s_thr.ensure(s * AI_NUM_CUTS));
    AI_TRY(s_costs.ensure(s * AI_NUM_CUTS));
    AI_TRY(s_mcut.ensure(s));
    AI_TRY(s_cu.ensure(s));
    return AI_OK;
  }

  // Level 0: the whole graph is one segment, rows in the graph's own order.
  int begin(bool force_single_segment) {
    const int n = (int)A->n;
    AI_TRY(alloc_rows());
    rowptr = A->rowptr;
    col = A->col;
    wraw = A->val;
    if (A->orig) {
      orig = A->orig;
    } else {
      AI_TRY(orig_id.alloc(n));
      hipLaunchKernelGGL(k_iota, dim3((n + AI_BLOCK - 1) / AI_BLOCK), dim3(AI_BLOCK), 0, st, orig_id.p, n);
      AI_KERNEL_CHECK();
      orig = orig_id.p;
    }
    AI_HIP(hipMemcpyAsync(final_order.p, orig, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    na = n;
    segs.clear();
    if (force_single_segment) segs.push_back(SegHost{0, n, 0, 0});
    return AI_OK;
  }

  // tasks + segment offsets for the current `segs`
  int build_tasks() {
    const int S_ = S();
    h_task_seg.clear();
    h_task_lo.clear();
    h_task_hi.clear();
    h_task0.assign(S_ + 1, 0);
    h_seg_start.assign(S_ + 1, 0);
    std::vector<int32_t> h_gstart(S_ + 1, 0);
    for (int s = 0; s < S_; ++s) {
      h_task0[s] = (int32_t)h_task_seg.size();
      h_seg_start[s] = segs[s].start;
      h_gstart[s] = segs[s].gstart;
      for (int lo = segs[s].start; lo < segs[s].start + segs[s].n; lo += AI_TASK_ROWS) {
        h_task_seg.push_back(s);
        h_task_lo.push_back(lo);
        h_task_hi.push_back(std::min(lo + AI_TASK_ROWS, segs[s].start + segs[s].n));
      }
    }
    h_task0[S_] = (int32_t)h_task_seg.size();
    h_seg_start[S_] = S_ ? segs[S_ - 1].start + segs[S_ - 1].n : 0;
    ntask = (int)h_task_seg.size();
    AI_TRY(alloc_segs(S_));
    AI_TRY(task_seg.ensure(ntask + 1));
    AI_TRY(task_lo.ensure(ntask + 1));
    AI_TRY(task_hi.ensure(ntask + 1));
    AI_TRY(task0.ensure(S_ + 1));
    AI_TRY(seg_start.ensure(S_ + 1));
    AI_TRY(pvol.ensure(ntask + 1));
    AI_TRY(pA.ensure(ntask + 1));
    AI_TRY(pB.ensure(ntask + 1));
    AI_TRY(pvolA.ensure(ntask + 1));
    AI_TRY(pvolB.ensure(ntask + 1));
    AI_TRY(psweep.ensure((size_t)(ntask + 1) * AI_SWEEP_VALS));
    AI_TRY(pmm.ensure(ntask + 1));
    if (ntask) {
      AI_HIP(hipMemcpyAsync(task_seg.p, h_task_seg.data(), ntask * sizeof(int32_t), hipMemcpyHostToDevice, st));
      AI_HIP(hipMemcpyAsync(task_lo.p, h_task_lo.data(), ntask * sizeof(int32_t), hipMemcpyHostToDevice, st));
      AI_HIP(hipMemcpyAsync(task_hi.p, h_task_hi.data(), ntask * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    AI_HIP(hipMemcpyAsync(task0.p, h_task0.data(), (S_ + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    AI_HIP(hipMemcpyAsync(seg_start.p, h_seg_start.data(), (S_ + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    AI_HIP(hipMemcpyAsync(s_gstart.p, h_gstart.data(), (S_ + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    // the host vectors must outlive the copies
    AI_HIP(hipStreamSynchronize(st));
    return AI_OK;
  }

  // degrees, scaled matrix, u1, connected components -> segs[s].mode
  int prepare(bool want_cc) {
    const int S_ = S();
    hipLaunchKernelGGL(k_degree, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, rowptr, wraw, deg.p, sinv.p, pvol.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_seg_sum, dim3(S_), dim3(AI_BLOCK), 0, st, task0.p, pvol.p, s_vol.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_scale, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, rowptr, col, wraw, deg.p, sinv.p,
                       s_vol.p, wm.p, sinv2.p, u1.p);
    AI_KERNEL_CHECK();
    std::vector<int32_t> ncomp(S_, 1);
    if (want_cc) {
      const unsigned gr = (unsigned)((na + AI_BLOCK - 1) / AI_BLOCK);
      const unsigned ge = (unsigned)(((int64_t)na * AI_LPR + AI_BLOCK - 1) / AI_BLOCK);
      hipLaunchKernelGGL(k_cc_init, dim3(gr), dim3(AI_BLOCK), 0, st, parent.p, na);
      AI_KERNEL_CHECK();
      hipLaunchKernelGGL(k_cc_hook, dim3(ge), dim3(AI_BLOCK), 0, st, rowptr, col, na, parent.p);
      AI_KERNEL_CHECK();
      hipLaunchKernelGGL(k_cc_compress, dim3(gr), dim3(AI_BLOCK), 0, st, parent.p, na);
      AI_KERNEL_CHECK();
      AI_HIP(hipMemsetAsync(s_ncomp.p, 0, (size_t)S_ * sizeof(int32_t), st));
      hipLaunchKernelGGL(k_cc_count, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, parent.p, s_ncomp.p);
      AI_KERNEL_CHECK();
      AI_HIP(hipMemcpyAsync(ncomp.data(), s_ncomp.p, (size_t)S_ * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      AI_HIP(hipStreamSynchronize(st));
    }
    std::vector<int32_t> mode(S_);
    for (int s = 0; s < S_; ++s) {
      segs[s].mode = (ncomp[s] > 1) ? 1 : 0;
      mode[s] = segs[s].mode;
    }
    AI_HIP(hipMemcpyAsync(s_mode.p, mode.data(), (size_t)S_ * sizeof(int32_t), hipMemcpyHostToDevice, st));
    AI_HIP(hipStreamSynchronize(st));
    return AI_OK;
  }

  int null_vectors() {
    const int S_ = S();
    bool any = false;
    for (auto& s : segs) any |= (s.mode == 1);
    if (!any) return AI_OK;
    const unsigned gr = (unsigned)((na + AI_BLOCK - 1) / AI_BLOCK);
    AI_HIP(hipMemsetAsync(rcnt.p, 0, (size_t)na * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_null_rootcount, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, s_mode.p, parent.p, rcnt.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_null_rootvals, dim3(gr), dim3(AI_BLOCK), 0, st, parent.p, rcnt.p, na, rc.p);
    AI_KERNEL_CHECK();
    AI_TRY(ai_exclusive_scan_i32(st, rc.p, ex.p, na, scantmp.p));
    hipLaunchKernelGGL(k_null_side, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, s_mode.p, seg_start.p, parent.p,
                       rcnt.p, ex.p, deg.p, side.p, pvolA.p, pvolB.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_seg_sum, dim3(S_), dim3(AI_BLOCK), 0, st, task0.p, pvolA.p, s_volA.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_seg_sum, dim3(S_), dim3(AI_BLOCK), 0, st, task0.p, pvolB.p, s_volB.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_null_vec, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, s_mode.p, s_volA.p, s_volB.p,
                       deg.p, side.p, ev.p);
    AI_KERNEL_CHECK();
    for (auto& s : segs) stats.null_solves += (s.mode == 1);
    return AI_OK;
  }

  double* vec(int j) { return slabs[(size_t)j / AI_SLAB_VECS] + (size_t)(j % AI_SLAB_VECS) * slab_stride; }
  int ensure_vec(int j) {
    while ((size_t)j / AI_SLAB_VECS >= slabs.size()) {
      double* p = nullptr;
      hipError_t e = hipMalloc((void**)&p, (size_t)AI_SLAB_VECS * slab_stride * sizeof(double));
      if (e != hipSuccess) {
        ai_set_error("Lanczos vector slab %zu (%zu bytes) could not be allocated: %s", slabs.size(),
                     (size_t)AI_SLAB_VECS * slab_stride * sizeof(double), hipGetErrorString(e));
        return AI_ERR_OOM;
      }
      slabs.push_back(p);
    }
    return AI_OK;
  }

  // Lock-step Lanczos over every mode-0 segment, then Ritz vectors into ev.
  // theta_out / iters_out / resid_out (optional): values of segment 0.
  int lanczos(double* theta_out, int* iters_out, double* resid_out) {
    const int S_ = S();
    int nl = 0, max_n = 0, min_n = 1 << 30;
    int64_t lz_rows = 0;
    for (auto& s : segs)
      if (s.mode == 0) {
        ++nl;
        max_n = std::max(max_n, s.n);
        min_n = std::min(min_n, s.n);
        lz_rows += s.n;
      }
    if (nl == 0) return AI_OK;
    stats.lanczos_solves += nl;
    mcap = std::max(1, std::min(opt.max_iter, max_n - 1));
    AI_TRY(alpha_hist.ensure((size_t)S_ * mcap));
    AI_TRY(b_hist.ensure((size_t)S_ * (mcap + 1)));
    AI_TRY(g_hist.ensure((size_t)S_ * (mcap + 1)));
    AI_TRY(coef.ensure((size_t)S_ * mcap));
    // vectors live in slabs whose stride is the row count of the first level that needs them
    // (levels only shrink)
    if (slab_stride == 0) slab_stride = (size_t)na;
    AI_HIP(hipMemsetAsync(s_frozen.p, 0, (size_t)S_ * sizeof(int32_t), st));
    AI_HIP(hipMemsetAsync(s_m.p, 0, (size_t)S_ * sizeof(int32_t), st));
    AI_HIP(hipMemsetAsync(s_g.p, 0, (size_t)S_ * sizeof(double), st));
    AI_HIP(hipMemsetAsync(s_rb.p, 0, (size_t)S_ * sizeof(double), st));
    AI_HIP(hipMemsetAsync(n_running.p, 0, sizeof(int32_t), st));
    SegDev sd = segdev();
    AI_TRY(ensure_vec(0));
    AI_TRY(ensure_vec(1));
    AI_HIP(hipEventRecord(ctx->ev[0], st));
    hipLaunchKernelGGL(k_lz_init, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, s_mode.p, orig, u1.p, vec(0), pB.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_lz_norm_check, dim3(S_), dim3(64), 0, st, sd, (const double2*)pB.p, alpha_hist.p, b_hist.p, g_hist.p, mcap, 0, 0,
                       opt.tol, opt.max_iter, n_running.p, rowptr, work.p);
    AI_KERNEL_CHECK();
    const bool dense_checks = (min_n <= 512);
    int next_check = dense_checks ? 1 : opt.check_every;
    int steps = 0;
    for (int j = 0; j < mcap; ++j) {
      AI_TRY(ensure_vec(j + 1));
      if (time_spmv) {
        while (evpool.size() < (size_t)2 * (j + 1)) {
          hipEvent_t e;
          AI_HIP(hipEventCreate(&e));
          evpool.push_back(e);
        }
        AI_HIP(hipEventRecord(evpool[2 * j], st));
      }
      hipLaunchKernelGGL(k_lz_spmv, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, ntask, s_mode.p, s_frozen.p,
                         s_g.p, s_rb.p, rowptr, col, wm.p, sinv2.p, u1.p, (const double*)vec(j), Y.p, pA.p);
      AI_KERNEL_CHECK();
      if (time_spmv) AI_HIP(hipEventRecord(evpool[2 * j + 1], st));
      hipLaunchKernelGGL(k_lz_alpha, dim3(S_), dim3(AI_BLOCK), 0, st, sd, (const double*)pA.p, alpha_hist.p, mcap, j);
      AI_KERNEL_CHECK();
      hipLaunchKernelGGL(k_lz_update, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, sd, u1.p, (const double*)Y.p,
                         (const double*)vec(j), (const double*)vec(j > 0 ? j - 1 : 0), vec(j + 1), pB.p);
      AI_KERNEL_CHECK();
      const int m = j + 1;
      const bool check = (m >= next_check) || (m == mcap);
      AI_HIP(hipMemsetAsync(n_running.p, 0, sizeof(int32_t), st));
      hipLaunchKernelGGL(k_lz_norm_check, dim3(S_), dim3(64), 0, st, sd, (const double2*)pB.p, alpha_hist.p, b_hist.p, g_hist.p, mcap, m,
                         check ? 1 : 0, opt.tol, opt.max_iter, n_running.p, rowptr, work.p);
      AI_KERNEL_CHECK();
      ++steps;
      if (check || dense_checks) {
        int32_t running = 0;
        AI_HIP(hipMemcpyAsync(&running, n_running.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        AI_HIP(hipStreamSynchronize(st));
        if (check) next_check = dense_checks ? m + 1 : m + std::max(opt.check_every, (m / 8 / opt.check_every) * opt.check_every);
        if (running == 0) break;
      }
    }
    stats.lanczos_steps += steps;
    (void)lz_rows;
    if (time_spmv) {
      AI_HIP(hipStreamSynchronize(st));
      for (int j = 0; j < steps; ++j) {
        float e = 0.f;
        AI_HIP(hipEventElapsedTime(&e, evpool[2 * j], evpool[2 * j + 1]));
        stats.ms_spmv += e;
      }
    }
    // ---- Ritz coefficients on the host (tiny), Ritz vectors on the device
    std::vector<int32_t> h_m(S_);
    std::vector<double> h_a((size_t)S_ * mcap), h_b((size_t)S_ * (mcap + 1)), h_g((size_t)S_ * (mcap + 1)), h_coef((size_t)S_ * mcap, 0.0),
        h_cu(S_, 0.0), h_resid(S_, 0.0);
    AI_HIP(hipMemcpyAsync(h_m.data(), s_m.p, (size_t)S_ * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    AI_HIP(hipMemcpyAsync(h_a.data(), alpha_hist.p, h_a.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    AI_HIP(hipMemcpyAsync(h_b.data(), b_hist.p, h_b.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    AI_HIP(hipMemcpyAsync(h_g.data(), g_hist.p, h_g.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    AI_HIP(hipMemcpyAsync(h_resid.data(), s_resid.p, (size_t)S_ * sizeof(double), hipMemcpyDeviceToHost, st));
    AI_HIP(hipStreamSynchronize(st));
    int max_m = 0;
    std::vector<double> sv;
    for (int s = 0; s < S_; ++s) {
      if (segs[s].mode != 0) continue;
      const int m = h_m[s];
      if (m <= 0) {
        ai_set_error("internal: Lanczos segment %d finished with an empty tridiagonal matrix", s);
        return AI_ERR_INTERNAL;
      }
      max_m = std::max(max_m, m);
      const double* a = &h_a[(size_t)s * mcap];
      const double* b = &h_b[(size_t)s * (mcap + 1)];
      const double* g = &h_g[(size_t)s * (mcap + 1)];
      double theta = 0.0;
      tridiag_top(a, b, m, &theta, sv);
      double cu = 0.0;
      for (int j = 0; j < m; ++j) {
        const double c = sv[j] / b[j];  // v_j = (R_j - g_j u1) / b_j
        h_coef[(size_t)s * mcap + j] = c;
        cu -= c * g[j];
      }
      h_cu[s] = cu;
      // a residual above tol is a failure only if T is smaller than the segment's own dimension
      if (h_resid[s] > opt.tol && m < segs[s].n - 1) ++stats.unconverged;
      stats.max_resid = std::max(stats.max_resid, h_resid[s]);
      if (s == 0) {
        if (theta_out) *theta_out = theta;
        if (iters_out) *iters_out = m;
        if (resid_out) *resid_out = h_resid[s];
      }
    }
    AI_HIP(hipMemcpyAsync(coef.p, h_coef.data(), h_coef.size() * sizeof(double), hipMemcpyHostToDevice, st));
    AI_HIP(hipMemcpyAsync(s_cu.p, h_cu.data(), (size_t)S_ * sizeof(double), hipMemcpyHostToDevice, st));
    for (int j0 = 0; j0 < max_m; j0 += AI_SLAB_VECS) {
      hipLaunchKernelGGL(k_ritz, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, s_mode.p, s_m.p, coef.p, mcap, s_cu.p,
                         u1.p, (const double*)slabs[(size_t)j0 / AI_SLAB_VECS], slab_stride, j0, AI_SLAB_VECS, j0 == 0 ? 1 : 0, ev.p);
      AI_KERNEL_CHECK();
    }
    AI_HIP(hipEventRecord(ctx->ev[1], st));
    AI_HIP(hipStreamSynchronize(st));  // h_coef / h_cu are read by the copies above
    float ms = 0.f;
    AI_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    stats.ms_eigen += ms;
    return AI_OK;
  }

  // min/max, bins, 10 costs, decision -> host vectors split / ntrue (raw = 1: ev used as given)
  int sweep(double T, int raw, std::vector<int32_t>& h_split, std::vector<int32_t>& h_ntrue) {
    const int S_ = S();
    AI_HIP(hipEventRecord(ctx->ev[2], st));
    hipLaunchKernelGGL(k_minmax, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, s_mode.p, ev.p, orig, pmm.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_minmax_final, dim3((S_ + 63) / 64), dim3(64), 0, st, task0.p, s_mode.p, pmm.p, S_, raw, s_scale.p, s_nosplit.p, s_thr.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_bin, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, s_scale.p, s_thr.p, ev.p, bin.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_sweep, dim3(ntask), dim3(AI_BLOCK), 0, st, task_seg.p, task_lo.p, task_hi.p, s_nosplit.p, rowptr, col, wraw, deg.p,
                       bin.p, psweep.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_sweep_final, dim3(S_), dim3(64), 0, st, task0.p, s_nosplit.p, psweep.p, T, s_costs.p, s_kstar.p, s_split.p, s_ntrue.p,
                       s_mcut.p);
    AI_KERNEL_CHECK();
    AI_HIP(hipEventRecord(ctx->ev[3], st));
    h_split.resize(S_);
    h_ntrue.resize(S_);
    AI_HIP(hipMemcpyAsync(h_split.data(), s_split.p, (size_t)S_ * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    AI_HIP(hipMemcpyAsync(h_ntrue.data(), s_ntrue.p, (size_t)S_ * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    AI_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    AI_HIP(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
    stats.ms_sweep += ms;
    return AI_OK;
  }
};

static bool eligible(int n, int64_t n_orig, double split_lim) {
  // normalized_cut.py:39-40: W.shape[0] > 2 and len(labels) / (num_points_orig + 1e-8) > split_lim
  return n > 2 && ((double)n / ((double)n_orig + 1e-8)) > split_lim;
}

}  // namespace

// ----------------------------------------------------------------------------- C ABI
static void fill_opts(Solver& S, const ai_ncut_opts* opts) {
  if (!opts) return;
  if (opts->tol > 0.0) S.opt.tol = opts->tol;
  if (opts->max_iter > 0) S.opt.max_iter = opts->max_iter;
  if (opts->check_every > 0) S.opt.check_every = opts->check_every;
  S.time_spmv = (opts->reserved & 1) != 0;
}

extern "C" int ai_ncut(ai_ctx* ctx, const ai_csr* csr, int64_t num_points_orig, double T, double split_lim, const ai_ncut_opts* opts,
                       int32_t* labels_out, int32_t* n_groups, ai_ncut_stats* stats_out) {
  if (!ctx || !csr || !labels_out || !n_groups || num_points_orig < 0) {
    ai_set_error("ai_ncut: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  Solver S(ctx, csr);
  fill_opts(S, opts);
  const int n = (int)csr->n;
  AI_TRY(S.begin(false));
  std::vector<int32_t> leaf_starts;
  if (eligible(n, num_points_orig, split_lim))
    S.segs.push_back(SegHost{0, n, 0, 0});
  else
    leaf_starts.push_back(0);
  hipStream_t st = ctx->stream;
  std::vector<int32_t> h_split, h_ntrue;
  while (S.S() > 0) {
    ++S.stats.levels;
    AI_TRY(S.build_tasks());
    AI_HIP(hipEventRecord(ctx->ev[4], st));
    AI_TRY(S.prepare(true));
    AI_HIP(hipEventRecord(ctx->ev[5], st));
    AI_TRY(S.null_vectors());
    AI_TRY(S.lanczos(nullptr, nullptr, nullptr));
    AI_TRY(S.sweep(T, 0, h_split, h_ntrue));
    // ---- children (deeper calls use split_lim = 0.01: normalized_cut.py:57-58 rely on the default)
    const int S_ = S.S();
    std::vector<SegHost> next;
    std::vector<int32_t> cA(S_, -1), cB(S_, -1);
    int cstart = 0;
    for (int s = 0; s < S_; ++s) {
      const SegHost& sg = S.segs[s];
      if (!h_split[s]) {
        leaf_starts.push_back(sg.gstart);
        continue;
      }
      const int na_ = h_ntrue[s], nb_ = sg.n - h_ntrue[s];
      if (na_ <= 0 || nb_ <= 0) {
        ai_set_error("internal: split of segment %d produced an empty side (%d / %d)", s, na_, nb_);
        return AI_ERR_INTERNAL;
      }
      if (eligible(na_, num_points_orig, 0.01)) {
        cA[s] = cstart;
        next.push_back(SegHost{cstart, na_, sg.gstart, 0});
        cstart += na_;
      } else {
        leaf_starts.push_back(sg.gstart);
      }
      if (eligible(nb_, num_points_orig, 0.01)) {
        cB[s] = cstart;
        next.push_back(SegHost{cstart, nb_, sg.gstart + na_, 0});
        cstart += nb_;
      } else {
        leaf_starts.push_back(sg.gstart + na_);
      }
    }
    AI_HIP(hipEventRecord(ctx->ev[6], st));
    AI_HIP(hipMemcpyAsync(S.s_childA.p, cA.data(), (size_t)S_ * sizeof(int32_t), hipMemcpyHostToDevice, st));
    AI_HIP(hipMemcpyAsync(S.s_childB.p, cB.data(), (size_t)S_ * sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_split_flags, dim3(S.ntask), dim3(AI_BLOCK), 0, st, S.task_seg.p, S.task_lo.p, S.task_hi.p, S.s_split.p, S.s_kstar.p,
                       S.bin.p, S.flag.p);
    AI_KERNEL_CHECK();
    AI_TRY(ai_exclusive_scan_i32(st, S.flag.p, S.fscan.p, S.na, S.scantmp.p));
    const int pp = S.pp;
    AI_TRY(S.b_orig[pp].ensure((size_t)std::max(cstart, 1)));
    hipLaunchKernelGGL(k_partition, dim3(S.ntask), dim3(AI_BLOCK), 0, st, S.task_seg.p, S.task_lo.p, S.task_hi.p, S.seg_start.p, S.s_gstart.p,
                       S.s_split.p, S.s_ntrue.p, S.s_childA.p, S.s_childB.p, S.flag.p, S.fscan.p, S.orig, S.final_order.p, S.map.p, S.b_orig[pp].p);
    AI_KERNEL_CHECK();
    if (cstart > 0) {
      const unsigned ge = (unsigned)(((int64_t)S.na * AI_LPR + AI_BLOCK - 1) / AI_BLOCK);
      AI_HIP(hipMemsetAsync(S.newcnt.p, 0, (size_t)(cstart + 1) * sizeof(int32_t), st));
      hipLaunchKernelGGL(k_rebuild_count, dim3(ge), dim3(AI_BLOCK), 0, st, S.rowptr, S.col, S.flag.p, S.map.p, S.na, S.newcnt.p);
      AI_KERNEL_CHECK();
      AI_TRY(S.b_rowptr[pp].ensure((size_t)cstart + 1));
      AI_TRY(ai_exclusive_scan_i32(st, S.newcnt.p, S.b_rowptr[pp].p, cstart, S.scantmp.p));
      int32_t new_nnz = 0;
      AI_HIP(hipMemcpyAsync(&new_nnz, S.b_rowptr[pp].p + cstart, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      AI_HIP(hipStreamSynchronize(st));
      AI_TRY(S.b_col[pp].ensure((size_t)std::max(new_nnz, 1)));
      AI_TRY(S.b_wraw[pp].ensure((size_t)std::max(new_nnz, 1)));
      hipLaunchKernelGGL(k_rebuild_fill, dim3(ge), dim3(AI_BLOCK), 0, st, S.rowptr, S.col, S.wraw, S.flag.p, S.map.p, S.na,
                         (const int32_t*)S.b_rowptr[pp].p, S.b_col[pp].p, S.b_wraw[pp].p);
      AI_KERNEL_CHECK();
    }
    AI_HIP(hipEventRecord(ctx->ev[7], st));
    AI_HIP(hipStreamSynchronize(st));  // cA / cB are read by the copies above
    float ms1 = 0.f, ms2 = 0.f;
    AI_HIP(hipEventElapsedTime(&ms1, ctx->ev[4], ctx->ev[5]));
    AI_HIP(hipEventElapsedTime(&ms2, ctx->ev[6], ctx->ev[7]));
    S.stats.ms_rebuild += ms1 + ms2;
    S.rowptr = S.b_rowptr[pp].p;
    S.col = S.b_col[pp].p;
    S.wraw = S.b_wraw[pp].p;
    S.orig = S.b_orig[pp].p;
    S.pp ^= 1;
    S.na = cstart;
    S.segs.swap(next);
  }
  // ---- groups = leaf ranges of the final ordering, left to right
  std::vector<int32_t> order((size_t)n);
  AI_HIP(hipMemcpyAsync(order.data(), S.final_order.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  std::sort(leaf_starts.begin(), leaf_starts.end());
  int g = -1;
  size_t li = 0;
  for (int p = 0; p < n; ++p) {
    while (li < leaf_starts.size() && leaf_starts[li] == p) {
      ++g;
      ++li;
    }
    if (g < 0 || order[p] < 0 || order[p] >= n) {
      ai_set_error("internal: final ordering is not a permutation (position %d)", p);
      return AI_ERR_INTERNAL;
    }
    labels_out[order[p]] = g;
  }
  *n_groups = g + 1;
  S.stats.n_groups = g + 1;
  {
    unsigned long long hw[2] = {0, 0};
    AI_HIP(hipMemcpyAsync(hw, S.work.p, sizeof(hw), hipMemcpyDeviceToHost, st));
    AI_HIP(hipStreamSynchronize(st));
    S.stats.spmv_rows = (int64_t)hw[0];
    S.stats.spmv_nnz = (int64_t)hw[1];
  }
  S.stats.ms_total = now_ms() - t0;
  if (stats_out) *stats_out = S.stats;
  return AI_OK;
}

extern "C" int ai_fiedler(ai_ctx* ctx, const ai_csr* csr, const ai_ncut_opts* opts, double* lambda2, double* ev_out, int32_t* iters,
                          double* resid) {
  if (!ctx || !csr || !ev_out) {
    ai_set_error("ai_fiedler: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  Solver S(ctx, csr);
  fill_opts(S, opts);
  const int n = (int)csr->n;
  AI_TRY(S.begin(true));
  AI_TRY(S.build_tasks());
  AI_TRY(S.prepare(true));
  AI_TRY(S.null_vectors());
  double theta = 1.0;
  int it = 0;
  double rs = 0.0;
  AI_TRY(S.lanczos(&theta, &it, &rs));
  hipStream_t st = ctx->stream;
  // unit norm + sign convention, then back to the caller's order
  hipLaunchKernelGGL(k_minmax, dim3(S.ntask), dim3(AI_BLOCK), 0, st, S.task_seg.p, S.task_lo.p, S.task_hi.p, S.s_mode.p, S.ev.p, S.orig, S.pmm.p);
  AI_KERNEL_CHECK();
  hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(64), 0, st, S.task0.p, S.s_mode.p, S.pmm.p, 1, 0, S.s_scale.p, S.s_nosplit.p, S.s_thr.p);
  AI_KERNEL_CHECK();
  double sc = 1.0;
  AI_HIP(hipMemcpyAsync(&sc, S.s_scale.p, sizeof(double), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  hipLaunchKernelGGL(k_scatter_d, dim3((n + AI_BLOCK - 1) / AI_BLOCK), dim3(AI_BLOCK), 0, st, (const double*)S.ev.p, S.orig, n, sc, S.Y.p);
  AI_KERNEL_CHECK();
  AI_HIP(hipMemcpyAsync(ev_out, S.Y.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  if (lambda2) *lambda2 = (S.segs[0].mode == 0) ? 1.0 - theta : 0.0;
  if (iters) *iters = it;
  if (resid) *resid = rs;
  return AI_OK;
}

extern "C" int ai_sweep(ai_ctx* ctx, const ai_csr* csr, const double* ev, double* costs, uint8_t* mask_out, double* mcut) {
  if (!ctx || !csr || !ev || !costs) {
    ai_set_error("ai_sweep: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  Solver S(ctx, csr);
  const int n = (int)csr->n;
  AI_TRY(S.begin(true));
  AI_TRY(S.build_tasks());
  AI_TRY(S.prepare(false));
  hipStream_t st = ctx->stream;
  // caller-order ev -> graph order
  AI_HIP(hipMemcpyAsync(S.Y.p, ev, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_gather_d, dim3((n + AI_BLOCK - 1) / AI_BLOCK), dim3(AI_BLOCK), 0, st, (const double*)S.Y.p, S.orig, n, S.ev.p);
  AI_KERNEL_CHECK();
  std::vector<int32_t> sp, nt;
  AI_TRY(S.sweep(INFINITY, 1, sp, nt));
  AI_HIP(hipMemcpyAsync(costs, S.s_costs.p, AI_NUM_CUTS * sizeof(double), hipMemcpyDeviceToHost, st));
  int32_t ks = 0;
  double mc = 0.0;
  AI_HIP(hipMemcpyAsync(&ks, S.s_kstar.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(&mc, S.s_mcut.p, sizeof(double), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  if (mcut) *mcut = mc;
  if (mask_out) {
    DevBuf<uint8_t> dm;
    AI_TRY(dm.alloc(n));
    if (isinf(mc)) {
      AI_HIP(hipMemsetAsync(dm.p, 0, n, st));
    } else {
      hipLaunchKernelGGL(k_scatter_mask, dim3((n + AI_BLOCK - 1) / AI_BLOCK), dim3(AI_BLOCK), 0, st, (const uint8_t*)S.bin.p, S.orig, n, ks, dm.p);
      AI_KERNEL_CHECK();
    }
    AI_HIP(hipMemcpyAsync(mask_out, dm.p, n, hipMemcpyDeviceToHost, st));
    AI_HIP(hipStreamSynchronize(st));
  }
  return AI_OK;
}

extern "C" int ai_lsym_apply(ai_ctx* ctx, const ai_csr* csr, const double* x, double* y) {
  if (!ctx || !csr || !x || !y) {
    ai_set_error("ai_lsym_apply: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  Solver S(ctx, csr);
  const int n = (int)csr->n;
  AI_TRY(S.begin(true));
  AI_TRY(S.build_tasks());
  AI_TRY(S.prepare(false));
  hipStream_t st = ctx->stream;
  DevBuf<double> xin, yout;
  AI_TRY(xin.alloc(n));
  AI_TRY(yout.alloc(n));
  AI_HIP(hipMemcpyAsync(S.Y.p, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  const unsigned gr = (unsigned)((n + AI_BLOCK - 1) / AI_BLOCK);
  hipLaunchKernelGGL(k_gather_d, dim3(gr), dim3(AI_BLOCK), 0, st, (const double*)S.Y.p, S.orig, n, xin.p);
  AI_KERNEL_CHECK();
  const unsigned ge = (unsigned)(((int64_t)n * AI_LPR + AI_BLOCK - 1) / AI_BLOCK);
  hipLaunchKernelGGL(k_lsym_apply, dim3(ge), dim3(AI_BLOCK), 0, st, S.rowptr, S.col, (const double*)S.wm.p, (const double*)S.sinv2.p,
                     (const double*)xin.p, n, yout.p);
  AI_KERNEL_CHECK();
  hipLaunchKernelGGL(k_scatter_d, dim3(gr), dim3(AI_BLOCK), 0, st, (const double*)yout.p, S.orig, n, 1.0, S.Y.p);
  AI_KERNEL_CHECK();
  AI_HIP(hipMemcpyAsync(y, S.Y.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  return AI_OK;
}

extern "C" int ai_bench_spmv(ai_ctx* ctx, const ai_csr* csr, int32_t reps, double* avg_ms, double* bytes_per_launch) {
  if (!ctx || !csr || reps <= 0 || !avg_ms) {
    ai_set_error("ai_bench_spmv: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  Solver S(ctx, csr);
  AI_TRY(S.begin(true));
  AI_TRY(S.build_tasks());
  AI_TRY(S.prepare(false));
  hipStream_t st = ctx->stream;
  S.slab_stride = (size_t)S.na;
  AI_TRY(S.ensure_vec(0));
  AI_HIP(hipMemsetAsync(S.s_frozen.p, 0, sizeof(int32_t), st));
  AI_HIP(hipMemsetAsync(S.s_g.p, 0, sizeof(double), st));
  const double one = 1.0;
  AI_HIP(hipMemcpyAsync(S.s_rb.p, &one, sizeof(double), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_lz_init, dim3(S.ntask), dim3(AI_BLOCK), 0, st, S.task_seg.p, S.task_lo.p, S.task_hi.p, S.s_mode.p, S.orig, S.u1.p,
                     S.vec(0), S.pB.p);
  AI_KERNEL_CHECK();
  auto launch = [&]() {
    hipLaunchKernelGGL(k_lz_spmv, dim3(S.ntask), dim3(AI_BLOCK), 0, st, S.task_seg.p, S.task_lo.p, S.task_hi.p, S.ntask, S.s_mode.p,
                       S.s_frozen.p, S.s_g.p, S.s_rb.p, S.rowptr, S.col, S.wm.p, S.sinv2.p, S.u1.p, (const double*)S.vec(0), S.Y.p, S.pA.p);
  };
  for (int i = 0; i < 3; ++i) launch();
  AI_KERNEL_CHECK();
  AI_HIP(hipEventRecord(ctx->ev[0], st));
  for (int i = 0; i < reps; ++i) launch();
  AI_HIP(hipEventRecord(ctx->ev[1], st));
  AI_KERNEL_CHECK();
  AI_HIP(hipStreamSynchronize(st));
  float ms = 0.f;
  AI_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
  *avg_ms = (double)ms / reps;
  if (bytes_per_launch) {
    // E (4 B index + 8 B value) + (N + 1) 4 B row pointers +
    // N x 8 B x {R_j read, sinv2, u1, Y written}
    const double N = (double)csr->n, E = (double)csr->nnz;
    *bytes_per_launch = E * 12.0 + (N + 1.0) * 4.0 + N * 8.0 * 4.0;
  }
  return AI_OK;
}
