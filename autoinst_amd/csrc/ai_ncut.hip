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
//   * a DISCONNECTED segment (union-find over the CSR) is split into its connected components in ONE step.
//     That is what the reference's recursion makes of it: eigsh(sigma=1e-10) returns the indicator vector
//     D^1/2 1_C of ONE component there (every component has its own computed "zero" eigenvalue of size ~1e-17
//     and shift-invert resolves them), the sweep cuts exactly that component off at cost 0, and the recursion
//     goes on with the remainder -- components are peeled off one at a time.  (The reference stops peeling when
//     the remainder falls to <= 1 % of the chunk; which components are left in that one remainder is decided by
//     round-off inside SuperLU, so it cannot be reproduced: here every component continues on its own.);
//   * a CONNECTED segment is solved by Lanczos on M = D^-1/2 W D^-1/2 = I - L without
//     re-orthogonalisation, every Lanczos vector kept in HBM, the known top eigenvector
//     u1 = D^1/2 1 / sqrt(vol) projected out of each new vector; the top Ritz pair of T_m is the
//     pair of L's 2nd-smallest eigenvalue.  All segments step together: two launches per step
//     for the whole frontier (fused SpMV; fused alpha / beta / three-term update), per-segment
//     sums by two-stage fixed-order reductions (no float atomics: reproducible run to run).
// tests/gpu_model.py is the NumPy model of this algorithm.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <set>

#include <hip/hip_ext.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "ai_common.h"

#define AI_FINE_ROWS 32      // rows per block in the 16-lanes-per-row kernels (2 rows in flight per lane group; 16 / 128 measured slower, 64 equal in throughput and 7 % slower for one chunk)
#define AI_COARSE_ROWS 512   // rows per block in the thread-per-row kernels (256 / 1024 measured within 2 %)
#define AI_ROW_ILP (AI_FINE_ROWS / (AI_BLOCK / AI_LPR))
#define AI_SLAB_VECS 32      // Lanczos vectors per HBM slab
#define AI_ROW_PF 4         // rounds of 16 entries per row loaded together in the 16-lanes-per-row kernels
#define AI_SWEEP_VALS 40     // per-task sweep partials: cut[10], assocA[10], assocB[10], cntA[10]
#define AI_MAX_CHECKS 4096   // convergence checks per level (one counter slot each)

namespace {

// A task = one block's contiguous row range inside ONE segment: {lo, hi, segment, first task of its segment}.
typedef int4 Task;
// For coarse tasks: the segment's fine-task range [x, y) and coarse-task range [z, w).
typedef int4 TaskRange;

// fixed-order sum of part[t0..t1) by one block; every thread returns the same value
__device__ __forceinline__ double ai_range_sum(const double* __restrict__ part, int t0, int t1, double* sm) {
  double a = 0.0;
  for (int t = t0 + threadIdx.x; t < t1; t += AI_BLOCK) a += part[t];
  return ai_block_sum(a, sm);
}

// ----------------------------------------------------------------------------- degrees, scaling
// deg_i = 1 + sum_j w_ij (W = w + I, normalized_cut.py:38,42); s_i = 1 / sqrt(deg_i) (:43)
__global__ __launch_bounds__(AI_BLOCK) void k_degree(const Task* __restrict__ tasks, const int32_t* __restrict__ rowptr,
                                                     const double* __restrict__ wraw, double* __restrict__ deg,
                                                     double* __restrict__ sinv, double* __restrict__ pvol) {
  __shared__ double sm[AI_BLOCK / 64];
  const Task tk = tasks[blockIdx.x];
  const int l = threadIdx.x & (AI_LPR - 1), r = threadIdx.x / AI_LPR;
  double acc = 0.0;
  for (int row = tk.x + r; row < tk.y; row += AI_BLOCK / AI_LPR) {
    const int p0 = rowptr[row], p1 = rowptr[row + 1];
    double s = 0.0;
    {
      // the first AI_ROW_PF rounds of a row (64 entries: nearly every row) are loaded together, then added in order
      double w[AI_ROW_PF];
#pragma unroll
      for (int q = 0; q < AI_ROW_PF; ++q) {
        const int p = p0 + l + q * AI_LPR;
        w[q] = (p < p1) ? wraw[p] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < AI_ROW_PF; ++q)
        if (p0 + l + q * AI_LPR < p1) s += w[q];
    }
    for (int p = p0 + l + AI_ROW_PF * AI_LPR; p < p1; p += AI_LPR) s += wraw[p];
    s = ai_group16_sum(s);
    if (l == 0) {
      const double d = s + 1.0;
      deg[row] = d;
      sinv[row] = 1.0 / sqrt(d);
      acc += d;
    }
  }
  const double tot = ai_block_sum(acc, sm);
  if (threadIdx.x == 0) pvol[blockIdx.x] = tot;
}

// one block per segment: out[s] = sum of part[task0[s] .. task0[s+1]) in a fixed order
__global__ __launch_bounds__(AI_BLOCK) void k_seg_sum(const int32_t* __restrict__ task0, const double* __restrict__ part,
                                                      double* __restrict__ out) {
  __shared__ double sm[AI_BLOCK / 64];
  const int s = blockIdx.x;
  const double tot = ai_range_sum(part, task0[s], task0[s + 1], sm);
  if (threadIdx.x == 0) out[s] = tot;
}

// wm_ij = (s_i * w_ij) * s_j  (row scaling then column scaling, like D2 * (D - W) * D2, :47);
// sinv2_i = s_i * s_i is the "+ I" term of W; u1_i = sqrt(deg_i / vol_seg)
__global__ __launch_bounds__(AI_BLOCK) void k_scale(const Task* __restrict__ tasks, const int32_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ col, const double* __restrict__ wraw,
                                                    const double* __restrict__ deg, const double* __restrict__ sinv,
                                                    const double* __restrict__ vol, double* __restrict__ wm,
                                                    double* __restrict__ sinv2, double* __restrict__ u1) {
  const Task tk = tasks[blockIdx.x];
  const double v = vol[tk.z];
  const int l = threadIdx.x & (AI_LPR - 1), r = threadIdx.x / AI_LPR;
  for (int row = tk.x + r; row < tk.y; row += AI_BLOCK / AI_LPR) {
    const int p0 = rowptr[row], p1 = rowptr[row + 1];
    const double si = sinv[row];
    {
      int c[AI_ROW_PF];
      double w[AI_ROW_PF], sj[AI_ROW_PF];
#pragma unroll
      for (int q = 0; q < AI_ROW_PF; ++q) {
        const int p = p0 + l + q * AI_LPR;
        const bool ok = p < p1;
        c[q] = ok ? col[p] : -1;
        w[q] = ok ? wraw[p] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < AI_ROW_PF; ++q) sj[q] = (c[q] >= 0) ? sinv[c[q]] : 0.0;
#pragma unroll
      for (int q = 0; q < AI_ROW_PF; ++q)
        if (c[q] >= 0) wm[p0 + l + q * AI_LPR] = (si * w[q]) * sj[q];
    }
    for (int p = p0 + l + AI_ROW_PF * AI_LPR; p < p1; p += AI_LPR) wm[p] = (si * wraw[p]) * sinv[col[p]];
    if (l == 0) {
      sinv2[row] = si * si;
      u1[row] = sqrt(deg[row] / v);
    }
  }
}

// ----------------------------------------------------------------------------- connected components
// Union-find with the smaller root as representative, so a component's label is its first row
// and labels do not depend on scheduling.  Plain loads may be stale inside a launch (a CU's L1 is
// not refreshed by other CUs' stores); that is harmless here: every value ever stored in
// parent[x] is an ancestor of x with a smaller-or-equal id, and hooking is decided by an
// agent-scope compare-and-swap whose failure returns the up-to-date parent.
__device__ __forceinline__ int32_t uf_find(int32_t* parent, int32_t x) {
  int32_t p = parent[x];
  while (p != x) {
    const int32_t gp = parent[p];
    if (gp != p) parent[x] = gp;  // path halving (benign race)
    x = p;
    p = gp;
  }
  return x;
}
__device__ __forceinline__ void uf_unite(int32_t* parent, int32_t a, int32_t b) {
  for (int guard = 0; guard < (1 << 22); ++guard) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) {
      const int32_t t = a;
      a = b;
      b = t;
    }
    const int32_t old = atomicCAS(&parent[a], a, b);  // hook the larger (apparent) root under the smaller id
    if (old == a) return;
    a = old;  // a was no longer a root: continue from its true parent
  }
}

// rows of segments that need a fresh labelling: parent = smallest neighbour id <= row (no cycles:
// strictly decreasing chains); rows of the other segments keep the labels carried over the split
__global__ __launch_bounds__(AI_BLOCK) void k_cc_init(const Task* __restrict__ tasks, const int32_t* __restrict__ need_cc,
                                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                      int32_t* __restrict__ parent) {
  const Task tk = tasks[blockIdx.x];
  if (!need_cc[tk.z]) return;
  const int l = threadIdx.x & (AI_LPR - 1), r = threadIdx.x / AI_LPR;
  for (int row = tk.x + r; row < tk.y; row += AI_BLOCK / AI_LPR) {
    int32_t mn = row;
    for (int p = rowptr[row] + l; p < rowptr[row + 1]; p += AI_LPR) mn = min(mn, col[p]);
    mn = min(mn, __shfl_xor(mn, 8, 16));
    mn = min(mn, __shfl_xor(mn, 4, 16));
    mn = min(mn, __shfl_xor(mn, 2, 16));
    mn = min(mn, __shfl_xor(mn, 1, 16));
    if (l == 0) parent[row] = mn;
  }
}

__global__ __launch_bounds__(AI_BLOCK) void k_cc_hook(const Task* __restrict__ tasks, const int32_t* __restrict__ need_cc,
                                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                      int32_t* parent) {
  const Task tk = tasks[blockIdx.x];
  if (!need_cc[tk.z]) return;
  const int l = threadIdx.x & (AI_LPR - 1), r = threadIdx.x / AI_LPR;
  for (int row = tk.x + r; row < tk.y; row += AI_BLOCK / AI_LPR) {
    for (int p = rowptr[row] + l; p < rowptr[row + 1]; p += AI_LPR) {
      const int32_t c = col[p];
      if (c < row) uf_unite(parent, row, c);
    }
  }
}

__global__ __launch_bounds__(AI_BLOCK) void k_cc_compress(const Task* __restrict__ tasks, const int32_t* __restrict__ need_cc,
                                                          int32_t* parent) {
  const Task tk = tasks[blockIdx.x];
  if (!need_cc[tk.z]) return;
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) {
    int32_t x = row;
    // chains are static in this launch (no hooking); stale values are still ancestors
    for (;;) {
      const int32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p == x) break;
      x = p;
    }
    parent[row] = x;
  }
}

// roots per segment (integer atomics: order independent)
__global__ __launch_bounds__(AI_BLOCK) void k_cc_count(const Task* __restrict__ tasks, const int32_t* __restrict__ parent,
                                                       int32_t* __restrict__ ncomp) {
  const Task tk = tasks[blockIdx.x];
  int c = 0;
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) c += (parent[row] == row);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&ncomp[tk.z], c);
}

// ----------------------------------------------------------------------------- null-space vector
// rcnt[root] = rows of the component; runs of equal roots inside a wave are added once
__global__ __launch_bounds__(AI_BLOCK) void k_null_rootcount(const Task* __restrict__ tasks, const int32_t* __restrict__ mode,
                                                             const int32_t* __restrict__ parent, int32_t* __restrict__ rcnt) {
  const Task tk = tasks[blockIdx.x];
  if (mode[tk.z] != 1) return;
  const int lane = threadIdx.x & 63;
  for (int base = tk.x; base < tk.y; base += AI_BLOCK) {
    const int row = base + threadIdx.x;
    const bool live = row < tk.y;
    const int32_t r = live ? parent[row] : -1;
    const int32_t prev = __shfl_up(r, 1, 64);
    const bool head = live && (lane == 0 || prev != r);
    const unsigned long long heads = __ballot(head);
    const unsigned long long lives = __ballot(live);
    if (head) {
      // run length = distance to the next head (or to the end of the live lanes)
      const unsigned long long later = (lane == 63) ? 0ull : (heads >> (lane + 1));
      const int nlive = __popcll(lives);
      const int len = later ? (__ffsll((long long)later)) : (nlive - lane);
      atomicAdd(&rcnt[r], len);
    }
  }
}

// rc[row] = size of the component if row is its root, else 0 (scanned to rank components by first row)
__global__ __launch_bounds__(AI_BLOCK) void k_null_rootvals(const int32_t* __restrict__ parent, const int32_t* __restrict__ rcnt,
                                                            int32_t n, int32_t* __restrict__ rc) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i < n) rc[i] = (parent[i] == i) ? rcnt[i] : 0;
}

// side A = first component + every later component that still ends within the first half of
// the segment's rows; partial volumes of both sides
__global__ __launch_bounds__(AI_BLOCK) void k_null_side(const Task* __restrict__ tasks, const int32_t* __restrict__ mode,
                                                        const int32_t* __restrict__ seg_start, const int32_t* __restrict__ parent,
                                                        const int32_t* __restrict__ rcnt, const int32_t* __restrict__ ex,
                                                        const double* __restrict__ deg, uint8_t* __restrict__ side,
                                                        double* __restrict__ pvolA, double* __restrict__ pvolB) {
  __shared__ double sm[AI_BLOCK / 64];
  const Task tk = tasks[blockIdx.x];
  const int s = tk.z;
  if (mode[s] != 1) {
    if (threadIdx.x == 0) {
      pvolA[blockIdx.x] = 0.0;
      pvolB[blockIdx.x] = 0.0;
    }
    return;
  }
  const int s0 = seg_start[s], ns = seg_start[s + 1] - s0;
  const int exs = ex[s0];
  double va = 0.0, vb = 0.0;
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) {
    const int r = parent[row];
    const int cb = ex[r] - exs;
    const bool inA = (r == s0) || (2 * (int64_t)(cb + rcnt[r]) <= (int64_t)ns);
    side[row] = inA ? 1 : 0;
    if (inA) va += deg[row]; else vb += deg[row];
  }
  const double ta = ai_block_sum(va, sm);
  const double tb = ai_block_sum(vb, sm);
  if (threadIdx.x == 0) {
    pvolA[blockIdx.x] = ta;
    pvolB[blockIdx.x] = tb;
  }
}

__global__ __launch_bounds__(AI_BLOCK) void k_null_vec(const Task* __restrict__ tasks, const int32_t* __restrict__ mode,
                                                       const double* __restrict__ volA, const double* __restrict__ volB,
                                                       const double* __restrict__ deg, const uint8_t* __restrict__ side,
                                                       double* __restrict__ ev) {
  const Task tk = tasks[blockIdx.x];
  const int s = tk.z;
  if (mode[s] != 1) return;
  const double ia = 1.0 / volA[s], ib = 1.0 / volB[s];
  const double rn = 1.0 / sqrt(ia + ib);  // ||D^1/2 (1_A/volA - 1_B/volB)||^2 = 1/volA + 1/volB
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) ev[row] = sqrt(deg[row]) * (side[row] ? ia : -ib) * rn;
}

// ----------------------------------------------------------------------------- Lanczos
// Notation: R_j is the stored (unnormalised, unprojected) j-th vector, g_j = u1 . R_j,
// b_j = ||R_j - g_j u1||, v_j = (R_j - g_j u1) / b_j the Lanczos vector.  T has diagonal
// alpha_j = v_j . M v_j and off-diagonals b_1, b_2, ...

// R_0 = hash(original id); partials of (R.R, u1.R)
__global__ __launch_bounds__(AI_BLOCK) void k_lz_init(const Task* __restrict__ ctasks, const int32_t* __restrict__ cactive,
                                                      const int32_t* __restrict__ orig, const double* __restrict__ u1,
                                                      double* __restrict__ R0, double2* __restrict__ pB) {
  __shared__ double sm[AI_BLOCK / 64];
  if (!cactive[blockIdx.x]) return;
  const Task tk = ctasks[blockIdx.x];
  double nn = 0.0, gg = 0.0;
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) {
    const double r = ai_hash_unit((uint32_t)orig[row]);
    R0[row] = r;
    nn = fma(r, r, nn);
    gg = fma(u1[row], r, gg);
  }
  const double tn = ai_block_sum(nn, sm);
  const double tg = ai_block_sum(gg, sm);
  if (threadIdx.x == 0) pB[blockIdx.x] = make_double2(tn, tg);
}

// THE hot kernel: z = M R_j for every running segment,
//   z_i = sum_k wm_ik R_j[k] + sinv2_i R_j[i]            (M = D^-1/2 (w + I) D^-1/2)
// and the per-block partial of R_j . z.  No per-segment scalar is needed here: with v = (R - g u1)/b
// and M u1 = u1,  alpha = v . M v = (R.z - g^2) / b^2  -- the update kernel finishes it.
// 16 lanes per row (neighbour counts ~30-40) read 64-B / 128-B runs of col / wm; each lane group
// keeps AI_ROW_ILP rows in flight so that the dependent chain rowptr -> col -> gather is overlapped
// four deep; gathers of R_j are served by L2 / MALL.
template <int LPR, int ILP, bool NOGATHER = false>
__device__ __forceinline__ void spmv_body(int t, const Task* __restrict__ ftasks, const int32_t* __restrict__ factive,
                                          const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                          const double* __restrict__ wm, const double* __restrict__ sinv2,
                                          const double* __restrict__ Rj, double* __restrict__ Z, double* __restrict__ pA,
                                          double* sm) {
  const int act = factive[t];
  const Task tk = ftasks[t];  // independent of the flag: both loads are in flight together
  if (!act) return;
  const int l = threadIdx.x & (LPR - 1), r = threadIdx.x / LPR;
  constexpr int GROUPS = AI_BLOCK / LPR;
  int p0[ILP], p1[ILP];
  double sum[ILP], ri[ILP], s2[ILP];
  int len = 0;
#pragma unroll
  for (int u = 0; u < ILP; ++u) {
    const int row = tk.x + r + u * GROUPS;
    const bool ok = row < tk.y;
    p0[u] = ok ? rowptr[row] : 0;
    p1[u] = ok ? rowptr[row + 1] : 0;
    // the diagonal "+ I" term needs only the row id: issue these loads before the dependent chain
    ri[u] = (ok && l == 0) ? Rj[row] : 0.0;
    s2[u] = (ok && l == 0) ? sinv2[row] : 0.0;
    sum[u] = 0.0;
  }
#pragma unroll
  for (int u = 0; u < ILP; ++u) len = max(len, p1[u] - p0[u]);
  for (int k = l; k < len; k += LPR) {
    int c[ILP];
    double w[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      const int p = p0[u] + k;
      const bool ok = p < p1[u];
      c[u] = ok ? col[p] : -1;
      w[u] = ok ? wm[p] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < ILP; ++u)
      if (c[u] >= 0) sum[u] = fma(w[u], NOGATHER ? (double)c[u] : Rj[c[u]], sum[u]);
  }
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < ILP; ++u) {
    double sg = sum[u];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sg += __shfl_xor(sg, o, LPR);
    const int row = tk.x + r + u * GROUPS;
    if (l == 0 && row < tk.y) {
      const double z = fma(s2[u], ri[u], sg);
      Z[row] = z;
      acc = fma(ri[u], z, acc);
    }
  }
  const double tot = ai_block_sum(acc, sm);
  if (threadIdx.x == 0) pA[t] = tot;
}

template <int LPR, int ILP, bool NOGATHER = false>
__global__ __launch_bounds__(AI_BLOCK) void k_lz_spmv_t(const Task* __restrict__ ftasks, const int32_t* __restrict__ factive, int ntask,
                                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                      const double* __restrict__ wm, const double* __restrict__ sinv2,
                                                      const double* __restrict__ Rj, double* __restrict__ Z,
                                                      double* __restrict__ pA) {
  __shared__ double sm[AI_BLOCK / 64];
  spmv_body<LPR, ILP, NOGATHER>(ai_xcd_task(blockIdx.x, ntask), ftasks, factive, rowptr, col, wm, sinv2, Rj, Z, pA, sm);
}

// ---- staged gather (the default SpMV): the 8-byte gathers of R_j, not the streamed bytes, bound the
// plain kernel.  The ~1200 entries of a 32-row task touch only ~150-300 DISTINCT
// columns (consecutive Morton-ordered rows share their 27 neighbour cells), so once per level every
// Lanczos task gets its sorted distinct-column list `ucol` and every entry a 16-bit index into it;
// a step then gathers each distinct R_j value ONCE into LDS (coalesced runs) and the per-entry
// gather becomes an LDS read.  Sums are formed in exactly the order of the plain kernel.
#define AI_ENC_MAXNNZ 4096  // entries of a task the encoder sorts in LDS
#define AI_ENC_XCAP 1024    // distinct columns of a task staged in LDS (8 KB)
#define AI_SPMV_PF 4         // rounds of 16 entries per row whose loads are in flight before the barrier
struct TaskEnc {
  int32_t uoff, ucnt;  // slice of the ucol pool; ucnt < 0: not encoded, the task gathers from global memory
};

__global__ __launch_bounds__(AI_BLOCK) void k_lz_encode(const Task* __restrict__ ftasks, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, int32_t pool_cap,
                                                        int32_t* __restrict__ pool_ctr, int32_t* __restrict__ ucol,
                                                        uint16_t* __restrict__ lidx, TaskEnc* __restrict__ enc) {
  __shared__ int32_t sk[AI_ENC_MAXNNZ];
  __shared__ int32_t uq[AI_ENC_XCAP];
  __shared__ int32_t wsum[AI_BLOCK / 64];
  __shared__ int32_t s_off;
  const Task tk = ftasks[blockIdx.x];
  const int p0 = rowptr[tk.x], cnt = rowptr[tk.y] - p0;
  if (cnt > AI_ENC_MAXNNZ || cnt <= 0) {
    if (threadIdx.x == 0) enc[blockIdx.x] = TaskEnc{0, -1};
    return;
  }
  int n = 64;
  while (n < cnt) n <<= 1;
  for (int i = threadIdx.x; i < n; i += AI_BLOCK) sk[i] = (i < cnt) ? col[p0 + i] : 0x7fffffff;
  __syncthreads();
  // bitonic sort of n (a power of two) keys in LDS
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < (n >> 1); i += AI_BLOCK) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
        const int32_t a = sk[lo], b = sk[hi];
        const bool up = (lo & k) == 0;
        if ((a > b) == up) {
          sk[lo] = b;
          sk[hi] = a;
        }
      }
      __syncthreads();
    }
  // distinct values: thread t owns the sorted positions [t * per, (t + 1) * per)
  const int per = n / AI_BLOCK > 0 ? n / AI_BLOCK : 1;
  int heads = 0;
  for (int q = 0; q < per; ++q) {
    const int i = threadIdx.x * per + q;
    if (i < cnt && (i == 0 || sk[i] != sk[i - 1])) ++heads;
  }
  int incl = heads;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if ((threadIdx.x & 63) >= o) incl += v;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int base = incl - heads, total = 0;
#pragma unroll
  for (int w = 0; w < AI_BLOCK / 64; ++w) {
    if (w < (int)(threadIdx.x >> 6)) base += wsum[w];
    total += wsum[w];
  }
  if (total > AI_ENC_XCAP) {
    if (threadIdx.x == 0) enc[blockIdx.x] = TaskEnc{0, -1};
    return;
  }
  for (int q = 0; q < per; ++q) {
    const int i = threadIdx.x * per + q;
    if (i < cnt && (i == 0 || sk[i] != sk[i - 1])) uq[base++] = sk[i];
  }
  if (threadIdx.x == 0) s_off = atomicAdd(pool_ctr, total);
  __syncthreads();
  const int off = s_off;
  if (off + total > pool_cap) {  // pool exhausted (never with the default sizing): plain gather for this task
    if (threadIdx.x == 0) enc[blockIdx.x] = TaskEnc{0, -1};
    return;
  }
  for (int i = threadIdx.x; i < total; i += AI_BLOCK) ucol[off + i] = uq[i];
  for (int e = threadIdx.x; e < cnt; e += AI_BLOCK) {
    const int32_t c = col[p0 + e];
    int lo = 0, hi = total - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (uq[mid] < c) lo = mid + 1; else hi = mid;
    }
    lidx[p0 + e] = (uint16_t)lo;
  }
  if (threadIdx.x == 0) enc[blockIdx.x] = TaskEnc{off, total};
}

template <int LPR, int ILP>
__global__ __launch_bounds__(AI_BLOCK) void k_lz_spmv_x(const Task* __restrict__ ftasks, const int32_t* __restrict__ factive, int ntask,
                                                        const TaskEnc* __restrict__ enc, const int32_t* __restrict__ ucol,
                                                        const uint16_t* __restrict__ lidx, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, const double* __restrict__ wm,
                                                        const double* __restrict__ sinv2, const double* __restrict__ Rj,
                                                        double* __restrict__ Z, double* __restrict__ pA,
                                                        unsigned long long* __restrict__ tstamp) {
  __shared__ double sm[AI_BLOCK / 64];
  __shared__ double xs[AI_ENC_XCAP];
  // profiling only (opts.reserved bit 1): every block stores when it started and ended on the device clock (plain
  // stores to its own slot: atomics on one address would serialise the blocks); k_ts_reduce takes min / max
  struct Stamp {
    unsigned long long* ts;
    __device__ explicit Stamp(unsigned long long* p) : ts(p) {
      if (ts && threadIdx.x == 0) ts[2 * blockIdx.x] = (unsigned long long)wall_clock64();
    }
    __device__ ~Stamp() {
      if (ts && threadIdx.x == 0) ts[2 * blockIdx.x + 1] = (unsigned long long)wall_clock64();
    }
  } stamp(tstamp);
  const int t = ai_xcd_task(blockIdx.x, ntask);
  const int act = factive[t];
  const Task tk = ftasks[t];
  const TaskEnc en = enc[t];
  if (!act) return;
  if (en.ucnt < 0) {
    spmv_body<LPR, ILP>(t, ftasks, factive, rowptr, col, wm, sinv2, Rj, Z, pA, sm);
    return;
  }
  for (int i = threadIdx.x; i < en.ucnt; i += AI_BLOCK) xs[i] = Rj[ucol[en.uoff + i]];
  const int l = threadIdx.x & (LPR - 1), r = threadIdx.x / LPR;
  constexpr int GROUPS = AI_BLOCK / LPR;
  int p0[ILP], p1[ILP];
  double sum[ILP], ri[ILP], s2[ILP];
  int len = 0;
#pragma unroll
  for (int u = 0; u < ILP; ++u) {
    const int row = tk.x + r + u * GROUPS;
    const bool ok = row < tk.y;
    p0[u] = ok ? rowptr[row] : 0;
    p1[u] = ok ? rowptr[row + 1] : 0;
    ri[u] = (ok && l == 0) ? Rj[row] : 0.0;
    s2[u] = (ok && l == 0) ? sinv2[row] : 0.0;
    sum[u] = 0.0;
  }
#pragma unroll
  for (int u = 0; u < ILP; ++u) len = max(len, p1[u] - p0[u]);
  // the streamed loads do not depend on xs: the first AI_SPMV_PF rounds (48 entries of a row: most rows
  // end there) are issued before the barrier, so that a block's dependent chain is
  // task -> (distinct columns | row pointers) -> (R_j gather | entries) -> sums
  constexpr int PF = AI_SPMV_PF;
  int cq[PF][ILP];
  double wq[PF][ILP];
#pragma unroll
  for (int q = 0; q < PF; ++q)
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      const int p = p0[u] + l + q * LPR;
      const bool ok = p < p1[u];
      cq[q][u] = ok ? (int)lidx[p] : -1;
      wq[q][u] = ok ? wm[p] : 0.0;
    }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PF; ++q)
#pragma unroll
    for (int u = 0; u < ILP; ++u)
      if (cq[q][u] >= 0) sum[u] = fma(wq[q][u], xs[cq[q][u]], sum[u]);
  for (int k = l + PF * LPR; k < len; k += LPR) {  // long rows: one more round at a time
#pragma unroll
    for (int u = 0; u < ILP; ++u) {
      const int p = p0[u] + k;
      if (p < p1[u]) sum[u] = fma(wm[p], xs[lidx[p]], sum[u]);
    }
  }
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < ILP; ++u) {
    double sg = sum[u];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sg += __shfl_xor(sg, o, LPR);
    const int row = tk.x + r + u * GROUPS;
    if (l == 0 && row < tk.y) {
      const double z = fma(s2[u], ri[u], sg);
      Z[row] = z;
      acc = fma(ri[u], z, acc);
    }
  }
  const double tot = ai_block_sum(acc, sm);
  if (threadIdx.x == 0) pA[t] = tot;
}

// span[0] = earliest block start, span[1] = latest block end of one launch (profiling only)
__global__ __launch_bounds__(AI_BLOCK) void k_ts_reduce(const unsigned long long* __restrict__ ts, int nblk, unsigned long long* __restrict__ span) {
  __shared__ unsigned long long smn[AI_BLOCK], smx[AI_BLOCK];
  unsigned long long mn = ~0ull, mx = 0ull;
  for (int b = threadIdx.x; b < nblk; b += AI_BLOCK) {
    mn = min(mn, ts[2 * b]);
    mx = max(mx, ts[2 * b + 1]);
  }
  smn[threadIdx.x] = mn;
  smx[threadIdx.x] = mx;
  __syncthreads();
  for (int o = AI_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      smn[threadIdx.x] = min(smn[threadIdx.x], smn[threadIdx.x + o]);
      smx[threadIdx.x] = max(smx[threadIdx.x], smx[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    span[0] = smn[0];
    span[1] = smx[0];
  }
}

struct LzSeg {
  int32_t* frozen;      // [S]
  int32_t* m;           // [S] size of T at freeze
  double* alpha_hist;   // [S][mcap]
  double* b_hist;       // [S][mcap + 1]
  double* g_hist;       // [S][mcap + 1]
  int32_t* factive;     // fine-task activity flags
  int32_t* cactive;     // coarse-task activity flags
  int mcap;
};

__device__ __forceinline__ void lz_freeze(const LzSeg& L, int s, int m, TaskRange rg, int lane, int nlanes) {
  if (lane == 0) {
    L.frozen[s] = 1;
    L.m[s] = m;
  }
  for (int t = rg.x + lane; t < rg.y; t += nlanes) L.factive[t] = 0;
  for (int t = rg.z + lane; t < rg.w; t += nlanes) L.cactive[t] = 0;
}

// Step j, second launch: every block of a segment re-derives the segment's scalars from the
// per-block partials in the same fixed order (so all blocks agree bit for bit):
//   g_j, b_j from the partials of R_j;  alpha_j = (R_j . z - g_j^2) / b_j^2;
// then  R_{j+1} = y - alpha_j v_j - b_j v_{j-1},  y = (z - g_j u1) / b_j,
// and the partials (R.R, u1.R) of R_{j+1}.  The segment's first block records alpha_j, b_j, g_j.
// A vanishing b_j (Krylov space exhausted) freezes the segment with T of size j.
__device__ __forceinline__ void update_body(int bid, const Task* __restrict__ ctasks, const TaskRange* __restrict__ cranges,
                                            const LzSeg& L, int j, const double* __restrict__ pA,
                                            const double2* __restrict__ pBcur, double2* __restrict__ pBnext,
                                            const double* __restrict__ u1, const double* __restrict__ Z,
                                            const double* __restrict__ Rj, const double* __restrict__ Rjm1,
                                            double* __restrict__ Rnext, double (*sm3)[AI_BLOCK / 64]) {
  constexpr int RPT = AI_COARSE_ROWS / AI_BLOCK;  // rows per thread
  const int act = L.cactive[bid];
  const Task tk = ctasks[bid];
  const TaskRange rg = cranges[bid];
  if (!act) return;
  const int s = tk.z;
  if (j >= (tk.w >> 1)) return;  // the segment's Krylov space is exhausted at its own dimension
  // row data does not depend on the segment scalars: get it moving before the reductions
  double zr[RPT], rr[RPT], rm[RPT], ur[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int row = tk.x + threadIdx.x + q * AI_BLOCK;
    const bool ok = row < tk.y;
    zr[q] = ok ? Z[row] : 0.0;
    rr[q] = ok ? Rj[row] : 0.0;
    rm[q] = (ok && j > 0) ? Rjm1[row] : 0.0;
    ur[q] = ok ? u1[row] : 0.0;
  }
  // previous step's scalars come from the history (written by an earlier launch)
  double gp = 0.0, rbp = 0.0;
  if (j > 0) {
    gp = L.g_hist[(size_t)s * (L.mcap + 1) + j - 1];
    rbp = 1.0 / L.b_hist[(size_t)s * (L.mcap + 1) + j - 1];
  }
  double a = 0.0, nn = 0.0, gg = 0.0;
  for (int t = rg.x + threadIdx.x; t < rg.y; t += AI_BLOCK) a += pA[t];
  for (int t = rg.z + threadIdx.x; t < rg.w; t += AI_BLOCK) {
    const double2 v = pBcur[t];
    nn += v.x;
    gg += v.y;
  }
  // three block sums behind one pair of barriers; every thread adds the wave partials in the same order
  a = ai_wave_sum(a);
  nn = ai_wave_sum(nn);
  gg = ai_wave_sum(gg);
  const int w = threadIdx.x >> 6, ln = threadIdx.x & 63;
  if (ln == 0) {
    sm3[0][w] = a;
    sm3[1][w] = nn;
    sm3[2][w] = gg;
  }
  __syncthreads();
  a = 0.0;
  nn = 0.0;
  double g = 0.0;
#pragma unroll
  for (int i = 0; i < AI_BLOCK / 64; ++i) {
    a += sm3[0][i];
    nn += sm3[1][i];
    g += sm3[2][i];
  }
  __syncthreads();
  const double b = sqrt(fmax(nn - g * g, 0.0));  // ||R_j - g u1||, u1 has unit norm
  if (j > 0 && !(b > 1e-14)) {
    if (tk.w & 1) lz_freeze(L, s, j, rg, threadIdx.x, AI_BLOCK);
    return;
  }
  const double rb = 1.0 / b;
  const double al = rb * rb * (a - g * g);
  if ((tk.w & 1) && threadIdx.x == 0) {
    L.alpha_hist[(size_t)s * L.mcap + j] = al;
    L.b_hist[(size_t)s * (L.mcap + 1) + j] = b;
    L.g_hist[(size_t)s * (L.mcap + 1) + j] = g;
  }
  double n2 = 0.0, g2 = 0.0;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int row = tk.x + threadIdx.x + q * AI_BLOCK;
    if (row < tk.y) {
      const double ui = ur[q];
      const double y = (zr[q] - g * ui) * rb;
      const double v = (rr[q] - g * ui) * rb;
      const double vm = (j > 0) ? (rm[q] - gp * ui) * rbp : 0.0;
      const double rnew = y - al * v - b * vm;
      Rnext[row] = rnew;
      n2 = fma(rnew, rnew, n2);
      g2 = fma(ui, rnew, g2);
    }
  }
  n2 = ai_wave_sum(n2);
  g2 = ai_wave_sum(g2);
  if (ln == 0) {
    sm3[0][w] = n2;
    sm3[1][w] = g2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tn = 0.0, tg = 0.0;
#pragma unroll
    for (int i = 0; i < AI_BLOCK / 64; ++i) {
      tn += sm3[0][i];
      tg += sm3[1][i];
    }
    pBnext[bid] = make_double2(tn, tg);
  }
}

__global__ __launch_bounds__(AI_BLOCK) void k_lz_update(const Task* __restrict__ ctasks, const TaskRange* __restrict__ cranges,
                                                        LzSeg L, int j, const double* __restrict__ pA,
                                                        const double2* __restrict__ pBcur, double2* __restrict__ pBnext,
                                                        const double* __restrict__ u1, const double* __restrict__ Z,
                                                        const double* __restrict__ Rj, const double* __restrict__ Rjm1,
                                                        double* __restrict__ Rnext) {
  __shared__ double sm3[3][AI_BLOCK / 64];
  update_body(blockIdx.x, ctasks, cranges, L, j, pA, pBcur, pBnext, u1, Z, Rj, Rjm1, Rnext, sm3);
}

// Number of eigenvalues of T_m (diag a[0..m), squared off-diagonals bb[1..m), both in LDS) that
// are < x, by sign changes of the leading principal minors p_i = det(T_i - x I), rescaled by
// powers of two.  The LDS reads do not depend on the recurrence, so they pipeline.
__device__ __forceinline__ int sturm_lt(const double* a, const double* bb, int m, double x) {
  // one wave per SIMD: the recurrence is bound by instruction issue, so magnitudes are looked at every
  // 8 rows only (|a - x| + b^2 < 4: eight rows move them by < 2^16; the rescale leaves 200 decades)
  double pm = 1.0, p = a[0] - x;
  int cnt = (p < 0.0) ? 1 : 0;
  for (int i0 = 1; i0 < m; i0 += 8) {
    double av[8], bv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = min(i0 + t, m - 1);
      av[t] = a[i];
      bv[t] = bb[i];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (i0 + t < m) {
        double pn = (av[t] - x) * p - bv[t] * pm;
        if (pn == 0.0) pn = (p > 0.0) ? -1e-300 : 1e-300;  // a zero takes the sign opposite to its predecessor
        cnt += ((pn < 0.0) != (p < 0.0)) ? 1 : 0;
        pm = p;
        p = pn;
      }
    }
    const double ap = fabs(p);
    if (ap > 1e100 || ap < 1e-100) {
      const double sc = (ap > 1e100) ? 0x1p-400 : 0x1p400;
      p *= sc;
      pm *= sc;
    }
  }
  return cnt;
}

// b_m = ||R_m - g_m u1|| per running segment from the update kernel's partials (main stream: the
// partials are overwritten two steps later, the check itself runs on the side stream)
__global__ __launch_bounds__(64) void k_lz_bnew(const TaskRange* __restrict__ seg_range, const int32_t* __restrict__ mode,
                                                const int32_t* __restrict__ frozen, const double2* __restrict__ pB,
                                                double* __restrict__ bnew_out) {
  const int s = blockIdx.x;
  if (mode[s] != 0 || frozen[s]) return;
  const TaskRange rg = seg_range[s];
  double nn = 0.0, gg = 0.0;
  for (int t = rg.z + threadIdx.x; t < rg.w; t += 64) {
    const double2 v = pB[t];
    nn += v.x;
    gg += v.y;
  }
  nn = ai_wave_sum(nn);
  gg = ai_wave_sum(gg);
  if (threadIdx.x == 0) bnew_out[s] = sqrt(fmax(nn - gg * gg, 0.0));
}

// Convergence check after step j (m = j + 1 rows of T), one 256-thread block per running segment, on
// the side stream: top eigenvalue of T_m by 256-way multisection (every thread one Sturm count, the
// four waves on the CU's four SIMDs); |s_m| by the recurrence from the bottom row upwards (the growing,
// hence stable, direction); residual = b_m |s_m|.  Freezes the segment when residual <= tol or T has
// reached the segment's dimension / the step cap.  slot[0] counts the segments still running; work[]
// accumulates rows and stored entries the SpMV kernel processed since the last check.  Reads only
// history entries [0, m), which later steps never touch.
#define AI_CHECK_THREADS 256
__global__ __launch_bounds__(AI_CHECK_THREADS) void k_lz_check(const int32_t* __restrict__ seg_start, const TaskRange* __restrict__ seg_range,
                                                               const int32_t* __restrict__ mode, LzSeg L, const double* __restrict__ bnew_in,
                                                               int m, double tol, int max_iter, int steps_since,
                                                               const int32_t* __restrict__ rowptr, double* __restrict__ theta_out,
                                                               double* __restrict__ resid_out, int32_t* __restrict__ slot,
                                                               unsigned long long* __restrict__ work, int with_rb) {
  constexpr int NT = AI_CHECK_THREADS, NW = NT / 64;
  __shared__ double smm[2][NW];
  __shared__ int sfirst[2][NW];
  const int s = blockIdx.x;
  if (mode[s] != 0) return;
  if (L.frozen[s]) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ns = seg_start[s + 1] - seg_start[s];
  const TaskRange rg = seg_range[s];
  if (tid == 0) {
    atomicAdd(&work[0], (unsigned long long)ns * (unsigned long long)steps_since);
    atomicAdd(&work[1], (unsigned long long)(rowptr[seg_start[s + 1]] - rowptr[seg_start[s]]) * (unsigned long long)steps_since);
  }
  const double bnew = bnew_in[s];
  const double* bh = L.b_hist + (size_t)s * (L.mcap + 1);
  const double* ah = L.alpha_hist + (size_t)s * L.mcap;
  const int cap = min(min(ns - 1, max_iter), L.mcap);
  const bool last = (m >= cap) || !(bnew > 1e-14);
  m = min(m, cap);  // steps launched past the segment's cap did nothing (graph replay works in batches)
  // T_m into LDS: a[0..m), b^2[0..m) (b^2[0] unused) and, when it fits, 1 / b[0..m)
  extern __shared__ double lds[];
  double* la = lds;
  double* lbb = lds + m;
  double* lrb = lds + 2 * m;
  double lo = -1e300, hi = -1e300;  // lambda_max >= max diagonal, <= Gershgorin bound
  for (int i = tid; i < m; i += NT) {
    const double ai = ah[i];
    const double bi = (i > 0) ? bh[i] : 0.0, br = (i + 1 < m) ? bh[i + 1] : 0.0;
    la[i] = ai;
    lbb[i] = bi * bi;
    if (with_rb) lrb[i] = (i > 0) ? 1.0 / bi : 0.0;
    lo = fmax(lo, ai);
    hi = fmax(hi, ai + fabs(bi) + fabs(br));
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmax(lo, __shfl_xor(lo, o, 64));
    hi = fmax(hi, __shfl_xor(hi, o, 64));
  }
  if (lane == 0) {
    smm[0][wv] = lo;
    smm[1][wv] = hi;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    lo = fmax(lo, smm[0][w]);
    hi = fmax(hi, smm[1][w]);
  }
  lo -= 1e-14 * fmax(fabs(lo), 1.0);
  hi += 1e-14 * fmax(fabs(hi), 1.0);
  // index of the first thread of the block whose shift exceeds every eigenvalue (Sturm count m), or -1
  int phase = 0;
  auto first_above = [&](bool mine) -> int {
    const unsigned long long above = __ballot(mine);
    if (lane == 0) sfirst[phase][wv] = above ? wv * 64 + __ffsll((long long)above) - 1 : (1 << 30);
    __syncthreads();
    int j = 1 << 30;
#pragma unroll
    for (int w = 0; w < NW; ++w) j = min(j, sfirst[phase][w]);
    phase ^= 1;  // the next call writes the other row: one barrier per call is enough
    return (j == (1 << 30)) ? -1 : j;
  };
  // Warm start from the previous check of this solve: T grew by rows and columns, so its top
  // eigenvalue did not decrease (interlacing), and it normally lies within the previous residual of
  // the previous value.  One round probes the ladder theta' + resid' * 2^j; the lowest rung above
  // every eigenvalue closes the bracket, the rung below it opens it.
  const double tprev = theta_out[s], rprev = resid_out[s];
  if (tprev > 0.0 && rprev > 0.0) {
    const double hi_g = hi, eps = 1e-15 * fmax(fabs(tprev), 1.0);
    const double lo_w = fmax(lo, tprev - 1e-14 * fmax(fabs(tprev), 1.0));
    const double x = fmin(hi_g, tprev + ldexp(rprev, tid) + eps);
    const int j0 = first_above(sturm_lt(la, lbb, m, x) == m);
    lo = lo_w;
    if (j0 >= 0) {
      hi = fmin(hi_g, tprev + ldexp(rprev, j0) + eps);
      if (j0 > 0) lo = fmax(lo_w, fmin(hi_g, tprev + ldexp(rprev, j0 - 1) + eps));
    }
  }
  for (int round = 0; round < 12; ++round) {
    // 5e-14 relative is enough: the residual estimate below needs the eigenvalue only to a small
    // fraction of T's top gap, and the host refines it inside a +-1e-13 bracket (tridiag_top)
    if (hi - lo <= 5e-14 * fmax(fabs(hi), 1e-300)) break;  // block-uniform
    const double w = (hi - lo) * (1.0 / (NT + 1));
    const double x = lo + (tid + 1) * w;
    const int j0 = first_above(sturm_lt(la, lbb, m, x) == m);
    if (j0 < 0) {
      lo = lo + (double)NT * w;
    } else {
      hi = lo + (j0 + 1) * w;
      lo = lo + j0 * w;
    }
  }
  if (wv != 0) return;
  const double theta = 0.5 * (lo + hi);
  // ---- |s_m| / ||s||: s_m = 1 at the start; `scale` follows the rescalings
  double sk1 = 0.0, sk = 1.0, sumsq = 1.0, scale = 1.0;  // s_{k+1}, s_k (every lane computes the same)
#pragma unroll 8
  for (int k = m - 1; k >= 1; --k) {
    const double bu = (k + 1 < m) ? bh[k + 1] : 0.0;
    const double num = (theta - la[k]) * sk - bu * sk1;
    const double sm1 = with_rb ? num * lrb[k] : num / bh[k];
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
  if (lane == 0) {
    theta_out[s] = theta;
    resid_out[s] = resid;
  }
  if (resid <= tol || last) {
    lz_freeze(L, s, m, rg, lane, 64);
  } else if (lane == 0) {
    atomicAdd(slot, 1);
  }
}

// ev = sum_j coef_j R_j + cu u1 for the rows of Lanczos segments (one slab of vectors per launch)
__global__ __launch_bounds__(AI_BLOCK) void k_ritz(const Task* __restrict__ ctasks, const int32_t* __restrict__ mode,
                                                   const int32_t* __restrict__ seg_m, const double* __restrict__ coef, int mcap,
                                                   const double* __restrict__ cu, const double* __restrict__ u1,
                                                   const double* __restrict__ slab, size_t stride, int j0, int nvec, int first,
                                                   double* __restrict__ ev) {
  const Task tk = ctasks[blockIdx.x];
  const int s = tk.z;
  if (mode[s] != 0) return;
  const int m = seg_m[s];
  const int jn = min(nvec, m - j0);
  if (jn <= 0 && !first) return;
  const double* cs = coef + (size_t)s * mcap + j0;
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) {
    double acc = first ? cu[s] * u1[row] : ev[row];
#pragma unroll 8
    for (int j = 0; j < jn; ++j) acc = fma(cs[j], slab[(size_t)j * stride + row], acc);  // same order, eight loads in flight
    ev[row] = acc;
  }
}

// ----------------------------------------------------------------------------- full re-orthogonalisation (ai_eigs_smallest)
// For k > 2 eigenpairs the three-term recurrence alone is not enough: once a Ritz pair has
// converged, copies of it re-enter.  Each new vector r = R_{j+1} is therefore orthogonalised
// against ALL kept Lanczos vectors v_i = (R_i - g_i u1) / b_i, i <= j, before it is normalised:
//   c_i = v_i . r = (R_i . r - g_i (u1 . r)) / b_i,   r <- r - sum_i c_i v_i.
#define FRO_CH 16

// part[task][i] = sum over the task's rows of R_i[row] * r[row], 16 vectors per block
__global__ __launch_bounds__(AI_BLOCK) void k_fro_dots(const Task* __restrict__ ctasks, const double* __restrict__ r,
                                                       double* const* __restrict__ slabs, size_t stride, int nvec,
                                                       double* __restrict__ part, int pitch) {
  constexpr int RPT = AI_COARSE_ROWS / AI_BLOCK;
  __shared__ double sm[FRO_CH][AI_BLOCK / 64];
  const Task tk = ctasks[blockIdx.x];
  const int i0 = blockIdx.y * FRO_CH;
  double rv[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int row = tk.x + threadIdx.x + q * AI_BLOCK;
    rv[q] = (row < tk.y) ? r[row] : 0.0;
  }
  double acc[FRO_CH];
#pragma unroll
  for (int ii = 0; ii < FRO_CH; ++ii) {
    acc[ii] = 0.0;
    const int i = i0 + ii;
    if (i < nvec) {
      const double* Ri = slabs[i / AI_SLAB_VECS] + (size_t)(i % AI_SLAB_VECS) * stride;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int row = tk.x + threadIdx.x + q * AI_BLOCK;
        if (row < tk.y) acc[ii] = fma(rv[q], Ri[row], acc[ii]);
      }
    }
  }
  const int w = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
  for (int ii = 0; ii < FRO_CH; ++ii) {
    const double v = ai_wave_sum(acc[ii]);
    if (ln == 0) sm[ii][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < FRO_CH && i0 + (int)threadIdx.x < nvec) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < AI_BLOCK / 64; ++i) t += sm[threadIdx.x][i];
    part[(size_t)blockIdx.x * pitch + i0 + threadIdx.x] = t;
  }
}

// coef_i = c_i / b_i and cu = sum_i c_i g_i / b_i (one block; fixed summation orders)
__global__ __launch_bounds__(AI_BLOCK) void k_fro_coef(int ntask, const double* __restrict__ part, int pitch, int nvec,
                                                       const double* __restrict__ g_hist, const double* __restrict__ b_hist,
                                                       const double2* __restrict__ pBr, double* __restrict__ coef,
                                                       double* __restrict__ cu) {
  __shared__ double sm[AI_BLOCK / 64];
  double gq = 0.0;
  for (int t = threadIdx.x; t < ntask; t += AI_BLOCK) gq += pBr[t].y;
  gq = ai_block_sum(gq, sm);  // u1 . r
  double cuacc = 0.0;
  for (int i = threadIdx.x; i < nvec; i += AI_BLOCK) {
    double d = 0.0;
    for (int t = 0; t < ntask; ++t) d += part[(size_t)t * pitch + i];
    const double rb = 1.0 / b_hist[i];
    const double c = (d - g_hist[i] * gq) * rb;
    coef[i] = c * rb;
    cuacc += c * g_hist[i] * rb;
  }
  const double tot = ai_block_sum(cuacc, sm);
  if (threadIdx.x == 0) cu[0] = tot;
}

// r <- r - sum_i coef_i R_i + cu u1; new partials (r.r, u1.r)
__global__ __launch_bounds__(AI_BLOCK) void k_fro_apply(const Task* __restrict__ ctasks, double* __restrict__ r,
                                                        double* const* __restrict__ slabs, size_t stride, int nvec,
                                                        const double* __restrict__ coef, const double* __restrict__ cu,
                                                        const double* __restrict__ u1, double2* __restrict__ pBout) {
  constexpr int RPT = AI_COARSE_ROWS / AI_BLOCK;
  __shared__ double sm[AI_BLOCK / 64];
  const Task tk = ctasks[blockIdx.x];
  double rv[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int row = tk.x + threadIdx.x + q * AI_BLOCK;
    rv[q] = (row < tk.y) ? r[row] : 0.0;
  }
  for (int i = 0; i < nvec; ++i) {
    const double* Ri = slabs[i / AI_SLAB_VECS] + (size_t)(i % AI_SLAB_VECS) * stride;
    const double c = coef[i];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int row = tk.x + threadIdx.x + q * AI_BLOCK;
      if (row < tk.y) rv[q] = fma(-c, Ri[row], rv[q]);
    }
  }
  const double cuv = cu[0];
  double nn = 0.0, gg = 0.0;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int row = tk.x + threadIdx.x + q * AI_BLOCK;
    if (row < tk.y) {
      const double ui = u1[row];
      const double v = fma(cuv, ui, rv[q]);
      r[row] = v;
      nn = fma(v, v, nn);
      gg = fma(ui, v, gg);
    }
  }
  const double tn = ai_block_sum(nn, sm);
  const double tg = ai_block_sum(gg, sm);
  if (threadIdx.x == 0) pBout[blockIdx.x] = make_double2(tn, tg);
}

// out[i][row] (+)= sum_j coefT[j][i] R_j[row] for up to 64 Ritz vectors at once, one slab per launch;
// the first launch starts from cu_i u1[row]
#define RITZ_MAXK 64
__global__ __launch_bounds__(AI_BLOCK) void k_ritz_multi(const Task* __restrict__ ctasks, int kv, const double* __restrict__ coefT,
                                                         const double* __restrict__ cuv, const double* __restrict__ u1,
                                                         const double* __restrict__ slab, size_t stride, int j0, int nvec, int m,
                                                         int first, double* __restrict__ out, size_t out_stride) {
  const Task tk = ctasks[blockIdx.x];
  const int jn = min(nvec, m - j0);
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) {
    double acc[RITZ_MAXK];
    const double ui = u1[row];
#pragma unroll
    for (int i = 0; i < RITZ_MAXK; ++i) acc[i] = (i < kv) ? (first ? cuv[i] * ui : out[(size_t)i * out_stride + row]) : 0.0;
    for (int j = 0; j < jn; ++j) {
      const double v = slab[(size_t)j * stride + row];
      const double* cj = coefT + (size_t)(j0 + j) * RITZ_MAXK;
#pragma unroll
      for (int i = 0; i < RITZ_MAXK; ++i) acc[i] = fma(cj[i], v, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < RITZ_MAXK; ++i)
      if (i < kv) out[(size_t)i * out_stride + row] = acc[i];
  }
}

// ----------------------------------------------------------------------------- threshold sweep
struct MinMaxPart {
  double mn, mx, sumsq, amax;
  int32_t amax_id;   // original id of the entry of largest magnitude (smallest id on ties)
  int32_t amax_neg;  // that entry is negative
};

__device__ __forceinline__ void mm_merge(MinMaxPart& r, const MinMaxPart& q) {
  r.mn = fmin(r.mn, q.mn);
  r.mx = fmax(r.mx, q.mx);
  r.sumsq += q.sumsq;
  if (q.amax > r.amax || (q.amax == r.amax && q.amax_id < r.amax_id)) {
    r.amax = q.amax;
    r.amax_id = q.amax_id;
    r.amax_neg = q.amax_neg;
  }
}

__global__ __launch_bounds__(AI_BLOCK) void k_minmax(const Task* __restrict__ ctasks, const int32_t* __restrict__ mode,
                                                     const double* __restrict__ ev, const int32_t* __restrict__ orig,
                                                     MinMaxPart* __restrict__ part) {
  __shared__ MinMaxPart sm[AI_BLOCK / 64];
  const Task tk = ctasks[blockIdx.x];
  if (mode && mode[tk.z] != 0) return;  // a disconnected segment has no eigenvector: it is split by components
  MinMaxPart r;
  r.mn = 1e300;
  r.mx = -1e300;
  r.sumsq = 0.0;
  r.amax = -1.0;
  r.amax_id = 0x7fffffff;
  r.amax_neg = 0;
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) {
    const double e = ev[row];
    MinMaxPart q;
    q.mn = e;
    q.mx = e;
    q.sumsq = e * e;
    q.amax = fabs(e);
    q.amax_id = orig[row];
    q.amax_neg = e < 0.0;
    mm_merge(r, q);
  }
  for (int o = 32; o > 0; o >>= 1) {
    MinMaxPart q;
    q.mn = __shfl_xor(r.mn, o, 64);
    q.mx = __shfl_xor(r.mx, o, 64);
    q.sumsq = __shfl_xor(r.sumsq, o, 64);
    q.amax = __shfl_xor(r.amax, o, 64);
    q.amax_id = __shfl_xor(r.amax_id, o, 64);
    q.amax_neg = __shfl_xor(r.amax_neg, o, 64);
    mm_merge(r, q);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    MinMaxPart t = sm[0];
    for (int i = 1; i < AI_BLOCK / 64; ++i) mm_merge(t, sm[i]);
    part[blockIdx.x] = t;
  }
}

// Per segment: unit norm + sign convention folded into one scale; np.allclose(mn, mx) test
// (normalized_cut.py:22); thresholds t_k = k * step + mn, step = (mx - mn) / 10, exactly as
// np.linspace(mn, mx, 10, endpoint=False) computes them (:28).
__global__ void k_minmax_final(const int32_t* __restrict__ ctask0, const int32_t* __restrict__ mode,
                               const MinMaxPart* __restrict__ part, int S, int raw, double* __restrict__ scale,
                               int32_t* __restrict__ nosplit, double* __restrict__ thr) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  if (mode && mode[s] != 0) {
    scale[s] = 1.0;
    nosplit[s] = 1;
    return;
  }
  MinMaxPart r = part[ctask0[s]];
  for (int t = ctask0[s] + 1; t < ctask0[s + 1]; ++t) mm_merge(r, part[t]);
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
__global__ __launch_bounds__(AI_BLOCK) void k_bin(const Task* __restrict__ ctasks, const int32_t* __restrict__ nosplit,
                                                  const double* __restrict__ scale, const double* __restrict__ thr,
                                                  const double* __restrict__ ev, uint8_t* __restrict__ bin) {
  const Task tk = ctasks[blockIdx.x];
  const int s = tk.z;
  if (nosplit[s]) return;  // no threshold will be applied to this segment (k_split_flags looks at split[] first)
  const double sc = scale[s];
  double th[AI_NUM_CUTS];
#pragma unroll
  for (int k = 0; k < AI_NUM_CUTS; ++k) th[k] = thr[s * AI_NUM_CUTS + k];
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) {
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
__global__ __launch_bounds__(AI_BLOCK) void k_sweep(const Task* __restrict__ ftasks, const int32_t* __restrict__ nosplit,
                                                    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    const double* __restrict__ wraw, const double* __restrict__ deg,
                                                    const uint8_t* __restrict__ bin, double* __restrict__ part) {
  // the block's partial cut sums go through LDS transposed, so that each of the 10 columns is
  // reduced by ONE wave (fixed order) instead of 40 block-wide reductions with two barriers each
  __shared__ double scut[AI_NUM_CUTS][AI_BLOCK + 1];
  __shared__ double sdeg[AI_FINE_ROWS];
  __shared__ int sbin[AI_FINE_ROWS];
  const Task tk = ftasks[blockIdx.x];
  if (nosplit[tk.z]) return;
  const int l = threadIdx.x & (AI_LPR - 1), r = threadIdx.x / AI_LPR;
  if (threadIdx.x < AI_FINE_ROWS) {
    const int row = tk.x + threadIdx.x;
    sbin[threadIdx.x] = (row < tk.y) ? (int)bin[row] : -1;
    sdeg[threadIdx.x] = (row < tk.y) ? deg[row] : 0.0;
  }
  double cut[AI_NUM_CUTS];
#pragma unroll
  for (int k = 0; k < AI_NUM_CUTS; ++k) cut[k] = 0.0;
  for (int row = tk.x + r; row < tk.y; row += AI_BLOCK / AI_LPR) {
    const int bi = bin[row];
    const int p0 = rowptr[row], p1 = rowptr[row + 1];
    {
      int c[AI_ROW_PF], bj[AI_ROW_PF];
      double w[AI_ROW_PF];
#pragma unroll
      for (int q = 0; q < AI_ROW_PF; ++q) {
        const int p = p0 + l + q * AI_LPR;
        const bool ok = p < p1;
        c[q] = ok ? col[p] : -1;
        w[q] = ok ? wraw[p] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < AI_ROW_PF; ++q) bj[q] = (c[q] >= 0) ? (int)bin[c[q]] : 0;
#pragma unroll
      for (int q = 0; q < AI_ROW_PF; ++q)
        if (c[q] >= 0) {
#pragma unroll
          for (int k = 0; k < AI_NUM_CUTS; ++k) cut[k] += (k >= bj[q] && k < bi) ? w[q] : 0.0;
        }
    }
    for (int p = p0 + l + AI_ROW_PF * AI_LPR; p < p1; p += AI_LPR) {
      const int bj = bin[col[p]];
      const double w = wraw[p];
#pragma unroll
      for (int k = 0; k < AI_NUM_CUTS; ++k) cut[k] += (k >= bj && k < bi) ? w : 0.0;
    }
  }
#pragma unroll
  for (int k = 0; k < AI_NUM_CUTS; ++k) scut[k][threadIdx.x] = cut[k];
  __syncthreads();
  double* out = part + (size_t)blockIdx.x * AI_SWEEP_VALS;
  const int w = threadIdx.x >> 6, ln = threadIdx.x & 63;
  for (int k = w; k < AI_NUM_CUTS; k += AI_BLOCK / 64) {
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < AI_BLOCK / 64; ++q) a += scut[k][ln + 64 * q];
    a = ai_wave_sum(a);
    if (ln == 0) out[k] = a;
  }
  // assocA_k / assocB_k / |A_k| over the task's rows in row order (30 threads, 32 rows each)
  if (threadIdx.x < 3 * AI_NUM_CUTS) {
    const int which = threadIdx.x / AI_NUM_CUTS, k = threadIdx.x % AI_NUM_CUTS;
    double a = 0.0;
    for (int i = 0; i < AI_FINE_ROWS; ++i) {
      const int bi = sbin[i];
      if (bi < 0) break;
      const bool inA = k < bi;
      a += (which == 0) ? (inA ? sdeg[i] : 0.0) : (which == 1) ? (inA ? 0.0 : sdeg[i]) : (inA ? 1.0 : 0.0);
    }
    out[(which + 1) * AI_NUM_CUTS + k] = a;
  }
}

// Per segment: ncut_k = cut_k / assocA_k + cut_k / assocB_k; first strictly smaller cost wins
// (normalized_cut.py:29-32); split iff mcut < T (:56).  One block per segment: column c of the
// 40 partial columns is summed by 6 threads over interleaved task stripes, then in stripe order.
#define SWF_STRIPES 6
__global__ __launch_bounds__(AI_BLOCK) void k_sweep_final(const int32_t* __restrict__ ftask0, const int32_t* __restrict__ nosplit,
                                                          const double* __restrict__ part, double T, double* __restrict__ costs,
                                                          int32_t* __restrict__ kstar, int32_t* __restrict__ split,
                                                          int32_t* __restrict__ ntrue, double* __restrict__ mcut_out) {
  __shared__ double acc[SWF_STRIPES][AI_SWEEP_VALS];
  const int s = blockIdx.x;
  if (nosplit[s]) {
    if (threadIdx.x == 0) {
      split[s] = 0;
      kstar[s] = 0;
      ntrue[s] = 0;
      mcut_out[s] = INFINITY;
    }
    if (threadIdx.x < AI_NUM_CUTS) costs[s * AI_NUM_CUTS + threadIdx.x] = NAN;
    return;
  }
  const int c = threadIdx.x % AI_SWEEP_VALS, stripe = threadIdx.x / AI_SWEEP_VALS;
  if (stripe < SWF_STRIPES) {
    double a = 0.0;
    for (int t = ftask0[s] + stripe; t < ftask0[s + 1]; t += SWF_STRIPES) a += part[(size_t)t * AI_SWEEP_VALS + c];
    acc[stripe][c] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double best = INFINITY;
    int kb = 0;
    double nb = 0.0;
    for (int k = 0; k < AI_NUM_CUTS; ++k) {
      double v[4];
      for (int q = 0; q < 4; ++q) {
        double a = 0.0;
        for (int st = 0; st < SWF_STRIPES; ++st) a += acc[st][q * AI_NUM_CUTS + k];
        v[q] = a;
      }
      const double cost = __dadd_rn(__ddiv_rn(v[0], v[1]), __ddiv_rn(v[0], v[2]));
      costs[s * AI_NUM_CUTS + k] = cost;
      if (cost < best) {
        best = cost;
        kb = k;
        nb = v[3];
      }
    }
    kstar[s] = kb;
    mcut_out[s] = best;
    split[s] = (best < T) ? 1 : 0;
    ntrue[s] = (int32_t)nb;
  }
}

// ----------------------------------------------------------------------------- partition + rebuild
__global__ __launch_bounds__(AI_BLOCK) void k_split_flags(const Task* __restrict__ ctasks, const int32_t* __restrict__ split,
                                                          const int32_t* __restrict__ kstar, const uint8_t* __restrict__ bin,
                                                          int32_t* __restrict__ flag) {
  const Task tk = ctasks[blockIdx.x];
  const int sp = split[tk.z], ks = kstar[tk.z];
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) flag[row] = (sp && (int)bin[row] > ks) ? 1 : 0;
}

// Stable partition inside each parent (mask side first, normalized_cut.py:57-59).  Writes the
// caller-order id of every row to its position in the final ordering and the row's index in the
// next level's compact order (-1: the row's segment is finished).
__global__ __launch_bounds__(AI_BLOCK) void k_partition(const Task* __restrict__ ctasks, const int32_t* __restrict__ seg_start,
                                                        const int32_t* __restrict__ seg_gstart, const int32_t* __restrict__ split,
                                                        const int32_t* __restrict__ ntrue, const int32_t* __restrict__ childA,
                                                        const int32_t* __restrict__ childB, const int32_t* __restrict__ flag,
                                                        const int32_t* __restrict__ fscan, const int32_t* __restrict__ orig,
                                                        int32_t* __restrict__ final_order, int32_t* __restrict__ map,
                                                        int32_t* __restrict__ orig_next, const int32_t* __restrict__ multi) {
  const Task tk = ctasks[blockIdx.x];
  const int s = tk.z;
  if (multi[s]) return;  // split by components: k_partition_multi
  const int s0 = seg_start[s], g0 = seg_gstart[s], nt = split[s] ? ntrue[s] : 0;
  const int cA = childA[s], cB = childB[s];
  const int f0 = fscan[s0];
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) {
    const int f = flag[row];
    const int rt = fscan[row] - f0;  // mask-side rows before this one
    const int rf = (row - s0) - rt;  // other-side rows before this one
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


// ---- a disconnected segment is split into ALL its connected components at once (see the file header).
// Component = union-find root = its first row; components keep the order of their first rows, rows keep
// their order inside a component: a stable sort of the rows by root id (segments stay where they are,
// because a root lies inside its segment's row range).

// rc[row] = 1 for the roots of the segments that are split by components
__global__ __launch_bounds__(AI_BLOCK) void k_comp_rootflag(const Task* __restrict__ ctasks, const int32_t* __restrict__ multi,
                                                            const int32_t* __restrict__ parent, int32_t* __restrict__ rc) {
  const Task tk = ctasks[blockIdx.x];
  const int m = multi[tk.z];
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK) rc[row] = (m && parent[row] == row) ? 1 : 0;
}

// component table in row order: (root row, rows of the component)
__global__ __launch_bounds__(AI_BLOCK) void k_comp_table(const Task* __restrict__ ctasks, const int32_t* __restrict__ multi,
                                                         const int32_t* __restrict__ parent, const int32_t* __restrict__ rootord,
                                                         const int32_t* __restrict__ rcnt, int32_t* __restrict__ troot,
                                                         int32_t* __restrict__ tsize) {
  const Task tk = ctasks[blockIdx.x];
  if (!multi[tk.z]) return;
  for (int row = tk.x + threadIdx.x; row < tk.y; row += AI_BLOCK)
    if (parent[row] == row) {
      const int o = rootord[row];
      troot[o] = row;
      tsize[o] = rcnt[row];
    }
}

// p = position of a row after the stable sort by root.  The row goes to position p of its parent's range in the
// final ordering and, if its component continues (cbase >= 0), to row cbase + (rank inside the component) of
// the next level's compact order.
__global__ __launch_bounds__(AI_BLOCK) void k_partition_multi(const Task* __restrict__ ctasks, const int32_t* __restrict__ multi,
                                                              const int32_t* __restrict__ seg_start, const int32_t* __restrict__ seg_gstart,
                                                              const int32_t* __restrict__ sorted_rows, const int32_t* __restrict__ parent,
                                                              const int32_t* __restrict__ rootord, const int32_t* __restrict__ cpos,
                                                              const int32_t* __restrict__ cbase, const int32_t* __restrict__ orig,
                                                              int32_t* __restrict__ final_order, int32_t* __restrict__ map,
                                                              int32_t* __restrict__ orig_next) {
  const Task tk = ctasks[blockIdx.x];
  const int s = tk.z;
  if (!multi[s]) return;
  const int s0 = seg_start[s], g0 = seg_gstart[s];
  for (int p = tk.x + threadIdx.x; p < tk.y; p += AI_BLOCK) {
    const int row = sorted_rows[p];
    const int o = rootord[parent[row]];
    const int32_t id = orig[row];
    final_order[g0 + (p - s0)] = id;
    const int base = cbase[o];
    const int32_t dst = (base >= 0) ? base + (p - cpos[o]) : -1;
    map[row] = dst;
    if (dst >= 0) orig_next[dst] = id;
  }
}

// parent_next[map[row]] = map[parent[row]]: a component's first row stays its first row under a
// stable partition, and when the cut ran between whole components it lands in the same child
__global__ __launch_bounds__(AI_BLOCK) void k_carry_parent(const int32_t* __restrict__ parent, const int32_t* __restrict__ map,
                                                           int32_t n, int32_t* __restrict__ parent_next) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t dst = map[i];
  if (dst >= 0) {
    const int32_t pr = map[parent[i]];
    parent_next[dst] = (pr >= 0) ? pr : dst;
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
  const int p0 = rowptr[row], p1 = rowptr[row + 1];
  {
    int cc[AI_ROW_PF];
#pragma unroll
    for (int q = 0; q < AI_ROW_PF; ++q) {
      const int p = p0 + l + q * AI_LPR;
      cc[q] = (p < p1) ? col[p] : -1;
    }
#pragma unroll
    for (int q = 0; q < AI_ROW_PF; ++q)
      if (cc[q] >= 0) c += (flag[cc[q]] == f) ? 1 : 0;
  }
  for (int p = p0 + l + AI_ROW_PF * AI_LPR; p < p1; p += AI_LPR) c += (flag[col[p]] == f) ? 1 : 0;
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
  // the first AI_ROW_PF rounds: every load of the chain col -> (flag, map), and the weights, in flight together
  int cq[AI_ROW_PF], mq[AI_ROW_PF];
  bool kq[AI_ROW_PF];
  double wq[AI_ROW_PF];
#pragma unroll
  for (int q = 0; q < AI_ROW_PF; ++q) {
    const int p = p0 + q * AI_LPR + l;
    const bool ok = p < p1;
    cq[q] = ok ? col[p] : -1;
    wq[q] = ok ? wraw[p] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < AI_ROW_PF; ++q) {
    kq[q] = (cq[q] >= 0) && (flag[cq[q]] == f);
    mq[q] = (cq[q] >= 0) ? map[cq[q]] : -1;
  }
#pragma unroll
  for (int q = 0; q < AI_ROW_PF; ++q) {
    if (q >= rounds) break;  // wave-uniform
    const unsigned long long bal = __ballot(kq[q]);
    const unsigned int gbits = (unsigned int)((bal >> (grp * AI_LPR)) & 0xffffull);
    const int before = __popc(gbits & ((1u << l) - 1u));
    if (kq[q]) {
      new_col[out + before] = mq[q];
      new_w[out + before] = wq[q];
    }
    out += __popc(gbits);
  }
  for (int it = AI_ROW_PF; it < rounds; ++it) {
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

// dst[i] = src[i] + add (src == nullptr: dst[i] = i + add): concatenation of several CSR graphs
__global__ __launch_bounds__(AI_BLOCK) void k_offset_copy(int32_t* __restrict__ dst, const int32_t* __restrict__ src, int64_t n,
                                                          int32_t add) {
  const int64_t i = (int64_t)blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i < n) dst[i] = (src ? src[i] : (int32_t)i) + add;
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

static void tridiag_top(const double* a, const double* b, int m, const double* hint, double* theta_out, std::vector<double>& s) {
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
  if (hint) {
    // the device check already located the eigenvalue: verify a tight bracket around it
    const double w = 1e-13 * std::max(fabs(*hint), 1.0);
    const double l2 = *hint - w, h2 = *hint + w;
    if (l2 > lo && sturm_lt_host(a, b, m, l2) < m) lo = l2;
    if (h2 < hi && sturm_lt_host(a, b, m, h2) == m) hi = h2;
  }
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
    for (int i = 0; i + 1 < m; ++i) {  // forward: L y = P x
      if (piv[i]) std::swap(x[i], x[i + 1]);
      x[i + 1] -= dl[i] * x[i];
    }
    x[m - 1] /= d[m - 1];  // backward: U z = y
    if (m >= 2) x[m - 2] = (x[m - 2] - du[m - 2] * x[m - 1]) / d[m - 2];
    for (int i = m - 3; i >= 0; --i) x[i] = (x[i] - du[i] * x[i + 1] - du2[i] * x[i + 2]) / d[i];
    double n2 = 0.0;
    for (int i = 0; i < m; ++i) n2 += x[i] * x[i];
    const double rn = 1.0 / sqrt(n2);
    for (int i = 0; i < m; ++i) x[i] *= rn;
  }
  s = x;
}

// idx-th largest eigenvalue (idx = 0: the largest) of T by bisection on Sturm counts
static double tridiag_eigval(const double* a, const double* b, int m, int idx, double lo, double hi) {
  const int need = m - idx;  // smallest x with count_lt(x) >= need is just above the wanted eigenvalue
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid <= lo || mid >= hi) break;
    if (sturm_lt_host(a, b, m, mid) >= need) hi = mid; else lo = mid;
  }
  return 0.5 * (lo + hi);
}

// eigenvector of T for the (already located) eigenvalue theta: pivoted LU + inverse iteration,
// kept orthogonal to `prev` (eigenvectors of neighbouring eigenvalues, as LAPACK dstein does)
static void tridiag_eigvec(const double* a, const double* b, int m, double theta, const std::vector<std::vector<double>>& prev,
                           const std::vector<int>& cluster, std::vector<double>& x) {
  x.assign(m, 0.0);
  if (m == 1) {
    x[0] = 1.0;
    return;
  }
  double nrm = 0.0;
  for (int i = 0; i < m; ++i) nrm = std::max(nrm, fabs(a[i]) + (i > 0 ? fabs(b[i]) : 0.0) + (i + 1 < m ? fabs(b[i + 1]) : 0.0));
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
  for (int i = 0; i < m; ++i) x[i] = 1.0 + 0.001 * ((i * 2654435761u) % 1000) / 1000.0;
  for (int iter = 0; iter < 5; ++iter) {
    for (int c : cluster) {  // stay orthogonal to the eigenvectors of the cluster found so far
      double dot = 0.0;
      for (int i = 0; i < m; ++i) dot += prev[c][i] * x[i];
      for (int i = 0; i < m; ++i) x[i] -= dot * prev[c][i];
    }
    for (int i = 0; i + 1 < m; ++i) {
      if (piv[i]) std::swap(x[i], x[i + 1]);
      x[i + 1] -= dl[i] * x[i];
    }
    x[m - 1] /= d[m - 1];
    if (m >= 2) x[m - 2] = (x[m - 2] - du[m - 2] * x[m - 1]) / d[m - 2];
    for (int i = m - 3; i >= 0; --i) x[i] = (x[i] - du[i] * x[i + 1] - du2[i] * x[i + 2]) / d[i];
    double n2 = 0.0;
    for (int i = 0; i < m; ++i) n2 += x[i] * x[i];
    const double rn = 1.0 / sqrt(n2);
    for (int i = 0; i < m; ++i) x[i] *= rn;
  }
  for (int c : cluster) {
    double dot = 0.0;
    for (int i = 0; i < m; ++i) dot += prev[c][i] * x[i];
    for (int i = 0; i < m; ++i) x[i] -= dot * prev[c][i];
  }
  double n2 = 0.0;
  for (int i = 0; i < m; ++i) n2 += x[i] * x[i];
  const double rn = 1.0 / sqrt(n2);
  for (int i = 0; i < m; ++i) x[i] *= rn;
}

// ----------------------------------------------------------------------------- host: driver
struct SegHost {
  int start, n, gstart;
  int mode;     // 0 Lanczos, 1 null vector
  int need_cc;  // 0: component labels were carried over a cut between whole components
  int chunk;    // which chunk of a batched call the segment belongs to
};

static double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

template <typename T>
struct Ptr {  // a device pointer into a larger blob (same `.p` spelling as DevBuf)
  T* p = nullptr;
};

struct TaskList {
  int rows_per_task = 0;
  int n = 0;
  std::vector<Task> h;
  std::vector<int32_t> h_seg0;
  Ptr<Task> d;
  Ptr<int32_t> d_seg0;
};

// Several small host arrays -> the context's pinned staging buffer -> ONE async copy into one
// device blob; the device pointers are handed out after the copy is queued.
struct Pack {
  char* stage;
  size_t cap, off = 0;
  struct Item {
    void** dst;
    size_t off;
  };
  std::vector<Item> items;
  bool overflow = false;
  Pack(char* stage_, size_t cap_) : stage(stage_), cap(cap_) {}
  template <typename T>
  void add(T** dst, const T* src, size_t count) {
    off = (off + 63) & ~(size_t)63;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    if (off + bytes > cap) {
      overflow = true;
      return;
    }
    if (count) memcpy(stage + off, src, count * sizeof(T));
    items.push_back(Item{(void**)dst, off});
    off += bytes;
  }
  int flush(DevBuf<char>& blob, hipStream_t st) {
    if (overflow) {
      ai_set_error("per-level task tables exceed the %zu-byte staging buffer (too many segments for this build)", cap);
      return AI_ERR_INTERNAL;
    }
    AI_TRY(blob.ensure(off + 64));
    AI_HIP(hipMemcpyAsync(blob.p, stage, off, hipMemcpyHostToDevice, st));
    for (auto& it : items) *it.dst = blob.p + it.off;
    return AI_OK;
  }
};

class Solver {
 public:
  Solver(ai_ctx* c, const ai_csr* a) : ctx(c), A(a), st(c->stream) {}

  ai_ctx* ctx;
  const ai_csr* A;
  hipStream_t st;
  ai_ncut_opts opt{1e-10, 4000, 16, 0};
  bool time_spmv = false;           // opts.reserved bit 0: HIP events around every SpMV launch
  bool clock_spmv = false;          // opts.reserved bit 1: every SpMV launch stamps its own span on the device clock
  DevBuf<unsigned long long> tstamps, tblock;  // {start, end} per launch of the current level; per block of the launch in flight
  double clock_khz = 0.0;
  std::vector<hipEvent_t> evpool;   // 2 per launch of the current level
  DevBuf<unsigned long long> work;  // [rows, nnz] processed by the SpMV kernel
  ai_ncut_stats stats{};

  // active set
  int na = 0;                 // active rows
  std::vector<SegHost> segs;  // active segments (host copy)
  int S() const { return (int)segs.size(); }
  const int32_t *rowptr = nullptr, *col = nullptr, *orig = nullptr;
  const double* wraw = nullptr;
  int32_t* parent = nullptr;  // component labels of the current level
  DevBuf<int32_t> b_rowptr[2], b_col[2], b_orig[2], b_parent[2];
  DevBuf<double> b_wraw[2];
  DevBuf<int32_t> orig_id;  // identity when the graph has no permutation of its own
  int pp = 0;               // ping-pong index of the NEXT level's buffers

  // per-row work arrays
  DevBuf<double> deg, sinv, sinv2, u1, wm, ev, Y;
  DevBuf<int32_t> rcnt, rc, ex, flag, fscan, map, newcnt, scantmp, final_order;
  DevBuf<uint8_t> side, bin;
  // tasks
  TaskList fine, coarse;  // every active row
  TaskList lzf, lzc;      // rows of the Lanczos-mode segments only (grids of the step kernels)
  std::vector<TaskRange> h_cranges, h_segrange;
  Ptr<TaskRange> cranges, segrange;
  DevBuf<char> blobA, blobB, blobC, blobD, resblob, lzres;
  DevBuf<int32_t> t_root, t_size;  // component table of the disconnected segments
  DevBuf<uint8_t> sorttmp;
  size_t rescap = 0;
  std::vector<int32_t> h_seg_start;
  Ptr<int32_t> seg_start, factive, cactive;
  DevBuf<double> pvol, pA, pvolA, pvolB, psweep;
  DevBuf<double2> pB[2];
  DevBuf<MinMaxPart> pmm;
  // per-segment device arrays
  Ptr<int32_t> s_mode, s_needcc, s_gstart, s_childA, s_childB, s_multi, c_pos, c_base;  // uploaded per level (blobs)
  Ptr<int32_t> s_split, s_ntrue, s_frozen, s_m;                  // downloaded per level (resblob / lzres)
  Ptr<double> s_mcut, s_resid, s_theta;
  DevBuf<int32_t> s_ncomp, s_nosplit, s_kstar, slots;
  DevBuf<double> s_vol, s_volA, s_volB, s_scale, s_thr, s_costs, s_cu;
  // Lanczos history + vectors
  DevBuf<double> alpha_hist, b_hist, g_hist, coef, bnew_buf;
  // staged-gather encoding of the Lanczos tasks (k_lz_encode), rebuilt per level
  DevBuf<uint16_t> lidx;
  DevBuf<int32_t> ucol, enc_ctr;
  DevBuf<TaskEnc> enc;
  int32_t ucol_cap = 0;
  bool enc_ready = false;
  int enc_min_tasks = -1;  // AI_SPMV_STAGE_MIN: smallest Lanczos frontier (in tasks) that is encoded
  int mcap = 0;
  std::vector<double*> slabs, owned_slabs;
  size_t slab_stride = 0;
  ~Solver() {
    for (double* p : owned_slabs) (void)hipFree(p);
    for (hipEvent_t e : evpool) (void)hipEventDestroy(e);
  }

  LzSeg lzseg() {
    LzSeg L;
    L.frozen = s_frozen.p;
    L.m = s_m.p;
    L.alpha_hist = alpha_hist.p;
    L.b_hist = b_hist.p;
    L.g_hist = g_hist.p;
    L.factive = factive.p;
    L.cactive = cactive.p;
    L.mcap = mcap;
    return L;
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
    AI_TRY(b_parent[0].alloc(n));
    AI_TRY(b_parent[1].alloc(n));
    // children never have more rows / entries than the chunk: size the ping-pong halves once
    for (int h = 0; h < 2; ++h) {
      AI_TRY(b_rowptr[h].alloc(n + 1));
      AI_TRY(b_orig[h].alloc(n));
      AI_TRY(b_col[h].alloc(e));
      AI_TRY(b_wraw[h].alloc(e));
    }
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
    AI_TRY(slots.alloc(AI_MAX_CHECKS));
    AI_TRY(lidx.alloc(e));
    ucol_cap = (int32_t)std::min<size_t>(e / 2 + 4096, (size_t)1 << 30);
    AI_TRY(ucol.alloc((size_t)ucol_cap));
    AI_TRY(enc_ctr.alloc(1));
    AI_TRY(work.alloc(2));
    AI_HIP(hipMemsetAsync(work.p, 0, 2 * sizeof(unsigned long long), st));
    return AI_OK;
  }

  int alloc_segs(int S_) {
    const size_t s = (size_t)S_ + 1;
    AI_TRY(s_ncomp.ensure(s));
    AI_TRY(s_nosplit.ensure(s));
    AI_TRY(s_kstar.ensure(s));
    AI_TRY(s_vol.ensure(s));
    AI_TRY(s_volA.ensure(s));
    AI_TRY(s_volB.ensure(s));
    AI_TRY(s_scale.ensure(s));
    AI_TRY(s_thr.ensure(s * AI_NUM_CUTS));
    AI_TRY(s_costs.ensure(s * AI_NUM_CUTS));
    AI_TRY(s_cu.ensure(s));
    if (s > rescap) {
      // results that the host reads every level sit together so that one copy fetches them
      rescap = std::max(s, (size_t)128);
      AI_TRY(resblob.alloc(rescap * 16));
      AI_TRY(lzres.alloc(rescap * 24));
    }
    s_split.p = (int32_t*)resblob.p;
    s_ntrue.p = s_split.p + rescap;
    s_mcut.p = (double*)(s_ntrue.p + rescap);
    s_m.p = (int32_t*)lzres.p;
    s_frozen.p = s_m.p + rescap;
    s_resid.p = (double*)(s_frozen.p + rescap);
    s_theta.p = s_resid.p + rescap;
    return AI_OK;
  }

  // Level 0: the whole graph is one segment, rows in the graph's own order.
  int begin(bool force_single_segment) {
    const int n = (int)A->n;
    AI_TRY(alloc_rows());
    rowptr = A->rowptr;
    col = A->col;
    wraw = A->val;
    parent = b_parent[0].p;
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
    if (force_single_segment) segs.push_back(SegHost{0, n, 0, 0, 1, 0});
    return AI_OK;
  }

  void make_tasks(TaskList& tl, int rows_per_task, bool lanczos_only = false) {
    const int S_ = S();
    tl.rows_per_task = rows_per_task;
    tl.h.clear();
    tl.h_seg0.assign(S_ + 1, 0);
    for (int s = 0; s < S_; ++s) {
      tl.h_seg0[s] = (int32_t)tl.h.size();
      if (lanczos_only && segs[s].mode != 0) continue;
      bool first = true;
      for (int lo = segs[s].start; lo < segs[s].start + segs[s].n; lo += rows_per_task) {
        Task t;
        t.x = lo;
        t.y = std::min(lo + rows_per_task, segs[s].start + segs[s].n);
        t.z = s;
        t.w = first ? 1 : 0;
        // Lanczos lists, bits 1..: coarse tasks carry the segment's step cap (its Krylov dimension / the
        // step limit), fine tasks their index among ALL fine tasks
        if (lanczos_only) {
          if (rows_per_task == AI_FINE_ROWS)
            t.w |= (fine.h_seg0[s] + (lo - segs[s].start) / AI_FINE_ROWS) << 1;
          else
            t.w |= std::max(1, std::min(opt.max_iter, segs[s].n - 1)) << 1;
        }
        first = false;
        tl.h.push_back(t);
      }
    }
    tl.h_seg0[S_] = (int32_t)tl.h.size();
    tl.n = (int)tl.h.size();
  }

  // task lists + segment offsets for the current `segs`
  int build_tasks() {
    const int S_ = S();
    make_tasks(fine, AI_FINE_ROWS);
    make_tasks(coarse, AI_COARSE_ROWS);
    h_seg_start.assign(S_ + 1, 0);
    std::vector<int32_t> h_gstart(S_ + 1, 0), h_needcc(S_ + 1, 0);
    for (int s = 0; s < S_; ++s) {
      h_seg_start[s] = segs[s].start;
      h_gstart[s] = segs[s].gstart;
      h_needcc[s] = segs[s].need_cc;
    }
    h_seg_start[S_] = S_ ? segs[S_ - 1].start + segs[S_ - 1].n : 0;
    AI_TRY(alloc_segs(S_));
    AI_TRY(pvol.ensure(fine.n + 1));
    AI_TRY(psweep.ensure((size_t)(fine.n + 1) * AI_SWEEP_VALS));
    AI_TRY(pvolA.ensure(coarse.n + 1));
    AI_TRY(pvolB.ensure(coarse.n + 1));
    AI_TRY(pmm.ensure(coarse.n + 1));
    Pack pk(ctx->stage, AI_STAGE_BYTES / 4);
    pk.add(&fine.d.p, fine.h.data(), (size_t)fine.n);
    pk.add(&coarse.d.p, coarse.h.data(), (size_t)coarse.n);
    pk.add(&fine.d_seg0.p, fine.h_seg0.data(), (size_t)S_ + 1);
    pk.add(&coarse.d_seg0.p, coarse.h_seg0.data(), (size_t)S_ + 1);
    pk.add(&seg_start.p, h_seg_start.data(), (size_t)S_ + 1);
    pk.add(&s_gstart.p, h_gstart.data(), (size_t)S_ + 1);
    pk.add(&s_needcc.p, h_needcc.data(), (size_t)S_ + 1);
    AI_TRY(pk.flush(blobA, st));
    return AI_OK;
  }

  // degrees, scaled matrix, u1, connected components -> segs[s].mode
  int prepare(bool want_cc) {
    const int S_ = S();
    hipLaunchKernelGGL(k_degree, dim3(fine.n), dim3(AI_BLOCK), 0, st, fine.d.p, rowptr, wraw, deg.p, sinv.p, pvol.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_seg_sum, dim3(S_), dim3(AI_BLOCK), 0, st, fine.d_seg0.p, pvol.p, s_vol.p);
    AI_KERNEL_CHECK();
    std::vector<int32_t> ncomp(S_, 1);
    if (want_cc) {
      bool any = false;
      for (auto& s : segs) any |= (s.need_cc != 0);
      if (any) {
        hipLaunchKernelGGL(k_cc_init, dim3(fine.n), dim3(AI_BLOCK), 0, st, fine.d.p, s_needcc.p, rowptr, col, parent);
        AI_KERNEL_CHECK();
        hipLaunchKernelGGL(k_cc_hook, dim3(fine.n), dim3(AI_BLOCK), 0, st, fine.d.p, s_needcc.p, rowptr, col, parent);
        AI_KERNEL_CHECK();
        hipLaunchKernelGGL(k_cc_compress, dim3(coarse.n), dim3(AI_BLOCK), 0, st, coarse.d.p, s_needcc.p, parent);
        AI_KERNEL_CHECK();
      }
      AI_HIP(hipMemsetAsync(s_ncomp.p, 0, (size_t)S_ * sizeof(int32_t), st));
      hipLaunchKernelGGL(k_cc_count, dim3(coarse.n), dim3(AI_BLOCK), 0, st, coarse.d.p, parent, s_ncomp.p);
      AI_KERNEL_CHECK();
      int32_t* dl = (int32_t*)(ctx->stage + AI_STAGE_BYTES / 2);
      AI_HIP(hipMemcpyAsync(dl, s_ncomp.p, (size_t)S_ * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      AI_HIP(hipStreamSynchronize(st));
      for (int s = 0; s < S_; ++s) ncomp[s] = dl[s];
    }
    std::vector<int32_t> mode(S_ + 1, 0);
    for (int s = 0; s < S_; ++s) {
      segs[s].mode = (ncomp[s] > 1) ? 1 : 0;
      mode[s] = segs[s].mode;
    }
    // step kernels run on the Lanczos-mode segments only: their own (compact) task lists, so that
    // a frontier with few connected segments launches few blocks and the XCD remap still spreads
    // them over the whole chip
    make_tasks(lzf, AI_FINE_ROWS, true);
    make_tasks(lzc, AI_COARSE_ROWS, true);
    h_segrange.assign(S_ + 1, TaskRange{0, 0, 0, 0});
    for (int s = 0; s < S_; ++s) h_segrange[s] = TaskRange{lzf.h_seg0[s], lzf.h_seg0[s + 1], lzc.h_seg0[s], lzc.h_seg0[s + 1]};
    h_cranges.resize(lzc.n);
    for (int t = 0; t < lzc.n; ++t) h_cranges[t] = h_segrange[lzc.h[t].z];
    std::vector<int32_t> ones((size_t)std::max(lzf.n, lzc.n) + 1, 1);
    AI_TRY(pA.ensure(lzf.n + 1));
    AI_TRY(pB[0].ensure(lzc.n + 1));
    AI_TRY(pB[1].ensure(lzc.n + 1));
    Pack pk(ctx->stage + AI_STAGE_BYTES / 4, AI_STAGE_BYTES / 4 - 16384);  // the tail holds the slab table
    pk.add(&s_mode.p, mode.data(), (size_t)S_ + 1);
    pk.add(&segrange.p, h_segrange.data(), (size_t)S_ + 1);
    pk.add(&lzf.d.p, lzf.h.data(), (size_t)lzf.n);
    pk.add(&lzc.d.p, lzc.h.data(), (size_t)lzc.n);
    pk.add(&cranges.p, h_cranges.data(), (size_t)lzc.n);
    pk.add(&factive.p, ones.data(), (size_t)lzf.n);
    pk.add(&cactive.p, ones.data(), (size_t)lzc.n);
    AI_TRY(pk.flush(blobB, st));
    // scaled matrix, 1 / deg and u1 are only read by the Lanczos kernels: rows of null-vector segments skip them
    if (lzf.n > 0) {
      hipLaunchKernelGGL(k_scale, dim3(lzf.n), dim3(AI_BLOCK), 0, st, (const Task*)lzf.d.p, rowptr, col, wraw, deg.p, sinv.p, s_vol.p, wm.p,
                         sinv2.p, u1.p);
      AI_KERNEL_CHECK();
    }
    if (enc_min_tasks < 0) {
      const char* v = getenv("AI_SPMV_STAGE_MIN");
      enc_min_tasks = v ? atoi(v) : 0;
    }
    enc_ready = false;
    if (lzf.n > 0 && lzf.n >= enc_min_tasks) {
      AI_TRY(enc.ensure((size_t)lzf.n));
      AI_HIP(hipMemsetAsync(enc_ctr.p, 0, sizeof(int32_t), st));
      hipLaunchKernelGGL(k_lz_encode, dim3(lzf.n), dim3(AI_BLOCK), 0, st, (const Task*)lzf.d.p, rowptr, col, ucol_cap, enc_ctr.p, ucol.p, lidx.p,
                         enc.p);
      AI_KERNEL_CHECK();
      enc_ready = true;
    }
    return AI_OK;
  }

  int null_vectors() {
    const int S_ = S();
    bool any = false;
    for (auto& s : segs) any |= (s.mode == 1);
    if (!any) return AI_OK;
    const unsigned gr = (unsigned)((na + AI_BLOCK - 1) / AI_BLOCK);
    AI_HIP(hipMemsetAsync(rcnt.p, 0, (size_t)na * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_null_rootcount, dim3(coarse.n), dim3(AI_BLOCK), 0, st, coarse.d.p, s_mode.p, (const int32_t*)parent, rcnt.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_null_rootvals, dim3(gr), dim3(AI_BLOCK), 0, st, (const int32_t*)parent, rcnt.p, na, rc.p);
    AI_KERNEL_CHECK();
    AI_TRY(ai_exclusive_scan_i32(st, rc.p, ex.p, na, scantmp.p));
    hipLaunchKernelGGL(k_null_side, dim3(coarse.n), dim3(AI_BLOCK), 0, st, coarse.d.p, s_mode.p, seg_start.p, (const int32_t*)parent, rcnt.p,
                       ex.p, deg.p, side.p, pvolA.p, pvolB.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_seg_sum, dim3(S_), dim3(AI_BLOCK), 0, st, coarse.d_seg0.p, pvolA.p, s_volA.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_seg_sum, dim3(S_), dim3(AI_BLOCK), 0, st, coarse.d_seg0.p, pvolB.p, s_volB.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_null_vec, dim3(coarse.n), dim3(AI_BLOCK), 0, st, coarse.d.p, s_mode.p, s_volA.p, s_volB.p, deg.p, side.p, ev.p);
    AI_KERNEL_CHECK();
    for (auto& s : segs) stats.null_solves += (s.mode == 1);
    return AI_OK;
  }

  double* vec(int j) { return slabs[(size_t)j / AI_SLAB_VECS] + (size_t)(j % AI_SLAB_VECS) * slab_stride; }
  int ensure_vec(int j) {
    while ((size_t)j / AI_SLAB_VECS >= slabs.size()) {
      const size_t bytes = (size_t)AI_SLAB_VECS * slab_stride * sizeof(double);
      double* p = nullptr;
      if (ai_arena* a = ai_current_arena()) {
        p = (double*)a->alloc(bytes);
      } else if (hipMalloc((void**)&p, bytes) == hipSuccess) {
        owned_slabs.push_back(p);
      } else {
        p = nullptr;
      }
      if (!p) {
        ai_set_error("Lanczos vector slab %zu (%zu bytes) could not be allocated", slabs.size(), bytes);
        return AI_ERR_OOM;
      }
      slabs.push_back(p);
    }
    return AI_OK;
  }

  // e0 / e1 (optional): HIP events that receive this dispatch's own start / stop timestamps
  int spmv_variant = -1;
  template <int LPR, int ILP, bool NOGATHER = false>
  int launch_spmv_t(int j, hipEvent_t e0, hipEvent_t e1) {
    static_assert((AI_BLOCK / LPR) * ILP == AI_FINE_ROWS, "a block covers exactly one fine task");
    if (e0) {
      hipExtLaunchKernelGGL((k_lz_spmv_t<LPR, ILP, NOGATHER>), dim3(lzf.n), dim3(AI_BLOCK), 0, st, e0, e1, 0, (const Task*)lzf.d.p, (const int32_t*)factive.p,
                            lzf.n, rowptr, col, (const double*)wm.p, (const double*)sinv2.p, (const double*)vec(j), Y.p, pA.p);
    } else {
      hipLaunchKernelGGL((k_lz_spmv_t<LPR, ILP, NOGATHER>), dim3(lzf.n), dim3(AI_BLOCK), 0, st, (const Task*)lzf.d.p, (const int32_t*)factive.p, lzf.n,
                         rowptr, col, (const double*)wm.p, (const double*)sinv2.p, (const double*)vec(j), Y.p, pA.p);
    }
    AI_KERNEL_CHECK();
    return AI_OK;
  }
  int launch_spmv(int j, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    if (spmv_variant < 0) {
      const char* v = getenv("AI_SPMV_VARIANT");
      spmv_variant = v ? atoi(v) : 0;
    }
    switch (spmv_variant) {
      case 9: return launch_spmv_t<16, AI_ROW_ILP, true>(j, e0, e1);  // timing only: no gather of R_j (wrong results)
      case 1: return launch_spmv_t<16, AI_ROW_ILP>(j, e0, e1);  // plain gather from global memory
      default: break;
    }
    if (!enc_ready) return launch_spmv_t<16, AI_ROW_ILP>(j, e0, e1);
    if (e0) {
      hipExtLaunchKernelGGL((k_lz_spmv_x<16, AI_ROW_ILP>), dim3(lzf.n), dim3(AI_BLOCK), 0, st, e0, e1, 0, (const Task*)lzf.d.p,
                            (const int32_t*)factive.p, lzf.n, (const TaskEnc*)enc.p, (const int32_t*)ucol.p, (const uint16_t*)lidx.p, rowptr, col,
                            (const double*)wm.p, (const double*)sinv2.p, (const double*)vec(j), Y.p, pA.p, (unsigned long long*)nullptr);
    } else {
      hipLaunchKernelGGL((k_lz_spmv_x<16, AI_ROW_ILP>), dim3(lzf.n), dim3(AI_BLOCK), 0, st, (const Task*)lzf.d.p, (const int32_t*)factive.p, lzf.n,
                         (const TaskEnc*)enc.p, (const int32_t*)ucol.p, (const uint16_t*)lidx.p, rowptr, col, (const double*)wm.p,
                         (const double*)sinv2.p, (const double*)vec(j), Y.p, pA.p,
                         (clock_spmv && tblock.p) ? tblock.p : (unsigned long long*)nullptr);
    }
    AI_KERNEL_CHECK();
    return AI_OK;
  }

  // Lock-step Lanczos over every mode-0 segment, then Ritz vectors into ev.
  // theta_out / iters_out / resid_out (optional): values of segment 0.
  int lanczos(double* theta_out, int* iters_out, double* resid_out) {
    const int S_ = S();
    int nl = 0, max_n = 0, min_n = 1 << 30;
    std::set<int> forced;  // steps at which some segment reaches its own dimension
    for (auto& s : segs)
      if (s.mode == 0) {
        ++nl;
        max_n = std::max(max_n, s.n);
        min_n = std::min(min_n, s.n);
        forced.insert(std::min(opt.max_iter, s.n - 1));
      }
    if (nl == 0) return AI_OK;
    stats.lanczos_solves += nl;
    mcap = std::max(1, std::min(opt.max_iter, max_n - 1));
    {
      // sized once for the usual frontier (children are > 1 % of the chunk, so <= ~100 segments)
      const size_t scap = (size_t)std::max(S_, 128), mc = (size_t)std::max(mcap, opt.max_iter);
      AI_TRY(alpha_hist.ensure(scap * mc));
      AI_TRY(b_hist.ensure(scap * (mc + 1)));
      AI_TRY(g_hist.ensure(scap * (mc + 1)));
      AI_TRY(coef.ensure(scap * mc));
      AI_TRY(bnew_buf.ensure((size_t)AI_CHECK_DEPTH * (scap + 1)));
    }
    // vectors live in slabs whose stride is the row count of the first level that needs them
    // (levels only shrink)
    if (slab_stride == 0) slab_stride = (size_t)na;
    AI_HIP(hipMemsetAsync(lzres.p, 0, rescap * 24, st));  // m, frozen, resid, theta
    AI_HIP(hipMemsetAsync(slots.p, 0, AI_MAX_CHECKS * sizeof(int32_t), st));
    LzSeg L = lzseg();
    AI_TRY(ensure_vec(0));
    AI_TRY(ensure_vec(1));
    AI_HIP(hipEventRecord(ctx->ev[0], st));
    hipLaunchKernelGGL(k_lz_init, dim3(lzc.n), dim3(AI_BLOCK), 0, st, lzc.d.p, cactive.p, orig, u1.p, vec(0), pB[0].p);
    AI_KERNEL_CHECK();
    if (clock_spmv) {
      // slot j: {earliest block start, latest block end} of launch j on the device's wall clock
      AI_TRY(tstamps.ensure((size_t)2 * (mcap + 1)));
      AI_TRY(tblock.ensure((size_t)2 * (lzf.n + 1)));
      if (clock_khz == 0.0) {
        int khz = 0;
        AI_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
        clock_khz = khz > 0 ? (double)khz : 100000.0;
      }
    }
    const bool dense_checks = (min_n <= 512);
    int next_check = dense_checks ? 1 : opt.check_every;
    int steps = 0, nchecks = 0, last_check_m = 0;
    // Checks are asynchronous: the check kernel's counter is copied to pinned host memory behind
    // an event and read a few steps later, so the stream never drains.  A finished segment's
    // blocks exit at their activity flag, so steps launched past the end cost next to nothing;
    // the host still never runs more than AI_RUNAHEAD steps past an unread check.
    static const int AI_RUNAHEAD = getenv("AI_RUNAHEAD") ? atoi(getenv("AI_RUNAHEAD")) : 10;
    struct Pending { int slot, m, ev; };
    std::vector<Pending> pending;  // FIFO of in-flight checks
    size_t phead = 0;
    bool done = false;
    auto reap = [&](bool block) -> int {
      while (phead < pending.size()) {
        const Pending pc = pending[phead];
        hipEvent_t e = ctx->chk_ev[pc.ev];
        if (block) {
          AI_HIP(hipEventSynchronize(e));
        } else {
          hipError_t q = hipEventQuery(e);
          if (q == hipErrorNotReady) break;
          AI_HIP(q);
        }
        if (ctx->pinned[pc.slot % AI_PINNED_INTS] == 0) done = true;
        ++phead;
        block = false;
      }
      return AI_OK;
    };
    for (int j = 0; j < mcap && !done; ++j) {
      AI_TRY(ensure_vec(j + 1));
      if (time_spmv) {
        while (evpool.size() < (size_t)2 * (j + 1)) {
          hipEvent_t e;
          AI_HIP(hipEventCreate(&e));
          evpool.push_back(e);
        }
        AI_TRY(launch_spmv(j, evpool[2 * j], evpool[2 * j + 1]));
      } else {
        AI_TRY(launch_spmv(j));
        if (clock_spmv && enc_ready && spmv_variant == 0) {
          hipLaunchKernelGGL(k_ts_reduce, dim3(1), dim3(AI_BLOCK), 0, st, (const unsigned long long*)tblock.p, lzf.n, tstamps.p + 2 * (size_t)j);
          AI_KERNEL_CHECK();
        }
      }
      hipLaunchKernelGGL(k_lz_update, dim3(lzc.n), dim3(AI_BLOCK), 0, st, lzc.d.p, cranges.p, L, j, (const double*)pA.p,
                         (const double2*)pB[j & 1].p, pB[(j + 1) & 1].p, u1.p, (const double*)Y.p, (const double*)vec(j),
                         (const double*)vec(j > 0 ? j - 1 : 0), vec(j + 1));
      AI_KERNEL_CHECK();
      ++steps;
      const int m = j + 1;
      const bool check = dense_checks || (m >= next_check) || (m == mcap) || forced.count(m) > 0;
      if (check && nchecks < AI_MAX_CHECKS) {
        // at most AI_CHECK_DEPTH checks in flight (their events and pinned slots are recycled)
        if (pending.size() - phead >= (size_t)AI_CHECK_DEPTH - 1) AI_TRY(reap(true));
        const int cd = nchecks % AI_CHECK_DEPTH;
        double* bn = bnew_buf.p + (size_t)cd * (S_ + 1);
        hipLaunchKernelGGL(k_lz_bnew, dim3(S_), dim3(64), 0, st, segrange.p, s_mode.p, s_frozen.p, (const double2*)pB[(j + 1) & 1].p, bn);
        AI_KERNEL_CHECK();
        AI_HIP(hipEventRecord(ctx->chk_ev1[cd], st));
        AI_HIP(hipStreamWaitEvent(ctx->side, ctx->chk_ev1[cd], 0));
        const int with_rb = (size_t)3 * m * sizeof(double) <= (size_t)64 * 1024;
        hipLaunchKernelGGL(k_lz_check, dim3(S_), dim3(AI_CHECK_THREADS), (size_t)(with_rb ? 3 : 2) * m * sizeof(double), ctx->side, seg_start.p, segrange.p,
                           s_mode.p, L, (const double*)bn, m, opt.tol, opt.max_iter, m - last_check_m, rowptr, s_theta.p, s_resid.p,
                           slots.p + nchecks, work.p, with_rb);
        AI_KERNEL_CHECK();
        AI_HIP(hipMemcpyAsync(&ctx->pinned[nchecks % AI_PINNED_INTS], slots.p + nchecks, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->side));
        AI_HIP(hipEventRecord(ctx->chk_ev[cd], ctx->side));
        pending.push_back(Pending{nchecks, m, cd});
        ++nchecks;
        last_check_m = m;
        if (m >= next_check) next_check = m + std::max(opt.check_every, (m / 8 / opt.check_every) * opt.check_every);
      }
      // read whatever has arrived; block only when the oldest unread check is AI_RUNAHEAD steps old
      const bool must = (phead < pending.size()) && (m - pending[phead].m >= AI_RUNAHEAD || m == mcap);
      AI_TRY(reap(must));
    }
    while (phead < pending.size()) AI_TRY(reap(true));
    stats.lanczos_steps += steps;
    if (getenv("AI_NCUT_DEBUG")) {
      fprintf(stderr, "[ai_ncut] level %lld: segments %d (lanczos %d, rows %d..%d), active rows %d, steps %d, checks %d\n",
              (long long)stats.levels, S_, nl, min_n, max_n, na, steps, nchecks);
    }
    if (clock_spmv && enc_ready && spmv_variant == 0) {
      std::vector<unsigned long long> hts((size_t)2 * steps);
      AI_HIP(hipMemcpyAsync(hts.data(), tstamps.p, hts.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      AI_HIP(hipStreamSynchronize(st));
      for (int j = 0; j < steps; ++j)
        if (hts[2 * j + 1] > hts[2 * j]) stats.ms_spmv += (double)(hts[2 * j + 1] - hts[2 * j]) / clock_khz;
    }
    if (time_spmv) {
      AI_HIP(hipStreamSynchronize(st));
      for (int j = 0; j < steps; ++j) {
        float e = 0.f;
        AI_HIP(hipEventElapsedTime(&e, evpool[2 * j], evpool[2 * j + 1]));
        stats.ms_spmv += e;
      }
    }
    // ---- Ritz coefficients on the host (tiny), Ritz vectors on the device
    std::vector<int32_t> h_m(S_), h_frozen(S_);
    std::vector<double> h_a((size_t)S_ * mcap), h_b((size_t)S_ * (mcap + 1)), h_g((size_t)S_ * (mcap + 1)), h_coef((size_t)S_ * mcap, 0.0),
        h_cu(S_, 0.0), h_resid(S_, 0.0), h_theta(S_, 0.0);
    char* dl = ctx->stage + AI_STAGE_BYTES / 2;
    AI_HIP(hipMemcpyAsync(dl, lzres.p, rescap * 24, hipMemcpyDeviceToHost, st));
    {
      // only the columns the steps have filled
      const size_t wa = (size_t)std::min(steps, mcap) * sizeof(double), wb = (size_t)std::min(steps + 1, mcap + 1) * sizeof(double);
      AI_HIP(hipMemcpy2DAsync(h_a.data(), (size_t)mcap * sizeof(double), alpha_hist.p, (size_t)mcap * sizeof(double), wa, S_, hipMemcpyDeviceToHost, st));
      AI_HIP(hipMemcpy2DAsync(h_b.data(), (size_t)(mcap + 1) * sizeof(double), b_hist.p, (size_t)(mcap + 1) * sizeof(double), wb, S_, hipMemcpyDeviceToHost, st));
      AI_HIP(hipMemcpy2DAsync(h_g.data(), (size_t)(mcap + 1) * sizeof(double), g_hist.p, (size_t)(mcap + 1) * sizeof(double), wb, S_, hipMemcpyDeviceToHost, st));
    }
    AI_HIP(hipStreamSynchronize(st));
    for (int s = 0; s < S_; ++s) {
      h_m[s] = ((const int32_t*)dl)[s];
      h_frozen[s] = ((const int32_t*)dl)[rescap + s];
      h_resid[s] = ((const double*)(dl + rescap * 8))[s];
      h_theta[s] = ((const double*)(dl + rescap * 16))[s];
    }
    int max_m = 0;
    std::vector<double> sv;
    for (int s = 0; s < S_; ++s) {
      if (segs[s].mode != 0) continue;
      int m = h_m[s];
      if (!h_frozen[s]) {
        // the check budget ran out before this segment was frozen: use everything computed
        m = steps;
        h_m[s] = m;
        ++stats.unconverged;
      }
      if (m <= 0) {
        ai_set_error("internal: Lanczos segment %d finished with an empty tridiagonal matrix", s);
        return AI_ERR_INTERNAL;
      }
      max_m = std::max(max_m, m);
      const double* a = &h_a[(size_t)s * mcap];
      const double* b = &h_b[(size_t)s * (mcap + 1)];
      const double* g = &h_g[(size_t)s * (mcap + 1)];
      double theta = 0.0;
      tridiag_top(a, b, m, (h_frozen[s] && h_m[s] == m) ? &h_theta[s] : nullptr, &theta, sv);
      double cu = 0.0;
      for (int j = 0; j < m; ++j) {
        const double c = sv[j] / b[j];  // v_j = (R_j - g_j u1) / b_j
        h_coef[(size_t)s * mcap + j] = c;
        cu -= c * g[j];
      }
      h_cu[s] = cu;
      // a residual above tol is a failure only if T is smaller than the segment's own dimension
      if (h_frozen[s] && h_resid[s] > opt.tol && m < segs[s].n - 1) ++stats.unconverged;
      stats.max_resid = std::max(stats.max_resid, h_resid[s]);
      if (s == 0) {
        if (theta_out) *theta_out = theta;
        if (iters_out) *iters_out = m;
        if (resid_out) *resid_out = h_resid[s];
      }
    }
    AI_HIP(hipMemcpyAsync(s_m.p, h_m.data(), (size_t)S_ * sizeof(int32_t), hipMemcpyHostToDevice, st));
    AI_HIP(hipMemcpy2DAsync(coef.p, (size_t)mcap * sizeof(double), h_coef.data(), (size_t)mcap * sizeof(double),
                            (size_t)std::max(max_m, 1) * sizeof(double), S_, hipMemcpyHostToDevice, st));
    AI_HIP(hipMemcpyAsync(s_cu.p, h_cu.data(), (size_t)S_ * sizeof(double), hipMemcpyHostToDevice, st));
    for (int j0 = 0; j0 < max_m; j0 += AI_SLAB_VECS) {
      hipLaunchKernelGGL(k_ritz, dim3(lzc.n), dim3(AI_BLOCK), 0, st, lzc.d.p, s_mode.p, s_m.p, coef.p, mcap, s_cu.p, u1.p,
                         (const double*)slabs[(size_t)j0 / AI_SLAB_VECS], slab_stride, j0, AI_SLAB_VECS, j0 == 0 ? 1 : 0, ev.p);
      AI_KERNEL_CHECK();
    }
    AI_HIP(hipEventRecord(ctx->ev[1], st));
    AI_HIP(hipStreamSynchronize(st));  // h_coef / h_cu are read by the copies above
    float ms = 0.f;
    AI_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    stats.ms_eigen += ms;
    return AI_OK;
  }

  // ---- k1 largest eigenpairs of the deflated M (= the k1 smallest non-zero of L) of ONE connected
  // segment: Lanczos with full re-orthogonalisation.  thetas[i] descending; Ritz vectors into
  // out (k1 x out_stride, compact row order).
  DevBuf<double*> d_slabtab;
  DevBuf<double> fro_part, fro_coef, fro_cu, coefT, cuv;
  size_t slabtab_n = 0;
  int sync_slabtab() {
    if (slabtab_n == slabs.size()) return AI_OK;
    if (slabs.size() > 500) {
      ai_set_error("internal: more than 500 Lanczos slabs");
      return AI_ERR_INTERNAL;
    }
    AI_TRY(d_slabtab.ensure(512));
    // the table only ever grows by appending, so rewriting the pinned copy while an earlier
    // upload is still in flight is harmless; no synchronisation needed
    double** pin = (double**)(ctx->stage + AI_STAGE_BYTES / 2 - 8192);
    for (size_t i = 0; i < slabs.size(); ++i) pin[i] = slabs[i];
    AI_HIP(hipMemcpyAsync(d_slabtab.p, pin, slabs.size() * sizeof(double*), hipMemcpyHostToDevice, st));
    slabtab_n = slabs.size();
    return AI_OK;
  }


  int lanczos_fro(int k1, std::vector<double>& thetas, std::vector<double>& resids, double* out, size_t out_stride, int* steps_out) {
    const int S_ = S();
    if (S_ != 1 || segs[0].mode != 0) {
      ai_set_error("internal: lanczos_fro needs one connected segment");
      return AI_ERR_INTERNAL;
    }
    const int n = segs[0].n;
    mcap = std::max(1, std::min(opt.max_iter, n - 1));
    if (k1 > mcap) k1 = mcap;
    {
      const size_t scap = 128, mc = (size_t)std::max(mcap, opt.max_iter);
      AI_TRY(alpha_hist.ensure(scap * mc));
      AI_TRY(b_hist.ensure(scap * (mc + 1)));
      AI_TRY(g_hist.ensure(scap * (mc + 1)));
      AI_TRY(bnew_buf.ensure((size_t)AI_CHECK_DEPTH * (scap + 1)));
    }
    const int pitch = mcap + 1;
    AI_TRY(fro_part.ensure((size_t)lzc.n * pitch));
    AI_TRY(fro_coef.ensure((size_t)pitch));
    AI_TRY(fro_cu.ensure(8));
    if (slab_stride == 0) slab_stride = (size_t)na;
    AI_HIP(hipMemsetAsync(lzres.p, 0, rescap * 24, st));
    LzSeg L = lzseg();
    AI_TRY(ensure_vec(0));
    AI_TRY(ensure_vec(1));
    hipLaunchKernelGGL(k_lz_init, dim3(lzc.n), dim3(AI_BLOCK), 0, st, lzc.d.p, cactive.p, orig, u1.p, vec(0), pB[0].p);
    AI_KERNEL_CHECK();
    std::vector<double> h_a, h_b, h_g;
    std::vector<std::vector<double>> svec;
    int m = 0;
    bool done = false;
    const int chk = std::max(8, opt.check_every);
    int next_check = std::max(2 * k1, chk);
    for (int j = 0; j < mcap && !done; ++j) {
      AI_TRY(ensure_vec(j + 1));
      AI_TRY(sync_slabtab());
      AI_TRY(launch_spmv(j));
      hipLaunchKernelGGL(k_lz_update, dim3(lzc.n), dim3(AI_BLOCK), 0, st, lzc.d.p, cranges.p, L, j, (const double*)pA.p,
                         (const double2*)pB[j & 1].p, pB[(j + 1) & 1].p, u1.p, (const double*)Y.p, (const double*)vec(j),
                         (const double*)vec(j > 0 ? j - 1 : 0), vec(j + 1));
      AI_KERNEL_CHECK();
      // full re-orthogonalisation of R_{j+1} against v_0 .. v_j (two passes: "twice is enough")
      for (int pass = 0; pass < 2; ++pass) {
        const int nvec = j + 1;
        hipLaunchKernelGGL(k_fro_dots, dim3(lzc.n, (nvec + FRO_CH - 1) / FRO_CH), dim3(AI_BLOCK), 0, st, lzc.d.p, (const double*)vec(j + 1),
                           (double* const*)d_slabtab.p, slab_stride, nvec, fro_part.p, pitch);
        AI_KERNEL_CHECK();
        hipLaunchKernelGGL(k_fro_coef, dim3(1), dim3(AI_BLOCK), 0, st, lzc.n, (const double*)fro_part.p, pitch, nvec, (const double*)g_hist.p,
                           (const double*)b_hist.p, (const double2*)pB[(j + 1) & 1].p, fro_coef.p, fro_cu.p);
        AI_KERNEL_CHECK();
        hipLaunchKernelGGL(k_fro_apply, dim3(lzc.n), dim3(AI_BLOCK), 0, st, lzc.d.p, vec(j + 1), (double* const*)d_slabtab.p, slab_stride, nvec,
                           (const double*)fro_coef.p, (const double*)fro_cu.p, u1.p, pB[(j + 1) & 1].p);
        AI_KERNEL_CHECK();
      }
      m = j + 1;
      if (m >= next_check || m == mcap) {
        next_check = m + chk;
        // T_m and b_m to the host; the innermost wanted pair converges last: look at it first
        double* bn = bnew_buf.p;
        hipLaunchKernelGGL(k_lz_bnew, dim3(1), dim3(64), 0, st, segrange.p, s_mode.p, s_frozen.p, (const double2*)pB[(j + 1) & 1].p, bn);
        AI_KERNEL_CHECK();
        h_a.resize(m);
        h_b.resize(m + 1);
        AI_HIP(hipMemcpyAsync(h_a.data(), alpha_hist.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st));
        AI_HIP(hipMemcpyAsync(h_b.data(), b_hist.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st));
        AI_HIP(hipMemcpyAsync(&h_b[m], bn, sizeof(double), hipMemcpyDeviceToHost, st));
        int32_t frozen = 0;
        AI_HIP(hipMemcpyAsync(&frozen, s_frozen.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        AI_HIP(hipStreamSynchronize(st));
        const int kk = std::min(k1, m);
        double glo = -1e300, ghi = -1e300;
        for (int i = 0; i < m; ++i) {
          glo = std::max(glo, h_a[i]);
          ghi = std::max(ghi, h_a[i] + (i > 0 ? fabs(h_b[i]) : 0.0) + (i + 1 < m ? fabs(h_b[i + 1]) : 0.0));
        }
        double gmin = 1e300;
        for (int i = 0; i < m; ++i) gmin = std::min(gmin, h_a[i] - (i > 0 ? fabs(h_b[i]) : 0.0) - (i + 1 < m ? fabs(h_b[i + 1]) : 0.0));
        ghi += 1e-14 * std::max(fabs(ghi), 1.0);
        gmin -= 1e-14 * std::max(fabs(gmin), 1.0);
        std::vector<double> sv;
        std::vector<std::vector<double>> none;
        std::vector<int> nocl;
        const double th_in = tridiag_eigval(h_a.data(), h_b.data(), m, kk - 1, gmin, ghi);
        tridiag_eigvec(h_a.data(), h_b.data(), m, th_in, none, nocl, sv);
        const double r_in = fabs(h_b[m] * sv[m - 1]);
        if (getenv("AI_NCUT_DEBUG")) fprintf(stderr, "[ai_eigs] m=%d innermost theta=%.12f resid=%.3e\n", m, th_in, r_in);
        if ((r_in <= opt.tol && kk == k1) || m == mcap || frozen) done = true;
        if (frozen) {
          // Krylov space exhausted inside the update kernel: T stops at the size recorded there
          int32_t fm = m;
          AI_HIP(hipMemcpyAsync(&fm, s_m.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
          AI_HIP(hipStreamSynchronize(st));
          m = std::max(1, std::min(m, (int)fm));
        }
      }
    }
    if (steps_out) *steps_out = m;
    // ---- all k1 Ritz pairs of T_m on the host
    h_a.resize(m);
    h_b.resize(m + 1);
    h_g.resize(m);
    AI_HIP(hipMemcpyAsync(h_a.data(), alpha_hist.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st));
    AI_HIP(hipMemcpyAsync(h_b.data(), b_hist.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st));
    AI_HIP(hipMemcpyAsync(h_g.data(), g_hist.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st));
    {
      hipLaunchKernelGGL(k_lz_bnew, dim3(1), dim3(64), 0, st, segrange.p, s_mode.p, s_frozen.p, (const double2*)pB[m & 1].p, bnew_buf.p);
      AI_KERNEL_CHECK();
      AI_HIP(hipMemcpyAsync(&h_b[m], bnew_buf.p, sizeof(double), hipMemcpyDeviceToHost, st));
    }
    AI_HIP(hipStreamSynchronize(st));
    const int kk = std::min(k1, m);
    double ghi = -1e300, gmin = 1e300, nrm = 0.0;
    for (int i = 0; i < m; ++i) {
      const double rad = (i > 0 ? fabs(h_b[i]) : 0.0) + (i + 1 < m ? fabs(h_b[i + 1]) : 0.0);
      ghi = std::max(ghi, h_a[i] + rad);
      gmin = std::min(gmin, h_a[i] - rad);
      nrm = std::max(nrm, fabs(h_a[i]) + rad);
    }
    ghi += 1e-14 * std::max(fabs(ghi), 1.0);
    gmin -= 1e-14 * std::max(fabs(gmin), 1.0);
    thetas.assign(kk, 0.0);
    resids.assign(kk, 0.0);
    svec.assign(kk, std::vector<double>());
    std::vector<double> h_coefT((size_t)m * RITZ_MAXK, 0.0), h_cuv(RITZ_MAXK, 0.0);
    for (int i = 0; i < kk; ++i) {
      thetas[i] = tridiag_eigval(h_a.data(), h_b.data(), m, i, gmin, ghi);
      std::vector<int> cluster;
      for (int c = 0; c < i; ++c)
        if (fabs(thetas[c] - thetas[i]) <= 1e-3 * nrm) cluster.push_back(c);
      tridiag_eigvec(h_a.data(), h_b.data(), m, thetas[i], svec, cluster, svec[i]);
      resids[i] = fabs(h_b[m] * svec[i][m - 1]);
      double cu = 0.0;
      for (int j = 0; j < m; ++j) {
        const double c = svec[i][j] / h_b[j];
        h_coefT[(size_t)j * RITZ_MAXK + i] = c;
        cu -= c * h_g[j];
      }
      h_cuv[i] = cu;
    }
    AI_TRY(coefT.ensure(h_coefT.size()));
    AI_TRY(cuv.ensure(RITZ_MAXK));
    AI_HIP(hipMemcpyAsync(coefT.p, h_coefT.data(), h_coefT.size() * sizeof(double), hipMemcpyHostToDevice, st));
    AI_HIP(hipMemcpyAsync(cuv.p, h_cuv.data(), RITZ_MAXK * sizeof(double), hipMemcpyHostToDevice, st));
    for (int j0 = 0; j0 < m; j0 += AI_SLAB_VECS) {
      hipLaunchKernelGGL(k_ritz_multi, dim3(lzc.n), dim3(AI_BLOCK), 0, st, lzc.d.p, kk, (const double*)coefT.p, (const double*)cuv.p, u1.p,
                         (const double*)slabs[(size_t)j0 / AI_SLAB_VECS], slab_stride, j0, AI_SLAB_VECS, m, j0 == 0 ? 1 : 0, out, out_stride);
      AI_KERNEL_CHECK();
    }
    AI_HIP(hipStreamSynchronize(st));
    return AI_OK;
  }

  // min/max, bins, 10 costs, decision -> host vectors (raw = 1: ev used as given)
  // skip_disconnected: segments of mode 1 have no eigenvector (they are split by components) and are left out
  int sweep(double T, int raw, bool skip_disconnected, std::vector<int32_t>& h_split, std::vector<int32_t>& h_ntrue, std::vector<double>& h_mcut) {
    const int S_ = S();
    const int32_t* md = skip_disconnected ? (const int32_t*)s_mode.p : nullptr;
    AI_HIP(hipEventRecord(ctx->ev[2], st));
    hipLaunchKernelGGL(k_minmax, dim3(coarse.n), dim3(AI_BLOCK), 0, st, coarse.d.p, md, ev.p, orig, pmm.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_minmax_final, dim3((S_ + 63) / 64), dim3(64), 0, st, coarse.d_seg0.p, md, pmm.p, S_, raw, s_scale.p, s_nosplit.p, s_thr.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_bin, dim3(coarse.n), dim3(AI_BLOCK), 0, st, coarse.d.p, s_nosplit.p, s_scale.p, s_thr.p, ev.p, bin.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_sweep, dim3(fine.n), dim3(AI_BLOCK), 0, st, fine.d.p, s_nosplit.p, rowptr, col, wraw, deg.p, bin.p, psweep.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_sweep_final, dim3(S_), dim3(AI_BLOCK), 0, st, fine.d_seg0.p, s_nosplit.p, psweep.p, T, s_costs.p, s_kstar.p, s_split.p,
                       s_ntrue.p, s_mcut.p);
    AI_KERNEL_CHECK();
    AI_HIP(hipEventRecord(ctx->ev[3], st));
    h_split.resize(S_);
    h_ntrue.resize(S_);
    h_mcut.resize(S_);
    {
      char* dl = ctx->stage + AI_STAGE_BYTES / 2;
      AI_HIP(hipMemcpyAsync(dl, resblob.p, rescap * 16, hipMemcpyDeviceToHost, st));
      AI_HIP(hipStreamSynchronize(st));
      const int32_t* sp = (const int32_t*)dl;
      const int32_t* nt = sp + rescap;
      const double* mc = (const double*)(nt + rescap);
      for (int s = 0; s < S_; ++s) {
        h_split[s] = sp[s];
        h_ntrue[s] = nt[s];
        h_mcut[s] = mc[s];
      }
    }
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

#include "ai_chfsi.inc"

// ----------------------------------------------------------------------------- C ABI
static void fill_opts(Solver& S, const ai_ncut_opts* opts) {
  if (!opts) return;
  if (opts->tol > 0.0) S.opt.tol = opts->tol;
  if (opts->max_iter > 0) S.opt.max_iter = std::min(opts->max_iter, 4000);  // T_m must fit the check kernel's 64 KB of LDS
  if (opts->check_every > 0) S.opt.check_every = opts->check_every;
  S.time_spmv = (opts->reserved & 1) != 0;
  S.clock_spmv = !S.time_spmv && (opts->reserved & 2) != 0;
}

// The recursion over one graph that holds `nchunks` independent chunks back to back (rows
// off[c] .. off[c+1]); every chunk starts as its own root segment and keeps its own original
// point count for the split_lim gate.  csr->orig holds chunk-LOCAL point ids.
static int ncut_impl(ai_ctx* ctx, const ai_csr* csr, int nchunks, const int64_t* off, const int64_t* n_orig, double T, double split_lim,
                     const ai_ncut_opts* opts, int32_t* const* labels_out, int32_t* n_groups, ai_ncut_stats* stats_out, double t0) {
  Solver S(ctx, csr);
  fill_opts(S, opts);
  const int n = (int)csr->n;
  AI_TRY(S.begin(false));
  std::vector<int32_t> leaf_starts;
  for (int c = 0; c < nchunks; ++c) {
    const int nc = (int)(off[c + 1] - off[c]);
    if (eligible(nc, n_orig[c], split_lim))
      S.segs.push_back(SegHost{(int)off[c], nc, (int)off[c], 0, 1, c});
    else
      leaf_starts.push_back((int32_t)off[c]);
  }
  // compact order of the first level = rows of the eligible chunks, back to back
  if ((int)S.segs.size() != nchunks) {
    int pos = 0;
    for (auto& sg : S.segs) {
      if (sg.start != pos) {
        ai_set_error("ai_ncut_batch: a chunk too small to be split (n <= 2 or below split_lim) must come after the others");
        return AI_ERR_BAD_ARG;
      }
      pos += sg.n;
    }
    S.na = pos;
  }
  hipStream_t st = ctx->stream;
  std::vector<int32_t> h_split, h_ntrue;
  std::vector<double> h_mcut;
  bool pending_rebuild = false;
  while (S.S() > 0) {
    ++S.stats.levels;
    AI_TRY(S.build_tasks());
    AI_HIP(hipEventRecord(ctx->ev[4], st));
    AI_TRY(S.prepare(true));  // synchronises: the previous level's rebuild events have completed
    if (pending_rebuild) {
      float ms2 = 0.f;
      AI_HIP(hipEventElapsedTime(&ms2, ctx->ev[6], ctx->ev[7]));
      S.stats.ms_rebuild += ms2;
      pending_rebuild = false;
    }
    AI_HIP(hipEventRecord(ctx->ev[5], st));
    AI_TRY(S.lanczos(nullptr, nullptr, nullptr));
    AI_TRY(S.sweep(T, 0, true, h_split, h_ntrue, h_mcut));
    {
      float ms1 = 0.f;
      AI_HIP(hipEventElapsedTime(&ms1, ctx->ev[4], ctx->ev[5]));
      S.stats.ms_rebuild += ms1;
    }
    const int S_ = S.S();
    // ---- disconnected segments: the component table (root row, rows) in row order.  The cut between whole
    // components costs exactly 0, and the reference splits iff mcut < T (normalized_cut.py:56): nothing for T <= 0.
    std::vector<int32_t> multi(S_ + 1, 0);
    bool any_multi = false;
    for (int s = 0; s < S_; ++s) {
      multi[s] = (S.segs[s].mode == 1 && T > 0.0) ? 1 : 0;
      any_multi |= (multi[s] != 0);
      S.stats.null_solves += (S.segs[s].mode == 1);
    }
    AI_HIP(hipEventRecord(ctx->ev[6], st));
    std::vector<int32_t> t_root, t_size;
    if (any_multi) {
      Pack pk(ctx->stage + 3 * (AI_STAGE_BYTES / 4), AI_STAGE_BYTES / 4);
      pk.add(&S.s_multi.p, multi.data(), (size_t)S_ + 1);
      AI_TRY(pk.flush(S.blobD, st));
      AI_HIP(hipMemsetAsync(S.rcnt.p, 0, (size_t)S.na * sizeof(int32_t), st));
      hipLaunchKernelGGL(k_null_rootcount, dim3(S.coarse.n), dim3(AI_BLOCK), 0, st, S.coarse.d.p, (const int32_t*)S.s_multi.p, (const int32_t*)S.parent,
                         S.rcnt.p);
      AI_KERNEL_CHECK();
      hipLaunchKernelGGL(k_comp_rootflag, dim3(S.coarse.n), dim3(AI_BLOCK), 0, st, S.coarse.d.p, (const int32_t*)S.s_multi.p, (const int32_t*)S.parent,
                         S.rc.p);
      AI_KERNEL_CHECK();
      AI_TRY(ai_exclusive_scan_i32(st, S.rc.p, S.ex.p, S.na, S.scantmp.p));  // ex[row] = ordinal of a root, ex[na] = components
      int32_t ncomp_all = 0;
      AI_HIP(hipMemcpyAsync(&ncomp_all, S.ex.p + S.na, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      AI_HIP(hipStreamSynchronize(st));
      if (ncomp_all < 2 || ncomp_all > S.na) {
        ai_set_error("internal: %d components in the disconnected segments of a level with %d rows", ncomp_all, S.na);
        return AI_ERR_INTERNAL;
      }
      AI_TRY(S.t_root.ensure((size_t)ncomp_all));
      AI_TRY(S.t_size.ensure((size_t)ncomp_all));
      hipLaunchKernelGGL(k_comp_table, dim3(S.coarse.n), dim3(AI_BLOCK), 0, st, S.coarse.d.p, (const int32_t*)S.s_multi.p, (const int32_t*)S.parent,
                         (const int32_t*)S.ex.p, (const int32_t*)S.rcnt.p, S.t_root.p, S.t_size.p);
      AI_KERNEL_CHECK();
      t_root.resize(ncomp_all);
      t_size.resize(ncomp_all);
      AI_HIP(hipMemcpyAsync(t_root.data(), S.t_root.p, (size_t)ncomp_all * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      AI_HIP(hipMemcpyAsync(t_size.data(), S.t_size.p, (size_t)ncomp_all * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      AI_HIP(hipStreamSynchronize(st));
    }
    // ---- children (deeper calls use split_lim = 0.01: normalized_cut.py:57-58 rely on the default)
    std::vector<SegHost> next;
    std::vector<int32_t> cA(S_, -1), cB(S_, -1);
    std::vector<int32_t> c_pos(t_root.size(), 0), c_base(t_root.size(), -1);
    int cstart = 0;
    bool any_carry = false;
    size_t ti = 0;  // next entry of the component table
    for (int s = 0; s < S_; ++s) {
      const SegHost& sg = S.segs[s];
      if (multi[s]) {
        // every connected component continues on its own, in the order of their first rows
        int off = 0;
        while (ti < t_root.size() && t_root[ti] < sg.start + sg.n) {
          const int nc = t_size[ti];
          if (t_root[ti] < sg.start || nc <= 0 || off + nc > sg.n) {
            ai_set_error("internal: component table does not tile segment %d", s);
            return AI_ERR_INTERNAL;
          }
          c_pos[ti] = sg.start + off;
          if (eligible(nc, n_orig[sg.chunk], 0.01)) {
            c_base[ti] = cstart;
            next.push_back(SegHost{cstart, nc, sg.gstart + off, 0, 0, sg.chunk});  // connected: its labels are carried
            cstart += nc;
            any_carry = true;
          } else {
            leaf_starts.push_back(sg.gstart + off);
          }
          off += nc;
          ++ti;
        }
        if (off != sg.n) {
          ai_set_error("internal: components of segment %d cover %d of %d rows", s, off, sg.n);
          return AI_ERR_INTERNAL;
        }
        continue;
      }
      if (!h_split[s]) {
        leaf_starts.push_back(sg.gstart);
        continue;
      }
      const int na_ = h_ntrue[s], nb_ = sg.n - h_ntrue[s];
      if (na_ <= 0 || nb_ <= 0) {
        ai_set_error("internal: split of segment %d produced an empty side (%d / %d)", s, na_, nb_);
        return AI_ERR_INTERNAL;
      }
      if (eligible(na_, n_orig[sg.chunk], 0.01)) {
        cA[s] = cstart;
        next.push_back(SegHost{cstart, na_, sg.gstart, 0, 1, sg.chunk});
        cstart += na_;
      } else {
        leaf_starts.push_back(sg.gstart);
      }
      if (eligible(nb_, n_orig[sg.chunk], 0.01)) {
        cB[s] = cstart;
        next.push_back(SegHost{cstart, nb_, sg.gstart + na_, 0, 1, sg.chunk});
        cstart += nb_;
      } else {
        leaf_starts.push_back(sg.gstart + na_);
      }
    }
    {
      Pack pk(ctx->stage + 3 * (AI_STAGE_BYTES / 4), AI_STAGE_BYTES / 4);
      pk.add(&S.s_childA.p, cA.data(), (size_t)S_);
      pk.add(&S.s_childB.p, cB.data(), (size_t)S_);
      pk.add(&S.s_multi.p, multi.data(), (size_t)S_ + 1);
      pk.add(&S.c_pos.p, c_pos.data(), c_pos.size());
      pk.add(&S.c_base.p, c_base.data(), c_base.size());
      AI_TRY(pk.flush(S.blobC, st));
    }
    hipLaunchKernelGGL(k_split_flags, dim3(S.coarse.n), dim3(AI_BLOCK), 0, st, S.coarse.d.p, S.s_split.p, S.s_kstar.p, S.bin.p, S.flag.p);
    AI_KERNEL_CHECK();
    AI_TRY(ai_exclusive_scan_i32(st, S.flag.p, S.fscan.p, S.na, S.scantmp.p));
    const int pp = S.pp;
    hipLaunchKernelGGL(k_partition, dim3(S.coarse.n), dim3(AI_BLOCK), 0, st, S.coarse.d.p, S.seg_start.p, S.s_gstart.p, S.s_split.p, S.s_ntrue.p,
                       S.s_childA.p, S.s_childB.p, S.flag.p, S.fscan.p, S.orig, S.final_order.p, S.map.p, S.b_orig[pp].p, (const int32_t*)S.s_multi.p);
    AI_KERNEL_CHECK();
    if (any_multi) {
      // stable sort of the level's rows by root id (rows of connected segments all carry their segment's first row)
      int bits = 1;
      while ((1ll << bits) < (long long)S.na) ++bits;
      hipLaunchKernelGGL(k_iota, dim3((unsigned)((S.na + AI_BLOCK - 1) / AI_BLOCK)), dim3(AI_BLOCK), 0, st, S.rc.p, S.na);
      AI_KERNEL_CHECK();
      size_t tmp_bytes = 0;
      AI_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const uint32_t*)S.parent, (uint32_t*)S.newcnt.p, (const int32_t*)S.rc.p, S.rcnt.p,
                                       (size_t)S.na, 0, bits, st));
      AI_TRY(S.sorttmp.ensure(tmp_bytes));
      AI_HIP(rocprim::radix_sort_pairs((void*)S.sorttmp.p, tmp_bytes, (const uint32_t*)S.parent, (uint32_t*)S.newcnt.p, (const int32_t*)S.rc.p, S.rcnt.p,
                                       (size_t)S.na, 0, bits, st));
      hipLaunchKernelGGL(k_partition_multi, dim3(S.coarse.n), dim3(AI_BLOCK), 0, st, S.coarse.d.p, (const int32_t*)S.s_multi.p, S.seg_start.p,
                         S.s_gstart.p, (const int32_t*)S.rcnt.p, (const int32_t*)S.parent, (const int32_t*)S.ex.p, (const int32_t*)S.c_pos.p,
                         (const int32_t*)S.c_base.p, S.orig, S.final_order.p, S.map.p, S.b_orig[pp].p);
      AI_KERNEL_CHECK();
    }
    int32_t* parent_next = (S.parent == S.b_parent[0].p) ? S.b_parent[1].p : S.b_parent[0].p;
    if (cstart > 0) {
      const unsigned gr = (unsigned)((S.na + AI_BLOCK - 1) / AI_BLOCK);
      const unsigned ge = (unsigned)(((int64_t)S.na * AI_LPR + AI_BLOCK - 1) / AI_BLOCK);
      if (any_carry) {
        hipLaunchKernelGGL(k_carry_parent, dim3(gr), dim3(AI_BLOCK), 0, st, (const int32_t*)S.parent, (const int32_t*)S.map.p, S.na, parent_next);
        AI_KERNEL_CHECK();
      }
      AI_HIP(hipMemsetAsync(S.newcnt.p, 0, (size_t)(cstart + 1) * sizeof(int32_t), st));
      hipLaunchKernelGGL(k_rebuild_count, dim3(ge), dim3(AI_BLOCK), 0, st, S.rowptr, S.col, S.flag.p, S.map.p, S.na, S.newcnt.p);
      AI_KERNEL_CHECK();
      AI_TRY(ai_exclusive_scan_i32(st, S.newcnt.p, S.b_rowptr[pp].p, cstart, S.scantmp.p));
      hipLaunchKernelGGL(k_rebuild_fill, dim3(ge), dim3(AI_BLOCK), 0, st, S.rowptr, S.col, S.wraw, S.flag.p, S.map.p, S.na,
                         (const int32_t*)S.b_rowptr[pp].p, S.b_col[pp].p, S.b_wraw[pp].p);
      AI_KERNEL_CHECK();
    }
    AI_HIP(hipEventRecord(ctx->ev[7], st));
    // no sync here: the next level's first host read (component counts) waits for all of this,
    // and the partition / rebuild time is collected there
    pending_rebuild = true;
    S.rowptr = S.b_rowptr[pp].p;
    S.col = S.b_col[pp].p;
    S.wraw = S.b_wraw[pp].p;
    S.orig = S.b_orig[pp].p;
    S.parent = parent_next;
    S.pp ^= 1;
    S.na = cstart;
    S.segs.swap(next);
  }
  // ---- groups = leaf ranges of the final ordering, left to right
  std::vector<int32_t> order((size_t)n);
  AI_HIP(hipMemcpyAsync(order.data(), S.final_order.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  if (pending_rebuild) {
    float ms2 = 0.f;
    AI_HIP(hipEventElapsedTime(&ms2, ctx->ev[6], ctx->ev[7]));
    S.stats.ms_rebuild += ms2;
  }
  std::sort(leaf_starts.begin(), leaf_starts.end());
  // groups of chunk c = the leaf ranges inside [off[c], off[c+1]), numbered from 0 in emission order
  size_t li = 0;
  int64_t total_groups = 0;
  for (int c = 0; c < nchunks; ++c) {
    int g = -1;
    const int32_t nloc = (int32_t)(off[c + 1] - off[c]);
    for (int64_t p = off[c]; p < off[c + 1]; ++p) {
      while (li < leaf_starts.size() && leaf_starts[li] == p) {
        ++g;
        ++li;
      }
      if (g < 0 || order[p] < 0 || order[p] >= nloc) {
        ai_set_error("internal: final ordering is not a permutation (chunk %d, position %lld)", c, (long long)p);
        return AI_ERR_INTERNAL;
      }
      labels_out[c][order[p]] = g;
    }
    n_groups[c] = g + 1;
    total_groups += g + 1;
  }
  S.stats.n_groups = total_groups;
  {
    unsigned long long hw[2] = {0, 0};
    AI_HIP(hipMemcpyAsync(hw, S.work.p, sizeof(hw), hipMemcpyDeviceToHost, st));
    AI_HIP(hipStreamSynchronize(st));
    S.stats.spmv_rows = (int64_t)hw[0];
    S.stats.spmv_nnz = (int64_t)hw[1];
  }
  S.stats.ms_total = now_ms() - t0;
  if (stats_out) *stats_out = S.stats;
  if (S.stats.unconverged > 0) {
    // the reference's eigsh raises ArpackNoConvergence (normalized_cut.py:49); labels and stats are filled all the same
    ai_set_error("ai_ncut: %lld Lanczos solve(s) reached max_iter = %d before the Ritz residual fell to %.3g (largest %.3g)",
                 (long long)S.stats.unconverged, S.opt.max_iter, S.opt.tol, S.stats.max_resid);
    return AI_ERR_NO_CONVERGENCE;
  }
  return AI_OK;
}

extern "C" int ai_ncut(ai_ctx* ctx, const ai_csr* csr, int64_t num_points_orig, double T, double split_lim, const ai_ncut_opts* opts,
                       int32_t* labels_out, int32_t* n_groups, ai_ncut_stats* stats_out) {
  if (!ctx || !csr || !labels_out || !n_groups || num_points_orig < 0) {
    ai_set_error("ai_ncut: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_CHECK_GRAPH(csr, "ai_ncut");
  AI_HIP(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  ArenaScope arena_scope(&ctx->arena);  // declared before the solver: its buffers die first
  const int64_t off[2] = {0, csr->n};
  int32_t* lab[1] = {labels_out};
  return ncut_impl(ctx, csr, 1, off, &num_points_orig, T, split_lim, opts, lab, n_groups, stats_out, t0);
}

extern "C" int ai_ncut_batch(ai_ctx* ctx, const ai_csr* const* graphs, int32_t count, const int64_t* num_points_orig, double T,
                             double split_lim, const ai_ncut_opts* opts, int32_t* const* labels_out, int32_t* n_groups,
                             ai_ncut_stats* stats_out) {
  if (!ctx || !graphs || count < 1 || !num_points_orig || !labels_out || !n_groups) {
    ai_set_error("ai_ncut_batch: bad argument");
    return AI_ERR_BAD_ARG;
  }
  int64_t N = 0, E = 0;
  for (int c = 0; c < count; ++c) {
    if (!graphs[c] || !labels_out[c] || num_points_orig[c] < 0) {
      ai_set_error("ai_ncut_batch: null graph / label buffer at position %d", c);
      return AI_ERR_BAD_ARG;
    }
    AI_CHECK_GRAPH(graphs[c], "ai_ncut_batch");
    N += graphs[c]->n;
    E += graphs[c]->nnz;
  }
  if (N >= ((int64_t)1 << 30) || E >= ((int64_t)1 << 31)) {
    ai_set_error("ai_ncut_batch: %lld rows / %lld entries exceed the int32 index range of this build", (long long)N, (long long)E);
    return AI_ERR_BAD_ARG;
  }
  AI_HIP(hipSetDevice(ctx->device));
  const double t0 = now_ms();
  ArenaScope arena_scope(&ctx->arena);
  hipStream_t st = ctx->stream;
  // The frontier's compact order starts with the rows of the chunks that can be split at all (normalized_cut.py:39-40);
  // chunks that cannot (n <= 2 or below split_lim) are placed behind them, whatever their position in the call.
  std::vector<int> order;
  for (int pass = 0; pass < 2; ++pass)
    for (int c = 0; c < count; ++c)
      if (eligible((int)graphs[c]->n, num_points_orig[c], split_lim) == (pass == 0)) order.push_back(c);
  std::vector<int64_t> norig_p(count);
  std::vector<int32_t*> labels_p(count);
  std::vector<int32_t> ngroups_p(count, 0);
  // one block-diagonal graph: the chunks back to back (column ids shifted by the chunk's first row)
  DevBuf<int32_t> rp, cl, og;
  DevBuf<double> vl;
  AI_TRY(rp.alloc((size_t)N + 1));
  AI_TRY(cl.alloc((size_t)E));
  AI_TRY(vl.alloc((size_t)E));
  AI_TRY(og.alloc((size_t)N));
  std::vector<int64_t> off(count + 1, 0);
  int64_t eoff = 0;
  for (int c = 0; c < count; ++c) {
    const ai_csr* g = graphs[order[c]];
    norig_p[c] = num_points_orig[order[c]];
    labels_p[c] = labels_out[order[c]];
    const int64_t n = g->n, e = g->nnz;
    off[c + 1] = off[c] + n;
    hipLaunchKernelGGL(k_offset_copy, dim3((unsigned)((n + 1 + AI_BLOCK - 1) / AI_BLOCK)), dim3(AI_BLOCK), 0, st, rp.p + off[c],
                       (const int32_t*)g->rowptr, n + 1, (int32_t)eoff);
    AI_KERNEL_CHECK();
    if (e) {
      hipLaunchKernelGGL(k_offset_copy, dim3((unsigned)((e + AI_BLOCK - 1) / AI_BLOCK)), dim3(AI_BLOCK), 0, st, cl.p + eoff, (const int32_t*)g->col,
                         e, (int32_t)off[c]);
      AI_KERNEL_CHECK();
      AI_HIP(hipMemcpyAsync(vl.p + eoff, g->val, (size_t)e * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(k_offset_copy, dim3((unsigned)((n + AI_BLOCK - 1) / AI_BLOCK)), dim3(AI_BLOCK), 0, st, og.p + off[c],
                       (const int32_t*)g->orig, n, 0);
    AI_KERNEL_CHECK();
    eoff += e;
  }
  ai_csr merged;
  merged.n = N;
  merged.nnz = E;
  merged.rowptr = rp.p;
  merged.col = cl.p;
  merged.val = vl.p;
  merged.orig = og.p;
  merged.device = ctx->device;
  const int rc = ncut_impl(ctx, &merged, count, off.data(), norig_p.data(), T, split_lim, opts, labels_p.data(), ngroups_p.data(), stats_out, t0);
  for (int c = 0; c < count; ++c) n_groups[order[c]] = ngroups_p[c];
  return rc;
}

extern "C" int ai_fiedler(ai_ctx* ctx, const ai_csr* csr, const ai_ncut_opts* opts, double* lambda2, double* ev_out, int32_t* iters,
                          double* resid) {
  if (!ctx || !csr || !ev_out) {
    ai_set_error("ai_fiedler: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_CHECK_GRAPH(csr, "ai_fiedler");
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);  // declared before the solver: its buffers die first
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
  hipLaunchKernelGGL(k_minmax, dim3(S.coarse.n), dim3(AI_BLOCK), 0, st, S.coarse.d.p, (const int32_t*)nullptr, S.ev.p, S.orig, S.pmm.p);
  AI_KERNEL_CHECK();
  hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(64), 0, st, S.coarse.d_seg0.p, (const int32_t*)nullptr, S.pmm.p, 1, 0, S.s_scale.p, S.s_nosplit.p,
                     S.s_thr.p);
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
  AI_CHECK_GRAPH(csr, "ai_sweep");
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);  // declared before the solver: its buffers die first
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
  std::vector<double> mcs;
  AI_TRY(S.sweep(INFINITY, 1, false, sp, nt, mcs));
  AI_HIP(hipMemcpyAsync(costs, S.s_costs.p, AI_NUM_CUTS * sizeof(double), hipMemcpyDeviceToHost, st));
  int32_t ks = 0;
  AI_HIP(hipMemcpyAsync(&ks, S.s_kstar.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  const double mc = mcs[0];
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
  AI_CHECK_GRAPH(csr, "ai_lsym_apply");
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);  // declared before the solver: its buffers die first
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
  AI_CHECK_GRAPH(csr, "ai_bench_spmv");
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);  // declared before the solver: its buffers die first
  Solver S(ctx, csr);
  AI_TRY(S.begin(true));
  AI_TRY(S.build_tasks());
  AI_TRY(S.prepare(false));
  hipStream_t st = ctx->stream;
  S.slab_stride = (size_t)S.na;
  AI_TRY(S.ensure_vec(0));
  hipLaunchKernelGGL(k_lz_init, dim3(S.lzc.n), dim3(AI_BLOCK), 0, st, S.lzc.d.p, S.cactive.p, S.orig, S.u1.p, S.vec(0), S.pB[0].p);
  AI_KERNEL_CHECK();
  for (int i = 0; i < 3; ++i) AI_TRY(S.launch_spmv(0));
  AI_HIP(hipEventRecord(ctx->ev[0], st));
  for (int i = 0; i < reps; ++i) AI_TRY(S.launch_spmv(0));
  AI_HIP(hipEventRecord(ctx->ev[1], st));
  AI_HIP(hipStreamSynchronize(st));
  float ms = 0.f;
  AI_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
  *avg_ms = (double)ms / reps;
  if (bytes_per_launch) {
    // E (4 B index + 8 B value) + (N + 1) 4 B row pointers +
    // N x 8 B x {R_j read, sinv2 read, z written}
    const double N = (double)csr->n, E = (double)csr->nnz;
    *bytes_per_launch = E * 12.0 + (N + 1.0) * 4.0 + N * 8.0 * 3.0;
  }
  return AI_OK;
}

namespace {
__global__ __launch_bounds__(AI_BLOCK) void k_comp_flag(const int32_t* __restrict__ parent, int32_t n, int32_t root, int32_t* __restrict__ flag) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i < n) flag[i] = (parent[i] == root) ? 1 : 0;
}
__global__ __launch_bounds__(AI_BLOCK) void k_comp_map(const int32_t* __restrict__ flag, const int32_t* __restrict__ fscan, int32_t n,
                                                       const int32_t* __restrict__ orig, int32_t* __restrict__ map,
                                                       int32_t* __restrict__ orig_sub) {
  const int i = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t d = flag[i] ? fscan[i] : -1;
  map[i] = d;
  if (d >= 0) orig_sub[d] = orig[i];
}

// eigenpairs 2 .. k1+1 of ONE connected graph: k1 (lambda, unit vector) pairs, vectors scattered into
// full-length rows of `vecs` (row stride n_full) at the positions csr->orig names
int eigs_connected(ai_ctx* ctx, const ai_csr* csr, const ai_ncut_opts* opts, int k1, int64_t n_full, std::vector<double>& lambdas,
                   std::vector<double>& vecs, int* steps_out, double* max_resid) {
  Solver S(ctx, csr);
  fill_opts(S, opts);
  const int n = (int)csr->n;
  AI_TRY(S.begin(true));
  AI_TRY(S.build_tasks());
  AI_TRY(S.prepare(false));
  hipStream_t st = ctx->stream;
  k1 = std::min(k1, n - 1);
  lambdas.clear();
  vecs.clear();
  if (k1 <= 0) return AI_OK;
  DevBuf<double> out;
  AI_TRY(out.alloc((size_t)k1 * n));
  std::vector<double> thetas, resids;
  int steps = 0;
  static const int force_fro = getenv("AI_EIGS_LANCZOS") ? atoi(getenv("AI_EIGS_LANCZOS")) : 0;
  if (!force_fro && n >= 1024 && k1 >= 3) {
    // many pairs of a large graph: Chebyshev-filtered subspace iteration (block of 64 / 128 vectors)
    ChfsiStats cs;
    if (k1 <= 32)
      AI_TRY(chfsi_solve<1>(S, k1, S.opt.tol, thetas, resids, out.p, (size_t)n, &cs));
    else
      AI_TRY(chfsi_solve<2>(S, k1, S.opt.tol, thetas, resids, out.p, (size_t)n, &cs));
    steps = cs.spmm;
    if (getenv("AI_NCUT_DEBUG"))
      fprintf(stderr, "[ai_eigs chfsi] %d outer iterations, %d polynomial degrees, %d SpMM launches, filter %.1f ms, orthonormalisation + Rayleigh-Ritz %.1f ms\n",
              cs.outer, cs.degrees, cs.spmm, cs.ms_filter, cs.ms_rr);
  } else {
    AI_TRY(S.lanczos_fro(k1, thetas, resids, out.p, (size_t)n, &steps));
  }
  const int got = (int)thetas.size();
  std::vector<double> h_out((size_t)got * n);
  std::vector<int32_t> h_orig(n);
  AI_HIP(hipMemcpyAsync(h_out.data(), out.p, h_out.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(h_orig.data(), S.orig, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  vecs.assign((size_t)got * n_full, 0.0);
  for (int i = 0; i < got; ++i) {
    lambdas.push_back(1.0 - thetas[i]);
    if (max_resid) *max_resid = std::max(*max_resid, resids[i]);
    const double* src = &h_out[(size_t)i * n];
    double n2 = 0.0;
    for (int r = 0; r < n; ++r) n2 += src[r] * src[r];
    const double rn = 1.0 / sqrt(n2);
    double* dst = &vecs[(size_t)i * n_full];
    for (int r = 0; r < n; ++r) dst[h_orig[r]] = src[r] * rn;
  }
  if (steps_out) *steps_out = std::max(*steps_out, steps);
  return AI_OK;
}
}  // namespace

extern "C" int ai_eigs_smallest(ai_ctx* ctx, const ai_csr* csr, int32_t k, const ai_ncut_opts* opts, double* evals, double* evecs,
                                int32_t* iters, double* max_resid) {
  if (!ctx || !csr || !evals || !evecs || k < 1 || k > RITZ_MAXK || k > csr->n) {
    ai_set_error("ai_eigs_smallest: bad argument (1 <= k <= %d, k <= n)", RITZ_MAXK);
    return AI_ERR_BAD_ARG;
  }
  AI_CHECK_GRAPH(csr, "ai_eigs_smallest");
  AI_HIP(hipSetDevice(ctx->device));
  ArenaScope arena_scope(&ctx->arena);
  Solver S(ctx, csr);
  fill_opts(S, opts);
  const int n = (int)csr->n;
  AI_TRY(S.begin(true));
  AI_TRY(S.build_tasks());
  AI_TRY(S.prepare(true));
  hipStream_t st = ctx->stream;
  // component structure and degrees on the host: every component contributes the eigenvalue 0
  // with eigenvector D^1/2 1_C / sqrt(vol_C)
  std::vector<int32_t> h_parent(n), h_orig(n);
  std::vector<double> h_deg(n);
  AI_HIP(hipMemcpyAsync(h_parent.data(), S.parent, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(h_orig.data(), S.orig, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  AI_HIP(hipMemcpyAsync(h_deg.data(), S.deg.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  AI_HIP(hipStreamSynchronize(st));
  std::vector<int32_t> roots;
  for (int i = 0; i < n; ++i)
    if (h_parent[i] == i) roots.push_back(i);
  const int ncomp = (int)roots.size();
  const int nzero = std::min(ncomp, (int)k);
  memset(evecs, 0, (size_t)k * n * sizeof(double));
  if (iters) *iters = 0;
  if (max_resid) *max_resid = 0.0;
  // zero pairs: the first min(k, components) components in row order (any k of them are a valid answer)
  {
    std::vector<int32_t> rank(n, -1);
    std::vector<double> vol(nzero, 0.0);
    for (int c = 0; c < nzero; ++c) rank[roots[c]] = c;
    for (int i = 0; i < n; ++i) {
      const int c = rank[h_parent[i]];
      if (c >= 0) vol[c] += h_deg[i];
    }
    for (int i = 0; i < n; ++i) {
      const int c = rank[h_parent[i]];
      if (c >= 0) evecs[(size_t)c * n + h_orig[i]] = sqrt(h_deg[i] / vol[c]);
    }
    for (int c = 0; c < nzero; ++c) evals[c] = 0.0;
  }
  const int need = k - nzero;  // non-zero eigenvalues still wanted
  if (need == 0) return AI_OK;
  // The non-zero spectrum is the union of the components' spectra: take the `need` smallest non-zero
  // pairs of every component (each a connected graph of its own) and merge.
  struct Cand {
    double lambda;
    int comp, idx;
  };
  std::vector<Cand> cands;
  std::vector<std::vector<double>> cvecs(ncomp);
  int steps = 0;
  double mr = 0.0;
  for (int c = 0; c < ncomp; ++c) {
    ai_csr sub;
    const ai_csr* use = csr;
    DevBuf<int32_t> s_rowptr, s_col, s_orig, s_cnt;
    DevBuf<double> s_val;
    if (ncomp > 1) {
      // sub-graph of component c (rows keep their order); sub.orig = the caller's ids of those rows
      const unsigned gr = (unsigned)((n + AI_BLOCK - 1) / AI_BLOCK);
      const unsigned ge = (unsigned)(((int64_t)n * AI_LPR + AI_BLOCK - 1) / AI_BLOCK);
      hipLaunchKernelGGL(k_comp_flag, dim3(gr), dim3(AI_BLOCK), 0, st, (const int32_t*)S.parent, n, roots[c], S.flag.p);
      AI_KERNEL_CHECK();
      AI_TRY(ai_exclusive_scan_i32(st, S.flag.p, S.fscan.p, n, S.scantmp.p));
      int32_t nc = 0;
      AI_HIP(hipMemcpyAsync(&nc, S.fscan.p + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      AI_HIP(hipStreamSynchronize(st));
      if (nc < 2) continue;  // a single point has no non-zero eigenvalue
      AI_TRY(s_orig.alloc(nc));
      AI_TRY(s_rowptr.alloc((size_t)nc + 1));
      AI_TRY(s_cnt.alloc((size_t)nc + 1));
      hipLaunchKernelGGL(k_comp_map, dim3(gr), dim3(AI_BLOCK), 0, st, (const int32_t*)S.flag.p, (const int32_t*)S.fscan.p, n, S.orig, S.map.p,
                         s_orig.p);
      AI_KERNEL_CHECK();
      AI_HIP(hipMemsetAsync(s_cnt.p, 0, ((size_t)nc + 1) * sizeof(int32_t), st));
      hipLaunchKernelGGL(k_rebuild_count, dim3(ge), dim3(AI_BLOCK), 0, st, S.rowptr, S.col, (const int32_t*)S.flag.p, (const int32_t*)S.map.p, n,
                         s_cnt.p);
      AI_KERNEL_CHECK();
      AI_TRY(ai_exclusive_scan_i32(st, s_cnt.p, s_rowptr.p, nc, S.scantmp.p));
      int32_t nnz_c = 0;
      AI_HIP(hipMemcpyAsync(&nnz_c, s_rowptr.p + nc, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      AI_HIP(hipStreamSynchronize(st));
      AI_TRY(s_col.alloc((size_t)std::max(nnz_c, 1)));
      AI_TRY(s_val.alloc((size_t)std::max(nnz_c, 1)));
      hipLaunchKernelGGL(k_rebuild_fill, dim3(ge), dim3(AI_BLOCK), 0, st, S.rowptr, S.col, S.wraw, (const int32_t*)S.flag.p,
                         (const int32_t*)S.map.p, n, (const int32_t*)s_rowptr.p, s_col.p, s_val.p);
      AI_KERNEL_CHECK();
      sub.n = nc;
      sub.nnz = nnz_c;
      sub.rowptr = s_rowptr.p;
      sub.col = s_col.p;
      sub.val = s_val.p;
      sub.orig = s_orig.p;
      sub.device = ctx->device;
      use = &sub;
    }
    std::vector<double> lam;
    AI_TRY(eigs_connected(ctx, use, opts, need, n, lam, cvecs[c], &steps, &mr));
    for (int i = 0; i < (int)lam.size(); ++i) cands.push_back(Cand{lam[i], c, i});
  }
  if ((int)cands.size() < need) {
    ai_set_error("ai_eigs_smallest: only %zu of %d eigenpairs could be formed", cands.size() + (size_t)nzero, k);
    return AI_ERR_NO_CONVERGENCE;
  }
  std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.lambda < b.lambda; });
  for (int i = 0; i < need; ++i) {
    evals[nzero + i] = cands[i].lambda;
    memcpy(evecs + (size_t)(nzero + i) * n, &cvecs[cands[i].comp][(size_t)cands[i].idx * n], (size_t)n * sizeof(double));
  }
  if (iters) *iters = steps;
  if (max_resid) *max_resid = mr;
  if (mr > S.opt.tol) {
    // the reference's eigsh raises ArpackNoConvergence in this situation; the pairs found so far are returned all the same
    ai_set_error("ai_eigs_smallest: largest residual %.3g after %d steps is above the tolerance %.3g", mr, steps, S.opt.tol);
    return AI_ERR_NO_CONVERGENCE;
  }
  return AI_OK;
}
