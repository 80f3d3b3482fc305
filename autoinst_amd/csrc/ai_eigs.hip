// ai_eigs_smallest: the k smallest eigenpairs of L_sym (BASELINE.json configs[4], k = 64; the reference itself only asks
// for k = 2, normalized_cut.py:49).  Every connected component contributes an explicit zero pair; the non-zero pairs of
// a component come from Chebyshev-filtered subspace iteration (ai_chfsi.inc) or, for small graphs / few pairs, Lanczos
// with full re-orthogonalisation (Solver::lanczos_fro).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <set>

#include <hip/hip_ext.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "ai_common.h"

#include "ai_ncut_params.h"
#include "ai_tridiag.h"

namespace {
#include "ai_ncut_kernels.inc"
#include "ai_ncut_solver.inc"
}  // namespace

#include "ai_chfsi.inc"

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

// rn[i] = 1 / ||row i of out|| (one block per row; fixed summation order)
__global__ __launch_bounds__(AI_BLOCK) void k_row_rnorm(const double* __restrict__ out, int32_t n, double* __restrict__ rn) {
  __shared__ double sm[AI_BLOCK / 64];
  const double* src = out + (size_t)blockIdx.x * n;
  double a = 0.0;
  for (int r = threadIdx.x; r < n; r += AI_BLOCK) a = fma(src[r], src[r], a);
  const double t = ai_block_sum(a, sm);
  if (threadIdx.x == 0) rn[blockIdx.x] = 1.0 / sqrt(t);
}
// full[i][orig[r]] = out[i][r] * rn[i]
__global__ __launch_bounds__(AI_BLOCK) void k_scatter_rows(const double* __restrict__ out, const double* __restrict__ rn,
                                                           const int32_t* __restrict__ orig, int32_t n, size_t n_full, double* __restrict__ full) {
  const int r = blockIdx.x * AI_BLOCK + threadIdx.x;
  if (r >= n) return;
  const int i = blockIdx.y;
  full[(size_t)i * n_full + orig[r]] = out[(size_t)i * n + r] * rn[i];
}

// eigenpairs 2 .. k1+1 of ONE connected graph: k1 (lambda, unit vector) pairs, vectors scattered into
// full-length rows of `vecs` (row stride n_full) at the positions csr->orig names
// (`direct`: host rows of n_full doubles that receive the vectors instead of `vecs` -- the one-component case, no staging copy)
int eigs_connected(ai_ctx* ctx, const ai_csr* csr, const ai_ncut_opts* opts, int k1, int64_t n_full, std::vector<double>& lambdas,
                   std::vector<double>& vecs, int* steps_out, double* max_resid, double* direct = nullptr) {
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
      fprintf(stderr, "[ai_eigs chfsi] %d outer iterations, %d polynomial degrees, %d SpMM launches, filter %.1f ms, orthonormalisation + Rayleigh-Ritz %.1f ms (host: %.1f ms in the 7 symmetric eigenproblems, %.1f ms in the Cholesky factors)\n",
              cs.outer, cs.degrees, cs.spmm, cs.ms_filter, cs.ms_rr, cs.ms_host_eig, cs.ms_host_chol);
  } else {
    AI_TRY(S.lanczos_fro(k1, thetas, resids, out.p, (size_t)n, &steps));
  }
  const int got = (int)thetas.size();
  // the solvers return at most k1 pairs, thetas descending (lambda = 1 - theta ascending): the direct path below writes `got` rows
  // into the caller's array in that order without the sort of the general path, so both properties are checked, not assumed
  if (got > k1) {
    ai_set_error("internal: the eigensolver returned %d pairs, %d were asked for", got, k1);
    return AI_ERR_INTERNAL;
  }
  for (int i = 1; i < got; ++i)
    if (thetas[i] > thetas[i - 1]) {
      ai_set_error("internal: the eigensolver's Ritz values are not in descending order (%.17g after %.17g at %d)", thetas[i], thetas[i - 1], i);
      return AI_ERR_INTERNAL;
    }
  // unit vectors at the caller's row positions, formed on the device: one transfer, no host pass over k1 x n doubles
  DevBuf<double> rn, full;
  AI_TRY(rn.alloc((size_t)std::max(got, 1)));
  AI_TRY(full.alloc((size_t)std::max(got, 1) * (size_t)n_full));
  if (got > 0) {
    if (n_full != n) AI_HIP(hipMemsetAsync(full.p, 0, (size_t)got * (size_t)n_full * sizeof(double), st));
    hipLaunchKernelGGL(k_row_rnorm, dim3((unsigned)got), dim3(AI_BLOCK), 0, st, (const double*)out.p, (int32_t)n, rn.p);
    AI_KERNEL_CHECK();
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((n + AI_BLOCK - 1) / AI_BLOCK), (unsigned)got), dim3(AI_BLOCK), 0, st, (const double*)out.p,
                       (const double*)rn.p, S.orig, (int32_t)n, (size_t)n_full, full.p);
    AI_KERNEL_CHECK();
    double* dst = direct;
    if (!dst) {
      vecs.resize((size_t)got * n_full);
      dst = vecs.data();
    }
    AI_HIP(hipMemcpyAsync(dst, full.p, (size_t)got * (size_t)n_full * sizeof(double), hipMemcpyDeviceToHost, st));
    AI_HIP(hipStreamSynchronize(st));
  }
  for (int i = 0; i < got; ++i) {
    lambdas.push_back(1.0 - thetas[i]);
    if (max_resid) *max_resid = std::max(*max_resid, resids[i]);
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
  memset(evecs, 0, (size_t)nzero * n * sizeof(double));  // the rows of the non-zero pairs are written whole below
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
  bool placed = false;
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
    // one component: its pairs come out in ascending order and go straight into the caller's rows
    double* direct = (ncomp == 1 && need <= n - 1) ? evecs + (size_t)nzero * n : nullptr;
    AI_TRY(eigs_connected(ctx, use, opts, need, n, lam, cvecs[c], &steps, &mr, direct));
    if (direct && (int)lam.size() == need) {
      for (int i = 0; i < need; ++i) evals[nzero + i] = lam[i];
      placed = true;
    }
    for (int i = 0; i < (int)lam.size(); ++i) cands.push_back(Cand{lam[i], c, i});
  }
  if ((int)cands.size() < need) {
    ai_set_error("ai_eigs_smallest: only %zu of %d eigenpairs could be formed", cands.size() + (size_t)nzero, k);
    return AI_ERR_NO_CONVERGENCE;
  }
  if (!placed) {
    std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.lambda < b.lambda; });
    for (int i = 0; i < need; ++i) {
      evals[nzero + i] = cands[i].lambda;
      memcpy(evecs + (size_t)(nzero + i) * n, &cvecs[cands[i].comp][(size_t)cands[i].idx * n], (size_t)n * sizeof(double));
    }
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

