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
#include <deque>
#include <map>
#include <set>

#include <hip/hip_ext.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "ai_common.h"

#include "ai_ncut_params.h"
#include "ai_tridiag.h"

namespace {
#include "ai_ncut_kernels.inc"
#include "ai_ncut_solver.inc"
#include "ai_flow_kernels.inc"
#include "ai_flow.inc"
}  // namespace

#ifdef AI_WITH_LOCKSTEP   // the level-synchronous recursion driver: only in the test-only build libautoinst_hip_lockstep.so (Makefile: make lockstep)
// ----------------------------------------------------------------------------- recursion driver + C ABI
static int ncut_lockstep(ai_ctx* ctx, const ai_csr* csr, int nchunks, const int64_t* off, const int64_t* n_orig, double T, double split_lim,
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
#endif  // AI_WITH_LOCKSTEP

// The shipped library has ONE recursion driver, the asynchronous frontier (ai_flow.inc).  The level-synchronous driver of rounds
// 1-2 (ncut_lockstep above: the same per-segment arithmetic, one recursion depth at a time) is compiled only into the test-only
// build libautoinst_hip_lockstep.so (-DAI_WITH_LOCKSTEP), where AI_NCUT_LOCKSTEP=1 selects it: tests/test_gpu_parity.py and
// tests/tools/fuzz_drivers.py load that build to check that both drivers give identical labels.
static int ncut_impl(ai_ctx* ctx, const ai_csr* csr, int nchunks, const int64_t* off, const int64_t* n_orig, double T, double split_lim,
                     const ai_ncut_opts* opts, int32_t* const* labels_out, int32_t* n_groups, ai_ncut_stats* stats_out, double t0) {
  static const int lockstep = getenv("AI_NCUT_LOCKSTEP") ? atoi(getenv("AI_NCUT_LOCKSTEP")) : 0;
#ifdef AI_WITH_LOCKSTEP
  if (lockstep) return ncut_lockstep(ctx, csr, nchunks, off, n_orig, T, split_lim, opts, labels_out, n_groups, stats_out, t0);
#else
  if (lockstep) {
    ai_set_error("AI_NCUT_LOCKSTEP=1: this build has no level-synchronous driver (load libautoinst_hip_lockstep.so: make -C autoinst_amd/csrc lockstep)");
    return AI_ERR_BAD_ARG;
  }
#endif
  return ncut_flow(ctx, csr, nchunks, off, n_orig, T, split_lim, opts, labels_out, n_groups, stats_out, t0);
}

extern "C" int ai_ncut(ai_ctx* ctx, const ai_csr* csr, int64_t num_points_orig, double T, double split_lim, const ai_ncut_opts* opts,
                       int32_t* labels_out, int32_t* n_groups, ai_ncut_stats* stats_out) {
  if (!ctx || !csr || !labels_out || !n_groups || num_points_orig < 0) {
    ai_set_error("ai_ncut: bad argument");
    return AI_ERR_BAD_ARG;
  }
  AI_CHECK_GRAPH(csr, "ai_ncut");
  // one chunk = a batch of one: the recursion keeps its even depths in a call-owned copy of the graph
  const ai_csr* graphs[1] = {csr};
  int32_t* lab[1] = {labels_out};
  return ai_ncut_batch(ctx, graphs, 1, &num_points_orig, T, split_lim, opts, lab, n_groups, stats_out);
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
  auto launch = [&]() -> int { return S.launch_spmv(0); };
  for (int i = 0; i < 3; ++i) AI_TRY(launch());
  AI_HIP(hipEventRecord(ctx->ev[0], st));
  for (int i = 0; i < reps; ++i) AI_TRY(launch());
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

