/*
 * autoinst_hip.h -- C ABI of libautoinst_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE hot path of artonson/autoinst: the per-chunk pairwise
 * affinity build and the recursive normalized cut.  The
 * reference is pure Python (no FFI of its own); each entry point cites the reference
 * lines it replaces (paths relative to the reference root).  Plain pointers and
 * sizes only -- no torch / numpy types.  All matrices are float64, like the
 * reference's arithmetic.
 *
 * Threading: an ai_ctx owns its HIP streams (one for the affinity build and the Lanczos steps, one for the convergence
 * checks, one for the harvest waves of the normalized cut) and one helper thread; it is not thread-safe, the library is
 * re-entrant across contexts.  One process per GPU for multi-GPU use.
 * Ownership: the caller owns every array it passes; nothing is retained after a
 * call returns.  Handles returned here are freed with the matching *_free/destroy.
 * Errors: every function returns AI_OK (0) or a negative ai_status; the text of the
 * last error of the calling thread is ai_last_error().
 */
#ifndef AUTOINST_HIP_H
#define AUTOINST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ai_ctx ai_ctx;
typedef struct ai_csr ai_csr; /* device-resident symmetric affinity graph */

/*
 * ABI version: bumped whenever a struct of this header changes size or a field changes meaning (6: ai_ncut_stats gained max_true_resid,
 * true_resid_limit, accepted_above_limit, check_timeouts, spmv_blocks and spmv_blocks_idle; 5: restarted_solves and hist_retries; 4: round 4; 3 added
 * ai_ncut_opts.window_rows).  A binding checks ai_abi_version() == AI_ABI_VERSION and ai_abi_sizeof(which) against its own
 * struct sizes when it loads the library (autoinst_amd/_ffi.py does): a caller built against an older header would otherwise
 * pass a shorter ai_ncut_opts and have the library read past it.
 */
#define AI_ABI_VERSION 6
int ai_abi_version(void);
int64_t ai_abi_sizeof(int which); /* 0: ai_ncut_opts, 1: ai_ncut_stats; -1 otherwise */

typedef enum {
  AI_OK = 0,
  AI_ERR_BAD_ARG = -1,
  AI_ERR_OOM = -2,
  AI_ERR_NO_CONVERGENCE = -3,
  AI_ERR_HIP = -4,
  AI_ERR_INTERNAL = -5
} ai_status;

/* where the caller's buffers live */
typedef enum { AI_MEM_HOST = 0, AI_MEM_DEVICE = 1 } ai_mem;

#define AI_NUM_CUTS 10 /* pipeline/ncuts/normalized_cut.py:54  get_min_ncut(ev, D, w, 10) */

int ai_version(void);
const char* ai_last_error(void);

int ai_ctx_create(int device, ai_ctx** out);
int ai_ctx_destroy(ai_ctx* ctx);
/*
 * Device memory a context holds between calls (no reference counterpart: the reference's arrays are NumPy's).
 * out[0] = bytes of the call workspace (one block once the largest call has been seen: a longer list of blocks is replaced
 * by one of the peak need + 6 % at the start of the next call), out[1] = its number of blocks, out[2] = bytes of graphs
 * handed out and not yet freed, out[3] = bytes of freed graph buffers kept for re-use.
 */
int ai_ctx_mem_info(ai_ctx* ctx, int64_t out[4]);

/*
 * Affinity build.  Replaces pipeline/ncuts/ncuts_utils.py:60-67 (cdist + radius mask +
 * spatial weights), :125-133 (DINO factor), :135-149 (TARL factor with the all-zero-row
 * exemption), :151-156 (product), :159 (remove_isolated_points: identity, A_ii = 1) and
 * :167 (CSR conversion):
 *     A_ij = 1[d_ij <= radius] * exp(-theta t_ij) * exp(-alpha d_ij) * exp(-gamma g_ij)
 * xyz: n x 3 row-major; tarl: n x tarl_dim or NULL (theta ignored); dino likewise.
 * A falsy alpha / theta / gamma (0.0) drops that factor, as the reference's `if CONFIG[..]`.
 * gamma != 0 with dino == NULL is AI_ERR_BAD_ARG (reference raises ValueError, :126-127).
 * The graph stays on the device in the library's own (cell-sorted) row order.
 */
int ai_affinity_build(ai_ctx* ctx, const double* xyz, int64_t n,
                      const double* tarl, int32_t tarl_dim,
                      const double* dino, int32_t dino_dim,
                      double alpha, double theta, double gamma, double radius,
                      int mem_kind, ai_csr** out);

/*
 * The same with the SAM factor of ncuts_utils.py:112-123 / utils/image/image_utils.py:64-89 (beta is 0.0 in
 * every shipped config, config.py:12,23,34,45, so the reference pipeline never takes this path): sam =
 * (n, sam_views) int32 SAM ids of one camera, -1 = no id in that view; per pair inside the radius the
 * factor is exp(-beta * fraction of the co-labelled views whose ids differ).  Factors are multiplied in
 * the reference's order tarl * spatial * sam * dino (:151-156).  beta != 0 without ids is a bad argument
 * (the reference raises ValueError, :116-117).
 */
int ai_affinity_build_sam(ai_ctx* ctx, const double* xyz, int64_t n, const double* tarl, int32_t tarl_dim,
                          const double* dino, int32_t dino_dim, const int32_t* sam, int32_t sam_views, double alpha,
                          double beta, double gamma, double theta, double radius, int mem_kind, ai_csr** out);

/*
 * One more camera (the loops over cameras at ncuts_utils.py:118-123 and :128-133; CAM_IDS has a single entry
 * in the reference's config.py:72): multiplies every stored value of an existing graph by that camera's
 * exp(-beta * SAM fraction) and exp(-gamma * ||dino_i - dino_j||).  The product is then associated
 * ((... * cam1) * cam2) instead of the reference's (cam1 * cam2): values agree to 1 ulp.
 */
int ai_affinity_apply_camera(ai_ctx* ctx, ai_csr* csr, const double* dino, int32_t dino_dim, const int32_t* sam,
                             int32_t sam_views, double beta, double gamma, int mem_kind);

/*
 * Upload a caller-built symmetric CSR (what ncuts_utils.py:167 hands to
 * normalized_cut at :168).  indptr has n+1 entries.  Row order is kept.
 */
int ai_csr_from_host(ai_ctx* ctx, int64_t n, const int64_t* indptr, const int32_t* indices,
                     const double* data, ai_csr** out);

int ai_csr_dims(const ai_csr* csr, int64_t* n, int64_t* nnz);

/*
 * Copy the graph out as scipy.sparse.csr_matrix(A) would hold it: rows and columns in
 * the caller's ORIGINAL point order, column indices ascending within a row.
 * indptr: n+1, indices/data: nnz (host buffers).
 */
int ai_csr_export(ai_ctx* ctx, const ai_csr* csr, int64_t* indptr, int32_t* indices, double* data);

/*
 * Release a graph.  The device buffers go back to the cache of the context that BUILT the graph, whichever
 * context (or NULL) is passed here (hipFree would synchronise the whole device and stall other host threads);
 * they are re-used by that context's next graph of similar size and returned to the driver by ai_ctx_destroy.
 * A graph that outlives its context keeps a valid handle (free it as usual) but no buffers: every other call
 * on it returns AI_ERR_BAD_ARG ("the graph's context was destroyed").
 */
int ai_csr_free(ai_ctx* ctx, ai_csr* csr);

typedef struct {
  double tol;           /* Ritz residual |beta_m s_m| at which a Lanczos solve stops (default 1e-10) */
  int32_t max_iter;     /* Lanczos step cap per solve (default 4000; larger values are clamped to 4000: the convergence check keeps T_m in 64 KB of LDS) */
  int32_t check_every;  /* size of T at a solve's FIRST convergence check (default 16, at most 64); later checks are placed by the residual trend, 4 .. 48 steps apart */
  int32_t reserved;     /* profiling, fills ms_spmv (for bench.py): bit 0 or 1 = every SpMV launch stamps its own span (first block in .. last block out) on the device clock, which does not perturb how launches of several streams overlap */
  int64_t window_rows;  /* ai_ncut_batch: rows (points) of the chunks that iterate at one time; further chunks of the call are admitted as earlier ones finish, so the Lanczos vector storage is sized for the window, not for the call (0 = 4 800 000; never less than the call's largest chunk) */
} ai_ncut_opts;

typedef struct {
  int64_t levels;          /* harvest waves run (asynchronous frontier) / frontier levels processed (level-synchronous driver) */
  int64_t lanczos_solves;  /* connected segments solved by Lanczos */
  int64_t null_solves;     /* disconnected segments, split into their connected components in one step */
  int64_t lanczos_steps;   /* Lanczos steps launched (= fused SpMV launches); every iterating segment of the call takes part in a step at its own step count */
  int64_t spmv_rows;       /* rows processed by the SpMV kernel, summed over launches (exact, counted on device) */
  int64_t spmv_nnz;        /* stored entries processed by the SpMV kernel, summed over launches */
  int64_t unconverged;     /* solves that hit max_iter before tol */
  int64_t n_groups;
  double ms_total;         /* host wall time of the call */
  double ms_eigen;         /* wall time during which Lanczos segments were iterating (host clock) */
  double ms_spmv;          /* device time in the fused SpMV kernel alone (only with reserved bit 0 or 1) */
  double ms_sweep;         /* level-synchronous driver only: device time in min/max + bin + sweep */
  double ms_rebuild;       /* level-synchronous driver only: device time in CC + partition + CSR rebuild */
  double max_resid;        /* largest accepted Ritz residual */
  int64_t restarted_solves; /* solves repeated because the Ritz pair of a 'converged' segment failed the true-residual test
                               ||M v - theta v|| <= max(1e-6, 100 tol) ||v||.  Healthy solves measure <= tol; the repeat starts from
                               the same vector and is bit-identical to an undisturbed solve.  A repeat that arrives at the same
                               residual bit for bit is accepted (dense graphs with clustered top eigenvalues: Lanczos without
                               reorthogonalisation gives 3e-7 there), so the labels never depend on the test */
  int64_t hist_retries;     /* waves whose packed Lanczos histories (device -> pinned host memory) failed their header check (size of T,
                               integer checksum) and were packed again; 0 in a healthy run, labels do not depend on it */
  double max_true_resid;    /* largest TRUE residual ||M v - theta v|| / ||v|| of any Ritz pair that was cut (max_resid is the Lanczos
                               ESTIMATE |beta_m s_m| that stopped the solve; this one is measured on the segment's own entries) */
  double true_resid_limit;  /* the bar the call enforced on it: max(1e-6, 100 tol).  Above it a pair is solved again and cut only if the
                               repeat reproduces the residual bit for bit (then it is the algorithm's own answer) */
  int64_t accepted_above_limit; /* pairs cut with a true residual above the limit for that reason (dense blobs whose top eigenvalues
                               cluster: 3e-7 without re-orthogonalisation); 0 on the benchmark's chunks */
  int64_t check_timeouts;   /* convergence-check blocks that gave up waiting for their launch's scanning blocks (they judge nothing and
                               the next launch judges instead); 0 unless the device is oversubscribed */
  int64_t spmv_blocks;      /* only with opts->reserved bit 0 or 1: blocks dispatched by the stamped SpMV launches ... */
  int64_t spmv_blocks_idle; /* ... and how many of them found their segment frozen since the task lists were written (they end after a
                               record and a flag load; the lists are rewritten at the next relist) */
} ai_ncut_stats;

/*
 * Recursive normalized cut.  Replaces pipeline/ncuts/normalized_cut.py:37-63
 * (normalized_cut), :13-34 (get_min_ncut), :4-11 (cut_cost / ncut_cost) and the
 * scipy eigsh call at :49.  labels_out[i] (i in the caller's ORIGINAL row order) is the
 * index of point i's group in the order the reference's recursion emits groups
 * (mask side first); *n_groups receives their number.  split_lim applies to the top
 * call only; deeper calls use 0.01 as the reference does (:57-58 rely on the default).
 * opts / stats may be NULL.
 * A solve that reaches opts->max_iter before its residual falls to opts->tol makes the call return
 * AI_ERR_NO_CONVERGENCE (the reference's eigsh raises ArpackNoConvergence, normalized_cut.py:49); labels,
 * n_groups and stats (stats->unconverged, max_resid) are filled from the best vectors all the same.
 */
int ai_ncut(ai_ctx* ctx, const ai_csr* csr, int64_t num_points_orig, double T, double split_lim,
            const ai_ncut_opts* opts, int32_t* labels_out, int32_t* n_groups, ai_ncut_stats* stats);

/*
 * The same recursion over `count` independent chunks at once (run_pipeline.py:160-179 loops over
 * them one by one).  The connected segments of all chunks iterate in ONE pool, each at its own Lanczos
 * step count, so every kernel launch is shared by all of them: a single chunk's launches are
 * latency-bound, a batch fills them.  Chunks beyond opts->window_rows wait inside the call and are
 * admitted as earlier ones finish (a whole map can be one call).  Per chunk c: graphs[c], num_points_orig[c], labels_out[c] (graphs[c]->n ints, group
 * ids from 0 in that chunk's own emission order), n_groups[c].  Results are those of `count`
 * separate ai_ncut calls.  stats (may be NULL) describes the whole batch.
 */
int ai_ncut_batch(ai_ctx* ctx, const ai_csr* const* graphs, int32_t count, const int64_t* num_points_orig,
                  double T, double split_lim, const ai_ncut_opts* opts, int32_t* const* labels_out,
                  int32_t* n_groups, ai_ncut_stats* stats);

/*
 * Building blocks exposed for parity tests (top-level call of the recursion only).
 * ai_fiedler: eigenpair of the 2nd-smallest eigenvalue of L_sym = D^-1/2 (D - W) D^-1/2,
 *   W = w + I (normalized_cut.py:38-53); ev_out (n, host, original order) has unit norm and
 *   the library's sign convention (entry of largest magnitude positive).  The graph must be
 *   connected, else a null-space vector is returned and *lambda2 = 0.
 * ai_sweep: the 10 threshold costs of normalized_cut.py:13-34 for a caller-given ev
 *   (host, original order); costs[10], mask_out[n] (uint8), *mcut.
 * ai_lsym_apply: y = L_sym x (host vectors, original order) through the SpMV kernel.
 */
int ai_fiedler(ai_ctx* ctx, const ai_csr* csr, const ai_ncut_opts* opts, double* lambda2,
               double* ev_out, int32_t* iters, double* resid);
int ai_sweep(ai_ctx* ctx, const ai_csr* csr, const double* ev, double* costs, uint8_t* mask_out,
             double* mcut);
int ai_lsym_apply(ai_ctx* ctx, const ai_csr* csr, const double* x, double* y);

/*
 * k smallest eigenpairs of L_sym = D^-1/2 (D - W) D^-1/2, W = w + I (BASELINE.json configs[4]; the
 * reference itself only ever asks for k = 2, normalized_cut.py:49).  1 <= k <= 64.
 * evals[k] ascending; evecs[j * n + i] = component i (caller's original order) of unit vector j.
 * Every connected component contributes one zero eigenvalue with eigenvector D^1/2 1_C / sqrt(vol_C)
 * (formed explicitly).  With >= k components the answer is k such pairs (any k of them are a
 * valid answer, as with SciPy).  Otherwise the remaining pairs are the smallest non-zero ones of the
 * union of the components' spectra: each component is solved on its own -- by Chebyshev-filtered subspace iteration
 * (n >= 1024 and >= 3 pairs wanted; stops when the true residuals of the first k - 1 pairs are <= opts->tol), else by Lanczos
 * with full re-orthogonalisation (stops when the innermost wanted pair's Ritz residual <= opts->tol) -- and the results are merged.
 */
int ai_eigs_smallest(ai_ctx* ctx, const ai_csr* csr, int32_t k, const ai_ncut_opts* opts, double* evals,
                     double* evecs, int32_t* iters, double* max_resid);

/*
 * "Next" rows on either side of the hot path.
 *
 * ai_radius_mean_pool: pipeline/utils/point_cloud/chunk_generation.py:243-256 -- out[i] (dim float64) =
 *   mean of the float32 feature rows of all source points with distance < radius from query i (a zero
 *   row when there is none); count_out[i] (may be NULL) = how many.  dim <= 384.
 * ai_nn1_project: pipeline/utils/point_cloud/point_cloud_utils.py:144-174 (kDTree_1NN_feature_reprojection)
 *   -- nn_index[i] = index of the source point nearest to fine point i, nn_dist[i] (may be NULL) its
 *   distance; the caller gathers labels / colours and applies max_radius.
 * Buffers are host or device according to mem_kind (all of one kind).
 */
int ai_radius_mean_pool(ai_ctx* ctx, const double* query_xyz, int64_t nq, const double* src_xyz, int64_t ns,
                        const float* src_feat, int32_t dim, double radius, int mem_kind, double* out,
                        int32_t* count_out);
int ai_nn1_project(ai_ctx* ctx, const double* to_xyz, int64_t nt, const double* from_xyz, int64_t nf,
                   int mem_kind, int32_t* nn_index, double* nn_dist);

/*
 * "Next" row 4: the per-point parts of the scorer and of the chunk merge.
 *
 * ai_label_pairs: the contingency table of two label arrays -- the distinct (a[i], b[i]) pairs in
 *   ascending (a, b) order with their counts.  Replaces the per-label np.unique / np.where /
 *   np.intersect1d / np.union1d passes of pipeline/metrics/metrics_class.py:302-309 (filter_labels),
 *   :60-114 (get_tp_fp), :181-235 (average_precision) and the np.unique of pred + gt * 2^32 of
 *   pipeline/metrics/modified_LSTQ.py:34-60.  a, b: host or device per mem_kind.  pair_a / pair_b /
 *   pair_count: HOST arrays of capacity `cap`; *n_pairs is always the full number of distinct pairs
 *   (call again with a larger cap when it exceeds cap).
 *
 * ai_merge_associate: the per-point work of one iteration of merge_chunks_unite_instances2
 *   (pipeline/utils/point_cloud/point_cloud_utils.py:397-463).  Instances are dense ids (0 = street /
 *   no instance, the reference's black colour) whose numeric order is the order of the reference's
 *   np.unique(colors, axis=0).  The map is cropped to the cube center +- side_length / 2 (inclusive,
 *   :405-417).  For id1 in [1, n_inst1), id2 in [1, n_inst2), row-major [id1 * n_inst2 + id2]:
 *     inter      = #chunk points of id2 inside the bounding box of the cropped points of id1 (:446-456);
 *     common     = #distinct scalar coordinate values the two instances share, so that the reference's
 *                  union (:458, np.unique of the concatenated arrays, flattened) is
 *                  n_scalars1[id1] + n_scalars2[id2] - common;
 *   n_points1[id1] = #cropped map points of id1 (0: the instance is not in the crop).
 *   Outputs are HOST arrays; the point / id arrays are host or device per mem_kind.
 *
 * ai_unique_points: PointCloud.remove_duplicated_points() (:489): indices (ascending) of the first point
 *   of every distinct coordinate triple.  keep_index (capacity n) is host or device per mem_kind.
 */
int ai_label_pairs(ai_ctx* ctx, const int32_t* a, const int32_t* b, int64_t n, int mem_kind, int64_t cap,
                   int32_t* pair_a, int32_t* pair_b, int64_t* pair_count, int64_t* n_pairs);
int ai_merge_associate(ai_ctx* ctx, const double* map_xyz, const int32_t* map_inst, int64_t n_map,
                       const double* chunk_xyz, const int32_t* chunk_inst, int64_t n_chunk, const double* center,
                       double side_length, int32_t n_inst1, int32_t n_inst2, int mem_kind, int32_t* inter,
                       int32_t* common, int32_t* n_scalars1, int32_t* n_scalars2, int32_t* n_points1);
int ai_unique_points(ai_ctx* ctx, const double* xyz, int64_t n, int mem_kind, int32_t* keep_index, int64_t* n_keep);

/*
 * Timing hook for bench.py: runs `reps` fused Lanczos SpMV steps on the whole graph as
 * one segment and returns the average kernel time (HIP events on the context's stream)
 * plus the algorithmic byte count of one launch.
 */
int ai_bench_spmv(ai_ctx* ctx, const ai_csr* csr, int32_t reps, double* avg_ms, double* bytes_per_launch);

/*
 * Timing hook for bench.py: the box's plain stream rate.  `reps` device-to-device copies of `bytes` bytes (16 bytes per lane per
 * access, four accesses in flight per thread, every block its own contiguous range), timed with HIP events on the context's stream;
 * *gbps = read + written bytes per second / 1e9.  What a roofline fraction of an HBM-bound kernel is quoted beside.
 */
int ai_bench_copy(ai_ctx* ctx, int64_t bytes, int32_t reps, double* gbps);

#ifdef __cplusplus
}
#endif
#endif /* AUTOINST_HIP_H */
