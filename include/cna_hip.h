/*
 * cna_hip.h -- C ABI of libcna_hip.so: the MI355X (gfx950) hot path of covarying
 * neighborhood analysis (reference: immunogenomics/cna 0.2.3).
 *
 * The reference is pure Python and has no FFI of its own; the boundary a
 * maintainer binds is therefore this library, called from the Python host through ctypes
 * (cna_amd/_ffi.py; INTEGRATION.md shows the stub a reference maintainer would add).
 * Every entry point names the reference lines it replaces (paths under
 * /root/reference/src/cna/tools/).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / numpy types;
 *   - every function returns 0 on success, a hipError_t (>0) or a CNA_E* code (<0) otherwise,
 *     with a human-readable message available from cna_last_error() (thread local);
 *   - the caller owns every host buffer; the library owns all device memory until
 *     cna_ctx_destroy();  one context per host thread; all work is issued on the context's
 *     own HIP stream and entry points that return host data synchronise that stream;
 *   - matrices on the device are "cell-major": one row per cell (neighbourhood), one column
 *     per sample, float64, leading dimension rounded up to a multiple of 4 with zero padding
 *     (the reference's frames are the transpose, samples x cells);
 *   - multi-GPU: one process per GPU; a context owns the contiguous block of graph rows
 *     [row0, row0+n_local) and the same rows of every matrix; sample-space objects are
 *     replicated.  Collectives (RCCL) are internal to the entry points that need them.
 */
#ifndef CNA_HIP_H
#define CNA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cna_ctx cna_ctx;

#define CNA_EINVAL  (-1)   /* bad argument / call order */
#define CNA_ENOMEM  (-2)
#define CNA_ERCCL   (-3)   /* RCCL failure or librccl.so not loadable */
#define CNA_ESTATE  (-4)   /* required earlier step missing */

/* matrix selectors for cna_matrix_shape / cna_fetch_matrix */
#define CNA_MAT_NAM   0    /* NAM after diffusion, all local cells x all samples           */
#define CNA_MAT_X     1    /* working matrix: selected NAM, then residualised NAM in place */
#define CNA_MAT_PROJ  2    /* result of cna_project_keep (cna_fetch_rows only)             */

/* kernel ids for the profiling counters (cna_prof_get) */
enum {
  CNA_K_COLSUM = 0, CNA_K_NAM_FIRST, CNA_K_NAM_STEP, CNA_K_BATCH_KURT, CNA_K_ZEROVAR,
  CNA_K_SELECT, CNA_K_RESID, CNA_K_STANDARDIZE, CNA_K_GRAM, CNA_K_GRAM_REDUCE, CNA_K_NCORRS,
  CNA_K_NULL_LOCAL, CNA_K_OBS_COUNTS, CNA_K_PERCELL_FDR, CNA_K_PROJECT, CNA_K_TRANSPOSE,
  CNA_K_ALLGATHER, CNA_K_CONDITION, CNA_K_GLOBAL_TEST,
  CNA_K_NAM_STEP_SPARSE,        /* the second walk step on the compressed state (k_nam_step_sparse); CNA_K_NAM_STEP is the dense gather */
  /* multi-GPU, so that a scaling run explains itself: CNA_K_ALLGATHER ("rccl") times the collectives of the main
     communicator (all-reduces of column sums / Gram / counts, the all-gather fallback of the state); the two below the
     exchange of state rows between walk steps (pack + ncclSend/Recv + unpack, on its own stream when it overlaps the step)
     and how long the main stream then STOOD STILL waiting for it (0 when the step's safe rows hid it) */
  CNA_K_HALO_EXCHANGE, CNA_K_HALO_WAIT,
  CNA_K_COUNT
};

/* ---- library / context ------------------------------------------------------------- */
const char* cna_last_error(void);
int  cna_abi_version(void);
int  cna_device_count(int* count);
int  cna_ctx_create(int device, cna_ctx** out);
int  cna_ctx_destroy(cna_ctx* ctx);
int  cna_ctx_sync(cna_ctx* ctx);
/* bytes of device memory currently held by the context */
int  cna_ctx_device_bytes(cna_ctx* ctx, int64_t* bytes);
/* Opt-in storage format of the diffusion state BETWEEN two steps of a walk (reference: the float64 `s` of
   _nam.py:31-34 between iterations of diffuse_stepwise): on != 0 keeps it in 4 bytes per entry from the second step on
   (one rank, more than 64 samples; anything else keeps 8 bytes).  Sums, the NAM and everything after it stay float64;
   the NAM then agrees with the 8-byte walk to ~1e-7 relative instead of bit for bit.  Default off. */
int  cna_set_state_f32(cna_ctx* ctx, int on);

/* ---- multi-GPU (RCCL over xGMI) ------------------------------------------------------ */
/* rank 0 creates a 128-byte id and ships it to the other ranks by any host channel */
int  cna_comm_unique_id(void* id128);
int  cna_comm_init(cna_ctx* ctx, int rank, int nranks, const void* id128);
/* Test communicator: the same collectives staged through a POSIX shared-memory segment `name`
 * (one slot of slot_bytes per rank), so that several ranks can share ONE GPU -- which RCCL refuses --
 * and the multi-rank paths of this library run on a single-GPU box.  Not for production use. */
int  cna_comm_init_shm(cna_ctx* ctx, int rank, int nranks, const char* name, int64_t slot_bytes);
/* which communicator the context holds (0 none, 1 RCCL, 2 the shared-memory test communicator) and the
 * number of ranks as the communicator itself reports it (ncclCommCount) -- bench.py prints both */
int  cna_comm_info(cna_ctx* ctx, int* backend, int* nranks);
/* Start-up check of an RCCL communicator with a time limit per collective: an all-reduce on the main stream, then a
 * ring send / receive on the halo stream's own communicator (cna_comm_init duplicates the communicator for it:
 * the exchange of the diffusion state between steps -- SURVEY.md 8e, the rows of _nam.py:33's s the other ranks
 * need -- runs under the step that produces it) while the main one carries a second all-reduce.  *halo_ok = 0: the
 * halo communicator is absent or did not answer on some rank and was aborted everywhere; exchanges then stay on
 * the main stream.  An error: the main communicator does not work.  No-op without RCCL or with one rank. */
int  cna_comm_selftest(cna_ctx* ctx, double timeout_s, int* halo_ok);

/* Neighbour ("halo") exchange of the diffusion state between steps instead of the all-gather.
 * After cna_graph_upload on every rank: send_rows = LOCAL row indices other ranks need, grouped by
 * destination rank (send_counts[nranks]); recv_rows = GLOBAL row indices this rank's cells
 * reference outside its block, grouped by owner rank (recv_counts[nranks]); the lists of two
 * peers must mirror each other.  NULL counts (and every cna_graph_upload) switch back to the
 * all-gather.  The walk itself is the reference's (_nam.py:31-34); only the data motion differs.
 * With ascending recv_rows that cover every foreign column of the block (the plan of cna_amd._order.halo_plan) the
 * diffusion state then holds n_local + sum(recv_counts) rows instead of n_global -- this rank's rows and, behind them,
 * the rows it receives, which land there directly (a rank owns its rows of A and of S); a block that
 * references a row outside both is refused (CNA_EINVAL).  CNA_COMPACT_STATE=0 keeps the global row space. */
int  cna_set_halo(cna_ctx* ctx, const int64_t* send_rows, const int64_t* send_counts,
                  const int64_t* recv_rows, const int64_t* recv_counts);

/* ---- graph: data.obsp['connectivities'] (_nam.py:12-19,25) ---------------------------- */
/* CSR rows [row0, row0+n_local) of the n_global x n_global kNN graph.  indptr has n_local+1
 * entries rebased so indptr[0]==0; column indices are global.  data is float32 (what scanpy
 * emits) or float64.  Replaces the scipy.sparse object the reference reads. */
int  cna_graph_upload(cna_ctx* ctx, int64_t n_global, int64_t row0, int64_t n_local,
                      const int64_t* indptr, const int32_t* indices,
                      const void* data, int data_is_f64);
/* Optional: the rows handed to cna_graph_upload are a renumbering of the caller's cells (the
 * reference never depends on cell order, _nam.py:25-34; a banded numbering makes the walk's
 * gathers cache-friendly).  orig_index[i] = caller's index of local row i (n_local entries, a
 * slice of one global permutation).  Effect: cna_percell_fdr returns its per-cell outputs in the
 * caller's numbering; every other per-cell array crosses the ABI in library row order.  NULL
 * (and every cna_graph_upload) resets to the identity. */
int  cna_set_cell_order(cna_ctx* ctx, const int64_t* orig_index);
/* The resident graph into another cell order WITHOUT a second trip over PCIe (one rank holding the whole graph, resident
 * in the caller's order): perm[i] = caller's cell that becomes row i (n_global entries, a permutation -- checked).  Same
 * state afterwards as cna_graph_upload of the renumbered rows followed by cna_set_cell_order(perm), except that the column
 * sums and the sample codes are carried over (permuted), not dropped: rows keep the order of their entries, so every
 * result keeps its bits (reference: results do not depend on the order of the cells, _nam.py:25-34). */
int  cna_graph_reorder(cna_ctx* ctx, const int64_t* perm);
/* Sharded callers (one process per GPU, each holding only the cells of its row block -- the layout
 * SURVEY.md 8(e) asks for; the reference has no counterpart, its AnnData is whole): with the local
 * view on, every per-cell array that leaves the library covers this rank's n_local rows only --
 * cna_percell_fdr / _pinned (no all-reduce of cells-sized vectors), the flags of cna_zero_variance,
 * cna_fetch_cell_stat -- and cna_set_cell_order takes indices into the local block.  Sample-space
 * results (Gram, histograms, p-values, medians) stay global.  Call before cna_set_cell_order. */
int  cna_set_local_view(cna_ctx* ctx, int on);
/* colsums = A.sum(axis=0) + self_weight (_nam.py:28), float64, all-reduced over ranks */
int  cna_colsums(cna_ctx* ctx, double self_weight);
int  cna_fetch_colsums(cna_ctx* ctx, double* out_n_global);

/* ---- NAM construction (_nam.py:44-76) ------------------------------------------------ */
/* codes[i] = column of cell i in pd.get_dummies(obs[sid]) (_nam.py:51), for ALL n_global cells;
 * counts[c] = cells per sample C (_nam.py:54). */
int  cna_set_samples(cna_ctx* ctx, const int32_t* codes, int n_samples, const double* counts);
/* start a new walk from the one-hot state with the sample codes already on the device (same
 * effect as calling cna_set_samples again with identical arguments, without the upload) */
int  cna_restart_nam(cna_ctx* ctx);
/* One diffusion step s <- A.(s/colsums) + w*s/colsums (_nam.py:31-34) of the sample indicators.
 * The first call after cna_set_samples starts from the one-hot matrix.
 *   want_kurt : also produce per-cell kurtosis over samples of s/C (_nam.py:59), readable with
 *               cna_fetch_cell_stat;
 *   may_continue : keep the scaled state for another step (exchanged across ranks);
 *   may_stop  : also write NAM = s/C (_nam.py:73) so the walk can end here. */
int  cna_nam_step(cna_ctx* ctx, int want_kurt, int may_continue, int may_stop);
/* nsteps steps with a fixed step count and no per-step host decision (nsteps given, no progress
 * output): cna_nam_step(0, more, last) x nsteps in one call */
int  cna_nam_steps(cna_ctx* ctx, int nsteps);
/* A hint for the walk in progress: y[0..n_samples) is the standardised phenotype that the analysis will hand to
 * cna_select_standardized[_fused] with every cell and every sample kept and nothing to regress out
 * (_association.py:182,77; _nam.py:122,159).  The next cna_nam_step that ends its walk (may_continue = 0) then leaves,
 * besides the NAM, what that call computes from it -- X, its digit planes, the coefficients X.y/N, the zero-variance
 * count -- and the call finds its pass done (its results agree with the separate pass to rounding: the row sums run
 * over the lanes in another order).  Wide sample axes only (more than 64 samples); y = NULL clears; a selection call
 * that asks for anything else simply runs its own pass from the NAM. */
int  cna_nam_select_hint(cna_ctx* ctx, const double* y, int n_samples);
/* per-cell statistic of the last kernel that produced one (kurtosis / batch kurtosis),
 * gathered over ranks: out has n_global entries (CNA_MAT_NAM rows) or n_x_total (CNA_MAT_X) */
/* The whole walk of _nam.py:57-70 with nsteps=None in one call: steps are taken until the median over the cells of
 * the per-cell kurtosis (np.median, _nam.py:59) falls by less than 3 from one step to the next (checked from the
 * third step on, _nam.py:64-68), at most maxnsteps (<= 16).  The medians and the rule are evaluated on the device
 * and steps are queued ahead of the verdict (those behind the last step return at once); the host reads one word
 * per batch of steps.  *steps_out: steps taken; medkurt_out (maxnsteps doubles, may be NULL): the median after
 * every step taken.  Leaves the NAM of the last step on the device, like cna_nam_step(.., may_stop=1). */
int  cna_nam_auto(cna_ctx* ctx, int maxnsteps, int* steps_out, double* medkurt_out);
/* The same in two halves: _launch queues the first four steps with their medians and the rule and returns at once (the
 * host goes on with its own work); _finish reads the verdict, queues two more steps at a time while the rule is not met,
 * and leaves the context as cna_nam_auto does.  Every entry point that reads or replaces the NAM finishes a pending
 * walk itself, so calling _finish is only needed for its outputs. */
int  cna_nam_auto_launch(cna_ctx* ctx, int maxnsteps);
int  cna_nam_auto_finish(cna_ctx* ctx, int* steps_out, double* medkurt_out);
/* _qc_nam's decision (_nam.py:94-96) on the statistic left by cna_batch_kurtosis(CNA_MAT_NAM, ...): np.median of it,
 * threshold = max(6, 2 median), and the number of cells that fail `kurtosis < threshold` (NaN fails) -- 0 means every
 * cell is kept and the per-cell vector need not be fetched (cna_fetch_cell_stat).  Median, threshold and count are
 * formed on the device; one wait. */
int  cna_stat_qc(cna_ctx* ctx, double* median_out, double* threshold_out, int64_t* n_dropped_out);
int  cna_fetch_cell_stat(cna_ctx* ctx, double* out, int64_t n_expected);
/* np.median of that statistic over all cells (or all kept cells, all ranks), computed on the device
 * by an exact radix select: NaN if any entry is NaN, mean of the two middle values for an even count
 * (medians of _nam.py:59,94,150) */
int  cna_stat_median(cna_ctx* ctx, double* median_out);

/* cna.tl.diffuse / diffuse_stepwise on an arbitrary dense cells x m state (_nam.py:21-41):
 * load the local rows, step, fetch the local rows (unscaled state s). */
int  cna_dense_load(cna_ctx* ctx, const double* s_local, int m);
int  cna_dense_step(cna_ctx* ctx);
int  cna_dense_fetch(cna_ctx* ctx, double* s_local_out);

/* ---- QC / selection (_nam.py:78-99, _association.py:175-191) --------------------------- */
/* _batch_kurtosis (_nam.py:78-82) of matrix `which`: per cell, Pearson kurtosis over the
 * per-batch means; batch_codes[s] in [0,n_batches) per column of that matrix. */
int  cna_batch_kurtosis(cna_ctx* ctx, int which, const int32_t* batch_codes, int n_batches);
/* cells whose NAM entries are constant over the selected samples (NAM.std(axis=0)==0,
 * _association.py:182): flags (1 byte per cell, all n_global cells, gathered over ranks) and
 * their total number. colmap NULL = all samples. */
int  cna_zero_variance(cna_ctx* ctx, const int32_t* colmap, int n_sel, uint8_t* flags_out,
                       int64_t* n_zero_out);
/* X[i', c'] = NAM[keep_idx[i'], colmap[c']]  (NAM.reindex(y.index)[filter], iloc[:, keep], drop;
 * _nam.py:99, _association.py:178-185).  keep_idx: local row indices, NULL = all rows. */
int  cna_select(cna_ctx* ctx, const int64_t* keep_idx, int64_t n_keep,
                const int32_t* colmap, int n_sel);
/* cna_select followed by centring and division by the per-cell std (ddof=1) in one pass, for the
 * case M = I (_association.py:178-185 + _nam.py:122,159); n_zero_out = number of selected cells with
 * zero variance over the selected samples, summed over ranks (if non-zero the caller drops them
 * with cna_zero_variance + cna_select and standardises again).  With y (n_sel doubles, the
 * standardised phenotype in X's column order) the rows are final when they leave the kernel, so
 * cna_ncorrs(y) is taken in the same pass: max_abs_out = max |ncorrs| over all ranks. */
/* one-shot: the next cna_select_standardized[_fused] over N selected samples also applies M = I - C.W
 * (factors as in cna_resid_lowrank; _nam.py:128-135) between the centring and the division by the std */
int  cna_set_resid_factors(cna_ctx* ctx, const double* C, const double* W, int r, int N);
/* cna_select plus, in the same pass, the number of selected cells whose selected entries have zero variance
 * (_association.py:182-185); non-zero: the caller redoes the step with cna_zero_variance + cna_select */
int  cna_select_checked(cna_ctx* ctx, const int64_t* keep_idx, int64_t n_keep, const int32_t* colmap, int n_sel,
                        int64_t* n_zero_out);
int  cna_select_standardized(cna_ctx* ctx, const int64_t* keep_idx, int64_t n_keep,
                             const int32_t* colmap, int n_sel, int64_t* n_zero_out,
                             const double* y /* or NULL */, double* max_abs_out /* or NULL */);
/* The same, plus -- when no selected cell has zero variance -- what the host would queue next from
 * values it has to wait for here anyway: the Gram kernels (cna_gram_launch), the local null's
 * thresholds np.arange(m/4, m, m/400) from m = max|ncorrs| (returned in thr_out[<= 512], *T_out of
 * them; _association.py:101), cna_null_local_prepare(null_P, edges(thr), thr) and
 * cna_percell_coef_launch.  *T_out = 0: only the selection (and possibly the Gram) was issued.
 * null_col0 >= 0: the conditioned phenotypes of this analysis are resident already -- the caller vouches for it
 * (null_flag NULL: cna_condition_phenotypes has returned), or *null_flag == 1 at the moment the pass could start
 * (cna_host_draw_then_condition sets it); the prepared pass is then launched on columns null_col0 ... as
 * cna_null_local_launch(ctx, null_col0, null_P, NULL, T, 0, NULL) would, *null_launched = 1. */
int  cna_select_standardized_fused(cna_ctx* ctx, const int64_t* keep_idx, int64_t n_keep, const int32_t* colmap,
                                   int n_sel, int64_t* n_zero_out, const double* y, double* max_abs_out,
                                   int null_P, int* T_out, double* thr_out, int* gram_queued, int* coef_queued,
                                   int null_col0, const int* null_flag, int* null_launched);
/* numpy's thresholds / bin edges of the local null for a given max|ncorrs| (host arithmetic only) */
int  cna_reference_thresholds(double maxabs, int cap, double* thr, double* edges);
/* upload a cells x samples matrix as X (cna.tl.svd_nam on a user NAM, _nam.py:102) */
int  cna_upload_x(cna_ctx* ctx, const double* x_local, int64_t n_rows, int n_cols);

/* ---- residualisation + PCA (_nam.py:102-177) ------------------------------------------ */
/* X <- (X - rowmean(X))  if center  (_nam.py:122);  then X <- X . M^T if M != NULL
 * (_nam.py:135,148; M is n_cols x n_cols row-major). */
int  cna_resid_apply(cna_ctx* ctx, const double* M, int center);
/* X <- X / std(X over samples, ddof=1)  (_nam.py:159); center!=0 subtracts the mean first
 * (svd_nam's own re-standardisation, _nam.py:103-104). */
/* the same residualisation for a projector given by its factors, M = I - C.W (C: N x r standardised batches /
 * covariates, W = (C^T C + ridge N L)^-1 C^T: r x N, _nam.py:128-148): x.M^T = x - (x.W^T).C^T row by row, fused
 * with the centring (_nam.py:122) and optionally the division by the std (_nam.py:159) and the neighbourhood
 * coefficients X.y/N with their max |.| (_association.py:77,101): one pass over X instead of three */
int  cna_resid_lowrank(cna_ctx* ctx, const double* C, const double* W, int r, int center, int standardize,
                       const double* y, double* max_abs_out);
/* One ridge of the schedule of _nam.py:142-156 in one pass over X and one wait: centre, apply M = I - C.W (factors as
 * above), take the batch kurtosis of the result (_nam.py:150; batch_codes: batch of every sample of X, -1 = none) and
 * its np.median (on the device, *median_out), and -- optimistically -- divide by the std (_nam.py:159) and take the
 * coefficients X.y/N (_association.py:77; *max_abs_out = max |coefficient|).  median <= 6: the schedule is over and X is
 * final.  Otherwise X has to be restored by the caller (cna_select_checked from the NAM) before the next ridge. */
int  cna_resid_lowrank_bk(cna_ctx* ctx, const double* C, const double* W, int r, const double* y, double* max_abs_out,
                          const int32_t* batch_codes, int n_batches, double* median_out);
/* Selection, QC and the first ridge of the demo's call shape (covariates AND batches, demo/demo.ipynb:149) in ONE pass over
 * the NAM: what cna_batch_kurtosis(CNA_MAT_NAM) + cna_stat_qc (_nam.py:85-99), cna_select_checked with every cell and every
 * sample in place (_association.py:178-185) and cna_resid_lowrank_bk (_nam.py:136-159, first ridge) compute in three passes.
 * At most seven batches (the kurtosis of so few batch means cannot reach 6: a row fails the QC only with a NaN kurtosis), no
 * sample without a batch, at most 128 samples.  Out: *n_qc_failed rows with a NaN batch kurtosis, *n_zero rows constant over
 * the samples, *median_out / *max_abs_out as cna_resid_lowrank_bk.  When both counts are 0 and the median is <= 6, X is final;
 * otherwise the caller runs the three calls (the NAM is untouched).  *done = 0: shape not covered, nothing queued. */
int  cna_select_resid_bk(cna_ctx* ctx, const double* C, const double* W, int r, const double* y, double* max_abs_out,
                         const int32_t* batch_codes, int n_batches, double* median_out, int64_t* n_qc_failed,
                         int64_t* n_zero, int* done);
int  cna_standardize(cna_ctx* ctx, int center);
/* G = X^T X over all cells of all ranks (NAM.dot(NAM.T), _nam.py:105), n_cols x n_cols row-major */
int  cna_gram(cna_ctx* ctx, double* G_out);
/* the same in two halves: queue the kernels, collect the matrix later (the copy does not wait
 * for work queued after the launch, e.g. the local-null kernel) */
int  cna_gram_launch(cna_ctx* ctx);
int  cna_gram_fetch(cna_ctx* ctx, double* G_out);
/* cna_gram_fetch, then the leading kmax eigenpairs of G on the host (cna_host_top_eig: svd_nam's np.linalg.svd, _nam.py:105,
 * as the global test consumes it) and, when they pass the acceptance rule -- residual <= resid_tol * lambda_1,
 * orthogonality <= resid_tol, every leading gap > gap_tol * lambda_1 -- cna_global_test_launch(U, kmax, ks, K, r)
 * (_association.py:35-61,84) in the same call: *accepted = 1, collect with cna_global_test_fetch.  *accepted = 0: G_out
 * is valid, nothing was queued, the caller takes the eigenvectors elsewhere (LAPACK).  use_native = 0: fetch only. */
int  cna_gram_pcs_tests(cna_ctx* ctx, int kmax, const int32_t* ks, int K, int r, int use_native, double resid_tol,
                        double gap_tol, double* G_out /* n_cols x n_cols */, double* U_out /* n_cols x kmax */, int* accepted);
/* out = X . W  (V = NAM^T U / sqrt(svs), _nam.py:106; W = U/sqrt(svs), n_cols x n_w row-major),
 * local rows, row-major n_x_local x n_w */
int  cna_project(cna_ctx* ctx, const double* W, int n_w, double* out_local);
/* cna_project with the result left on the device: read it with cna_fetch_rows(CNA_MAT_PROJ, ...) */
int  cna_project_keep(cna_ctx* ctx, const double* W, int n_w);

/* ---- association (_association.py:77-120, _stats.py:34-83) ----------------------------- */
/* *yes = 1 when the working matrix X on the device is the standardised NAM of the resident walk with every cell and
 * every sample kept and nothing regressed out (what cna_select_standardized leaves for a call without covariates and
 * batches, _nam.py:122,159): it depends on the NAM only, not on the phenotype, so a further analysis of the same dataset
 * keeps it and takes its coefficients with cna_ncorrs (_association.py:77) instead of repeating the selection pass. */
int  cna_x_identity(cna_ctx* ctx, int* yes);
/* ncorrs = (y[:,None]*NAMresid).mean(axis=0) (_association.py:77); kept on the device and
 * optionally copied out (local rows); max_abs = max|ncorrs| over all ranks (_association.py:101). */
int  cna_ncorrs(cna_ctx* ctx, const double* y, double* out_local, double* max_abs);
/* Local null: for P' permuted, conditioned, standardised phenotypes Yc (n_cols x P row-major,
 * _association.py:96-97) count, per permutation p and threshold t,
 *   tails[p][t] = #{cells i : (|X_i . Yc_p| / n_cols)^2 >= edges[t]}
 * = tail_counts(thresholds, nullncorrs) of _stats.py:34-62 with
 * edges[t] = thr_t^2 - 1e-8 - 1e-5*thr_t^2 (ascending).  The cells x P' matrix of
 * _association.py:99 is never materialised.  tails_out is P x T int64, summed over ranks. */
int  cna_null_local(cna_ctx* ctx, const double* Yc, int P, const double* edges, int T,
                    int64_t* tails_out);
/* ---- the permutation test with the phenotypes resident on the device -------------------
 * cna_condition_phenotypes: M is N x N, Y is N x P row-major (observed phenotype in column 0, the
 * permuted ones after it, _association.py:80-83); computes, per column, zcond = M.z / std(M.z, ddof=1)
 * (_association.py:51-52,96-97) and keeps the result on the device.  Sample space only: it runs on
 * the context's second stream and may be called before the working matrix exists (while the
 * diffusion kernels execute); N must equal the matrix's column count when the tests run.
 * cna_null_local_resident: cna_null_local on columns [col0, col0+P) of that resident matrix;
 * tails_out (P x T) and tail_sums_out (T: sum over permutations, all the FDR needs) may each be NULL.
 * cna_global_test: _reg/_stats/_minp_stats (_association.py:35-61) for every resident column:
 * U is n_cols x kmax row-major (first kmax sample-PCs), ks[K] the PC counts tried, r the number
 * of conditioning columns; out: min over k of the F-test p-value (scipy.stats.f.sf semantics),
 * its r2 and the index of the chosen k (-1 when every p is NaN).  All three are replicated work
 * on every rank (sample space). */
int  cna_condition_phenotypes(cna_ctx* ctx, const double* M, const double* Y, int N, int P);
int  cna_null_local_resident(cna_ctx* ctx, int col0, int P, const double* edges, int T, int64_t* tails_out,
                             int64_t* tail_sums_out);
/* cna_null_local_resident in two halves: queue the pass (returns at once; at most one pending),
 * collect its results later.  Between the two the host may call cna_gram_fetch and cna_global_test,
 * which run beside the local-null kernel on a second stream. */
int  cna_null_local_launch(cna_ctx* ctx, int col0, int P, const double* edges, int T, int want_tails,
                           const double* thr /* or NULL: also queue cna_obs_counts(edges, thr) */);
/* the part of a launch that needs only the thresholds (exact cuts, their upload, the observed
 * counts), for callers that know them before the phenotypes are conditioned; follow it with
 * cna_null_local_launch(ctx, col0, P, NULL, T, want_tails, NULL).  No other entry point that uses the
 * context's main scratch (selection, ncorrs, percell) may run in between. */
int  cna_null_local_prepare(cna_ctx* ctx, int P, const double* edges, int T, int want_tails, const double* thr);
int  cna_null_local_fetch(cna_ctx* ctx, int64_t* tails_out, int64_t* tail_sums_out,
                          int64_t* ranks_out, int64_t* num_detected_out /* both NULL unless thr was given */);
/* Error path of _association.py:84-120 on the caller's side: a pass that was launched (cna_null_local_launch, or the
 * fused selection call) and never fetched because the caller raised in between is waited for and dropped, together
 * with a prepared half; the context is ready for the next analysis.  No-op when nothing is pending. */
int  cna_null_local_discard(cna_ctx* ctx);
/* Diagnostics of the last local-null pass that wanted only the sums over permutations (the
 * analysis): such a pass forms the products on the integer matrix cores (csrc/null_i8.hip: 24-bit
 * fixed point, exact int32 accumulation, outputs within the error bound of a cut recomputed in f64)
 * ; the f64 kernel runs the pass again when the integer one gives up.  used_out: 1 when the integer path ran; rechecked_out: outputs it
 * sent to the f64 recheck; fallback_out: 1 when it gave up (queue overflow) and the f64 kernel did
 * the work.  Waits for the device.  Environment CNA_NULL_F64=1 disables the integer path. */
int  cna_null_local_i8_stats(cna_ctx* ctx, int* used_out, int64_t* rechecked_out, int* fallback_out);
int  cna_global_test(cna_ctx* ctx, const double* U, int kmax, const int32_t* ks, int K, int r,
                     double* minp_out, double* r2_out, int32_t* kidx_out);
/* The same in two halves (launch returns at once; U and ks are copied before it returns). */
int  cna_global_test_launch(cna_ctx* ctx, const double* U, int kmax, const int32_t* ks, int K, int r);
int  cna_global_test_fetch(cna_ctx* ctx, double* minp_out, double* r2_out, int32_t* kidx_out);
/* ranks[t] = #{i : ncorrs_i^2 >= edges[t]} (_stats.py:74) and
 * num_detected[t] = #{i : |ncorrs_i| > thr[t]} (_association.py:108), summed over ranks */
int  cna_obs_counts(cna_ctx* ctx, const double* edges, const double* thr, int T,
                    int64_t* ranks_out, int64_t* num_detected_out);
/* data.obs[key] and data.obs[key+'_fdr'] (_association.py:230-237) for ALL n_global cells:
 * coef = NaN for cells not kept else ncorrs; fdr = min{fdr_t : thr_t <= |coef|} else 1.
 * runmin_fdr[t] = min(fdr[0..t]).  kept rows are the ones given to cna_select. */
int  cna_percell_fdr(cna_ctx* ctx, const double* thr, const double* runmin_fdr, int T,
                     double* coef_out_global, double* fdr_out_global);
/* same, into pinned host buffers owned by the context (n_global doubles each; valid until the next
 * call on this context): the D2H copies run at PCIe speed and the caller copies or consumes them */
int  cna_percell_fdr_pinned(cna_ctx* ctx, const double* thr, const double* runmin_fdr, int T,
                            double** coef_ptr, double** fdr_ptr);
/* The coefficient column alone, early: it depends on the observed phenotype only (not on the null),
 * so after cna_ncorrs / cna_select_standardized(y) the caller may queue it ahead of the local-null
 * kernel (launch: returns at once; the copy runs on the second stream under that kernel) and
 * collect the pinned pointer with _wait -- from any thread.  A later cna_percell_fdr_pinned then
 * delivers the FDR column only and returns the same coefficient pointer.  Single rank or local view. */
int  cna_percell_coef_launch(cna_ctx* ctx);
int  cna_percell_coef_wait(cna_ctx* ctx, double** coef_ptr);
/* The FDR column of a pending local-null pass (cna_null_local_launch after cna_percell_coef_launch: the column
 * follows the pass on the device) copied into dst[n] -- the storage of data.obs[key + '_fdr'],
 * _association.py:236-237 -- as soon as it has reached the pinned block: callable from a helper thread while
 * the main thread is inside the samples x samples SVD.  *done = 0: not applicable, nothing copied.
 * cna_percell_fdr_copied_early: *yes = 1 when the column cna_percell_fdr_pinned last returned is that copy. */
int  cna_percell_fdr_copy_early(cna_ctx* ctx, double* dst, int64_t n, int nthreads, int* done);
int  cna_percell_fdr_copied_early(cna_ctx* ctx, int* yes);

/* ---- the fixed-shape analysis in two library calls (round 6) ----------------------------------------------------------
 * cna.tl.association (_association.py:193-242) is ONE straight-line function; for the call shape that needs no decision
 * of the host between its stages -- nsteps given, one batch, covariates allowed, a seed, the local test on -- the stages
 * are queued here without the interpreter in between.  The caller validates the sample-level inputs (check_inputs,
 * _association.py:131-173), uploads graph and sample codes as usual (cna_graph_upload, cna_colsums, cna_set_samples /
 * cna_restart_nam), starts the permutation draw (cna_host_draw_start) and then calls
 *
 *   cna_assoc_begin   the walk of _nam.py:57-70 with a fixed step count: cna_nam_select_hint(y_hint) when given, then
 *                     cna_nam_steps(nsteps); nsteps = 0 keeps the NAM the device holds (the caller's NAM cache).
 *                     Returns once the kernels are queued: the caller builds the projector of _nam.py:128-135 and the
 *                     storage of the two data.obs columns meanwhile.
 *   cna_assoc_finish  compute_nam_and_reindex's selection (_association.py:175-191) + _resid_nam (_nam.py:118-177, no
 *                     batches) as cna_select_standardized_fused; the Gram matrix, its leading eigenpairs and the global
 *                     F-tests (_nam.py:105, _association.py:35-88) as cna_gram_pcs_tests + cna_global_test_fetch; the
 *                     conditioned phenotypes (cna_host_draw_then_condition, or cna_condition_phenotypes once the draw is
 *                     there); the local null (_association.py:91-120) as cna_null_local_launch / _fetch; the FDR table
 *                     (_stats.py:79-80, _association.py:105-108); and the two per-cell columns (_association.py:230-237),
 *                     copied into the caller's storage (coef_dst / fdr_dst) as they arrive.  Blocks until all of it is
 *                     done.  Every stage is the entry point named, in the order the Python host issues them: same bits.
 *
 * status (cna_assoc_out): CNA_ASSOC_DONE; CNA_ASSOC_GENERAL -- a selected cell has zero variance, max|ncorrs| is not
 * finite or the thresholds are out of range: nothing beyond the selection pass was issued, coef_dst / fdr_dst hold
 * nothing, the caller takes its general path (which reports what the reference reports); CNA_ASSOC_NEED_PCS -- the
 * library's eigen-solver stepped aside (cna_gram_pcs_tests: *accepted = 0): everything but the global test is done, G is
 * valid, the caller supplies eigenvectors (LAPACK) through cna_global_test_launch / _fetch; CNA_ASSOC_STALE -- an input
 * listed in `verify_*` no longer hashes to what the device copy was made from: as CNA_ASSOC_GENERAL, after a fresh upload.
 * cna_assoc_run = cna_assoc_begin + cna_assoc_finish for callers with nothing to do in between. */
#define CNA_ASSOC_DONE      0
#define CNA_ASSOC_GENERAL   1
#define CNA_ASSOC_NEED_PCS  2
#define CNA_ASSOC_STALE     3
#define CNA_ASSOC_MAXT      512
typedef struct cna_assoc_args {
  const int32_t* colmap;     /* NAM column of every analysed sample (NAM.reindex(y.index)[filter], _association.py:178-181); NULL: all, in place */
  int32_t n_sel;             /* analysed samples N */
  int32_t r;                 /* conditioning columns (covariates): M = I - C.W, _nam.py:128-135; 0: M = I */
  const double* y;           /* standardised phenotype (numpy ddof=0, _association.py:22), N */
  const double* M;           /* projector, N x N row-major */
  const double* resid_C;     /* N x r standardised covariates (NULL when r = 0) */
  const double* resid_W;     /* r x N: (C^T C)^-1 C^T */
  const int32_t* ks;         /* PC counts of the global test (_association.py:25-28) */
  int32_t K;
  int32_t Nnull;
  const double* table;       /* N x (Nnull + 1) row-major: [y | permuted phenotypes] (_association.py:80-83) */
  int32_t draw_pending;      /* 1: cna_host_draw_start is still filling `table` (this call joins it, the caller collects it) */
  int32_t conditioned;       /* 1: cna_condition_phenotypes(M, table, N, Nnull + 1) has already returned */
  int32_t use_native_eig;    /* cna_gram_pcs_tests' arguments */
  int32_t coef_first;        /* (unused since the eigenpairs run on a thread of their own: the coefficient column never waits for them) */
  double resid_tol, gap_tol;
  double* coef_dst;          /* storage of data.obs[key] / data.obs[key + '_fdr'] (n_dst doubles each), or NULL: */
  double* fdr_dst;           /*   pinned pointers are returned in cna_assoc_out instead */
  int64_t n_dst;
  int32_t copy_threads;
  int32_t n_verify;          /* content checks of the inputs the device copy was made from (the reference re-reads the graph and
                                the ids on every call, _nam.py:25-28,51): cna_host_hash64 of verify_ptr[i] (verify_bytes[i] bytes)
                                must equal verify_hash[i]; taken on a thread of this call while the device works.  A mismatch:
                                status CNA_ASSOC_STALE, nothing is written to coef_dst / fdr_dst */
  const void* verify_ptr[4];
  int64_t verify_bytes[4];
  uint64_t verify_hash[4];
  int32_t verify_threads;
  int32_t reserved;
  double* G;                 /* out: N x N Gram matrix */
  double* U;                 /* out: N x max(ks) leading eigenvectors (valid when eig_accepted) */
  double* minp;              /* out: Nnull + 1 (column 0: the observed phenotype) */
  double* r2;
  int32_t* kidx;
} cna_assoc_args;
typedef struct cna_assoc_out {
  int32_t status, T, eig_accepted, null_fused;
  int32_t coef_in_dst, fdr_in_dst;
  int64_t n_zero;
  double max_abs;
  double* coef_ptr;          /* pinned, valid until the next per-cell call on the context (when not copied to coef_dst) */
  double* fdr_ptr;
  double t_ms[16];           /* when the stages of this call were reached, ms from its entry: [0] phenotypes posted, [1] selection
                                pass back (the walk is over), [2] local null queued, [3] inputs verified, [4] coefficient column
                                out, [5] local null over + FDR column out, [6] null results, [7] eigenpairs + F-tests joined,
                                [8] exit; on the eigenpairs thread: [9] Gram matrix on the host, [10] eigenpairs done and F-tests queued, [11] F-tests fetched; on the draw thread (negative: before this call was entered): [12] permutations drawn,
                                [13] phenotypes conditioned (the flag the fused selection call reads) */
  double thr[CNA_ASSOC_MAXT], fdr[CNA_ASSOC_MAXT], runmin[CNA_ASSOC_MAXT];
  int64_t tail_sums[CNA_ASSOC_MAXT], ranks[CNA_ASSOC_MAXT], num_detected[CNA_ASSOC_MAXT];
} cna_assoc_out;
int  cna_assoc_begin(cna_ctx* ctx, int nsteps, const double* y_hint, int n_hint);
/* the same walk in parts: steps first .. first + count - 1 (0-based) of `total`; y_hint applies to the part that holds the last
 * step.  A caller queues the steps that need only graph and sample codes before it has validated the phenotype. */
int  cna_assoc_begin_part(cna_ctx* ctx, int first, int count, int total, const double* y_hint, int n_hint);
int  cna_assoc_finish(cna_ctx* ctx, const cna_assoc_args* args, cna_assoc_out* out);
int  cna_assoc_run(cna_ctx* ctx, int nsteps, const double* y_hint, int n_hint, const cna_assoc_args* args, cna_assoc_out* out);
/* blocks until the request of cna_host_draw_start (and its follow-up) is done WITHOUT collecting it: the caller's
 * cna_host_draw_wait still returns its status (0 / -1 as cna_host_draw_wait; -2: nothing was started) */
int  cna_host_draw_join(void);
/* out2 = {when the last draw finished, when its follow-up (the conditioning) did}, CLOCK_MONOTONIC seconds: diagnostics */
void cna_host_draw_times(double* out2);

/* ---- device -> host for the lazily materialised result fields (a20) -------------------- */
int  cna_matrix_shape(cna_ctx* ctx, int which, int64_t* n_rows_local, int* n_cols);
/* transposed!=0 writes samples x cells (the reference's orientation), else cells x samples */
int  cna_fetch_matrix(cna_ctx* ctx, int which, double* out, int transposed);
/* the same with rows and columns picked and ordered on the device: out[i][j] = M[rows[i]][cols[j]]
 * (n_out x n_cols, or its transpose); rows / cols NULL = all, in order.  rows are local row indices
 * of the matrix.  One gather kernel + one contiguous copy: the caller's cell order and orientation
 * (res.nam, res.namresid, cna.tl.nam: _nam.py:73,193) cost no cells x samples reshuffle on the host. */
int  cna_fetch_rows(cna_ctx* ctx, int which, const int64_t* rows, int64_t n_out,
                    const int32_t* cols, int n_cols, double* out, int transposed);

/* concatenate count_local doubles from every rank, in rank order, into out_all on every rank
 * (row blocks of the lazily fetched matrices when the job spans several GPUs) */
int  cna_allgather_host(cna_ctx* ctx, const double* local, int64_t count_local, double* out_all,
                        int64_t count_total);

/* ---- host-side helper: the permutation draw's random stream (_stats.py:10) ----------------- */
/* n values of np.random.randn from numpy's legacy generator, bit for bit, vectorised (host code
 * only, no device involved; csrc/host_rng.c).  key[624] / *pos / *has_gauss / *gauss: the state as
 * np.random.get_state() reports it, advanced in place to where numpy's own draw would leave it.
 * No context: callable from any thread. */
int  cna_host_legacy_randn(uint32_t* key, int* pos, int* has_gauss, double* gauss, int64_t n, double* out);
/* threads the two host helpers of csrc/host_rng.c may use for LARGE draws (default 1; results do not depend on it) */
void cna_host_set_threads(int n);
/* out[rows[i]][c] = y[argsort(R[:, c])[i]] for the m x num draws R (row-major): the permuted phenotypes of
 * conditional_permutation / grouplevel_permutation (_stats.py:11-17,31) for large draws, columns split over
 * threads (host only; rows NULL = identity) */
int  cna_host_argsort_gather(const double* R, int m, int num, const double* y, double* out, int64_t ld_out,
                             const int64_t* rows);
/* The whole draw of conditional_permutation (reference _stats.py:4-18: per level of the batch vector, in np.unique
 * order, Y[members][argsort(randn(len(members), num), axis=0)]) on the library's own host thread, so that it runs
 * beside the caller's interpreter instead of inside it.  key / pos: numpy's MT19937 state memory, freshly seeded (no
 * cached normal), num even; members[lev_off[l] .. lev_off[l+1]) = rows of level l; out[r * ld_out + p] receives the
 * permuted phenotypes.  All pointers stay valid until cna_host_draw_wait() has returned (0; -1: the draw failed and the
 * generator state is undefined; -2: nothing started).  cna_host_draw_start: 0 = accepted, -1 = not accepted, nothing
 * touched.  One request at a time per process. */
int  cna_host_draw_start(uint32_t* key, int* pos, const double* y, int m, int num, int nlev, const int64_t* lev_off,
                         const int64_t* members, double* out, int64_t ld_out, int threads);
/* cna_host_draw_start that also records WHICH row of y every output was taken from (idx_out: m x num int32 row-major): the
 * permutations of a seeded draw depend on (seed, m, num, levels) only -- not on y -- so a caller that tests many phenotypes
 * with one seed (the demo does: demo/demo.ipynb:156,251) replays them with cna_host_gather_rows, out[r][p] = y[idx[r][p]]
 * (= Y[bix], _stats.py:16-17), instead of drawing and sorting again; numpy's generator is then put where the recorded draw
 * left it (the caller keeps that state next to idx). */
int  cna_host_draw_start_idx(uint32_t* key, int* pos, const double* y, int m, int num, int nlev, const int64_t* lev_off,
                             const int64_t* members, double* out, int64_t ld_out, int threads, int32_t* idx_out);
int  cna_host_gather_rows(const double* y, const int32_t* idx, int m, int num, double* out, int64_t ld_out, int nthreads);
/* follow-up of the request under way: the worker conditions the phenotypes itself when the draw is there --
 * cna_condition_phenotypes(ctx, M, table, N, cols) with table = the N x cols matrix [y | permutations] being filled --
 * and stores 1 / -1 (failed) in *flag.  0 accepted, -1 nothing to follow.  cna_host_draw_wait covers it. */
int  cna_host_draw_then_condition(cna_ctx* ctx, const double* M, const double* table, int N, int cols, int* flag);
int  cna_host_draw_wait(void);

/* ---- host-side helpers: graph identity and the device cell order (csrc/host_graph.c) ------- */
/* 64-bit content hash of a buffer, computed on up to nthreads threads (the value does not depend on the
 * thread count).  The engine keys the resident graph on the hash of ALL of data / indices / indptr, so an
 * in-place edit of any entry re-uploads the graph -- the reference reads `data.obsp['connectivities']`
 * afresh on every call (_nam.py:25-28). */
uint64_t cna_host_hash64(const void* p, int64_t nbytes, int nthreads);
/* memcpy on up to nthreads threads (the per-cell result columns into the caller's frame,
 * _association.py:230-237 of the reference assigns them to data.obs) */
int  cna_host_copy(void* dst, const void* src, int64_t nbytes, int nthreads);
/* dst[i] = bins[i] > 0 ? runmin[bins[i] - 1] : 1.0 on nthreads threads: the per-cell FDR column (_association.py:234-237)
 * from the per-cell threshold counts #{t : thr_t <= |coef_i|} and the running minimum of the FDR table */
int  cna_host_expand_u16(double* dst, const uint16_t* bins, int64_t n, const double* runmin, int T, int nthreads);
/* Device cell order: clusters of B cells grown greedily by "most edges into the cluster" so that the
 * rows of one block share neighbours (edges / distinct neighbour rows of a block: 3.0 at B = 64 against
 * 2.0 for reverse Cuthill-McKee).  indptr int64[n+1], indices int32 of the rows' columns (columns outside
 * [0, n) are ignored); order_out[i] = caller's index of device row i.  Returns the number of cells placed
 * in full clusters, -1 when out of memory.  Only the numbering changes: every row keeps its neighbours in
 * the caller's CSR order, so no sum is reordered (_nam.py:33). */
int64_t cna_host_cluster_order(int64_t n, const int64_t* indptr, const int32_t* indices, int B, int64_t* order_out);
/* The same on several threads: the cells are first split into n / 65536 (<= 64) connected regions by a multi-source
 * breadth-first search, every region is ordered as above on its own (in parallel), full clusters of all regions first.
 * The result depends on n and the graph only, not on nthreads.  Below 131072 cells: the sequential order. */
int64_t cna_host_cluster_order_mt(int64_t n, const int64_t* indptr, const int32_t* indices, int B, int nthreads,
                                  int64_t* order_out);
/* graph of the clusters of a cell order (cluster c = order[c * B .. (c + 1) * B)): per cluster the clusters its cells
 * have edges into and how many; first call with col == NULL fills ptr[nc + 1] and returns the entry count, the second
 * fills col / cnt.  Host only; the blocks of a sharded run are packed from it (cna_amd._order.partition_order). */
int64_t cna_host_cluster_graph(int64_t n, const int64_t* indptr, const int32_t* indices, const int64_t* order, int B,
                               int64_t* ptr, int32_t* col, int64_t* cnt);
/* Rows [r0, r1) of the graph in the device order: out row i = caller's row perm[r0 + i], columns relabelled
 * through col_map and left in their original order; values (vbytes = 4 | 8) copied bit for bit.  Sizes from
 * cna_host_permuted_nnz.  Threaded; integer work only. */
int64_t cna_host_permuted_nnz(const int64_t* perm, int64_t r0, int64_t r1, const int64_t* indptr);
int  cna_host_permute_rows(const int64_t* perm, int64_t r0, int64_t r1, const int64_t* indptr, const int32_t* indices,
                           const void* data, int vbytes, const int64_t* col_map, int64_t* out_indptr,
                           int32_t* out_indices, void* out_data, int nthreads);

/* ---- benchmark / test input: kNN connectivities graph built on the device (csrc/knn.hip) ---- */
/* Stand-in for scanpy.pp.neighbors (demo/demo.ipynb:590, makedata.ipynb:117) on synthetic points: exact
 * brute-force kNN of X (n x d float32, d <= 64; k counts the point itself, 2 <= k <= 65), UMAP smooth-kNN
 * weights, fuzzy union A + A^T - A o A^T; CSR with sorted int32 column indices, float32 values, empty
 * diagonal.  indices_out / data_out: room for 2 n (k-1) entries.  An input generator, not a parity target. */
int  cna_knn_graph(cna_ctx* ctx, const float* X, int64_t n, int d, int k, int64_t* indptr_out,
                   int32_t* indices_out, float* data_out, int64_t* nnz_out);

/* ---- measurement ------------------------------------------------------------------------ */
/* HIP-event timing of every kernel launch on the context's stream (bench.py roofline).  on = 1: every kernel group;
 * on = 2: the walk kernels (CNA_K_NAM_FIRST / _STEP / _STEP_SPARSE) and the communication spans only -- two event records
 * per span cost the host ~0.1 ms of a 1.2 ms analysis when all ~20 groups are timed; 0: off. */
int  cna_prof_enable(cna_ctx* ctx, int on);
int  cna_prof_reset(cna_ctx* ctx);
int  cna_prof_get(cna_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);
const char* cna_kernel_name(int kernel_id);

/* svd_nam's `np.linalg.svd(NAM.dot(NAM.T))` (_nam.py:105) as the global test consumes it (_association.py:35-48: the
 * first k <= max(ks) vectors, through squared projections only): the k leading eigenpairs of the symmetric n x n matrix
 * G (row-major) on the host -- Householder tridiagonalisation, bisection, inverse iteration (csrc/host_eig.c) -- with
 * the evidence the caller needs to accept them or fall back to LAPACK: U_out n x k row-major (column t belongs to the
 * t-th largest eigenvalue, sign arbitrary), lam_out the k + 1 largest eigenvalues, *resid_out / *ortho_out = largest
 * residual and largest orthogonality defect of the eigenvectors of the tridiagonal stage (the one that can fail; the
 * reflectors around it are orthogonal to rounding).  Returns 0; 1 = breakdown, 2 = bad arguments (k + 1 <= min(n, 260)),
 * -1 = no memory.  Needs no context and no GPU. */
int  cna_host_top_eig(const double* G, int n, int k, double* U_out, double* lam_out, double* resid_out, double* ortho_out);
/* The same two figures measured on G itself: max_t ||G u_t - lam_t u_t||_inf, max |U^T U - I| (tests). */
int  cna_host_eig_check(const double* G, int n, int k, const double* U, const double* lam, double* resid_out, double* ortho_out);

#ifdef __cplusplus
}
#endif
#endif /* CNA_HIP_H */
