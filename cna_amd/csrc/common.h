// Shared declarations of libcna_hip.so (gfx950 only).  See include/cna_hip.h for the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <mutex>
#include <atomic>
#include <vector>
#include "../../include/cna_hip.h"

#define WAVE 64
#define I8_QMAX 8355000.0   // largest |q| of the 24-bit fixed point of null_i8.hip: 127*65536 + 127*256 + 127 = 8355711

void cna_set_error(const std::string& msg);

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      cna_set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ +  \
                    ":" + std::to_string(__LINE__) + ")");                               \
      return (int)_e;                                                                      \
    }                                                                                      \
  } while (0)

#define CNA_TRY(expr)            \
  do {                           \
    int _r = (expr);             \
    if (_r != 0) return _r;      \
  } while (0)

#define CNA_FAIL(code, msg)      \
  do {                           \
    cna_set_error(msg);          \
    return (code);               \
  } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct ProfSpan {
  int kid;
  hipEvent_t a, b;
};

// X^T X in parts (mfma.hip: gram_plan / launch_gram_range / launch_gram_finish)
struct GramPlan {
  int Nx = 0, nt = 0, ntri = 0, ldp = 0, slab_rows = 32, nblocks = 1, nw = 16;
  bool use_blk = false;
  int64_t nslab = 0;
  int32_t* tiles_dev = nullptr;
  double* partial = nullptr;
  size_t smem = 0;
};

struct cna_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;   // D2H of small results that must not queue behind long kernels
  hipStream_t coef_stream = nullptr;   // the early coefficient column's copy: the helper thread waits on copy_stream and must not wait for this
  hipEvent_t gram_done = nullptr;
  hipEvent_t scal_ready = nullptr;     // the selection pass's two scalars are in h_scal (the Gram kernels may already be queued behind them)
  double* gram_buf = nullptr;          // Gram matrix of the last cna_gram_launch
  hipEvent_t null_done = nullptr;      // results of the last local-null launch are in h_res
  void* h_hint = nullptr;              // pinned staging of cna_nam_select_hint's phenotype (8 KB)
  void* h_gram = nullptr;              // pinned staging of cna_gram_fetch (the caller's matrix is pageable: the runtime would stage the copy itself, slower)
  int64_t h_gram_cap = 0;
  double* gram_mirror = nullptr;  // set around a Gram launch: the reduction also writes G here (the pinned h_gram)
  bool gram_mirrored = false;     // ... and says so
  double t_gram_fetched = 0.0;   // steady-clock seconds at which cna_gram_pcs_tests had the Gram matrix on the host (stage marks of cna_assoc_finish)
  void* h_scal = nullptr;              // pinned: a few words for scalar results a call waits for (a copy into pageable memory -- a stack
                                       // variable -- is staged by the runtime: 20-25 us each, measured between the selection pass and the Gram kernel)
  void* h_res = nullptr;               // pinned host staging for asynchronously fetched results
  int64_t h_res_cap = 0;
  int null_P = 0, null_T = 0, null_has_tails = 0;
  int null_col0 = 0;                   // first column of Zc of the pending pass
  int64_t null_status_off = -1;        // >= 0: h_res + off holds the integer pass's status word of the pending pass (0: its sums stand)
  std::atomic<int> null_pending{0};   // (read by the helper thread's cna_percell_fdr_copy_early, like the four flags below)
  int64_t gram_cap = 0;
  int gram_n = 0;
  std::atomic<int64_t> dev_bytes{0};   // (two host threads may reserve buffers of one context at once: the F-tests from the eigenvector thread)
  std::recursive_mutex alloc_mu;   // dev_alloc / dev_free / dev_reserve (two host threads may share the context)

  // ---- communicator
  int rank = 0, nranks = 1;
  void* comm = nullptr;  // ncclComm_t
  void* comm_halo = nullptr;   // ncclComm_t: a duplicate of `comm` (ncclCommSplit) for the halo stream; null: exchanges stay on the main stream
  void* shm = nullptr;   // host-staged communicator of cna_comm_init_shm (several ranks on one GPU: tests)
  // neighbour ("halo") exchange of the diffusion state, replacing the all-gather (cna_set_halo)
  bool halo_on = false;
  int64_t* halo_send_idx = nullptr;   // device: local rows to ship, grouped by destination rank
  int64_t* halo_recv_idx = nullptr;   // device: global rows received, grouped by source rank
  std::vector<int64_t> halo_send_cnt, halo_recv_cnt;
  int64_t halo_ns = 0, halo_nr = 0;
  // overlap of the exchange with the step that produces it: the rows other ranks wait for are walked first, their
  // exchange runs on halo_stream while the rest of the block is walked (cna_nam_step)
  int32_t* halo_rows_b = nullptr;     // device: local rows some other rank asked for (ascending, each once)
  int32_t* halo_rows_i = nullptr;     // device: the other local rows (ascending)
  int64_t halo_nb = 0, halo_ni = 0;
  hipStream_t halo_stream = nullptr;
  hipEvent_t halo_e1 = nullptr, halo_e2 = nullptr;
  void* halo_sbuf = nullptr;
  void* halo_rbuf = nullptr;
  int64_t halo_sbuf_cap = 0, halo_rbuf_cap = 0;

  // ---- graph (local row block)
  int64_t n_global = 0, row0 = 0, n_local = 0, rows_per_rank = 0, n_pad = 0, nnz = 0;
  int64_t* indptr = nullptr;
  int32_t* indices = nullptr;
  void* data = nullptr;
  int data_f64 = 0;
  double self_weight = 1.0;
  double* colsum = nullptr;  // n_pad, replicated
  int64_t* orig_idx = nullptr;  // n_local: caller's cell index of local row i; null = identity
  bool local_view = false;      // per-cell outputs cover this rank's rows only (cna_set_local_view)
  bool have_colsum = false;

  // ---- samples
  int N = 0, ld = 0;
  int32_t* sid = nullptr;    // n_global
  int64_t sid_n = 0;
  double* counts = nullptr;  // N
  void* cellinfo = nullptr;  // n_global x {1/colsums, sid}: one gather per edge in the first step
  int64_t cellinfo_cap = 0;
  bool cellinfo_valid = false;

  // ---- diffusion state: scaled state T = s/colsums.  Rows: every global row (one rank; the all-gather exchange), or --
  // with the halo exchange (t_compact, round 5; SURVEY 8e: "GPU g owns rows ... of A and of S") -- this rank's n_local
  // rows followed by the halo rows in the order of halo_recv_idx; the walk steps then read the graph through idx_t,
  // the column indices renumbered into that row space.
  bool t_compact = false;
  int64_t t_rows = 0;            // rows of T (and of sp_cnt) when compact
  int32_t* idx_t = nullptr;      // nnz: column indices in the compact row space
  int64_t idx_t_n = 0;
  // rows of the block that read NO foreign state row (safe) / at least one (need): a step can walk the safe rows while the
  // exchange that feeds it is still in flight (halo_wait_pending: recorded in halo_e2, the main stream has not waited yet)
  int32_t* halo_rows_safe = nullptr;
  int32_t* halo_rows_need = nullptr;
  int64_t halo_nsafe = 0, halo_nneed = 0;
  bool halo_wait_pending = false;
  bool halo_safe_launch = false;        // set around a launch over halo_rows_safe (launch_nam_step then does not wait)
  bool sp_dense_interior_off = false;   // set around the launch over the rows no other rank asked for (cna_nam_step)
  int64_t sp_pair_rows = 0;      // rows of sp_pair (pairs exist for this rank's own rows only when compact)
  double* T[2] = {nullptr, nullptr};
  int64_t t_cap = 0;  // doubles allocated per buffer
  int t_cur = 0, t_width = 0, t_ld = 0, steps_done = 0;
  bool t_valid = false;
  // opt-in (cna_set_state_f32): the scaled state BETWEEN two steps of a walk is stored in 4 bytes per entry
  // (sums still run in f64).  Halves what the dense step gathers; the NAM then equals the f64 one to ~1e-7 relative
  // instead of bit for bit.  t_f32[i]: what T[i] holds right now.
  bool state_f32_mode = false;
  bool t_f32[2] = {false, false};
  double* dense_s = nullptr;  // unscaled local state of the dense diffusion (n_local x t_ld)
  int64_t dense_cap = 0;

  // ---- NAM (n_local x ld)
  double* nam = nullptr;
  int64_t nam_cap = 0;
  bool nam_valid = false;

  // ---- X (nx x ldx): selected NAM -> residualised NAM
  double* X = nullptr;
  int64_t x_cap = 0;
  // projector factors for the next cna_select_standardized (cna_set_resid_factors): W | C^T, rk x Nx each
  double* resid_f = nullptr;
  int64_t resid_f_cap = 0;
  int resid_rk = 0, resid_n = 0;
  double* X2 = nullptr;          // second working matrix: target of the k-split residualisation (N > 256), swapped with X
  int64_t x2_cap = 0;
  int64_t nx = 0;
  int Nx = 0, ldx = 0;
  int64_t* keep_idx = nullptr;    // active map: local NAM row of each X row; null = identity
  int64_t* keep_store = nullptr;  // allocation behind keep_idx
  int64_t keep_cap = 0;
  bool x_valid = false, x_from_nam = false;
  bool x_ident = false;            // X = standardised NAM, every cell, samples in place, nothing regressed out (cna_x_identity)
  // The selection pass as a by-product of the walk's last step (cna_nam_select_hint): when the caller says which
  // standardised phenotype the analysis will use and nothing will be filtered or regressed out, the last step's
  // write-out also leaves X = centred / standardised NAM, its digit planes, the coefficients X.y/N and the
  // zero-variance count -- what cna_select_standardized(all cells, samples in place, y) would produce from the NAM
  // it has just written -- and that call then finds its work done (the 1.5 ms pass at 2M x 200 leaves the path).
  std::vector<double> byp_hint;    // y of a pending hint (consumed by the next last step)
  std::vector<double> byp_y;       // y the by-product on the device was made for
  bool byp_valid = false;          // X / planes / coefficients on the device are that by-product
  bool byp_with_q = false;
  bool byp_skip_nam = false;       // ... and it does not write the NAM (nam_lazy afterwards)
  bool nam_lazy = false;           // the NAM is one more run of the last step away (c_api.hip:need_nam)
  int lazy_steps_before = 0;       // steps_done when that step was launched
  bool byp_arm = false;            // the launch being issued is that last step (launch_nam_step reads it)
  void* pair_buf = nullptr;        // {count, max bits} of every rank: the selection pass's two counters in one collective
  void* byp_buf = nullptr;         // device: [0] zero-variance rows, [1] max |coef| bits, then y (1024 doubles)

  // ---- per-cell vectors
  double* stat = nullptr;  // n_pad
  int stat_space = -1;     // CNA_MAT_NAM / CNA_MAT_X
  // walk with the stop rule on the device (cna_nam_auto): bookkeeping block (rows.hip:AutoState) and, while such a
  // walk is being queued, the address of its `stopped_at` word for the step kernels
  void* auto_state = nullptr;
  const int* auto_stop = nullptr;
  bool auto_pending = false;   // cna_nam_auto_launch has queued steps whose verdict nobody has read yet
  int auto_max = 0, auto_queued = 0;
  double* ncorrs = nullptr;
  int64_t ncorrs_cap = 0;
  bool ncorrs_valid = false;

  // ---- resident conditioned phenotypes Zc (ldx x zc_ld), zc_cols valid columns
  double* zc = nullptr;
  int64_t zc_cap = 0;
  int zc_ld = 0, zc_cols = 0, zc_rows = 0;
  void* gt = nullptr;        // global-test staging (U, ks, outputs)
  int64_t gt_cap = 0;

  // ---- scratch
  void* scratch = nullptr;
  int64_t scratch_cap = 0;
  void* proj = nullptr;           // result of cna_project_keep (rows of X x proj_cols, leading dimension proj_ld)
  int64_t proj_cap = 0, proj_rows = 0;
  int proj_ld = 0, proj_cols = 0;
  bool proj_valid = false;
  int null_prepared = 0;          // cna_null_local_prepare done, launch still to come
  double null_cut0 = 0, null_inv_step = 0, null_eps = 0;
  int null_has_obs = 0;
  int64_t null_obs_off = 0;
  void* h_gt = nullptr;           // pinned staging of cna_global_test_launch / _fetch: U, ks | minp, r2, kidx
  int64_t h_gt_cap = 0;
  hipEvent_t gt_done = nullptr;
  hipEvent_t stage_done = nullptr;   // uploads out of h_res' staging tail (cna_null_local_prepare) have been issued and finished
  int gt_pending_P = 0;           // > 0: a launched global test waits to be fetched (its number of columns)
  int64_t gt_off_out = 0;
  double* coef_dev = nullptr;     // early copy of the per-cell coefficients (cna_percell_coef_launch): 2 x n_pad
  int64_t coef_dev_cap = 0;
  hipEvent_t coef_ready = nullptr, coef_copied = nullptr;
  std::atomic<bool> coef_early{false};        // h_cell[0, n_out) already holds the coefficients of the current ncorrs
  std::atomic<bool> fdr_inline{false};        // ... and h_cell[n_out, 2 n_out) the per-cell FDRs of the last local-null pass
  double null_thr0 = 0, null_thr_step = 0;   // linear guess over the thresholds of the prepared pass
  std::atomic<bool> fdr_early_copied{false};  // cna_percell_fdr_copy_early took the FDR column of the pending pass ...
  std::atomic<bool> fdr_early_served{false};  // ... and cna_percell_fdr_pinned then returned that same column
  // the FDR column behind a local-null pass that was launched with the coefficient column already out (fdr_inline):
  // the per-cell threshold counts (16 bits, caller's order) cross PCIe while the null runs, the FDR table follows it,
  // and the host puts the two together (cna_percell_fdr_copy_early / cna_percell_fdr_pinned)
  unsigned short* bins_dev = nullptr;
  int64_t bins_cap = 0;
  unsigned short* h_bins = nullptr;   // pinned
  int64_t h_bins_cap = 0;
  double* h_tab = nullptr;            // pinned: running minimum of the FDR table (512 doubles)
  hipEvent_t bins_copied = nullptr;
  bool bins_pending = false;          // the copy on coef_stream may still read the thresholds in c->scratch
  double* fdr_early_dst = nullptr;    // where cna_percell_fdr_copy_early put the column
  std::atomic<int> fdr_early_inflight{0};   // a helper thread is inside cna_percell_fdr_copy_early
  void* h_cell = nullptr;         // pinned: per-cell outputs of cna_percell_fdr_pinned (coef | fdr)
  int64_t h_cell_cap = 0;
  // compressed copy of the state after the first walk step (single GPU, wide sample axis)
  void* sp_pair = nullptr;
  void* sp_cnt = nullptr;
  int64_t sp_rows = 0;
  void* i8_buf = nullptr;         // digit planes, queue and slabs of the integer local-null path (null_i8.hip)
  int64_t i8_cap = 0;
  void* xq = nullptr;             // digit planes of X ([row][digit][32 KS] bytes) and per-row {max |x|, sum |q|}, written by
  int64_t xq_cap = 0;             //   the pass that produced X (k_select_std) or by k_quant_x
  void* xq_scale = nullptr;
  int64_t xq_scale_cap = 0;
  bool xq_valid = false;          // they describe the current X
  int64_t xq_rows = 0;
  int xq_KS = 0;
  bool i8_last = false;           // the last local-null pass took the integer path
  unsigned long long* i8_qcount = nullptr;   // device: [0] outputs sent to the f64 recheck by the last pass, [1] low word = status
  void* null_part = nullptr;      // per-block counter slabs of the local-null kernel
  int64_t null_part_cap = 0;
  void* rp16_buf = nullptr;       // operands of rows16.hip:k_rowpass16 (k_rowpass16_prep)
  int64_t rp16_cap = 0;
  void* scratch2 = nullptr;
  int64_t scratch2_cap = 0;
  // The Gram matrix of the selection by-product taken UNDER the walk's last step (cna_nam_step): that step runs in
  // row ranges on the main stream, and behind each range's event the Gram kernel of the same rows runs on gram_stream
  // (the gather leaves the matrix pipe idle); partial tiles carry over from range to range in gram_part, so the sum
  // is the one launch_gram would form.  gram_pre: c->gram_buf holds (will hold, after gram_pre_done) X^T X of the
  // by-product X, not yet summed over the ranks.
  hipStream_t gram_stream = nullptr;
  int gram_stream_state = 0;             // 0: not created yet, 1: ready, -1: unavailable
  hipEvent_t gram_pre_done = nullptr, range_done = nullptr;
  bool gram_pre = false;
  bool gram_pre_pending = false;         // kernels of a ranged product may still run on gram_stream (nobody has waited for gram_pre_done yet)
  void* gram_part = nullptr;
  int64_t gram_part_cap = 0;
  GramPlan gram_pre_plan;
  int gram_tiles_nt = -1;        // upper-triangular tile table of the Gram kernel (depends on nt only)
  void* gram_tiles_ptr = nullptr;
  int64_t gram_tiles_cap = 0;

  // ---- profiling
  bool prof = false;
  uint64_t prof_mask = ~0ull;      // kernel ids that are timed while prof is on (cna_prof_enable level 2: the walk and the communication spans)
  double prof_ms[CNA_K_COUNT] = {0};
  int64_t prof_n[CNA_K_COUNT] = {0};
  std::vector<ProfSpan> prof_pending;
  std::vector<hipEvent_t> ev_pool;
  std::mutex prof_mu;            // the helper thread's cna_condition_phenotypes records spans while the main thread launches kernels
};

// profiling helpers (c_api.hip)
void prof_begin(cna_ctx* c, int kid, hipStream_t st);
void prof_end(cna_ctx* c, int kid, hipStream_t st);
struct ProfScope {
  cna_ctx* c;
  int kid;
  hipStream_t st;
  ProfScope(cna_ctx* c_, int k, hipStream_t s = nullptr) : c(c_), kid(k), st(s ? s : c_->stream) {
    if (c->prof && kid >= 0 && (c->prof_mask >> kid & 1)) prof_begin(c, kid, st);
  }
  ~ProfScope() { if (c->prof && kid >= 0 && (c->prof_mask >> kid & 1)) prof_end(c, kid, st); }
};

int dev_alloc(cna_ctx* c, void** p, size_t bytes);
int dev_free(cna_ctx* c, void* p, size_t bytes);
// grow-only buffer: (re)allocates *p when cap < need (contents discarded)
int dev_reserve(cna_ctx* c, void** p, int64_t* cap_bytes, int64_t need_bytes);

// ---- collectives (comm.hip)
inline bool comm_active(const cna_ctx* c) { return c->comm != nullptr || c->shm != nullptr; }
int comm_halo_exchange(cna_ctx* c, const double* sendbuf, double* recvbuf, int64_t doubles_per_row, hipStream_t st = nullptr);
// column indices of the local graph block -> rows of the compact state (cna_ctx::t_compact): own rows 0 .. n_local - 1, the
// rows of the ascending receive list behind them; *bad = 1 when an index is neither (diffuse.hip)
// the main stream waits for an exchange still in flight on the halo stream (c_api.hip); no-op when none is
int halo_settle(cna_ctx* c);
// flags[row] = 1 when the row's graph block references a column outside [row0, row0 + n_local) (diffuse.hip)
int launch_rows_need_halo(cna_ctx* c, unsigned char* flags_dev);
int launch_remap_indices(cna_ctx* c, const int64_t* recv_rows_dev, int64_t nr, int32_t* out, int* bad);
int launch_pack_rows(cna_ctx* c, const double* src, const int64_t* idx, int64_t nrows, int ld, double* dst, hipStream_t st = nullptr);
int launch_unpack_rows(cna_ctx* c, const double* src, const int64_t* idx, int64_t nrows, int ld, double* dst, hipStream_t st = nullptr);
int comm_allreduce_f64_sum(cna_ctx* c, double* buf, size_t count);
int comm_allreduce_f64_max(cna_ctx* c, double* buf, size_t count);
int comm_allreduce_i64_sum(cna_ctx* c, int64_t* buf, size_t count);
int comm_allgather_bytes(cna_ctx* c, const void* send, void* recv, size_t bytes_per_rank);

// ---- kernel launchers
// diffuse.hip
int launch_colsum(cna_ctx* c);
int graph_reorder_device(cna_ctx* c, const int64_t* perm_dev);
int launch_add_scalar(cna_ctx* c, double* v, int64_t n, double s);
int launch_nam_step(cna_ctx* c, bool first, bool want_kurt, bool write_t, bool write_nam, bool dense,
                    const int32_t* rows = nullptr, int64_t n_rows = 0,       // rows: this launch's rows of the block (null: all)
                    int64_t base = 0, int64_t count = -1,                    // ... or the contiguous rows [base, base + count) (count < 0: all)
                    hipStream_t st = nullptr, bool timed = true);
int64_t nam_step_turn_rows(const cna_ctx* c, int64_t n_rows);               // rows one turn of all eight XCDs covers in the wide step kernels (0: not that kernel)
int launch_scale_rows(cna_ctx* c, const double* s_local, double* t_global, int m, int ld);
// rows.hip
int launch_batch_kurtosis(cna_ctx* c, const double* mat, int64_t rows, int ncols, int ld,
                          const int32_t* order_dev, const int32_t* boff_dev, int n_batches, double* out);
int launch_zero_variance(cna_ctx* c, const int32_t* colmap_dev, int n_sel, uint8_t* flags_dev,
                         unsigned long long* count_dev);
int launch_select(cna_ctx* c, const int32_t* colmap_dev);
int launch_select_zv(cna_ctx* c, const int32_t* colmap_dev, unsigned long long* count_dev);
int launch_gather_rows(cna_ctx* c, const double* src, int ld, const int64_t* rows_dev, int64_t n_out,
                       const int32_t* cols_dev, int n_cols, double* dst, int transposed);
int launch_digit_hist(cna_ctx* c, const double* v, int64_t n, unsigned long long prefix, int shift,
                      unsigned long long* hist_dev);
int launch_max_fold(cna_ctx* c, const unsigned long long* blockmax, int nblocks, unsigned long long* out);
int launch_pair_pack(cna_ctx* c, const unsigned long long* cnt, const unsigned long long* maxbits, unsigned long long* slot);
int launch_pair_fold(cna_ctx* c, const unsigned long long* slots, int nranks, unsigned long long* cnt, unsigned long long* maxbits);
// exact median of v[0..n) (np.median semantics) on the device, then the walk's stop rule for step `step` (0-based):
// nothing returns to the host; `state` is the AutoState block of the walk, `hist` 2 x 257 words of scratch
int launch_auto_median(cna_ctx* c, const double* v, int64_t n, void* state, unsigned long long* hist, int step, int min_steps,
                       bool sum_over_ranks = false);
int launch_qc_count(cna_ctx* c, const double* v, int64_t n, void* state);   // threshold max(6, 2 median) and the entries not below it
int auto_state_result_offset();                                             // {median, threshold, count} in the state block
size_t auto_state_bytes();
int auto_state_stopped_offset();
int launch_select_std(cna_ctx* c, const int32_t* colmap_dev, unsigned long long* nzero_dev, const double* y_dev,
                      unsigned long long* maxbits_dev, const double* W_dev, const double* Ct_dev, int rk,
                      unsigned char* xq = nullptr, void* xscale = nullptr, int Kp = 0);
int launch_standardize(cna_ctx* c, int center);
int launch_ncorrs(cna_ctx* c, const double* y_dev, unsigned long long* maxbits_dev);
int launch_resid_lowrank(cna_ctx* c, const double* W_dev, const double* Ct_dev, int r, int center, int standardize,
                         const double* y_dev, unsigned long long* maxbits_dev, const int32_t* bk_order = nullptr,
                         const int32_t* bk_boff = nullptr, int nb = 0, double* bk_out = nullptr);
int launch_obs_counts(cna_ctx* c, const double* edges_dev, const double* thr_dev, int T, double thr0,
                      double inv_step, unsigned long long* hist_dev /* 2*T */);
int launch_suffix_sum(cna_ctx* c, const unsigned long long* hist, int P, int T, int64_t* tails);
int launch_tail_sums(cna_ctx* c, const int64_t* tails, int P, int T, int64_t* sums);
int launch_fdr_table(cna_ctx* c, const int64_t* sums, const int64_t* ranks, int T, int P, double* fdr, double* runmin);
// rows16.hip: sixteen rows per wave, projector on the matrix cores; 1 = queued, 0 = not eligible
int launch_rowpass16(cna_ctx* c, const double* src, int lds, double* dst, int ldd, int64_t nrows, int N, const double* W_dev,
                     const double* Ct_dev, int r, int center, int standardize, int write_out, const double* y_dev,
                     unsigned long long* maxword, const int32_t* bk_order, const int32_t* bk_boff, int nb, double* bk_out,
                     unsigned long long* qc_counters = nullptr);   // non-null: src is the NAM; [0] += rows with NaN batch kurtosis, [1] += constant rows
int launch_percell_bins(cna_ctx* c, hipStream_t st, const double* coef_local, const double* thr_dev, int T, double thr0,
                        double inv_step, unsigned short* bins);
extern "C" int cna_host_expand_u16(double* dst, const uint16_t* bins, int64_t n, const double* runmin, int T, int nthreads);
int launch_unpermute2(cna_ctx* c, const double* a, const double* b, const int64_t* idx, int64_t n, double* oa,
                      double* ob);
int launch_percell_fdr(cna_ctx* c, const double* thr_dev, const double* runmin_dev, int T, double thr0,
                       double inv_step, double* coef_local, double* fdr_local);
int launch_transpose(cna_ctx* c, const double* in, int64_t rows, int cols, int ld, double* out);
// mfma.hip
int launch_xb(cna_ctx* c, const double* B_dev, int ldb, int n_out, bool center, double* out, int ld_out);
int launch_gram(cna_ctx* c, double* G_dev);
int launch_copy_f64(cna_ctx* c, const double* src, double* dst, int64_t n);   // rows.hip: dst may be mapped host memory
// the same product range by range (rows [row0, row1), boundaries multiples of *unit_rows) on stream st, partial tiles
// carried in c->gram_part; bit-identical to launch_gram
int gram_pre_begin(cna_ctx* c, int64_t* unit_rows);
int gram_pre_range(cna_ctx* c, int64_t row0, int64_t row1, hipStream_t st);
int gram_pre_finish(cna_ctx* c, double* G_dev, hipStream_t st);
// selection (all cells, samples in place, no projector) + standardisation + coefficients + digit planes + Gram in one
// kernel (k_selgram_blk); gram_fused_ok: the shapes it covers (c->nx, c->ld set)
bool gram_fused_ok(const cna_ctx* c, int Nx, int ldx, int Kp);
int launch_selgram(cna_ctx* c, double* G_dev, unsigned long long* nzero, const double* y, unsigned long long* maxbits,
                   unsigned char* xq, void* xscale, int Kp);
int launch_null_local(cna_ctx* c, const double* Yc_dev, int ldy, int P, const double* cuts_dev, int T,
                      double cut0, double inv_step, double eps, unsigned long long* hist_dev, const int* guard = nullptr);
// null_i8.hip
bool null_i8_eligible(const cna_ctx* c, int P, int T, double cut0, double inv_step, double eps);
bool null_i8_enabled();
int ensure_xq(cna_ctx* c, int KS);
int launch_null_local_i8(cna_ctx* c, const double* Yc_dev, int ldy, int P, const double* cuts_dev, int T, double cut0,
                         double inv_step, double eps, int64_t** sums_out, int** status_out);

// stats.hip
int launch_condition(cna_ctx* c, hipStream_t st, const double* M_dev, const double* Y_dev, int N, int P,
                     double* Zc_dev, int ldy);
int64_t global_test_scratch_doubles(int P, int kmax, int K);
int launch_global_test(cna_ctx* c, hipStream_t st, const double* Zc_dev, int ldy, int N, int P, const double* U_dev,
                       int kmax, const int32_t* ks_dev, int K, int r, double* work, double* minp_dev, double* r2_dev,
                       int32_t* kidx_dev);

// ---- device helpers shared by the kernel files
#ifdef __HIPCC__
// Sum over the 64 lanes of a wave, result in every lane.  Four DPP steps (xor 1, xor 2, mirror
// within 8, mirror within 16) give every lane its 16-lane row total without touching LDS
// (ds_bpermute-based __shfl_xor costs two LDS round trips per step for a double); the four
// row totals are then combined through scalar reads.  Fixed order -> deterministic.
__device__ __forceinline__ double dpp_add(double v, const int ctrl_sel) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  int lo2, hi2;
  switch (ctrl_sel) {
    case 0: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
    case 1: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
    case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break; // row_half_mirror
    default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break; // row_mirror
  }
  return v + __hiloint2double(hi2, lo2);
}
// the partner value of a DPP step (see dpp_add) without the addition
__device__ __forceinline__ double dpp_partner(double v, const int ctrl_sel) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  int lo2, hi2;
  switch (ctrl_sel) {
    case 0: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;
    case 1: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;
    case 2: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break;
    default: lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break;
  }
  return __hiloint2double(hi2, lo2);
}
// max over the 64 lanes of a wave (non-negative inputs; NaN-free), in every lane: DPP steps inside the rows of 16,
// scalar reads across them -- no LDS traffic (a __shfl_xor of a double is two ds_bpermute round trips per step)
__device__ __forceinline__ double wave_max_d(double v) {
  v = fmax(v, dpp_partner(v, 0));
  v = fmax(v, dpp_partner(v, 1));
  v = fmax(v, dpp_partner(v, 2));
  v = fmax(v, dpp_partner(v, 3));
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return fmax(fmax(r0, r1), fmax(r2, r3));
}
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add(v, 0);
  v = dpp_add(v, 1);
  v = dpp_add(v, 2);
  v = dpp_add(v, 3);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
  const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
  const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (r0 + r1) + (r2 + r3);
}
#endif
