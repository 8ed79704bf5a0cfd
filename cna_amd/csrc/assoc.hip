// The fixed-shape analysis in two library calls (include/cna_hip.h: cna_assoc_begin / cna_assoc_finish).
//
// The reference's association() (_association.py:193-242) is one straight-line function.  For the call shape that needs
// no decision of the host between its stages -- a fixed number of walk steps, one batch, covariates allowed, a seed, the
// local test on -- this file issues the stages back to back: every stage is the public entry point the Python host would
// call (same kernels, same order, same bits), minus the interpreter between them.  Nothing here touches a kernel.
#include "common.h"
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

extern "C" {
int cna_host_draw_join(void);
void cna_host_draw_times(double* out2);
int cna_host_draw_then_condition(cna_ctx* ctx, const double* M, const double* table, int N, int cols, int* flag);
int cna_host_copy(void* dst, const void* src, int64_t nbytes, int nthreads);
uint64_t cna_host_hash64(const void* p, int64_t nbytes, int nthreads);

int cna_assoc_begin(cna_ctx* c, int nsteps, const double* y_hint, int n_hint) {
  if (!c) CNA_FAIL(CNA_EINVAL, "null context");
  if (nsteps < 0) CNA_FAIL(CNA_EINVAL, "cna_assoc_begin: nsteps < 0");
  if (nsteps == 0) return 0;                       // the NAM of the resident walk stays (the caller's NAM cache)
  // the hint is consumed by the step that ends a walk of two or more steps (c_api.hip:cna_nam_step); never leave one behind
  if (y_hint && nsteps >= 2) CNA_TRY(cna_nam_select_hint(c, y_hint, n_hint));
  return cna_nam_steps(c, nsteps);
}

// Steps first .. first + count - 1 (0-based) of a walk of `total` steps: the caller queues the steps that need nothing but
// graph and sample codes before it has looked at the phenotype, and the last one -- with the hint -- once it has.
int cna_assoc_begin_part(cna_ctx* c, int first, int count, int total, const double* y_hint, int n_hint) {
  if (!c) CNA_FAIL(CNA_EINVAL, "null context");
  if (first < 0 || count < 0 || total < 1 || first + count > total) CNA_FAIL(CNA_EINVAL, "cna_assoc_begin_part: bad step range");
  for (int i = first; i < first + count; ++i) {
    const bool last = i + 1 == total;
    if (last && y_hint && total >= 2) CNA_TRY(cna_nam_select_hint(c, y_hint, n_hint));
    CNA_TRY(cna_nam_step(c, 0, last ? 0 : 1, last ? 1 : 0));
  }
  return 0;
}

namespace {
// every way out of cna_assoc_finish passes here: the draw thread writes the flag on this call's stack and reads the
// caller's table / M until its follow-up is done
struct DrawJoin {
  bool pending;
  int rc = 0;
  explicit DrawJoin(bool p) : pending(p) {}
  int join() {
    if (pending) {
      pending = false;
      rc = cna_host_draw_join();
      if (rc == -2) rc = 0;                        // (already collected by the caller: nothing to wait for)
    }
    return rc;
  }
  ~DrawJoin() { join(); }
};
}  // namespace

int cna_assoc_finish(cna_ctx* c, const cna_assoc_args* a, cna_assoc_out* o) {
#pragma clang fp contract(off)
  if (!c) CNA_FAIL(CNA_EINVAL, "null context");
  if (!a || !o) CNA_FAIL(CNA_EINVAL, "cna_assoc_finish: args and out are required");
  const auto t_entry = std::chrono::steady_clock::now();
  std::memset(o, 0, sizeof(*o));
  auto mark = [&](int i) { o->t_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count(); };
  const int N = a->n_sel;
  if (N < 2 || N > 1024 || !a->y || !a->M || !a->ks || a->K < 1 || a->Nnull < 1 || !a->table || !a->G || !a->U || !a->minp ||
      !a->r2 || !a->kidx || a->r < 0 || (a->r > 0 && (!a->resid_C || !a->resid_W)) || a->n_verify < 0 || a->n_verify > 4)
    CNA_FAIL(CNA_EINVAL, "cna_assoc_finish: bad arguments");
  if (!a->colmap && N != c->N) CNA_FAIL(CNA_EINVAL, "cna_assoc_finish: without a column map every sample is analysed");
  int kmax = 0;
  for (int i = 0; i < a->K; ++i) kmax = a->ks[i] > kmax ? a->ks[i] : kmax;
  const int P1 = a->Nnull + 1;
  const int Ploc = a->Nnull < 1000 ? a->Nnull : 1000;       // P' = min(1000, Nnull), _association.py:95

  // The inputs the device copy was made from, hashed while the device works; joined before anything is written where the
  // caller can see it.  (Declared before the draw's guard: joined after it on every way out.)
  struct Verifier {
    std::thread th;
    std::atomic<int> stale{0};
    void join() { if (th.joinable()) th.join(); }
    ~Verifier() { join(); }
  } verifier;
  if (a->n_verify > 0) {
    const int vt = a->verify_threads > 0 ? a->verify_threads : 1;
    verifier.th = std::thread([a, vt, &verifier]() {
      for (int i = 0; i < a->n_verify; ++i)
        if (cna_host_hash64(a->verify_ptr[i], a->verify_bytes[i], vt) != a->verify_hash[i]) { verifier.stale.store(1); return; }
    });
  }

  DrawJoin draw(a->draw_pending != 0);
  int flag = 0;                                             // 1: the draw thread has conditioned the phenotypes
  bool then_posted = false;
  bool conditioned = a->conditioned != 0;
  if (!conditioned) {
    if (a->draw_pending) then_posted = cna_host_draw_then_condition(c, a->M, a->table, N, P1, &flag) == 0;
    if (!then_posted) {
      if (draw.join() != 0) CNA_FAIL(CNA_ENOMEM, "cna_assoc_finish: the permutation draw failed");
      CNA_TRY(cna_condition_phenotypes(c, a->M, a->table, N, P1));
      conditioned = true;
    }
  }
  mark(0);

  // compute_nam_and_reindex + _resid_nam without batches: one pass (M = I - C.W applied inside it)
  CNA_TRY(cna_set_resid_factors(c, a->r > 0 ? a->resid_C : nullptr, a->r > 0 ? a->resid_W : nullptr, a->r, N));
  CNA_TRY(cna_null_local_discard(c));                       // (a pass an earlier analysis left behind when it raised)
  int T = 0, gq = 0, cq = 0, nl = 0;
  int64_t nz = 0;
  double m = 0.0;
  CNA_TRY(cna_select_standardized_fused(c, nullptr, 0, a->colmap, N, &nz, a->y, &m, Ploc, &T, o->thr, &gq, &cq, 1,
                                        then_posted ? &flag : nullptr, &nl));
  mark(1);
  o->n_zero = nz;
  o->max_abs = m;
  o->T = T;
  o->null_fused = nl;
  if (nz != 0 || !(m == m) || T < 1) {
    // zero-variance cells to drop, NaN coefficients, thresholds out of range: the general path decides (and raises what
    // the reference raises); whatever the fused call queued beyond the selection is collected and dropped
    o->status = CNA_ASSOC_GENERAL;
    CNA_TRY(cna_null_local_discard(c));
    draw.join();
    mark(8);
    return 0;
  }
  if (!nl) {
    if (then_posted) {
      if (draw.join() != 0) CNA_FAIL(CNA_ENOMEM, "cna_assoc_finish: the permutation draw failed");
      if (__atomic_load_n(&flag, __ATOMIC_ACQUIRE) != 1) CNA_TRY(cna_condition_phenotypes(c, a->M, a->table, N, P1));
    }
    CNA_TRY(cna_null_local_launch(c, 1, Ploc, nullptr, T, 0, nullptr));
  }
  mark(2);

  // Gram matrix -> leading eigenpairs -> F-tests on a thread of their own (second stream, their own buffers: what the
  // general path's small-block schedule does from the interpreter, _association.py: tail_first): the per-cell columns
  // and the local null's results need none of it
  struct Eig {
    std::thread th;
    int rc = 0, acc = 0;
    std::string err;
    void join() { if (th.joinable()) th.join(); }
    ~Eig() { join(); }
  } eig;
  eig.th = std::thread([&]() {
    eig.rc = cna_gram_pcs_tests(c, kmax, a->ks, a->K, a->r, a->use_native_eig, a->resid_tol, a->gap_tol, a->G, a->U, &eig.acc);
    mark(10);
    o->t_ms[9] = (c->t_gram_fetched - std::chrono::duration<double>(t_entry.time_since_epoch()).count()) * 1e3;
    if (eig.rc == 0 && eig.acc) eig.rc = cna_global_test_fetch(c, a->minp, a->r2, a->kidx);
    mark(11);
    if (eig.rc != 0) eig.err = cna_last_error();
  });

  verifier.join();
  mark(3);
  if (verifier.stale.load()) {
    o->status = CNA_ASSOC_STALE;
    eig.join();
    CNA_TRY(cna_null_local_discard(c));
    draw.join();
    mark(8);
    return 0;
  }

  const int64_t n_out = c->local_view ? c->n_local : c->n_global;
  const bool to_dst = a->coef_dst && a->fdr_dst && a->n_dst == n_out && n_out > 0;
  const int threads = a->copy_threads > 0 ? a->copy_threads : 1;
  int rc_main = 0;
  auto run = [&]() -> int {
    if (cq) {                                               // (else: replicated multi-rank outputs, assembled with the FDR column below)
      double* cp = nullptr;
      CNA_TRY(cna_percell_coef_wait(c, &cp));
      o->coef_ptr = cp;
      if (to_dst) {
        if (cna_host_copy(a->coef_dst, cp, 8 * n_out, threads) != 0) std::memcpy(a->coef_dst, cp, 8 * (size_t)n_out);
        o->coef_in_dst = 1;
      }
    }
    mark(4);
    // the FDR column follows the local null on the device (c_api.hip:null_local_go, fdr_inline): put together in the
    // caller's storage the moment the pass is done
    int fdr_done = 0;
    if (to_dst && cq) CNA_TRY(cna_percell_fdr_copy_early(c, a->fdr_dst, n_out, threads, &fdr_done));
    mark(5);
    CNA_TRY(cna_null_local_fetch(c, nullptr, o->tail_sums, o->ranks, o->num_detected));
    mark(6);
    // fdr[t] = mean over permutations of tails / ranks (_stats.py:79-80) from the per-threshold sums; running minimum
    // for the per-cell lookup (_association.py:234-237): numpy's `tail_sums / ranks / Nloc` and np.fmin.accumulate
    double runv = 0.0;
    for (int t = 0; t < T; ++t) {
      const double q = (double)o->tail_sums[t] / (double)o->ranks[t];
      const double f = q / (double)Ploc;
      o->fdr[t] = f;
      runv = t == 0 ? f : std::fmin(runv, f);
      o->runmin[t] = runv;
    }
    if (fdr_done) {
      o->fdr_in_dst = 1;
      o->fdr_ptr = a->fdr_dst;
    } else {
      double *cp = nullptr, *fp = nullptr;
      CNA_TRY(cna_percell_fdr_pinned(c, o->thr, o->runmin, T, &cp, &fp));
      o->coef_ptr = cp;
      o->fdr_ptr = fp;
      if (to_dst && cp && fp) {
        if (!o->coef_in_dst) {
          if (cna_host_copy(a->coef_dst, cp, 8 * n_out, threads) != 0) std::memcpy(a->coef_dst, cp, 8 * (size_t)n_out);
          o->coef_in_dst = 1;
        }
        if (fp != a->fdr_dst && cna_host_copy(a->fdr_dst, fp, 8 * n_out, threads) != 0) std::memcpy(a->fdr_dst, fp, 8 * (size_t)n_out);
        o->fdr_in_dst = 1;
      }
    }
    return 0;
  };
  rc_main = run();
  eig.join();
  mark(7);
  if (rc_main != 0) return rc_main;
  if (eig.rc != 0) {
    cna_set_error(eig.err);
    return eig.rc;
  }
  o->eig_accepted = eig.acc;
  if (!eig.acc) o->status = CNA_ASSOC_NEED_PCS;
  if (draw.join() != 0) CNA_FAIL(CNA_ENOMEM, "cna_assoc_finish: the permutation draw failed");
  mark(8);
  {
    double td[2];
    cna_host_draw_times(td);
    const double t0 = std::chrono::duration<double>(t_entry.time_since_epoch()).count();
    o->t_ms[12] = (td[0] - t0) * 1e3;
    o->t_ms[13] = (td[1] - t0) * 1e3;
  }
  return 0;
}

int cna_assoc_run(cna_ctx* c, int nsteps, const double* y_hint, int n_hint, const cna_assoc_args* args, cna_assoc_out* out) {
  CNA_TRY(cna_assoc_begin(c, nsteps, y_hint, n_hint));
  return cna_assoc_finish(c, args, out);
}
}  // extern "C"
