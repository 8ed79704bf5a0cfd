/* host_eig.c -- the leading eigenpairs of the samples x samples Gram matrix, on the host, without LAPACK.
 *
 * Reference: svd_nam takes `np.linalg.svd(NAM.dot(NAM.T))` (/root/reference/src/cna/tools/_nam.py:105) and the global
 * test consumes the first k <= max(ks) left singular vectors only, and only through squared projections
 * (_association.py:35-48).  LAPACK's dsyevr for the 16 leading pairs of a 200 x 200 matrix costs 1.1-1.25 ms on one
 * host thread -- a third of a rank's step when eight GPUs share the 2M-cell problem -- of which the
 * tridiagonalisation is a third, the 16 eigenvectors of the tridiagonal matrix (MRRR) another, and the rest
 * back-transformation and bookkeeping.  This file does the same job directly:
 *
 *   1. Householder tridiagonalisation of the lower triangle, rows contiguous; the symmetric matrix-vector product of a
 *      step is one pass over the rows with a fused dot + axpy, the rank-2 update another, and that second pass
 *      collects the next column so that no strided read is left;
 *   2. the k + 1 largest eigenvalues by bisection on Sturm counts, all of them at once (the k + 1 independent
 *      recurrences fill the SIMD lanes and hide the division latency);
 *   3. their eigenvectors by inverse iteration with partial pivoting, re-orthogonalised inside clusters (the rule of
 *      LAPACK's dstein: eigenvalues closer than 1e-3 ||T||_1), then the reflectors applied in reverse;
 *   4. CHECKED: residual max_t ||G u_t - lambda_t u_t||_inf and orthogonality max |U^T U - I| are returned, and the
 *      caller (tools/_nam.py:_top_pcs) falls back to dsyevr when they are not at rounding level, when a gap of the
 *      leading spectrum is too small for individual vectors to be defined, or when k > n / 4.
 *
 * Signs of the vectors are arbitrary (nothing downstream of this routine depends on them; the fields that show PC
 * signs come from LAPACK's SVD, `GramPCs`).  No threads, no shared state: re-entrant (the work space is per thread). */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifdef CNA_NO_CLONES          /* sanitizer builds: an ifunc resolver runs before the sanitizer's runtime is up */
#define CLONES
#elif !defined(CLONES)
#define CLONES __attribute__((target_clones("avx512f", "fma", "default")))
#endif
#define MAXT 260                    /* most eigenpairs asked for at once (k + 1 <= n / 4 + 1 <= 257) */

/* ---- 1. tridiagonalisation: Q^T A Q = T, Q = H_0 H_1 ... H_{n-3}, H_j = I - tau_j v_j v_j^T on rows j+1 .. n-1 ----
 * One pass over the trailing lower triangle per step: the rank-2 update of step j - 1 (A22 -= v w^T + w v^T) is applied
 * to a row at the moment step j's symmetric product p = A22 v_j reads it -- the updated row is stored, its first entry
 * (the next column) kept aside, and its dot / axpy contributions to p taken from registers.  Four rows at a time over
 * the columns they share: a row of the triangle is ~n/4 entries long on average, so per-row loop overhead, not
 * arithmetic, is what a row-at-a-time version spends its time on (2.3 -> 1.1 ms at n = 200 for the blocking alone). */
CLONES
static void tridiagonalise(double* A, int n, double* d, double* e, double* V, double* tau, double* work) {
  double* x = work;               /* column j below the diagonal, every earlier update applied */
  double* cap = work + n;         /* first column of the trailing block as the pass leaves it (update j not yet in) */
  double* p = work + 2 * n;
  double* w = work + 3 * n;
  double* vp = work + 4 * n;      /* pending update: v_{j-1}, w_{j-1} without their first entries */
  double* wp = work + 5 * n;
  for (int i = 1; i < n; ++i) x[i - 1] = A[(size_t)i * n];
  for (int i = 0; i < n; ++i) vp[i] = wp[i] = 0.0;
  d[0] = A[0];
  for (int j = 0; j + 1 < n; ++j) {
    const int m = n - j - 1;
    double* v = V + (size_t)j * n;
    const double alpha = x[0];
    double s2 = 0.0;
#pragma omp simd reduction(+ : s2)
    for (int i = 1; i < m; ++i) s2 += x[i] * x[i];
    double tj = 0.0;
    v[0] = 1.0;
    if (s2 == 0.0) {                                   /* nothing to annihilate: H = I */
      e[j] = alpha;
      for (int i = 1; i < m; ++i) v[i] = 0.0;
    } else {
      const double beta = -copysign(sqrt(alpha * alpha + s2), alpha);
      tj = (beta - alpha) / beta;
      const double sc = 1.0 / (alpha - beta);
      e[j] = beta;
#pragma omp simd
      for (int i = 1; i < m; ++i) v[i] = x[i] * sc;
    }
    tau[j] = tj;
    for (int i = 0; i < m; ++i) p[i] = 0.0;
    int i0 = 0;
    for (; i0 + 4 <= m; i0 += 4) {
      double* r0 = A + (size_t)(j + 1 + i0) * n + (j + 1);
      double* r1 = r0 + n;
      double* r2 = r1 + n;
      double* r3 = r2 + n;
      const double a0 = vp[i0], a1 = vp[i0 + 1], a2 = vp[i0 + 2], a3 = vp[i0 + 3];
      const double b0 = wp[i0], b1 = wp[i0 + 1], b2 = wp[i0 + 2], b3 = wp[i0 + 3];
      const double v0 = v[i0], v1 = v[i0 + 1], v2 = v[i0 + 2], v3 = v[i0 + 3];
      double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma omp simd reduction(+ : t0, t1, t2, t3)
      for (int c = 0; c < i0; ++c) {
        const double vc = v[c], pc = vp[c], qc = wp[c];
        const double y0 = r0[c] - (a0 * qc + b0 * pc);
        const double y1 = r1[c] - (a1 * qc + b1 * pc);
        const double y2 = r2[c] - (a2 * qc + b2 * pc);
        const double y3 = r3[c] - (a3 * qc + b3 * pc);
        r0[c] = y0; r1[c] = y1; r2[c] = y2; r3[c] = y3;
        t0 += y0 * vc; t1 += y1 * vc; t2 += y2 * vc; t3 += y3 * vc;
        p[c] += y0 * v0 + y1 * v1 + y2 * v2 + y3 * v3;
      }
      double* rr[4] = {r0, r1, r2, r3};
      double tt[4] = {t0, t1, t2, t3};
      for (int a = 0; a < 4; ++a) {                  /* the 4 x 4 corner on the diagonal */
        const int i = i0 + a;
        const double va = v[i], pa = vp[i], qa = wp[i];
        for (int b = 0; b <= a; ++b) {
          const int c = i0 + b;
          const double y = rr[a][c] - (pa * wp[c] + qa * vp[c]);
          rr[a][c] = y;
          if (b < a) { tt[a] += y * v[c]; p[c] += y * va; }
          else p[i] += tt[a] + y * va;
        }
        cap[i] = rr[a][0];
      }
    }
    for (int i = i0; i < m; ++i) {
      double* row = A + (size_t)(j + 1 + i) * n + (j + 1);
      const double vi = v[i], pa = vp[i], qa = wp[i];
      double t = 0.0;
      for (int c = 0; c < i; ++c) {
        const double y = row[c] - (pa * wp[c] + qa * vp[c]);
        row[c] = y;
        t += y * v[c];
        p[c] += y * vi;
      }
      const double y = row[i] - 2.0 * pa * qa;
      row[i] = y;
      p[i] += t + y * vi;
      cap[i] = row[0];
    }
    double pv = 0.0;
#pragma omp simd reduction(+ : pv)
    for (int i = 0; i < m; ++i) {
      p[i] *= tj;
      pv += p[i] * v[i];
    }
    const double a2 = -0.5 * tj * pv;
#pragma omp simd
    for (int i = 0; i < m; ++i) w[i] = p[i] + a2 * v[i];
    /* what the next step needs of THIS step's update right away: the diagonal entry and the column below it */
    d[j + 1] = cap[0] - 2.0 * v[0] * w[0];
    for (int i = 1; i < m; ++i) x[i - 1] = cap[i] - (v[i] * w[0] + w[i] * v[0]);
    for (int i = 1; i < m; ++i) { vp[i - 1] = v[i]; wp[i - 1] = w[i]; }
  }
}

/* ---- 2. eigenvalues: the nt largest by bisection, all targets in one pass over the matrix per step ---- */
CLONES
static void sturm_counts(const double* d, const double* e2, int n, const double* xs, int* cnt, int nt, double pivmin) {
  double q[MAXT];
  for (int t = 0; t < nt; ++t) {
    q[t] = d[0] - xs[t];
    if (fabs(q[t]) < pivmin) q[t] = -pivmin;
    cnt[t] = q[t] < 0.0;
  }
  for (int i = 1; i < n; ++i) {
    const double di = d[i], ei = e2[i - 1];
#pragma omp simd
    for (int t = 0; t < nt; ++t) {
      double qt = di - xs[t] - ei / q[t];
      qt = fabs(qt) < pivmin ? -pivmin : qt;
      q[t] = qt;
      cnt[t] += qt < 0.0;
    }
  }
}

static int largest_eigenvalues(const double* d, const double* e, int n, int nt, double* lam, double* norm1_out) {
  double* e2 = (double*)malloc(sizeof(double) * (size_t)(n > 1 ? n : 1));
  if (!e2) return -1;
  double gl = d[0], gu = d[0], emax = 0.0, norm1 = 0.0;
  for (int i = 0; i < n; ++i) {
    const double a = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < n ? fabs(e[i]) : 0.0);
    if (d[i] - a < gl) gl = d[i] - a;
    if (d[i] + a > gu) gu = d[i] + a;
    if (fabs(d[i]) + a > norm1) norm1 = fabs(d[i]) + a;
    if (i + 1 < n) { e2[i] = e[i] * e[i]; if (e2[i] > emax) emax = e2[i]; }
  }
  *norm1_out = norm1;
  const double pivmin = DBL_MIN * (emax > 1.0 ? emax : 1.0);
  const double tn = fmax(fabs(gl), fabs(gu));
  gl -= 2.1 * tn * DBL_EPSILON * n + 2.1 * pivmin;
  gu += 2.1 * tn * DBL_EPSILON * n + 2.1 * pivmin;
  double lo[MAXT], hi[MAXT], mid[MAXT];
  int cnt[MAXT];
  for (int t = 0; t < nt; ++t) { lo[t] = gl; hi[t] = gu; }
  for (int it = 0; it < 200; ++it) {
    int open = 0;
    for (int t = 0; t < nt; ++t) {
      mid[t] = 0.5 * (lo[t] + hi[t]);
      open |= hi[t] - lo[t] > 2.0 * DBL_EPSILON * fmax(fabs(lo[t]), fabs(hi[t])) + 2.0 * pivmin && mid[t] > lo[t] && mid[t] < hi[t];
    }
    if (!open) break;
    sturm_counts(d, e2, n, mid, cnt, nt, pivmin);
    for (int t = 0; t < nt; ++t) {                 /* target t: the (n - 1 - t)-th eigenvalue in ascending order */
      if (cnt[t] <= n - 1 - t) lo[t] = mid[t]; else hi[t] = mid[t];
    }
  }
  for (int t = 0; t < nt; ++t) lam[t] = 0.5 * (lo[t] + hi[t]);
  free(e2);
  return 0;
}

/* ---- 3. eigenvectors of T by inverse iteration (LU with partial pivoting of T - x I) ---- */
struct lu3 { double *a, *b, *c2, *l; unsigned char* sw; };

static void lu_factor(const double* d, const double* e, int n, double x, double tiny, struct lu3* f) {
  double* a = f->a; double* b = f->b; double* c2 = f->c2; double* l = f->l;
  for (int i = 0; i < n; ++i) a[i] = d[i] - x;
  for (int i = 0; i + 1 < n; ++i) { b[i] = e[i]; c2[i] = 0.0; }
  for (int i = 0; i + 1 < n; ++i) {
    const double sub = e[i];                       /* entry (i+1, i) */
    if (fabs(a[i]) >= fabs(sub)) {
      if (a[i] == 0.0) a[i] = tiny;
      const double mult = sub / a[i];
      l[i] = mult;
      f->sw[i] = 0;
      a[i + 1] -= mult * b[i];
    } else {                                       /* rows i and i+1 change places */
      const double mult = a[i] / sub;
      l[i] = mult;
      f->sw[i] = 1;
      const double ai1 = a[i + 1], bi = b[i], bi1 = i + 2 < n ? b[i + 1] : 0.0;
      a[i] = sub;
      b[i] = ai1;
      c2[i] = bi1;
      a[i + 1] = bi - mult * ai1;
      if (i + 2 < n) b[i + 1] = -mult * bi1;
    }
  }
  if (fabs(a[n - 1]) < tiny) a[n - 1] = a[n - 1] < 0.0 ? -tiny : tiny;
  for (int i = 0; i + 1 < n; ++i)
    if (fabs(a[i]) < tiny) a[i] = a[i] < 0.0 ? -tiny : tiny;
}

static void lu_solve(const struct lu3* f, int n, double* y) {
  for (int i = 0; i + 1 < n; ++i) {
    if (f->sw[i]) { const double t = y[i]; y[i] = y[i + 1]; y[i + 1] = t; }
    y[i + 1] -= f->l[i] * y[i];
  }
  y[n - 1] /= f->a[n - 1];
  if (n >= 2) y[n - 2] = (y[n - 2] - f->b[n - 2] * y[n - 1]) / f->a[n - 2];
  for (int i = n - 3; i >= 0; --i) y[i] = (y[i] - f->b[i] * y[i + 1] - f->c2[i] * y[i + 2]) / f->a[i];
}

static int tridiagonal_vectors(const double* d, const double* e, int n, int nt, const double* lam, double norm1, double* Y /* nt x n */) {
  double* buf = (double*)malloc(sizeof(double) * (size_t)n * 4 + (size_t)n);
  if (!buf) return -1;
  struct lu3 f = {buf, buf + n, buf + 2 * n, buf + 3 * n, (unsigned char*)(buf + 4 * (size_t)n)};
  const double tiny = fmax(DBL_EPSILON * norm1, DBL_MIN * 1e16);
  const double ortol = 1e-3 * norm1;
  int group = 0;
  double xprev = 0.0;
  int rc = 0;
  for (int t = 0; t < nt && rc == 0; ++t) {
    double x = lam[t];
    if (t > 0) {
      if (fabs(lam[t] - lam[t - 1]) >= ortol) group = t;
      const double pertol = 10.0 * fabs(DBL_EPSILON * x);
      if (xprev - x < pertol) x = xprev - pertol;      /* (descending order) keep coincident shifts apart, as dstein does */
    }
    xprev = x;
    lu_factor(d, e, n, x, tiny, &f);
    double* y = Y + (size_t)t * n;
    /* a start vector with every component present, different for every eigenvalue */
    unsigned long long s = 0x9E3779B97F4A7C15ull * (unsigned long long)(t + 1);
    for (int i = 0; i < n; ++i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      y[i] = 0.5 + (double)(s >> 11) * (1.0 / 9007199254740992.0);
    }
    /* the shift is an eigenvalue to rounding: the first solve already lands in the eigenspace (growth ~ 1 / eps), the
     * next two are dstein's two extra steps */
    for (int it = 0; it < 3; ++it) {
      double nrm = 0.0;
      for (int i = 0; i < n; ++i) nrm += y[i] * y[i];
      nrm = sqrt(nrm);
      if (!(nrm > 0.0) || !isfinite(nrm)) { rc = 1; break; }
      const double scl = 1.0 / nrm;
      for (int i = 0; i < n; ++i) y[i] *= scl;
      lu_solve(&f, n, y);
      for (int g = group; g < t; ++g) {            /* modified Gram-Schmidt against the vectors of the same cluster */
        const double* z = Y + (size_t)g * n;
        double dot = 0.0;
        for (int i = 0; i < n; ++i) dot += y[i] * z[i];
        for (int i = 0; i < n; ++i) y[i] -= dot * z[i];
      }
    }
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) nrm += y[i] * y[i];
    nrm = sqrt(nrm);
    if (!(nrm > 0.0) || !isfinite(nrm)) { rc = 1; break; }
    const double scl = 1.0 / nrm;
    for (int i = 0; i < n; ++i) y[i] *= scl;
  }
  free(buf);
  return rc;
}

/* ---- back-transformation: u = H_0 H_1 ... H_{n-3} y, vectors as rows of Y, four of them per pass over a reflector ---- */
CLONES
static void apply_reflectors(const double* V, const double* tau, int n, int nt, double* Y) {
  for (int j = n - 2; j >= 0; --j) {
    if (tau[j] == 0.0) continue;
    const int m = n - j - 1;
    const double* v = V + (size_t)j * n;
    const double tj = tau[j];
    int t = 0;
    for (; t + 4 <= nt; t += 4) {
      double* y0 = Y + (size_t)t * n + (j + 1);
      double* y1 = y0 + n;
      double* y2 = y1 + n;
      double* y3 = y2 + n;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma omp simd reduction(+ : s0, s1, s2, s3)
      for (int i = 0; i < m; ++i) {
        const double vi = v[i];
        s0 += vi * y0[i]; s1 += vi * y1[i]; s2 += vi * y2[i]; s3 += vi * y3[i];
      }
      s0 *= tj; s1 *= tj; s2 *= tj; s3 *= tj;
#pragma omp simd
      for (int i = 0; i < m; ++i) {
        const double vi = v[i];
        y0[i] -= s0 * vi; y1[i] -= s1 * vi; y2[i] -= s2 * vi; y3[i] -= s3 * vi;
      }
    }
    for (; t < nt; ++t) {
      double* y = Y + (size_t)t * n + (j + 1);
      double s = 0.0;
#pragma omp simd reduction(+ : s)
      for (int i = 0; i < m; ++i) s += v[i] * y[i];
      s *= tj;
#pragma omp simd
      for (int i = 0; i < m; ++i) y[i] -= s * v[i];
    }
  }
}

/* The evidence for the caller, taken where the work is O(n k): residuals and orthogonality of the eigenvectors OF THE
 * TRIDIAGONAL MATRIX (the stage that can fail to converge or lose orthogonality in a cluster).  The reflectors on either
 * side are orthogonal to rounding unconditionally, so the pairs of G inherit both figures. */
static void check_tridiagonal_pairs(const double* d, const double* e, int n, int k, const double* Y, const double* lam,
                                    double* resid_out, double* ortho_out) {
  double rmax = 0.0, omax = 0.0;
  for (int t = 0; t < k; ++t) {
    const double* y = Y + (size_t)t * n;
    for (int i = 0; i < n; ++i) {
      double s = (d[i] - lam[t]) * y[i];
      if (i > 0) s += e[i - 1] * y[i - 1];
      if (i + 1 < n) s += e[i] * y[i + 1];
      const double r = fabs(s);
      if (!(r <= rmax)) rmax = r;                   /* (NaN propagates) */
    }
    for (int g = 0; g <= t; ++g) {
      const double* z = Y + (size_t)g * n;
      double s = 0.0;
      for (int c = 0; c < n; ++c) s += y[c] * z[c];
      const double o = fabs(s - (g == t ? 1.0 : 0.0));
      if (!(o <= omax)) omax = o;
    }
  }
  *resid_out = rmax;
  *ortho_out = omax;
}

CLONES
static void check_pairs(const double* G, int n, int k, const double* Y, const double* lam, double* resid_out, double* ortho_out) {
  double rmax = 0.0, omax = 0.0;
  for (int t = 0; t < k; ++t) {
    const double* u = Y + (size_t)t * n;
    for (int i = 0; i < n; ++i) {
      const double* row = G + (size_t)i * n;
      double s = 0.0;
#pragma omp simd reduction(+ : s)
      for (int c = 0; c < n; ++c) s += row[c] * u[c];
      const double r = fabs(s - lam[t] * u[i]);
      if (!(r <= rmax)) rmax = r;                   /* (NaN propagates) */
    }
    for (int g = 0; g <= t; ++g) {
      const double* z = Y + (size_t)g * n;
      double s = 0.0;
#pragma omp simd reduction(+ : s)
      for (int c = 0; c < n; ++c) s += u[c] * z[c];
      const double o = fabs(s - (g == t ? 1.0 : 0.0));
      if (!(o <= omax)) omax = o;
    }
  }
  *resid_out = rmax;
  *ortho_out = omax;
}

static pthread_key_t g_ws_key;
static int g_ws_key_ok = 0;
static pthread_once_t g_ws_once = PTHREAD_ONCE_INIT;
static void ws_make_key(void) { g_ws_key_ok = pthread_key_create(&g_ws_key, free) == 0; }

/* G: n x n symmetric, row-major (only its lower triangle is read for the decomposition; all of it for the check).
 * U_out: n x k row-major, column t = eigenvector of the t-th LARGEST eigenvalue.  lam_out: k + 1 values, the k leading
 * eigenvalues and the next one (for the caller's gap test; k + 1 <= n).  resid_out: max_t ||T y_t - lam_t y_t||_inf and
 * ortho_out: max |Y^T Y - I| of the tridiagonal stage (see check_tridiagonal_pairs); `cna_host_eig_check` measures the
 * same two figures on G itself.  Returns 0, -1 out of memory, 1 breakdown of the inverse iteration, 2 bad arguments. */
int cna_host_top_eig(const double* G, int n, int k, double* U_out, double* lam_out, double* resid_out, double* ortho_out) {
  if (!G || !U_out || !lam_out || n < 2 || k < 1 || k + 1 > n || k + 1 > MAXT) return 2;
  const size_t nn = (size_t)n * n;
  /* the work space (2 n^2 + ...: 0.7 MB at 200 samples) stays with the thread: a fresh allocation of that size is an
   * mmap whose pages fault in one by one on every call -- as long as the whole decomposition at 100 samples */
  static __thread double* t_buf = NULL;
  static __thread size_t t_cap = 0;
  const size_t need = 2 * nn + (size_t)n * 10 + (size_t)(k + 1) * n;
  if (need > t_cap) {
    free(t_buf);
    t_buf = (double*)malloc(sizeof(double) * need);
    t_cap = t_buf ? need : 0;
    pthread_once(&g_ws_once, ws_make_key);          /* ... and goes when the thread does */
    if (g_ws_key_ok) pthread_setspecific(g_ws_key, t_buf);
  }
  double* A = t_buf;
  if (!A) return -1;
  double* V = A + nn;
  double* d = V + nn;
  double* e = d + n;
  double* tau = e + n;
  double* work = tau + n;                          /* 6 n */
  double* Y = work + 6 * (size_t)n;                /* (k + 1) x n */
  memcpy(A, G, sizeof(double) * nn);
  memset(tau, 0, sizeof(double) * (size_t)n);
  e[n - 1] = 0.0;
  tridiagonalise(A, n, d, e, V, tau, work);
  double norm1 = 0.0;
  int rc = largest_eigenvalues(d, e, n, k + 1, lam_out, &norm1);
  if (rc == 0) rc = tridiagonal_vectors(d, e, n, k, lam_out, norm1, Y);
  if (rc == 0) {
    double r = 0.0, o = 0.0;
    check_tridiagonal_pairs(d, e, n, k, Y, lam_out, &r, &o);
    if (resid_out) *resid_out = r;
    if (ortho_out) *ortho_out = o;
    apply_reflectors(V, tau, n, k, Y);
    for (int t = 0; t < k; ++t)
      for (int i = 0; i < n; ++i) U_out[(size_t)i * k + t] = Y[(size_t)t * n + i];
  }
  return rc;
}

/* max_t ||G u_t - lam_t u_t||_inf and max |U^T U - I| of k pairs against G itself (tests; 2 n^2 k flops). */
int cna_host_eig_check(const double* G, int n, int k, const double* U /* n x k row-major */, const double* lam,
                       double* resid_out, double* ortho_out) {
  if (!G || !U || !lam || n < 1 || k < 1 || !resid_out || !ortho_out) return 2;
  double* Y = (double*)malloc(sizeof(double) * (size_t)n * k);
  if (!Y) return -1;
  for (int t = 0; t < k; ++t)
    for (int i = 0; i < n; ++i) Y[(size_t)t * n + i] = U[(size_t)i * k + t];
  check_pairs(G, n, k, Y, lam, resid_out, ortho_out);
  free(Y);
  return 0;
}
