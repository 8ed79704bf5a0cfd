/* Host-side helper (plain C, no device code): numpy's legacy standard-normal stream, restated so
 * that it vectorises.
 *
 * The permutation null of the reference is  argsort(np.random.randn(m, Nnull), axis=0)  drawn from
 * numpy's global legacy generator (reference _stats.py:10, _association.py:79-83); results are only
 * bit-identical to the reference if exactly that stream is consumed.  numpy produces it one value
 * at a time (MT19937 word -> 53-bit double -> polar Box-Muller with rejection), ~13 ns per normal,
 * which at 50 x 1000 draws is longer than the GPU needs for the whole random walk.  The stream has
 * a fixed shape, though: every candidate pair consumes exactly four 32-bit words whether it is
 * accepted or not, so words, doubles and candidates can be produced a 624-word block at a time with
 * SIMD, accepted candidates compacted in order, and log / sqrt applied to the survivors.
 *
 * Exactness: same MT19937 recurrence and tempering, same double construction
 * (a>>5, b>>6 -> (a*2^26+b)/2^53), same operation order in  x = 2u-1,  r2 = x1*x1 + x2*x2  (this file
 * is compiled with -ffp-contract=off: numpy's baseline build has no FMA),  f = sqrt(-2*log(r2)/r2)
 * with libm's own scalar log (the function numpy calls), outputs f*x2 then f*x1, the cached second
 * value and the generator position handed back.  tests/test_host_rng.py compares with numpy bit for
 * bit, including odd counts, a pending cached value and every starting position in a block. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MT_N 624
#define MT_M 397
#define MATRIX_A 0x9908b0dfu
#define UPPER_MASK 0x80000000u
#define LOWER_MASK 0x7fffffffu

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#ifdef CNA_NO_CLONES          /* sanitizer builds: an ifunc resolver runs before the sanitizer's runtime is up */
#define CLONES
#else
#define CLONES __attribute__((target_clones("avx2", "default")))
#endif
#else
#define CLONES
#endif

/* next block of 624 raw words (numpy: mt19937_gen) */
CLONES static void mt_reload(uint32_t* restrict key) {
  int kk;
  for (kk = 0; kk < MT_N - MT_M; kk++) {
    const uint32_t y = (key[kk] & UPPER_MASK) | (key[kk + 1] & LOWER_MASK);
    key[kk] = key[kk + MT_M] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & MATRIX_A);
  }
  /* the middle part reads words written 227 places earlier: split so that each piece only reads
   * finished words and the compiler may vectorise it */
  for (; kk < MT_N - 1; kk++) {
    const uint32_t y = (key[kk] & UPPER_MASK) | (key[kk + 1] & LOWER_MASK);
    key[kk] = key[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & MATRIX_A);
  }
  {
    const uint32_t y = (key[MT_N - 1] & UPPER_MASK) | (key[0] & LOWER_MASK);
    key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & MATRIX_A);
  }
}

CLONES static void temper(const uint32_t* restrict in, uint32_t* restrict out, int n) {
  for (int i = 0; i < n; i++) {
    uint32_t y = in[i];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    out[i] = y;
  }
}

/* nc candidate pairs from 4*nc tempered words: x1, x2 and r2 = x1*x1 + x2*x2 */
CLONES static void candidates(const uint32_t* restrict w, int nc, double* restrict x1, double* restrict x2,
                              double* restrict r2) {
  for (int c = 0; c < nc; c++) {
    const double u1 = ((double)(int32_t)(w[4 * c] >> 5) * 67108864.0 + (double)(int32_t)(w[4 * c + 1] >> 6)) /
                      9007199254740992.0;
    const double u2 = ((double)(int32_t)(w[4 * c + 2] >> 5) * 67108864.0 + (double)(int32_t)(w[4 * c + 3] >> 6)) /
                      9007199254740992.0;
    const double a = 2.0 * u1 - 1.0;
    const double b = 2.0 * u2 - 1.0;
    x1[c] = a;
    x2[c] = b;
    r2[c] = a * a + b * b;
  }
}

#include <pthread.h>
#include <sched.h>
#include <time.h>
static int g_host_threads = 1;
/* the library's draw thread runs with the count its request carries: an override of ITS OWN, so that it never races
 * with cna_host_set_threads or with a draw on the caller's thread */
static __thread int t_host_threads = 0;
static inline int host_threads(void) { return t_host_threads > 0 ? t_host_threads : g_host_threads; }

struct norm_job { const double *x1, *x2, *r2; double* out; int64_t n_out, a, b; };
static void* norm_worker(void* arg) {
  struct norm_job* j = (struct norm_job*)arg;
  for (int64_t p = j->a; p < j->b; ++p) {
    const double f = sqrt(-2.0 * log(j->r2[p]) / j->r2[p]);
    j->out[2 * p] = f * j->x2[p];
    if (2 * p + 1 < j->n_out) j->out[2 * p + 1] = f * j->x1[p];
  }
  return NULL;
}

/* ---- the same stream on several threads (round 6) ----------------------------------------------------------------
 * Every candidate consumes exactly four words, accepted or not, and the generator's blocks follow from the key alone
 * (a block costs 0.04 us to produce): thread t re-derives the key of its first block and then produces, tests and
 * compacts the candidates of ITS blocks into a buffer of its own; the accepted counts per block are summed in order,
 * which tells every thread where its pairs go; each thread then turns its own pairs into normals.  Nothing a thread
 * computed is read by another one (handing the accepted pairs of ONE producer to helper threads moved 2.4 MB of
 * dirty cache lines between cores and was slower than no helpers at all: 0.97 against 0.69 ms for 100 000 pairs,
 * tools/micro/bench_draw.c).  Requires the remaining words of the current block to be a multiple of four (always
 * true behind np.random.seed), so that no candidate straddles two blocks.  The number of blocks is estimated from the
 * acceptance rate pi/4 with a wide margin; should it fall short the sequential code below does the draw.
 * 1: done (key, pos, cached value updated), 0: not applicable / fell short (nothing changed). */
struct rp_shared {
  const uint32_t* key0; int pos0, c0, nt;
  int64_t nblocks, pairs, n_out;
  double* out;
  int32_t* cnt; int64_t* offs;
  int arrived, go;                                   /* atomics */
};
struct rp_job { struct rp_shared* sh; int tid; int64_t b0, b1; double* buf; int64_t cap, nacc; int bad; };
static void rp_block(const struct rp_shared* sh, uint32_t* st, int64_t* reloads, int64_t b, uint32_t* words, int* nc) {
  if (b == 0) {
    temper(sh->key0 + sh->pos0, words, 4 * sh->c0);
    *nc = sh->c0;
  } else {
    while (*reloads < b) { mt_reload(st); ++*reloads; }
    temper(st, words, MT_N);
    *nc = MT_N / 4;
  }
}
/* waiting for the other threads of a draw: they are busy for tens of microseconds -- yield; should one of them have been
 * descheduled for long (a CPU quota), stop burning the quota it is waiting for */
static inline void rp_wait_step(int* spins) {
  if (++*spins < 4096) { sched_yield(); return; }
  struct timespec ts = {0, 20000};
  nanosleep(&ts, NULL);
}
static void* rp_worker(void* arg) {
  struct rp_job* j = (struct rp_job*)arg;
  struct rp_shared* sh = j->sh;
  uint32_t st[MT_N], words[MT_N];
  double x1[MT_N / 4], x2[MT_N / 4], r2[MT_N / 4];
  memcpy(st, sh->key0, sizeof(st));
  int64_t reloads = 0, found = 0;
  double *ax1 = j->buf, *ax2 = j->buf + j->cap, *ar2 = j->buf + 2 * j->cap;
  if (j->buf) {
    for (int64_t b = j->b0; b < j->b1; ++b) {
      int nc;
      rp_block(sh, st, &reloads, b, words, &nc);
      candidates(words, nc, x1, x2, r2);
      const int64_t before = found;
      for (int u = 0; u < nc; ++u) {
        const double r = r2[u];
        ax1[found] = x1[u];
        ax2[found] = x2[u];
        ar2[found] = r;
        found += (r < 1.0) & (r != 0.0);
      }
      sh->cnt[b] = (int32_t)(found - before);
    }
  }
  j->nacc = found;
  /* all counts in -> thread 0 sums them -> everybody places its pairs */
  __atomic_fetch_add(&sh->arrived, 1, __ATOMIC_ACQ_REL);
  if (j->tid == 0) {
    for (int spins = 0; __atomic_load_n(&sh->arrived, __ATOMIC_ACQUIRE) < sh->nt;) rp_wait_step(&spins);
    int64_t tot = 0;
    for (int64_t b = 0; b < sh->nblocks; ++b) { sh->offs[b] = tot; tot += sh->cnt[b]; }
    sh->offs[sh->nblocks] = tot;
    __atomic_store_n(&sh->go, 1, __ATOMIC_RELEASE);
  } else {
    for (int spins = 0; !__atomic_load_n(&sh->go, __ATOMIC_ACQUIRE);) rp_wait_step(&spins);
  }
  if (!j->buf || sh->offs[sh->nblocks] < sh->pairs) return NULL;          /* fell short (or no memory): the caller redoes it */
  const int64_t g0 = sh->offs[j->b0];
  int64_t cnt = found;
  if (g0 + cnt > sh->pairs) cnt = sh->pairs > g0 ? sh->pairs - g0 : 0;
  struct norm_job nj = {ax1, ax2, ar2, sh->out + 2 * g0, sh->n_out - 2 * g0, 0, cnt};
  norm_worker(&nj);
  return NULL;
}
static int randn_parallel(uint32_t* key, int* pos, int* has_gauss, double* gauss, int64_t n_out, double* out, int64_t pairs, int nt) {
  const int c0 = (MT_N - *pos) / 4;
  if ((MT_N - *pos) % 4 != 0 || nt < 2) return 0;
  if (nt > 16) nt = 16;
  const double need = (double)pairs / 0.78539816339744831 + 8.0 * sqrt((double)pairs) + 2.0 * (MT_N / 4);
  int64_t nblocks = 1 + (int64_t)((need - c0) / (MT_N / 4)) + 1;
  if (nblocks < 2) nblocks = 2;
  if (nblocks < nt) nt = (int)nblocks;
  struct rp_shared sh = {key, *pos, c0, nt, nblocks, pairs, n_out, out, NULL, NULL, 0, 0};
  sh.cnt = (int32_t*)calloc((size_t)nblocks, sizeof(int32_t));
  sh.offs = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nblocks + 1));
  struct rp_job jobs[16];
  pthread_t th[16];
  int started[16];
  int ok = sh.cnt && sh.offs;
  for (int t = 0; t < nt; ++t) {
    jobs[t].sh = &sh; jobs[t].tid = t; jobs[t].bad = 0; jobs[t].nacc = 0;
    jobs[t].b0 = nblocks * t / nt; jobs[t].b1 = nblocks * (t + 1) / nt;
    jobs[t].cap = (jobs[t].b1 - jobs[t].b0) * (MT_N / 4) + 1;
    jobs[t].buf = ok ? (double*)malloc(sizeof(double) * 3 * (size_t)jobs[t].cap) : NULL;
    if (!jobs[t].buf) ok = 0;
    started[t] = 0;
  }
  if (!ok) {
    for (int t = 0; t < nt; ++t) free(jobs[t].buf);
    free(sh.cnt); free(sh.offs);
    return 0;
  }
  /* a thread that cannot be started leaves the draw to the sequential code: the others would wait for it */
  int nstarted = 1;
  for (int t = 1; t < nt; ++t) { started[t] = pthread_create(&th[t], NULL, rp_worker, &jobs[t]) == 0; nstarted += started[t]; }
  if (nstarted < nt) {
    for (int t = 1; t < nt; ++t) if (!started[t]) { jobs[t].nacc = 0; __atomic_fetch_add(&sh.arrived, 1, __ATOMIC_ACQ_REL); }
  }
  rp_worker(&jobs[0]);
  for (int t = 1; t < nt; ++t) if (started[t]) pthread_join(th[t], NULL);
  int done = nstarted == nt && sh.offs[nblocks] >= pairs;
  if (done) {
    /* numpy stops right behind the candidate that completed the last pair: find it, and the key of its block */
    int64_t bs = 0;
    while (sh.offs[bs + 1] < pairs) ++bs;
    uint32_t st[MT_N], words[MT_N];
    double x1[MT_N / 4], x2[MT_N / 4], r2[MT_N / 4];
    memcpy(st, key, sizeof(st));
    int64_t reloads = 0;
    int nc;
    rp_block(&sh, st, &reloads, bs, words, &nc);
    candidates(words, nc, x1, x2, r2);
    int64_t want = pairs - sh.offs[bs];
    int u = 0;
    for (; u < nc; ++u) {
      if (r2[u] < 1.0 && r2[u] != 0.0 && --want == 0) break;
    }
    if (bs > 0) memcpy(key, st, sizeof(st));
    *pos = (bs == 0 ? *pos : 0) + 4 * (u + 1);
    if (2 * pairs > n_out) {                           /* odd count: the last second value stays cached */
      const double f = sqrt(-2.0 * log(r2[u]) / r2[u]);
      *has_gauss = 1;
      *gauss = f * x1[u];
    }
  }
  for (int t = 0; t < nt; ++t) free(jobs[t].buf);
  free(sh.cnt); free(sh.offs);
  return done;
}

/* n standard normals of numpy's legacy generator (RandomState.randn / standard_normal) into out.
 * key[624], *pos (0..624), *has_gauss, *gauss: the generator state as np.random.get_state() reports
 * it, updated in place to what numpy's own state would be after the same draws.  Returns 0, or -1
 * on a bad state / allocation failure (state untouched). */
int cna_host_legacy_randn(uint32_t* key, int* pos, int* has_gauss, double* gauss, int64_t n, double* out) {
  if (!key || !pos || !has_gauss || !gauss || n < 0 || (n > 0 && !out)) return -1;
  if (*pos < 0 || *pos > MT_N) return -1;
  int64_t done = 0;
  if (n > 0 && *has_gauss) {
    out[done++] = *gauss;
    *has_gauss = 0;
    *gauss = 0.0;
  }
  const int64_t pairs = (n - done + 1) / 2;          /* accepted candidate pairs still to find */
  if (pairs == 0) return 0;
  if (host_threads() > 1 && pairs >= (1 << 13) &&
      randn_parallel(key, pos, has_gauss, gauss, n - done, out + done, pairs, host_threads()) == 1)
    return 0;
  double* acc = (double*)malloc(sizeof(double) * 3 * (size_t)pairs);
  if (!acc) return -1;
  double *ax1 = acc, *ax2 = acc + pairs, *ar2 = acc + 2 * pairs;

  uint32_t words[MT_N + 4];                          /* tempered, not yet consumed: carry (< 4) + one block */
  double x1[MT_N / 4 + 1], x2[MT_N / 4 + 1], r2[MT_N / 4 + 1];
  int have = MT_N - *pos;                            /* unconsumed words of the current block */
  temper(key + *pos, words, have);
  int p = *pos;                                      /* position in the current block of words[0] */
  int64_t found = 0;
  while (1) {
    const int nc = have / 4;
    candidates(words, nc, x1, x2, r2);
    int used = 0;                                    /* candidates consumed */
    /* accepted candidates are compacted without a branch (one in five is rejected: 0.28 of the 0.44 ms this loop took
     * for 100 000 pairs went into mispredictions, tools/micro/bench_draw.c): every candidate is stored at the cursor,
     * which advances only behind an accepted one */
    for (; used < nc && found < pairs; used++) {
      const double r = r2[used];
      ax1[found] = x1[used];
      ax2[found] = x2[used];
      ar2[found] = r;
      found += (r < 1.0) & (r != 0.0);
    }
    if (found == pairs) {                            /* numpy stops right after this candidate */
      p += 4 * used;
      /* p may exceed 624 when the candidate straddled into the block generated last: key already is
       * that block and p - 624 its position */
      *pos = p > MT_N ? p - MT_N : p;
      break;
    }
    /* all complete candidates used up: keep the (< 4) leftover words, draw the next block */
    const int left = have - 4 * nc;
    memmove(words, words + 4 * nc, sizeof(uint32_t) * (size_t)left);
    p += 4 * nc;                                     /* == 624 - left */
    mt_reload(key);
    temper(key, words + left, MT_N);
    have = left + MT_N;
    p -= MT_N;                                       /* words[0] sits at position p (<= 0) relative to the new block */
  }
  /* survivors -> normals, in order: the second value of a pair is numpy's cached one.  Every pair is
   * independent (log and sqrt dominate the whole draw), so large draws split the pairs over threads; the
   * values and their positions do not depend on the thread count. */
  {
    struct norm_job jobs[64];
    pthread_t th[64];
    int started[64];
    int nt = host_threads();
    if (nt > 64) nt = 64;
    if (nt < 1 || pairs < (1 << 18)) nt = 1;         /* (helpers read what ONE producer wrote: slower than none below ~0.25M pairs) */                  /* (25 000 pairs = 50 samples x 1000 permutations split over two to four threads) */
    for (int t = 0; t < nt; ++t) {
      jobs[t].x1 = ax1; jobs[t].x2 = ax2; jobs[t].r2 = ar2; jobs[t].out = out + done; jobs[t].n_out = n - done;
      jobs[t].a = pairs * t / nt; jobs[t].b = pairs * (t + 1) / nt;
      started[t] = 0;
    }
    for (int t = 1; t < nt; ++t) started[t] = pthread_create(&th[t], NULL, norm_worker, &jobs[t]) == 0;
    norm_worker(&jobs[0]);
    for (int t = 1; t < nt; ++t) {
      if (started[t]) pthread_join(th[t], NULL);
      else norm_worker(&jobs[t]);
    }
    if (2 * pairs > n - done) {                        /* odd count: the last second value stays cached */
      const double f = sqrt(-2.0 * log(ar2[pairs - 1]) / ar2[pairs - 1]);
      *has_gauss = 1;
      *gauss = f * ax1[pairs - 1];
    }
  }
  free(acc);
  return 0;
}

/* threads the host helpers of this file may use (large draws only); default 1 */
void cna_host_set_threads(int n) { g_host_threads = n < 1 ? 1 : n; }

/* out[rows[i] * ld_out + c] = y[ argsort(R[:, c])[i] ] for every column c of the m x num matrix R of normal
 * draws (row-major): the permuted phenotypes of conditional_permutation / grouplevel_permutation
 * (_stats.py:11-17,31: Y[m][argsort(randn(len(m), num), axis=0)]) for LARGE draws (10 000 permutations of
 * 200 samples: numpy's argsort along axis 0 takes 24 ms on one thread), columns split over threads, each
 * column merge-sorted as (value, row) pairs.  rows == NULL: identity.  The draws are distinct doubles (a tie
 * has probability ~1e-16 per pair), so every correct sort yields numpy's permutation; ties keep row order. */
struct sort_pair { double v; int32_t i; int32_t pad; };
struct sort_job {
  const double* R; const double* y; double* out; const int64_t* rows;
  int64_t ld_out; int m, num, c0, c1;
  int32_t* idx_out; int64_t ld_idx;                 /* optional: the row of y every output was taken from (cna_host_draw_start_idx) */
};

#define SORT_CB 64                                        /* columns handled together: row segments of 512 bytes */

/* Short columns (m <= NET_MAX rows, the sample axis of a cohort): a sorting network over SORT_CB columns at once.
 * Every entry carries its row in the low NET_BITS bits of its mantissa, so that min / max on the doubles themselves move
 * (value, row) pairs -- two vector instructions per comparator and four columns, no branches, against ~40 ns per
 * element of a comparison sort.  Entries whose remaining bits coincide would be ordered by row instead of by their
 * true low bits: such a column (probability ~1e-8) is reported and redone exactly by the caller. */
#define NET_MAX 512
#define NET_BITS 9
#define NET_SIZES 7
static int g_net_n[NET_SIZES] = {8, 16, 32, 64, 128, 256, 512};
static int* g_net_pairs[NET_SIZES];                      /* comparators (i, j), i < j, of Batcher's odd-even merge sort */
static int g_net_count[NET_SIZES];
static pthread_once_t g_net_once = PTHREAD_ONCE_INIT;

static void net_build(void) {
  for (int k = 0; k < NET_SIZES; ++k) {
    const int n = g_net_n[k];
    int cap = 0;
    for (int pass = 0; pass < 2; ++pass) {
      int cnt = 0;
      for (int p = 1; p < n; p *= 2)
        for (int q = p; q >= 1; q /= 2)
          for (int j = q % p; j <= n - 1 - q; j += 2 * q)
            for (int i = 0; i < (q < n - j - q ? q : n - j - q); ++i)
              if ((i + j) / (2 * p) == (i + j + q) / (2 * p)) {
                if (pass) { g_net_pairs[k][2 * cnt] = i + j; g_net_pairs[k][2 * cnt + 1] = i + j + q; }
                ++cnt;
              }
      if (!pass) { cap = cnt; g_net_pairs[k] = (int*)malloc(sizeof(int) * 2 * (size_t)(cap > 0 ? cap : 1)); if (!g_net_pairs[k]) { g_net_count[k] = -1; break; } }
      else g_net_count[k] = cnt;
    }
  }
}

/* A: n x SORT_CB doubles, row-major (row r of every column side by side); sorted along r, column by column */
CLONES static void net_sort(double* restrict A, const int* restrict pairs, int count) {
  for (int c = 0; c < count; ++c) {
    double* restrict lo = A + (size_t)pairs[2 * c] * SORT_CB;
    double* restrict hi = A + (size_t)pairs[2 * c + 1] * SORT_CB;
    for (int k = 0; k < SORT_CB; ++k) {
      const double a = lo[k], b = hi[k];
      lo[k] = a < b ? a : b;
      hi[k] = a < b ? b : a;
    }
  }
}

/* longer columns: stable merge sort of (value, row) pairs; idx[r] = row of the r-th smallest */
static void sort_column(const double* v, int m, struct sort_pair* a, struct sort_pair* b, int32_t* idx) {
  for (int i = 0; i < m; ++i) { a[i].v = v[i]; a[i].i = i; }
  for (int lo = 0; lo < m; lo += 8) {                /* runs of 8 by insertion */
    const int hi = lo + 8 < m ? lo + 8 : m;
    for (int i = lo + 1; i < hi; ++i) {
      const struct sort_pair x = a[i];
      int k = i - 1;
      while (k >= lo && a[k].v > x.v) { a[k + 1] = a[k]; --k; }
      a[k + 1] = x;
    }
  }
  struct sort_pair *src = a, *dst = b;
  for (int wd = 8; wd < m; wd *= 2) {                /* bottom-up merges (stable) */
    for (int lo = 0; lo < m; lo += 2 * wd) {
      const int mid = lo + wd < m ? lo + wd : m, hi = lo + 2 * wd < m ? lo + 2 * wd : m;
      int i = lo, k = mid, o = lo;
      while (i < mid && k < hi) dst[o++] = src[k].v < src[i].v ? src[k++] : src[i++];
      while (i < mid) dst[o++] = src[i++];
      while (k < hi) dst[o++] = src[k++];
    }
    struct sort_pair* t = src; src = dst; dst = t;
  }
  for (int i = 0; i < m; ++i) idx[i] = src[i].i;
}

typedef uint64_t __attribute__((may_alias)) u64a;
static void* sort_worker(void* arg) {
  struct sort_job* j = (struct sort_job*)arg;
  const int m = j->m;
  int net = -1;
  if (m <= NET_MAX && m >= 2) {
    pthread_once(&g_net_once, net_build);
    for (int k = 0; k < NET_SIZES && net < 0; ++k) if (g_net_n[k] >= m && g_net_count[k] > 0) net = k;
  }
  const int n = net >= 0 ? g_net_n[net] : 0;
  struct sort_pair* a = (struct sort_pair*)malloc(sizeof(struct sort_pair) * (size_t)m * 2);
  double* blk = (double*)malloc(sizeof(double) * ((size_t)m * SORT_CB * 2 + (size_t)n * SORT_CB + m));  /* draws | results | network | one column */
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
  int32_t* resi = j->idx_out ? (int32_t*)malloc(sizeof(int32_t) * (size_t)m * SORT_CB) : NULL;
  if (!a || !blk || !idx || (j->idx_out && !resi)) { free(a); free(blk); free(idx); free(resi); return (void*)1; }
  struct sort_pair* b = a + m;
  double* res = blk + (size_t)m * SORT_CB;
  double* A = res + (size_t)m * SORT_CB;
  double* col = A + (size_t)n * SORT_CB;
  const uint64_t lowmask = ((uint64_t)1 << NET_BITS) - 1;
  for (int cb = j->c0; cb < j->c1; cb += SORT_CB) {
    const int w = j->c1 - cb < SORT_CB ? j->c1 - cb : SORT_CB;
    unsigned char redo[SORT_CB];
    memset(redo, net < 0, sizeof(redo));
    if (net >= 0) {
      u64a* Au = (u64a*)A;
      for (int i = 0; i < n; ++i) {
        u64a* dst = Au + (size_t)i * SORT_CB;
        if (i < m) {
          const u64a* src = (const u64a*)(j->R + (size_t)i * j->num + cb);
          for (int cc = 0; cc < w; ++cc) dst[cc] = (src[cc] & ~lowmask) | (uint64_t)i;
          for (int cc = w; cc < SORT_CB; ++cc) dst[cc] = (uint64_t)i;
        } else {
          for (int cc = 0; cc < SORT_CB; ++cc) A[(size_t)i * SORT_CB + cc] = 1e300;      /* padding rows sort to the end */
        }
      }
      net_sort(A, g_net_pairs[net], g_net_count[net]);
      uint64_t same[SORT_CB];
      for (int cc = 0; cc < SORT_CB; ++cc) same[cc] = 0;
      for (int i = 1; i < m; ++i) {                          /* neighbours the compared bits do not separate */
        const u64a* r1 = Au + (size_t)i * SORT_CB;
        const u64a* r0 = r1 - SORT_CB;
        for (int cc = 0; cc < SORT_CB; ++cc) same[cc] |= (((r1[cc] ^ r0[cc]) & ~lowmask) == 0);
      }
      for (int cc = 0; cc < w; ++cc) redo[cc] = same[cc] != 0 || !(A[(size_t)(m - 1) * SORT_CB + cc] < 1e299);
      for (int i = 0; i < m; ++i) {
        const u64a* srow = Au + (size_t)i * SORT_CB;
        double* rrow = res + (size_t)i * SORT_CB;
        for (int cc = 0; cc < w; ++cc) rrow[cc] = j->y[srow[cc] & lowmask];
        if (resi) for (int cc = 0; cc < w; ++cc) resi[(size_t)i * SORT_CB + cc] = (int32_t)(srow[cc] & lowmask);
      }
    }
    for (int cc = 0; cc < w; ++cc) {
      if (!redo[cc]) continue;
      for (int i = 0; i < m; ++i) col[i] = j->R[(size_t)i * j->num + cb + cc];
      sort_column(col, m, a, b, idx);
      for (int i = 0; i < m; ++i) res[(size_t)i * SORT_CB + cc] = j->y[idx[i]];
      if (resi) for (int i = 0; i < m; ++i) resi[(size_t)i * SORT_CB + cc] = idx[i];
    }
    for (int i = 0; i < m; ++i) {
      const int64_t row = j->rows ? j->rows[i] : i;
      memcpy(j->out + (size_t)row * j->ld_out + cb, res + (size_t)i * SORT_CB, sizeof(double) * (size_t)w);
      if (resi) {                                          /* (rows of the level -> rows of the whole phenotype) */
        int32_t* d = j->idx_out + (size_t)row * j->ld_idx + cb;
        const int32_t* s = resi + (size_t)i * SORT_CB;
        for (int cc = 0; cc < w; ++cc) d[cc] = j->rows ? (int32_t)j->rows[s[cc]] : s[cc];
      }
    }
  }
  free(a); free(blk); free(idx); free(resi);
  return NULL;
}

static int argsort_gather_idx(const double* R, int m, int num, const double* y, double* out, int64_t ld_out,
                              const int64_t* rows, int32_t* idx_out, int64_t ld_idx);
int cna_host_argsort_gather(const double* R, int m, int num, const double* y, double* out, int64_t ld_out,
                            const int64_t* rows) {
  return argsort_gather_idx(R, m, num, y, out, ld_out, rows, NULL, 0);
}
static int argsort_gather_idx(const double* R, int m, int num, const double* y, double* out, int64_t ld_out,
                              const int64_t* rows, int32_t* idx_out, int64_t ld_idx) {
  if (m < 0 || num < 0 || !R || !y || !out) return -1;
  if (m == 0 || num == 0) return 0;
  struct sort_job jobs[64];
  pthread_t th[64];
  int started[64];
  int nt = host_threads();
  if (nt > 64) nt = 64;
  if (nt > num / 128) nt = num / 128;                    /* at least two blocks of 64 columns per thread */
  if ((int64_t)m * num < 32768) nt = 1;                  /* (a thread costs ~30 us to start) */
  if (nt < 1) nt = 1;
  for (int t = 0; t < nt; ++t) {
    jobs[t].R = R; jobs[t].y = y; jobs[t].out = out; jobs[t].rows = rows; jobs[t].ld_out = ld_out;
    jobs[t].m = m; jobs[t].num = num; jobs[t].idx_out = idx_out; jobs[t].ld_idx = ld_idx;
    jobs[t].c0 = (int)((int64_t)num * t / nt); jobs[t].c1 = (int)((int64_t)num * (t + 1) / nt);
    started[t] = 0;
  }
  int bad = 0;
  for (int t = 1; t < nt; ++t) started[t] = pthread_create(&th[t], NULL, sort_worker, &jobs[t]) == 0;
  bad |= sort_worker(&jobs[0]) != NULL;
  for (int t = 1; t < nt; ++t) {
    void* rv = NULL;
    if (started[t]) pthread_join(th[t], &rv);
    else rv = sort_worker(&jobs[t]);
    bad |= rv != NULL;
  }
  return bad ? -1 : 0;
}


/* ---- the permutation draw of conditional_permutation (_stats.py:4-18) off the interpreter ---------------------
 * One persistent worker thread takes a request -- the generator state (numpy's own memory, freshly seeded: no
 * cached normal pending), the standardised phenotype y[m], the levels of the batch vector as member lists
 * (np.unique order), an even number `num` of permutations -- and fills out[r * ld_out + p] = permuted y: per level
 * one randn(len(level), num) block of the legacy stream, then y[members][argsort(block, axis=0)] written to the
 * members' rows: exactly the reference's RNG consumption and values.  The caller goes on with its own work and
 * collects the result with cna_host_draw_wait().  Nothing here needs the interpreter lock: on a small cohort the
 * draw is the longest host item of an analysis, and a Python helper thread got to it 0.35 ms late and took
 * 0.54 ms for 0.39 ms of work (tools/host_trace.py at 200k cells x 50 samples). */
struct draw_req {
  uint32_t* key; int* pos; const double* y; int m, num, nlev; const int64_t* lev_off; const int64_t* members;
  double* out; int64_t ld_out; int threads;
  int32_t* idx_out;                                  /* m x num, or NULL */
};
struct cna_ctx;
extern int cna_condition_phenotypes(struct cna_ctx* c, const double* M, const double* Y, int N, int P);
struct draw_then { struct cna_ctx* ctx; const double* M; const double* table; int N, cols; int* flag; };
static pthread_mutex_t g_draw_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_draw_cv = PTHREAD_COND_INITIALIZER;
static pthread_t g_draw_th;
static int g_draw_alive = 0, g_draw_state = 0, g_draw_rc = 0;        /* state 0 idle, 1 posted, 2 running, 3 done */
static int g_then_state = 0;                                           /* follow-up: 0 none, 1 posted, 2 running, 3 done */
static struct draw_req g_draw_req;
static struct draw_then g_draw_then;

static int draw_run(const struct draw_req* q) {
  int64_t mmax = 0;
  for (int l = 0; l < q->nlev; ++l) { const int64_t ml = q->lev_off[l + 1] - q->lev_off[l]; if (ml > mmax) mmax = ml; }
  double* R = (double*)malloc(sizeof(double) * (size_t)(mmax > 0 ? mmax : 1) * (size_t)q->num);
  double* ysub = (double*)malloc(sizeof(double) * (size_t)(mmax > 0 ? mmax : 1));
  if (!R || !ysub) { free(R); free(ysub); return -1; }
  int rc = 0;
  t_host_threads = q->threads > 0 ? q->threads : 1;
  for (int l = 0; l < q->nlev && rc == 0; ++l) {
    const int64_t* mem = q->members + q->lev_off[l];
    const int ml = (int)(q->lev_off[l + 1] - q->lev_off[l]);
    if (ml == 0) continue;
    int hg = 0;
    double g = 0.0;
    if (cna_host_legacy_randn(q->key, q->pos, &hg, &g, (int64_t)ml * q->num, R) != 0 || hg != 0) { rc = -1; break; }
    for (int i = 0; i < ml; ++i) ysub[i] = q->y[mem[i]];
    if (argsort_gather_idx(R, ml, q->num, ysub, q->out, q->ld_out, mem, q->idx_out, q->num) != 0) rc = -1;
  }
  t_host_threads = 0;
  free(R); free(ysub);
  return rc;
}

/* when the last draw and its follow-up finished (CLOCK_MONOTONIC seconds = std::chrono::steady_clock): stage marks of
 * cna_assoc_finish (cna_assoc_out.t_ms[12], [13]) */
#include <time.h>
static double g_draw_times[2];
static double mono_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
void cna_host_draw_times(double* out2) { out2[0] = g_draw_times[0]; out2[1] = g_draw_times[1]; }

static void* draw_thread(void* arg) {
  (void)arg;
  pthread_mutex_lock(&g_draw_mu);
  for (;;) {
    while (!(g_draw_state == 1 || (g_draw_state == 3 && g_then_state == 1))) pthread_cond_wait(&g_draw_cv, &g_draw_mu);
    if (g_draw_state == 1) {
      g_draw_state = 2;
      const struct draw_req q = g_draw_req;
      pthread_mutex_unlock(&g_draw_mu);
      const int rc = draw_run(&q);
      g_draw_times[0] = mono_now();
      pthread_mutex_lock(&g_draw_mu);
      g_draw_rc = rc;
      g_draw_state = 3;
    } else {
      g_then_state = 2;
      const struct draw_then t = g_draw_then;
      const int ok = g_draw_rc == 0;
      pthread_mutex_unlock(&g_draw_mu);
      const int rc = ok ? cna_condition_phenotypes(t.ctx, t.M, t.table, t.N, t.cols) : -1;
      g_draw_times[1] = mono_now();
      __atomic_store_n(t.flag, rc == 0 ? 1 : -1, __ATOMIC_RELEASE);
      pthread_mutex_lock(&g_draw_mu);
      g_then_state = 3;
    }
    pthread_cond_broadcast(&g_draw_cv);
  }
  return NULL;
}

/* a forked child has no worker thread (threads do not survive fork): it starts its own on first use */
static void draw_atfork_child(void) {
  pthread_mutex_init(&g_draw_mu, NULL);
  pthread_cond_init(&g_draw_cv, NULL);
  g_draw_alive = 0;
  g_draw_state = 0;
  g_then_state = 0;
}
static pthread_once_t g_draw_fork_once = PTHREAD_ONCE_INIT;
static void draw_register_atfork(void) { pthread_atfork(NULL, NULL, draw_atfork_child); }

/* 0: the request is with the worker (every pointer must stay valid until cna_host_draw_wait returns);
 * -1: bad arguments / an earlier request not collected / no thread: the caller draws the usual way, the generator
 * state has not been touched */
int cna_host_draw_start_idx(uint32_t* key, int* pos, const double* y, int m, int num, int nlev, const int64_t* lev_off,
                            const int64_t* members, double* out, int64_t ld_out, int threads, int32_t* idx_out);
int cna_host_draw_start(uint32_t* key, int* pos, const double* y, int m, int num, int nlev, const int64_t* lev_off,
                        const int64_t* members, double* out, int64_t ld_out, int threads) {
  return cna_host_draw_start_idx(key, pos, y, m, num, nlev, lev_off, members, out, ld_out, threads, NULL);
}
/* the same, also recording WHICH row of y every output is (idx_out: m x num int32, row-major; rows of no level are left
 * untouched): the permutations of a seeded draw depend on (seed, m, num, levels) only, so a caller that analyses many
 * phenotypes with one seed replays them with cna_host_gather_rows instead of drawing again */
int cna_host_draw_start_idx(uint32_t* key, int* pos, const double* y, int m, int num, int nlev, const int64_t* lev_off,
                            const int64_t* members, double* out, int64_t ld_out, int threads, int32_t* idx_out) {
  if (!key || !pos || !y || !lev_off || !members || !out || m < 1 || num < 2 || (num & 1) || nlev < 1 || ld_out < num) return -1;
  if (*pos < 0 || *pos > MT_N) return -1;
  for (int l = 0; l < nlev; ++l) if (lev_off[l + 1] < lev_off[l] || lev_off[l + 1] > m) return -1;
  for (int64_t i = 0; i < lev_off[nlev]; ++i) if (members[i] < 0 || members[i] >= m) return -1;
  pthread_once(&g_draw_fork_once, draw_register_atfork);
  pthread_mutex_lock(&g_draw_mu);
  if (g_draw_state != 0) { pthread_mutex_unlock(&g_draw_mu); return -1; }
  if (!g_draw_alive) {
    if (pthread_create(&g_draw_th, NULL, draw_thread, NULL) != 0) { pthread_mutex_unlock(&g_draw_mu); return -1; }
    pthread_detach(g_draw_th);
    g_draw_alive = 1;
  }
  g_draw_req = (struct draw_req){key, pos, y, m, num, nlev, lev_off, members, out, ld_out, threads, idx_out};
  g_draw_state = 1;
  g_then_state = 0;
  pthread_cond_broadcast(&g_draw_cv);
  pthread_mutex_unlock(&g_draw_mu);
  return 0;
}

/* A follow-up for the request under way (or finished and not yet collected): once the draw is there the worker
 * conditions the phenotypes itself -- cna_condition_phenotypes(ctx, M, table, N, cols), table = the N x cols matrix
 * [y | permutations] the draw fills -- and stores 1 (done) or -1 (failed) in *flag, which the caller and
 * cna_select_standardized_fused read.  0: accepted; -1: nothing to follow (no request, or one follow-up already). */
int cna_host_draw_then_condition(struct cna_ctx* ctx, const double* M, const double* table, int N, int cols, int* flag) {
  if (!ctx || !M || !table || !flag || N < 2 || cols < 1) return -1;
  pthread_mutex_lock(&g_draw_mu);
  if (g_draw_state == 0 || g_then_state != 0) { pthread_mutex_unlock(&g_draw_mu); return -1; }
  g_draw_then = (struct draw_then){ctx, M, table, N, cols, flag};
  g_then_state = 1;
  pthread_cond_broadcast(&g_draw_cv);
  pthread_mutex_unlock(&g_draw_mu);
  return 0;
}

/* blocks until the request of cna_host_draw_start (and its follow-up, if one was posted) is done; 0, or -1 when the
 * draw failed (allocation) -- the generator state is then undefined and the caller must raise; -2: nothing was started */
int cna_host_draw_wait(void) {
  pthread_mutex_lock(&g_draw_mu);
  if (g_draw_state == 0) { pthread_mutex_unlock(&g_draw_mu); return -2; }
  while (!(g_draw_state == 3 && (g_then_state == 0 || g_then_state == 3))) pthread_cond_wait(&g_draw_cv, &g_draw_mu);
  const int rc = g_draw_rc;
  g_draw_state = 0;
  g_then_state = 0;
  pthread_mutex_unlock(&g_draw_mu);
  return rc;
}

/* as cna_host_draw_wait, but the request stays collectable: for a library-side consumer of the table (cna_assoc_finish)
 * whose caller still collects the draw itself (it writes the generator state back) */
int cna_host_draw_join(void) {
  pthread_mutex_lock(&g_draw_mu);
  if (g_draw_state == 0) { pthread_mutex_unlock(&g_draw_mu); return -2; }
  while (!(g_draw_state == 3 && (g_then_state == 0 || g_then_state == 3))) pthread_cond_wait(&g_draw_cv, &g_draw_mu);
  const int rc = g_draw_rc;
  pthread_mutex_unlock(&g_draw_mu);
  return rc;
}

/* out[r * ld_out + p] = y[idx[r * num + p]] (m x num): the permuted phenotypes of a draw whose source rows were recorded by
 * cna_host_draw_start_idx -- conditional_permutation's Y[bix] (_stats.py:16-17) without the random numbers and the sorts.
 * Rows are split over up to nthreads threads (large draws). */
struct gather_job { const double* y; const int32_t* idx; double* out; int64_t ld_out; int num, r0, r1; };
static void* gather_worker(void* arg) {
  const struct gather_job* j = (const struct gather_job*)arg;
  for (int r = j->r0; r < j->r1; ++r) {
    const int32_t* s = j->idx + (size_t)r * j->num;
    double* d = j->out + (size_t)r * j->ld_out;
    for (int p = 0; p < j->num; ++p) d[p] = j->y[s[p]];
  }
  return NULL;
}
int cna_host_gather_rows(const double* y, const int32_t* idx, int m, int num, double* out, int64_t ld_out, int nthreads) {
  if (!y || !idx || !out || m < 1 || num < 1 || ld_out < num) return -1;
  for (int64_t i = 0; i < (int64_t)m * num; ++i) if (idx[i] < 0 || idx[i] >= m) return -1;
  if (nthreads > 16) nthreads = 16;
  if ((int64_t)m * num < 262144 || nthreads < 2) nthreads = 1;
  if (nthreads > m) nthreads = m;
  struct gather_job jobs[16];
  pthread_t th[16];
  int started[16];
  for (int t = 0; t < nthreads; ++t) {
    jobs[t] = (struct gather_job){y, idx, out, ld_out, num, (int)((int64_t)m * t / nthreads), (int)((int64_t)m * (t + 1) / nthreads)};
    started[t] = 0;
  }
  for (int t = 1; t < nthreads; ++t) started[t] = pthread_create(&th[t], NULL, gather_worker, &jobs[t]) == 0;
  gather_worker(&jobs[0]);
  for (int t = 1; t < nthreads; ++t) { if (started[t]) pthread_join(th[t], NULL); else gather_worker(&jobs[t]); }
  return 0;
}
