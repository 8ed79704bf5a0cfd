/* Stress driver for the threaded host side of libcna_hip (csrc/host_rng.c, host_graph.c, host_eig.c), built with
 * -fsanitize=thread or -fsanitize=address by `make -C cna_amd/csrc tsan|asan` and run by tests/test_host_sanitizers.py.
 * SURVEY.md 5 lists race detection among the auxiliary subsystems; the reference has no threads, this library has
 * five kinds of helper threads on the host.  What is driven here, all at once where the product allows it:
 *   - the library's draw thread: cna_host_draw_start / _then_condition / _wait in a loop, with 1 and 4 worker threads,
 *     while the calling thread draws from its OWN generator state, sorts with cna_host_argsort_gather and flips
 *     cna_host_set_threads (the overlap of a Python-path draw with the native one);
 *   - the multi-threaded cluster order, content hash, threaded copy and 16-bit expansion of host_graph.c;
 *   - cna_host_top_eig from four threads at once (its work space is per thread);
 *   - fork after use: the child has no draw thread and must start its own; the parent goes on.
 * The GPU side (c_api.hip) is not built here: hipcc's device runtime does not run under the sanitizers.
 * Exit code 0 and no sanitizer report = clean. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

struct cna_ctx;
int cna_host_legacy_randn(uint32_t* key, int* pos, int* has_gauss, double* gauss, int64_t n, double* out);
void cna_host_set_threads(int n);
int cna_host_argsort_gather(const double* R, int m, int num, const double* y, double* out, int64_t ld_out, const int64_t* rows);
int cna_host_draw_start(uint32_t* key, int* pos, const double* y, int m, int num, int nlev, const int64_t* lev_off,
                        const int64_t* members, double* out, int64_t ld_out, int threads);
int cna_host_draw_then_condition(struct cna_ctx* ctx, const double* M, const double* table, int N, int cols, int* flag);
int cna_host_draw_wait(void);
uint64_t cna_host_hash64(const void* p, int64_t nbytes, int nthreads);
int cna_host_copy(void* dst, const void* src, int64_t nbytes, int nthreads);
int cna_host_expand_u16(double* dst, const uint16_t* bins, int64_t n, const double* runmin, int T, int nthreads);
int64_t cna_host_cluster_order(int64_t n, const int64_t* indptr, const int32_t* indices, int B, int64_t* order_out);
int64_t cna_host_cluster_order_mt(int64_t n, const int64_t* indptr, const int32_t* indices, int B, int nthreads, int64_t* order_out);
int cna_host_top_eig(const double* G, int n, int k, double* U_out, double* lam_out, double* resid_out, double* ortho_out);

/* what the draw thread calls when asked to condition the phenotypes itself: the GPU entry point, stubbed -- it reads
 * the table the draw has just filled (as the real one uploads it) */
static double g_cond_sum;
int cna_condition_phenotypes(struct cna_ctx* c, const double* M, const double* Y, int N, int P) {
  (void)c;
  double s = 0.0;
  for (int i = 0; i < N * P; ++i) s += Y[i];
  for (int i = 0; i < N * N; ++i) s += M[i];
  g_cond_sum = s;
  return 0;
}

static void mt_seed(uint32_t* key, int* pos, uint32_t seed) {       /* init_genrand */
  key[0] = seed;
  for (int i = 1; i < 624; ++i) key[i] = 1812433253u * (key[i - 1] ^ (key[i - 1] >> 30)) + (uint32_t)i;
  *pos = 624;
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "host_stress: check failed at line %d: %s\n", __LINE__, #c); exit(3); } } while (0)

static void draw_rounds(int rounds, int threads) {
  enum { N = 200, P = 1000 };
  static uint32_t key[625];
  static double y[N], table[N * (P + 1)], M[N * N], own[N * 64], sorted[N * 64];
  int64_t off[2] = {0, N}, mem[N];
  for (int i = 0; i < N; ++i) { y[i] = sin(i * 0.37); mem[i] = i; }
  for (int i = 0; i < N * N; ++i) M[i] = (i % (N + 1)) == 0;
  for (int r = 0; r < rounds; ++r) {
    int pos;
    mt_seed(key, &pos, 1000u + (uint32_t)r);
    for (int i = 0; i < N; ++i) table[(size_t)i * (P + 1)] = y[i];
    int flag = 0;
    CHECK(cna_host_draw_start(key, &pos, y, N, P, 1, off, mem, table + 1, P + 1, threads) == 0);
    CHECK(cna_host_draw_then_condition((struct cna_ctx*)&flag, M, table, N, P + 1, &flag) == 0);
    /* meanwhile, on this thread: a draw of its own from its own state, the global thread count flipped under it */
    uint32_t k2[625];
    int p2, hg = 0;
    double g = 0.0;
    mt_seed(k2, &p2, 7u + (uint32_t)r);
    cna_host_set_threads(1 + (r & 3));
    CHECK(cna_host_legacy_randn(k2, &p2, &hg, &g, (int64_t)N * 64, own) == 0);
    CHECK(cna_host_argsort_gather(own, N, 64, y, sorted, 64, NULL) == 0);
    cna_host_set_threads(1);
    CHECK(cna_host_draw_wait() == 0);
    CHECK(__atomic_load_n(&flag, __ATOMIC_ACQUIRE) == 1);
    /* every column of the table is a permutation of y */
    double s0 = 0.0, s1 = 0.0;
    for (int i = 0; i < N; ++i) { s0 += y[i]; s1 += table[(size_t)i * (P + 1) + 1 + (r % P)]; }
    CHECK(fabs(s0 - s1) < 1e-9);
    CHECK(cna_host_draw_wait() == -2);                                /* nothing pending any more */
  }
}

static void graph_rounds(void) {
  const int64_t n = 150000;
  const int deg = 12;
  int64_t* indptr = malloc(sizeof(int64_t) * (n + 1));
  int32_t* idx = malloc(sizeof(int32_t) * n * deg);
  uint64_t s = 12345;
  for (int64_t i = 0; i <= n; ++i) indptr[i] = i * deg;
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < deg; ++d) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      int64_t j = i + (int64_t)((s >> 33) % 400) - 200;
      if (j < 0) j = 0;
      if (j >= n) j = n - 1;
      idx[i * deg + d] = (int32_t)j;
    }
  int64_t* o1 = malloc(sizeof(int64_t) * n);
  int64_t* o2 = malloc(sizeof(int64_t) * n);
  CHECK(cna_host_cluster_order_mt(n, indptr, idx, 512, 8, o1) >= 0);
  CHECK(cna_host_cluster_order_mt(n, indptr, idx, 512, 3, o2) >= 0);
  CHECK(memcmp(o1, o2, sizeof(int64_t) * n) == 0);                   /* the order depends on the graph, not on the thread count */
  char* seen = calloc(n, 1);
  for (int64_t i = 0; i < n; ++i) { CHECK(o1[i] >= 0 && o1[i] < n && !seen[o1[i]]); seen[o1[i]] = 1; }
  const int64_t nb = sizeof(int32_t) * n * deg;
  CHECK(cna_host_hash64(idx, nb, 8) == cna_host_hash64(idx, nb, 8));
  const uint64_t h1 = cna_host_hash64(idx, nb, 8);
  idx[nb / 8] ^= 1;
  CHECK(cna_host_hash64(idx, nb, 8) != h1);
  int32_t* cp = malloc(nb);
  CHECK(cna_host_copy(cp, idx, nb, 6) == 0 && memcmp(cp, idx, nb) == 0);
  uint16_t* bins = malloc(sizeof(uint16_t) * n);
  double* dst = malloc(sizeof(double) * n);
  double runmin[301];
  for (int t = 0; t <= 300; ++t) runmin[t] = 1.0 / (1 + t);
  for (int64_t i = 0; i < n; ++i) bins[i] = (uint16_t)(i % 301);
  CHECK(cna_host_expand_u16(dst, bins, n, runmin, 300, 8) == 0);
  free(indptr); free(idx); free(o1); free(o2); free(seen); free(cp); free(bins); free(dst);
}

static void* eig_thread(void* arg) {
  const int n = 96 + 8 * (int)(intptr_t)arg, k = 8;
  double* G = malloc(sizeof(double) * n * n);
  double* U = malloc(sizeof(double) * n * k);
  double lam[9], r, o;
  for (int rep = 0; rep < 20; ++rep) {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) {
        const double v = (i == j ? 10.0 + 100.0 / (1 + i) : 1.0 / (1 + i + j + rep));
        G[i * n + j] = G[j * n + i] = v;
      }
    if (cna_host_top_eig(G, n, k, U, lam, &r, &o) != 0 || !(r < 1e-9) || !(o < 1e-9)) return (void*)1;
  }
  free(G); free(U);
  return NULL;
}

int main(void) {
  draw_rounds(40, 1);
  draw_rounds(40, 4);
  graph_rounds();
  pthread_t th[4];
  for (intptr_t t = 0; t < 4; ++t) CHECK(pthread_create(&th[t], NULL, eig_thread, (void*)t) == 0);
  for (int t = 0; t < 4; ++t) { void* rc; pthread_join(th[t], &rc); CHECK(rc == NULL); }
  /* fork after use: the child has no worker thread and starts its own */
  fflush(NULL);
  const pid_t pid = fork();
  CHECK(pid >= 0);
  if (pid == 0) {
    draw_rounds(5, 2);
    _exit(0);
  }
  int status = 0;
  CHECK(waitpid(pid, &status, 0) == pid && WIFEXITED(status) && WEXITSTATUS(status) == 0);
  draw_rounds(5, 4);                                                  /* ... and the parent's thread is still there */
  printf("host_stress ok\n");
  return 0;
}
