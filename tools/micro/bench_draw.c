/* Phases of the permutation draw (200 samples x 1000 permutations) on this machine:  gcc -O3 -ffp-contract=off -fno-math-errno -o /tmp/bench_draw tools/micro/bench_draw.c -lm -lpthread && /tmp/bench_draw */
#include "../../cna_amd/csrc/host_rng.c"
#include <stdio.h>
int cna_condition_phenotypes(struct cna_ctx* c, const double* M, const double* Y, int N, int P) { (void)c; (void)M; (void)Y; (void)N; (void)P; return 0; }
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
int main(void) {
  uint32_t key[MT_N], words[MT_N];
  static double x1[MT_N / 4 + 1], x2[MT_N / 4 + 1], r2[MT_N / 4 + 1];
  for (int i = 0; i < MT_N; ++i) key[i] = 1812433253u * (i + 1);
  const int reps = 200, blocks = 815;                   /* 815 blocks of 156 candidates = 127k candidates = 100k pairs */
  double t0 = now_ms();
  for (int r = 0; r < reps; ++r) for (int b = 0; b < blocks; ++b) mt_reload(key);
  double t1 = now_ms();
  for (int r = 0; r < reps; ++r) for (int b = 0; b < blocks; ++b) temper(key, words, MT_N);
  double t2 = now_ms();
  for (int r = 0; r < reps; ++r) for (int b = 0; b < blocks; ++b) candidates(words, MT_N / 4, x1, x2, r2);
  double t3 = now_ms();
  static double a1[200000], a2[200000], ar[200000];
  volatile int64_t sink = 0;
  for (int r = 0; r < reps; ++r) { int64_t found = 0; for (int b = 0; b < blocks; ++b) { mt_reload(key); temper(key, words, MT_N); candidates(words, MT_N / 4, x1, x2, r2);
      for (int u = 0; u < MT_N / 4; ++u) { const double q = r2[u]; if (q < 1.0 && q != 0.0) { a1[found] = x1[u]; a2[found] = x2[u]; ar[found] = q; found++; } } } sink += found; }
  double t4 = now_ms();
  static double out[400000];
  int64_t pairs = 100000;
  for (int r = 0; r < reps; ++r) { struct norm_job j = {a1, a2, ar, out, 2 * pairs, 0, pairs}; norm_worker(&j); }
  double t5 = now_ms();
  {
    uint32_t k2[MT_N]; for (int i = 0; i < MT_N; ++i) k2[i] = 1812433253u * (i + 7);
    static double R[200 * 1000], outp[200 * 1000], y[200];
    for (int i = 0; i < 200; ++i) y[i] = i;
    for (int nt = 1; nt <= 4; nt *= 2) {
      double tb = 0, tc = 0;
      for (int r = 0; r < reps; ++r) {
        t_host_threads = nt;
        int pos = MT_N, hg = 0; double g = 0;
        double b = now_ms();
        cna_host_legacy_randn(k2, &pos, &hg, &g, 200 * 1000, R);
        double c = now_ms();
        argsort_gather_idx(R, 200, 1000, y, outp, 1000, NULL, NULL, 0);
        double d = now_ms();
        t_host_threads = 0;
        tb += c - b; tc += d - c;
      }
      printf("%d thread(s): normal stream %.3f  argsort + gather %.3f ms\n", nt, tb / reps, tc / reps);
    }
  }
  printf("per 100k pairs: mt_reload %.3f  temper %.3f  candidates %.3f  all + compaction %.3f  log/sqrt %.3f ms\n", (t1 - t0) / reps, (t2 - t1) / reps, (t3 - t2) / reps, (t4 - t3) / reps, (t5 - t4) / reps);
  return 0;
}
