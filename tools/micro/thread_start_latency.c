#include <pthread.h>
#include <stdio.h>
#include <time.h>
#include <math.h>
#include <stdlib.h>
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static double t_create, t_started[8], t_done[8];
static double *X, *O; static long NP = 100000;
static void work(long a, long b) { for (long p = a; p < b; ++p) { double f = sqrt(-2.0 * log(X[p]) / X[p]); O[2 * p] = f * X[p]; O[2 * p + 1] = f * 0.5; } }
static void* th(void* p) { long t = (long)p; t_started[t] = now_ms(); work(NP * t / 4, NP * (t + 1) / 4); t_done[t] = now_ms(); return 0; }
int main() {
  X = malloc(8 * NP); O = malloc(16 * NP);
  for (long i = 0; i < NP; ++i) X[i] = 0.01 + 0.98 * rand() / RAND_MAX;
  work(0, NP);
  double a = now_ms(); work(0, NP); double b = now_ms();
  printf("solo %.3f ms\n", b - a);
  for (int rep = 0; rep < 5; ++rep) {
    /* busy phase first, like the producer */
    volatile double s = 0; double q = now_ms(); while (now_ms() - q < 0.4) s += 1;
    pthread_t h[4]; t_create = now_ms();
    for (long t = 1; t < 4; ++t) pthread_create(&h[t], 0, th, (void*)t);
    double c1 = now_ms();
    th((void*)0);
    for (int t = 1; t < 4; ++t) pthread_join(h[t], 0);
    double e = now_ms();
    printf("creates took %.3f; total %.3f; member start/done rel. to create: ", c1 - t_create, e - t_create);
    for (int t = 0; t < 4; ++t) printf("[%d] %.3f/%.3f ", t, t_started[t] - t_create, t_done[t] - t_create);
    printf("\n");
  }
}
