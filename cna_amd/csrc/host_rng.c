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
#define CLONES __attribute__((target_clones("avx2", "default")))
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
    for (; used < nc && found < pairs; used++) {
      const double r = r2[used];
      if (r < 1.0 && r != 0.0) {
        ax1[found] = x1[used];
        ax2[found] = x2[used];
        ar2[found] = r;
        found++;
      }
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
  /* survivors -> normals, in order: the second value of a pair is numpy's cached one */
  for (int64_t j = 0; j < pairs; j++) {
    const double f = sqrt(-2.0 * log(ar2[j]) / ar2[j]);
    const double second = f * ax1[j];
    out[done++] = f * ax2[j];
    if (done < n) {
      out[done++] = second;
    } else {
      *has_gauss = 1;
      *gauss = second;
    }
  }
  free(acc);
  return 0;
}
