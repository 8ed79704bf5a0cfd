/* Host-side helpers for graph preparation (plain C, no device code, no floating-point arithmetic of
 * the analysis): nothing here computes a result of the path, it identifies and orders the input.
 *
 *   cna_host_hash64     content hash of a buffer on several threads: the engine recognises "this very
 *                       graph is already resident" by the full content of data / indices / indptr, so an
 *                       in-place edit of a single entry is seen (the reference reads the matrix afresh on
 *                       every call, /root/reference/src/cna/tools/_nam.py:25-28)
 *   cna_host_cluster_order   (below) the cell order of the device copy of the graph
 */
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define P1 0x9E3779B185EBCA87ull
#define P2 0xC2B2AE3D27D4EB4Full
#define P3 0x165667B19E3779F9ull

static inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t mixin(uint64_t acc, uint64_t w) { return rotl(acc + w * P2, 31) * P1; }
static inline uint64_t fin(uint64_t h) {
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

/* one chunk: four independent multiply-rotate lanes over 32-byte stripes, then the tail bytewise */
static uint64_t hash_chunk(const unsigned char* p, size_t n, uint64_t seed) {
  uint64_t a = seed + P1, b = seed ^ P2, c = seed + P3, d = seed - P1;
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    uint64_t w[4];
    memcpy(w, p + i, 32);
    a = mixin(a, w[0]); b = mixin(b, w[1]); c = mixin(c, w[2]); d = mixin(d, w[3]);
  }
  uint64_t h = rotl(a, 1) + rotl(b, 7) + rotl(c, 12) + rotl(d, 18) + (uint64_t)n;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = mixin(h, w);
  }
  for (; i < n; ++i) h = mixin(h, (uint64_t)p[i] + 0x100);
  return fin(h);
}

#define HASH_CHUNK ((size_t)1 << 20)

struct hash_job {
  const unsigned char* p;
  size_t n, nchunks;
  uint64_t* out;
  int tid, nthreads;
};

static void* hash_worker(void* arg) {
  struct hash_job* j = (struct hash_job*)arg;
  for (size_t c = (size_t)j->tid; c < j->nchunks; c += (size_t)j->nthreads) {
    const size_t off = c * HASH_CHUNK;
    const size_t len = j->n - off < HASH_CHUNK ? j->n - off : HASH_CHUNK;
    j->out[c] = hash_chunk(j->p + off, len, (uint64_t)c);
  }
  return NULL;
}

/* 64-bit content hash of nbytes at p (1 MiB chunks hashed on up to nthreads threads, chunk digests
 * chained in order: the result does not depend on the thread count). */
uint64_t cna_host_hash64(const void* p, int64_t nbytes, int nthreads) {
  if (nbytes <= 0 || !p) return fin(P3);
  const size_t n = (size_t)nbytes;
  const size_t nchunks = (n + HASH_CHUNK - 1) / HASH_CHUNK;
  uint64_t* dig = (uint64_t*)malloc(8 * nchunks);
  if (!dig) return 0;
  if (nthreads > 64) nthreads = 64;
  if ((size_t)nthreads > nchunks) nthreads = (int)nchunks;
  if (nthreads < 1) nthreads = 1;
  struct hash_job jobs[64];
  pthread_t th[64];
  int started[64];
  for (int t = 0; t < nthreads; ++t) {
    jobs[t].p = (const unsigned char*)p; jobs[t].n = n; jobs[t].nchunks = nchunks; jobs[t].out = dig;
    jobs[t].tid = t; jobs[t].nthreads = nthreads;
    started[t] = 0;
  }
  for (int t = 1; t < nthreads; ++t) started[t] = pthread_create(&th[t], NULL, hash_worker, &jobs[t]) == 0;
  hash_worker(&jobs[0]);
  for (int t = 1; t < nthreads; ++t) {
    if (started[t]) pthread_join(th[t], NULL);
    else hash_worker(&jobs[t]);            /* could not start the thread: do its share here */
  }
  uint64_t h = P1 ^ (uint64_t)n;
  for (size_t c = 0; c < nchunks; ++c) h = mixin(h, dig[c]);
  free(dig);
  return fin(h);
}

/* memcpy on several threads: the per-cell result columns (8 bytes x cells) from the context's pinned
 * staging into the caller's frame */
struct copy_job { char* d; const char* s; size_t n; };
static void* copy_worker(void* a) {
  struct copy_job* j = (struct copy_job*)a;
  memcpy(j->d, j->s, j->n);
  return NULL;
}
int cna_host_copy(void* dst, const void* src, int64_t nbytes, int nthreads) {
  if (nbytes <= 0) return 0;
  if (!dst || !src) return 1;
  if (nthreads > 16) nthreads = 16;
  if ((int64_t)nthreads > nbytes / (1 << 20)) nthreads = (int)(nbytes / (1 << 20));
  if (nthreads < 1) nthreads = 1;
  struct copy_job jobs[16];
  pthread_t th[16];
  int started[16];
  const size_t part = (((size_t)nbytes / nthreads) + 63) & ~(size_t)63;
  for (int t = 0; t < nthreads; ++t) {
    const size_t b = (size_t)t * part;
    jobs[t].d = (char*)dst + b; jobs[t].s = (const char*)src + b;
    jobs[t].n = b >= (size_t)nbytes ? 0 : ((size_t)nbytes - b < part || t == nthreads - 1 ? (size_t)nbytes - b : part);
    started[t] = 0;
  }
  for (int t = 1; t < nthreads; ++t) started[t] = pthread_create(&th[t], NULL, copy_worker, &jobs[t]) == 0;
  copy_worker(&jobs[0]);
  for (int t = 1; t < nthreads; ++t) {
    if (started[t]) pthread_join(th[t], NULL);
    else copy_worker(&jobs[t]);
  }
  return 0;
}

/* dst[i] = bins[i] > 0 ? runmin[bins[i] - 1] : 1.0 on several threads: the per-cell FDR column (_association.py:234-237)
 * from the per-cell threshold counts, which leave the device while the local null still runs, and the FDR table,
 * which follows it (2 bytes per cell cross PCIe instead of 8, and none of them after the null) */
struct expand_job { double* d; const uint16_t* b; size_t n; const double* tab; int T; };
/* branch-free: a 1024-entry table t2[h] = 1 for h = 0 (and beyond T), runmin[h - 1] otherwise */
#ifndef CNA_NO_CLONES          /* (sanitizer builds: no ifunc resolvers) */
__attribute__((target_clones("avx2", "default")))
#endif
static void expand_span(double* d, const uint16_t* b, size_t n, const double* t2) {
  for (size_t i = 0; i < n; ++i) d[i] = t2[b[i] & 1023u];
}
static void* expand_worker(void* a) {
  struct expand_job* j = (struct expand_job*)a;
  expand_span(j->d, j->b, j->n, j->tab);
  return NULL;
}
int cna_host_expand_u16(double* dst, const uint16_t* bins, int64_t n, const double* runmin, int T, int nthreads) {
  if (n <= 0) return 0;
  if (!dst || !bins || !runmin) return 1;
  if (nthreads > 16) nthreads = 16;
  if ((int64_t)nthreads > n / (1 << 17)) nthreads = (int)(n / (1 << 17));
  if (nthreads < 1) nthreads = 1;
  struct expand_job jobs[16];
  pthread_t th[16];
  int started[16];
  double t2[1024];
  for (int h = 0; h < 1024; ++h) t2[h] = (h > 0 && h <= T && h <= 512) ? runmin[h - 1] : 1.0;
  const size_t part = (((size_t)n / nthreads) + 63) & ~(size_t)63;
  for (int t = 0; t < nthreads; ++t) {
    const size_t b = (size_t)t * part;
    jobs[t].d = dst + b; jobs[t].b = bins + b; jobs[t].tab = t2; jobs[t].T = T;
    jobs[t].n = b >= (size_t)n ? 0 : ((size_t)n - b < part || t == nthreads - 1 ? (size_t)n - b : part);
    started[t] = 0;
  }
  for (int t = 1; t < nthreads; ++t) started[t] = pthread_create(&th[t], NULL, expand_worker, &jobs[t]) == 0;
  expand_worker(&jobs[0]);
  for (int t = 1; t < nthreads; ++t) {
    if (started[t]) pthread_join(th[t], NULL);
    else expand_worker(&jobs[t]);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Cell order of the device copy: clusters of `B` cells that share neighbours.
 *
 * The walk kernel (csrc/diffuse_lds.hip) gives a workgroup a block of B consecutive device rows and
 * stages the state rows of ALL their neighbours in LDS once; what it saves over a row-by-row gather is
 * edges / distinct neighbour rows of the block.  A bandwidth-reducing order (reverse Cuthill-McKee)
 * reaches 2.0 (B = 64) ... 2.4 (B = 128) on a k = 30 graph of 8-d points; growing each block greedily --
 * start from a seed, keep adding the not yet placed cell with the most edges into the block -- reaches
 * 3.0 ... 3.6 (tools/kbench_reorder.py).  Integer work only; the numbering never changes a result (rows
 * keep their neighbours in the caller's order).
 *
 * Seeds: the oldest not yet placed cell among those that were candidates of an earlier cluster (FIFO),
 * else the next unplaced cell in index order -- successive clusters are therefore graph neighbours, which
 * keeps a workgroup's neighbours' neighbours in the XCD's L2.  Clusters that run out of candidates before
 * reaching B cells are collected and laid out, in creation order, behind the full ones.
 *
 * n cells; CSR (indptr int64[n+1], indices int32) of the graph rows; only columns < n are followed (a
 * sharded caller passes the diagonal part of its row block).  order_out[i] = caller's index of device
 * row i.  Returns the number of cells in full clusters, or -1 (out of memory). */
int64_t cna_host_cluster_order(int64_t n, const int64_t* indptr, const int32_t* indices, int B, int64_t* order_out) {
  if (n <= 0) return 0;
  if (B < 1) B = 1;
  /* state[j]: -1 = placed, else the number of edges from the growing cluster into j (one random access per edge;
   * with `placed` and `links` as two arrays the pass took 0.59 s at 2M cells, three quarters of a cold call) */
  int32_t* state = (int32_t*)calloc((size_t)n, 4);
  int32_t* touched = (int32_t*)malloc(4 * (size_t)n);         /* candidates of the current cluster */
  int32_t* fifo = (int32_t*)malloc(4 * (size_t)n);            /* seeds-to-be: every cell enters at most once */
  unsigned char* queued = (unsigned char*)calloc((size_t)n, 1);
  int64_t* shorts = (int64_t*)malloc(8 * (size_t)n);          /* members of short clusters */
  int32_t* members = (int32_t*)malloc(4 * (size_t)B);
  /* lazy bucket queue: level[c] holds cells whose link count was c when pushed; stale entries are skipped when
   * popped.  Pushes per cluster <= edges of its members.  A link count can exceed B when a row lists a column
   * more than once (non-canonical CSR): counts are filed under min(count, B + 1). */
  int32_t** level = (int32_t**)calloc((size_t)B + 2, sizeof(int32_t*));
  int64_t* lsize = (int64_t*)calloc((size_t)B + 2, 8);
  int64_t* lcap = (int64_t*)calloc((size_t)B + 2, 8);
  int64_t n_full = 0, n_short = 0, fifo_head = 0, fifo_tail = 0, next_index = 0;
  int64_t rc = -1;
  if (!state || !touched || !fifo || !queued || !shorts || !members || !level || !lsize || !lcap) goto done;
  for (;;) {
    int64_t seed = -1;
    while (fifo_head < fifo_tail) {
      const int32_t c = fifo[fifo_head++];
      if (state[c] >= 0) { seed = c; break; }
    }
    if (seed < 0) {
      while (next_index < n && state[next_index] < 0) ++next_index;
      if (next_index >= n) break;
      seed = next_index;
    }
    int m = 0, top = 0;
    int64_t ntouched = 0;
    int32_t cur = (int32_t)seed;
    for (;;) {
      if (state[cur] == 0) touched[ntouched++] = cur;         /* (a seed nobody linked to: reset below like the others) */
      state[cur] = -1;
      members[m++] = cur;
      if (m == B) break;
      for (int64_t e = indptr[cur]; e < indptr[cur + 1]; ++e) {
        const int32_t j = indices[e];
        if (j < 0 || j >= n) continue;
        const int32_t sj = state[j];
        if (sj < 0) continue;
        if (sj == 0) touched[ntouched++] = j;
        state[j] = sj + 1;
        const int c = sj + 1 > B + 1 ? B + 1 : sj + 1;
        if (lsize[c] == lcap[c]) {
          lcap[c] = lcap[c] ? 2 * lcap[c] : 256;
          int32_t* grown = (int32_t*)realloc(level[c], 4 * (size_t)lcap[c]);
          if (!grown) goto done;
          level[c] = grown;
        }
        level[c][lsize[c]++] = j;
        if (c > top) top = c;
      }
      int32_t pick = -1;
      while (top > 0) {
        if (lsize[top] == 0) { --top; continue; }
        const int32_t j = level[top][--lsize[top]];
        const int32_t sj = state[j];
        if (sj >= 0 && (sj > B + 1 ? B + 1 : sj) == top) { pick = j; break; }
      }
      if (pick < 0) break;                                    /* no candidate left: a short cluster */
      cur = pick;
    }
    for (int c = 0; c <= B + 1; ++c) lsize[c] = 0;
    for (int64_t t = 0; t < ntouched; ++t) {
      const int32_t j = touched[t];
      if (state[j] >= 0) {
        state[j] = 0;
        if (!queued[j]) { queued[j] = 1; fifo[fifo_tail++] = j; }
      }
    }
    if (m == B) {
      for (int i = 0; i < m; ++i) order_out[n_full + i] = members[i];
      n_full += m;
    } else {
      for (int i = 0; i < m; ++i) shorts[n_short + i] = members[i];
      n_short += m;
    }
  }
  for (int64_t i = 0; i < n_short; ++i) order_out[n_full + i] = shorts[i];
  rc = n_full;
done:
  if (level) for (int c = 0; c <= B + 1; ++c) free(level[c]);
  free(level); free(lsize); free(lcap); free(state); free(touched); free(fifo); free(queued);
  free(shorts); free(members);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * The same order on several threads (round 3: the sequential pass was three quarters of a cold call at 2M cells).
 *   1. the cells are split into P regions by a multi-source breadth-first search from P evenly spaced seeds: a cell
 *      joins the region that reaches it first, the lowest-numbered one among those that reach it in the same round
 *      (level-synchronous rounds, atomic minimum) -- connected, compact regions whatever the caller's numbering;
 *      cells no seed reaches (other components) are dealt to the regions by index;
 *   2. every region is ordered by the greedy cluster growth above, edges that leave the region ignored -- regions are
 *      independent, so they run in parallel;
 *   3. layout: the full clusters of region 0, 1, ..., then the short clusters of all regions.
 * P depends on n only (n / 65536, at most 64; one region below 131072 cells = the sequential order): the result
 * does not depend on the number of threads.  Returns the number of cells in full clusters, -1: out of memory. */
struct co_shared {
  int64_t n; const int64_t* indptr; const int32_t* indices; int B, P, nthreads;
  int32_t* region; int32_t* cand;
  int32_t* frontier; int64_t nfront;             /* current round */
  int32_t** next; int64_t* nnext; int64_t* capnext;   /* per thread */
  pthread_barrier_t bar;
  /* phase 2 */
  int64_t* rstart; int32_t* members;             /* cells of region p: members[rstart[p] .. rstart[p+1]) ascending */
  int32_t* state; unsigned char* queued;
  int64_t* out_full; int64_t* n_full; int64_t* out_short; int64_t* n_short;   /* per region, written at rstart[p] */
  volatile int failed; volatile int64_t next_region;
  int gate;                                      /* start gate of the workers (co_worker) */
};
struct co_job { struct co_shared* sh; int tid; };

static int co_grow(struct co_shared* sh, int p) {
  const int B = sh->B;
  const int64_t r0 = sh->rstart[p], cnt = sh->rstart[p + 1] - r0;
  if (cnt == 0) { sh->n_full[p] = sh->n_short[p] = 0; return 0; }
  const int32_t* mem = sh->members + r0;
  int32_t* state = sh->state;
  const int32_t* region = sh->region;
  int32_t* touched = (int32_t*)malloc(4 * (size_t)cnt);
  int32_t* fifo = (int32_t*)malloc(4 * (size_t)cnt);
  int32_t* cl = (int32_t*)malloc(4 * (size_t)B);
  int32_t** level = (int32_t**)calloc((size_t)B + 2, sizeof(int32_t*));
  int64_t* lsize = (int64_t*)calloc((size_t)B + 2, 8);
  int64_t* lcap = (int64_t*)calloc((size_t)B + 2, 8);
  int64_t* full = sh->out_full + r0;
  int64_t* shorts = sh->out_short + r0;
  int64_t n_full = 0, n_short = 0, fifo_head = 0, fifo_tail = 0, next_i = 0;
  int rc = -1;
  if (!touched || !fifo || !cl || !level || !lsize || !lcap) goto done;
  for (;;) {
    int64_t seed = -1;
    while (fifo_head < fifo_tail) {
      const int32_t c = fifo[fifo_head++];
      if (state[c] >= 0) { seed = c; break; }
    }
    if (seed < 0) {
      while (next_i < cnt && state[mem[next_i]] < 0) ++next_i;
      if (next_i >= cnt) break;
      seed = mem[next_i];
    }
    int m = 0, top = 0;
    int64_t ntouched = 0;
    int32_t cur = (int32_t)seed;
    for (;;) {
      if (state[cur] == 0) touched[ntouched++] = cur;
      state[cur] = -1;
      cl[m++] = cur;
      if (m == B) break;
      for (int64_t e = sh->indptr[cur]; e < sh->indptr[cur + 1]; ++e) {
        const int32_t j = sh->indices[e];
        if (j < 0 || j >= sh->n || region[j] != p) continue;
        const int32_t sj = state[j];
        if (sj < 0) continue;
        if (sj == 0) touched[ntouched++] = j;
        state[j] = sj + 1;
        const int c = sj + 1 > B + 1 ? B + 1 : sj + 1;
        if (lsize[c] == lcap[c]) {
          lcap[c] = lcap[c] ? 2 * lcap[c] : 256;
          int32_t* grown = (int32_t*)realloc(level[c], 4 * (size_t)lcap[c]);
          if (!grown) goto done;
          level[c] = grown;
        }
        level[c][lsize[c]++] = j;
        if (c > top) top = c;
      }
      int32_t pick = -1;
      while (top > 0) {
        if (lsize[top] == 0) { --top; continue; }
        const int32_t j = level[top][--lsize[top]];
        const int32_t sj = state[j];
        if (sj >= 0 && (sj > B + 1 ? B + 1 : sj) == top) { pick = j; break; }
      }
      if (pick < 0) break;
      cur = pick;
    }
    for (int c = 0; c <= B + 1; ++c) lsize[c] = 0;
    for (int64_t t = 0; t < ntouched; ++t) {
      const int32_t j = touched[t];
      if (state[j] >= 0) {
        state[j] = 0;
        if (!sh->queued[j]) { sh->queued[j] = 1; fifo[fifo_tail++] = j; }
      }
    }
    if (m == B) { for (int i = 0; i < m; ++i) full[n_full + i] = cl[i]; n_full += m; }
    else { for (int i = 0; i < m; ++i) shorts[n_short + i] = cl[i]; n_short += m; }
  }
  sh->n_full[p] = n_full;
  sh->n_short[p] = n_short;
  rc = 0;
done:
  if (level) for (int c = 0; c <= B + 1; ++c) free(level[c]);
  free(level); free(lsize); free(lcap); free(touched); free(fifo); free(cl);
  return rc;
}

static void* co_worker(void* arg) {
  struct co_job* jb = (struct co_job*)arg;
  struct co_shared* sh = jb->sh;
  const int t = jb->tid, T = sh->nthreads;
  /* start gate: the barrier below counts nthreads participants, so nobody enters it before every thread exists
     (1: go; -1: a thread could not be created, leave) */
  for (;;) {
    const int g = __atomic_load_n(&sh->gate, __ATOMIC_ACQUIRE);
    if (g > 0) break;
    if (g < 0) return NULL;
    sched_yield();
  }
  /* ---- 1. regions: level-synchronous multi-source BFS */
  for (;;) {
    const int64_t nf = sh->nfront;
    if (nf == 0) break;
    const int64_t a = nf * t / T, b = nf * (t + 1) / T;
    for (int64_t i = a; i < b; ++i) {                       /* candidates: the lowest region that reaches a cell */
      const int32_t u = sh->frontier[i], ru = sh->region[u];
      for (int64_t e = sh->indptr[u]; e < sh->indptr[u + 1]; ++e) {
        const int32_t v = sh->indices[e];
        if (v < 0 || v >= sh->n || __atomic_load_n(&sh->region[v], __ATOMIC_RELAXED) >= 0) continue;
        int32_t c = __atomic_load_n(&sh->cand[v], __ATOMIC_RELAXED);
        while (ru < c && !__atomic_compare_exchange_n(&sh->cand[v], &c, ru, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
      }
    }
    pthread_barrier_wait(&sh->bar);
    sh->nnext[t] = 0;
    for (int64_t i = a; i < b && !sh->failed; ++i) {        /* claim: exactly one thread appends a reached cell */
      const int32_t u = sh->frontier[i];
      for (int64_t e = sh->indptr[u]; e < sh->indptr[u + 1]; ++e) {
        const int32_t v = sh->indices[e];
        if (v < 0 || v >= sh->n) continue;
        int32_t expect = -1;
        const int32_t c = sh->cand[v];
        if (c == 0x7fffffff) continue;
        if (__atomic_compare_exchange_n(&sh->region[v], &expect, c, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
          if (sh->nnext[t] == sh->capnext[t]) {
            sh->capnext[t] = sh->capnext[t] ? 2 * sh->capnext[t] : 4096;
            int32_t* g = (int32_t*)realloc(sh->next[t], 4 * (size_t)sh->capnext[t]);
            if (!g) { sh->failed = 1; break; }
            sh->next[t] = g;
          }
          sh->next[t][sh->nnext[t]++] = v;
        }
      }
    }
    pthread_barrier_wait(&sh->bar);
    if (t == 0) {                                           /* the next round's frontier (its order does not matter) */
      int64_t tot = 0;
      for (int k = 0; k < T; ++k) { if (sh->nnext[k]) memcpy(sh->frontier + tot, sh->next[k], 4 * (size_t)sh->nnext[k]); tot += sh->nnext[k]; }   /* (a thread that found nothing has no list yet: memcpy(_, NULL, 0) is undefined) */
      sh->nfront = sh->failed ? 0 : tot;
    }
    pthread_barrier_wait(&sh->bar);
  }
  pthread_barrier_wait(&sh->bar);
  if (t == 0 && !sh->failed) {                              /* leftovers by index; members of every region, ascending */
    const int64_t n = sh->n;
    for (int64_t i = 0; i < n; ++i) if (sh->region[i] < 0) sh->region[i] = (int32_t)(i * sh->P / n);
    for (int p = 0; p <= sh->P; ++p) sh->rstart[p] = 0;
    for (int64_t i = 0; i < n; ++i) sh->rstart[sh->region[i] + 1]++;
    for (int p = 0; p < sh->P; ++p) sh->rstart[p + 1] += sh->rstart[p];
    int64_t* cur = (int64_t*)malloc(8 * (size_t)sh->P);
    if (!cur) sh->failed = 1;
    else {
      memcpy(cur, sh->rstart, 8 * (size_t)sh->P);
      for (int64_t i = 0; i < n; ++i) sh->members[cur[sh->region[i]]++] = (int32_t)i;
      free(cur);
    }
  }
  pthread_barrier_wait(&sh->bar);
  /* ---- 2. greedy cluster growth, region by region (largest first would balance better; regions are similar) */
  while (!sh->failed) {
    const int64_t p = __atomic_fetch_add(&sh->next_region, 1, __ATOMIC_RELAXED);
    if (p >= sh->P) break;
    if (co_grow(sh, (int)p) != 0) sh->failed = 1;
  }
  return NULL;
}

int64_t cna_host_cluster_order_mt(int64_t n, const int64_t* indptr, const int32_t* indices, int B, int nthreads,
                                  int64_t* order_out) {
  if (n <= 0) return 0;
  int P = (int)(n / 65536);
  if (P > 64) P = 64;
  if (P < 2 || nthreads < 1) return cna_host_cluster_order(n, indptr, indices, B, order_out);
  if (B < 1) B = 1;
  if (nthreads > 64) nthreads = 64;
  struct co_shared sh;
  memset(&sh, 0, sizeof(sh));
  sh.n = n; sh.indptr = indptr; sh.indices = indices; sh.B = B; sh.P = P; sh.nthreads = nthreads;
  sh.region = (int32_t*)malloc(4 * (size_t)n);
  sh.cand = (int32_t*)malloc(4 * (size_t)n);
  sh.frontier = (int32_t*)malloc(4 * (size_t)n);
  sh.next = (int32_t**)calloc((size_t)nthreads, sizeof(int32_t*));
  sh.nnext = (int64_t*)calloc((size_t)nthreads, 8);
  sh.capnext = (int64_t*)calloc((size_t)nthreads, 8);
  sh.rstart = (int64_t*)calloc((size_t)P + 1, 8);
  sh.members = (int32_t*)malloc(4 * (size_t)n);
  sh.state = (int32_t*)calloc((size_t)n, 4);
  sh.queued = (unsigned char*)calloc((size_t)n, 1);
  sh.out_full = (int64_t*)malloc(8 * (size_t)n);
  sh.out_short = (int64_t*)malloc(8 * (size_t)n);
  sh.n_full = (int64_t*)calloc((size_t)P, 8);
  sh.n_short = (int64_t*)calloc((size_t)P, 8);
  int64_t rc = -1;
  pthread_t th[64];
  struct co_job jobs[64];
  int started = 0, bar_ok = 0, seq_fallback = 0;
  if (!sh.region || !sh.cand || !sh.frontier || !sh.next || !sh.nnext || !sh.capnext || !sh.rstart || !sh.members ||
      !sh.state || !sh.queued || !sh.out_full || !sh.out_short || !sh.n_full || !sh.n_short) goto done;
  for (int64_t i = 0; i < n; ++i) { sh.region[i] = -1; sh.cand[i] = 0x7fffffff; }
  for (int p = 0; p < P; ++p) {                              /* seeds: evenly spaced indices (distinct since n >= 65536 P) */
    const int64_t sidx = (int64_t)p * n / P;
    sh.region[sidx] = p;
    sh.frontier[p] = (int32_t)sidx;
  }
  sh.nfront = P;
  if (pthread_barrier_init(&sh.bar, NULL, (unsigned)nthreads) != 0) goto done;
  bar_ok = 1;
  for (int t = 0; t < nthreads; ++t) { jobs[t].sh = &sh; jobs[t].tid = t; }
  for (int t = 1; t < nthreads; ++t) {
    if (pthread_create(&th[t], NULL, co_worker, &jobs[t]) != 0) {
      /* cannot run with fewer participants than the barrier counts: the threads that exist are still at the start
         gate -- send them home and take the sequential path */
      __atomic_store_n(&sh.gate, -1, __ATOMIC_RELEASE);
      for (int k = 1; k < t; ++k) pthread_join(th[k], NULL);
      started = 0;
      seq_fallback = 1;
      goto done;
    }
    started = t;
  }
  __atomic_store_n(&sh.gate, 1, __ATOMIC_RELEASE);
  co_worker(&jobs[0]);
  for (int t = 1; t <= started; ++t) pthread_join(th[t], NULL);
  started = 0;
  if (sh.failed) goto done;
  {
    int64_t pos = 0, nf = 0;
    for (int p = 0; p < P; ++p) { memcpy(order_out + pos, sh.out_full + sh.rstart[p], 8 * (size_t)sh.n_full[p]); pos += sh.n_full[p]; }
    nf = pos;
    for (int p = 0; p < P; ++p) { memcpy(order_out + pos, sh.out_short + sh.rstart[p], 8 * (size_t)sh.n_short[p]); pos += sh.n_short[p]; }
    rc = pos == n ? nf : -1;
  }
done:
  if (bar_ok) pthread_barrier_destroy(&sh.bar);
  if (sh.next) for (int t = 0; t < nthreads; ++t) free(sh.next[t]);
  free(sh.next); free(sh.nnext); free(sh.capnext); free(sh.region); free(sh.cand); free(sh.frontier); free(sh.rstart);
  free(sh.members); free(sh.state); free(sh.queued); free(sh.out_full); free(sh.out_short); free(sh.n_full); free(sh.n_short);
  (void)seq_fallback;                            /* (sh.failed stays 0 on that path: the sequential order below) */
  if (rc < 0 && !sh.failed) return cna_host_cluster_order(n, indptr, indices, B, order_out);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Rows [r0, r1) of P A P^T for the device order: out row i = caller's row perm[r0 + i] with its columns
 * relabelled through col_map (caller's column -> device column) and LEFT IN THEIR ORIGINAL ORDER (no sum
 * is reordered).  indptr of A as int64; vbytes = 4 or 8 (float32 / float64 values, copied bit for bit).
 * out_indptr int64[r1 - r0 + 1] (rebased to 0), out_indices / out_data sized by the caller from
 * cna_host_permuted_nnz.  Threads split the rows; the result does not depend on the thread count. */
int64_t cna_host_permuted_nnz(const int64_t* perm, int64_t r0, int64_t r1, const int64_t* indptr) {
  int64_t s = 0;
  for (int64_t i = r0; i < r1; ++i) s += indptr[perm[i] + 1] - indptr[perm[i]];
  return s;
}

struct perm_job {
  const int64_t* perm; const int64_t* indptr; const int32_t* indices; const char* data; const int64_t* col_map;
  const int64_t* out_indptr; int32_t* out_indices; char* out_data;
  int64_t r0, row_a, row_b;
  int vbytes;
};

static void* perm_worker(void* arg) {
  struct perm_job* j = (struct perm_job*)arg;
  for (int64_t i = j->row_a; i < j->row_b; ++i) {
    const int64_t src = j->indptr[j->perm[j->r0 + i]];
    const int64_t len = j->indptr[j->perm[j->r0 + i] + 1] - src;
    const int64_t dst = j->out_indptr[i];
    for (int64_t e = 0; e < len; ++e) j->out_indices[dst + e] = (int32_t)j->col_map[j->indices[src + e]];
    memcpy(j->out_data + (size_t)dst * j->vbytes, j->data + (size_t)src * j->vbytes, (size_t)len * j->vbytes);
  }
  return NULL;
}

int cna_host_permute_rows(const int64_t* perm, int64_t r0, int64_t r1, const int64_t* indptr, const int32_t* indices,
                          const void* data, int vbytes, const int64_t* col_map, int64_t* out_indptr,
                          int32_t* out_indices, void* out_data, int nthreads) {
  const int64_t nr = r1 - r0;
  if (nr < 0 || (vbytes != 4 && vbytes != 8)) return -1;
  out_indptr[0] = 0;
  for (int64_t i = 0; i < nr; ++i) out_indptr[i + 1] = out_indptr[i] + (indptr[perm[r0 + i] + 1] - indptr[perm[r0 + i]]);
  if (nthreads > 64) nthreads = 64;
  if (nthreads < 1 || nr < 4096) nthreads = 1;
  struct perm_job jobs[64];
  pthread_t th[64];
  int started[64];
  const int64_t total = out_indptr[nr];
  int64_t row = 0;
  for (int t = 0; t < nthreads; ++t) {                 /* equal shares of the EDGES, cut at row boundaries */
    const int64_t want = total * (t + 1) / nthreads;
    int64_t b = row;
    if (t == nthreads - 1) b = nr;
    else while (b < nr && out_indptr[b] < want) ++b;
    jobs[t].perm = perm; jobs[t].indptr = indptr; jobs[t].indices = indices; jobs[t].data = (const char*)data;
    jobs[t].col_map = col_map; jobs[t].out_indptr = out_indptr; jobs[t].out_indices = out_indices;
    jobs[t].out_data = (char*)out_data; jobs[t].r0 = r0; jobs[t].row_a = row; jobs[t].row_b = b; jobs[t].vbytes = vbytes;
    row = b;
    started[t] = 0;
  }
  for (int t = 1; t < nthreads; ++t) started[t] = pthread_create(&th[t], NULL, perm_worker, &jobs[t]) == 0;
  perm_worker(&jobs[0]);
  for (int t = 1; t < nthreads; ++t) {
    if (started[t]) pthread_join(th[t], NULL);
    else perm_worker(&jobs[t]);
  }
  return 0;
}

/* Graph of the clusters of a cell order: order[c * B .. (c + 1) * B) are the cells of cluster c (the last one may be
 * short).  For every cluster the other clusters its cells have edges into, with the number of such edges:
 * ptr[nc + 1], then (col, cnt) lists.  Two calls: col == NULL counts (returns the number of entries and fills ptr),
 * the second fills.  Used by cna_amd._order.partition_order (blocks of cells for a sharded run).
 * -1: out of memory / bad arguments. */
int64_t cna_host_cluster_graph(int64_t n, const int64_t* indptr, const int32_t* indices, const int64_t* order, int B,
                               int64_t* ptr, int32_t* col, int64_t* cnt) {
  if (n < 0 || B < 1 || !indptr || !indices || !order || !ptr) return -1;
  const int64_t nc = (n + B - 1) / B;
  int32_t* cl = (int32_t*)malloc(4 * (size_t)(n > 0 ? n : 1));
  int64_t* acc = (int64_t*)calloc((size_t)(nc > 0 ? nc : 1), 8);
  int32_t* touched = (int32_t*)malloc(4 * (size_t)(nc > 0 ? nc : 1));
  if (!cl || !acc || !touched) { free(cl); free(acc); free(touched); return -1; }
  for (int64_t p = 0; p < n; ++p) cl[order[p]] = (int32_t)(p / B);
  int64_t total = 0;
  for (int64_t c = 0; c < nc; ++c) {
    int64_t nt = 0;
    const int64_t p1 = (c + 1) * B < n ? (c + 1) * B : n;
    for (int64_t p = c * B; p < p1; ++p) {
      const int64_t i = order[p];
      for (int64_t e = indptr[i]; e < indptr[i + 1]; ++e) {
        const int32_t j = indices[e];
        if (j < 0 || j >= n) continue;
        const int32_t d = cl[j];
        if (d == c) continue;
        if (acc[d]++ == 0) touched[nt++] = d;
      }
    }
    if (col) {
      if (ptr[c] != total) { free(cl); free(acc); free(touched); return -1; }
      for (int64_t k = 0; k < nt; ++k) { col[total + k] = touched[k]; cnt[total + k] = acc[touched[k]]; }
    } else {
      ptr[c] = total;
    }
    for (int64_t k = 0; k < nt; ++k) acc[touched[k]] = 0;
    total += nt;
  }
  if (!col) ptr[nc] = total;
  free(cl); free(acc); free(touched);
  return total;
}
