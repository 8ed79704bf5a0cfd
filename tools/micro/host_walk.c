/* Host-side planners of the walk-step PROBES under tools/micro/ (gather_lds.hip, walk_lds.hip, walk_tiles.hip): which
 * neighbour rows a block of destination rows shares, in the layouts those kernels stage through LDS.  The kernels were
 * measured and dropped (HISTORY.md 5), so this code left the product library in round 6 -- it is built into
 * tools/micro/libmicro_host.so by tools/micro/micro_host.py when a probe is re-run.  Integer work only; nothing here
 * computes a result of the path (reference: the sums of /root/reference/src/cna/tools/_nam.py:33 are never reordered). */
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Per block of B consecutive local rows: the distinct columns its rows reference ("sources", in order of
 * first appearance, at most `cap` per block) and, per edge, the position of its column in that list
 * (0xFFFF: the block's list was full -- the kernel fetches such a neighbour row from memory instead).
 *   indptr int64[n_local+1], indices int32 (device numbering, < n_cols)
 *   src_ptr int64[nblocks+1] out; src out, capacity nnz; slot uint16[nnz] out.
 * Returns the total number of sources, or -1. */
int64_t micro_block_sources(int64_t n_local, int64_t n_cols, const int64_t* indptr, const int32_t* indices, int B,
                               int cap, int64_t* src_ptr, int32_t* src, uint16_t* slot) {
  if (cap > 0xFFFE) cap = 0xFFFE;
  int32_t* stamp = (int32_t*)malloc(4 * (size_t)(n_cols > 0 ? n_cols : 1));    /* block that last listed the column */
  uint16_t* where = (uint16_t*)malloc(2 * (size_t)(n_cols > 0 ? n_cols : 1));
  if (!stamp || !where) return -1;
  memset(stamp, 0xff, 4 * (size_t)(n_cols > 0 ? n_cols : 1));
  const int64_t nblocks = (n_local + B - 1) / B;
  int64_t total = 0;
  for (int64_t b = 0; b < nblocks; ++b) {
    src_ptr[b] = total;
    int count = 0;
    const int64_t r1 = (b + 1) * B < n_local ? (b + 1) * B : n_local;
    for (int64_t e = indptr[b * B]; e < indptr[r1]; ++e) {
      const int32_t j = indices[e];
      if (stamp[j] == (int32_t)b) { slot[e] = where[j]; continue; }
      if (count < cap) {
        stamp[j] = (int32_t)b;
        where[j] = (uint16_t)count;
        slot[e] = (uint16_t)count;
        src[total + count] = j;
        ++count;
      } else {
        slot[e] = 0xFFFF;
      }
    }
    total += count;
  }
  src_ptr[nblocks] = total;
  free(stamp); free(where);
  return total;
}

/* ------------------------------------------------------------------------------------------------
 * Blocks of the LDS-staged walk step (csrc/walk_lds.hip): runs of consecutive device rows whose
 * neighbour rows ("sources") one workgroup stages in LDS.  A block takes rows while it has fewer than
 * `bmax` of them and its distinct sources still number at most `cap`; it never crosses a multiple of
 * `super` rows (so the blocks of one cluster of the device order stay together and the work splits over
 * threads without changing the result).  Per block the sources are listed in ascending order (adjacent
 * rows of the state are adjacent in memory: the staging loads coalesce), per edge `slot` is the position of
 * its column in the block's list (0xFFFF: a single row with more than `cap` distinct columns -- the kernel
 * reads those from memory).
 *   blk_row int64[<= n_local + 1], src_ptr int64[<= n_local + 1], src int32[<= nnz], slot uint16[nnz]
 * Returns the number of blocks (total sources = src_ptr[blocks]), or -1. */
struct wb_job {
  int64_t n_local, n_cols; const int64_t* indptr; const int32_t* indices; int bmax, cap, super;
  int64_t* cnt_blocks; int64_t* cnt_src;          /* per super-block (pass 1 out, pass 2 in as offsets) */
  int64_t* blk_row; int64_t* src_ptr; int32_t* src; uint16_t* slot;
  int64_t nsuper; int tid, nthreads, fill, failed;
};

static int cmp_i32(const void* a, const void* b) {
  const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}

static void* wb_worker(void* arg) {
  struct wb_job* j = (struct wb_job*)arg;
  const size_t nc = (size_t)(j->n_cols > 0 ? j->n_cols : 1);
  int64_t* stamp = (int64_t*)malloc(8 * nc);
  uint16_t* where = (uint16_t*)malloc(2 * nc);
  int32_t* list = (int32_t*)malloc(4 * (size_t)(j->cap + 1));
  if (!stamp || !where || !list) { j->failed = 1; free(stamp); free(where); free(list); return NULL; }
  memset(stamp, 0xff, 8 * nc);
  int64_t epoch = 0;
  for (int64_t s = j->tid; s < j->nsuper; s += j->nthreads) {
    const int64_t ra = s * j->super, rb = (s + 1) * j->super < j->n_local ? (s + 1) * j->super : j->n_local;
    int64_t nb = 0, ns = 0;
    int64_t bi = j->fill ? j->cnt_blocks[s] : 0, so = j->fill ? j->cnt_src[s] : 0;
    int64_t r = ra;
    while (r < rb) {
      ++epoch;
      int count = 0;
      int64_t r1 = r;
      while (r1 < rb && r1 - r < j->bmax) {
        int added = 0, ok = 1;
        for (int64_t e = j->indptr[r1]; e < j->indptr[r1 + 1]; ++e) {
          const int32_t c = j->indices[e];
          if (stamp[c] == epoch) continue;
          if (count + added >= j->cap) { ok = 0; break; }
          stamp[c] = epoch;
          list[count + added++] = c;
        }
        if (!ok && r1 > r) {                        /* the row does not fit any more: undo it, close the block */
          for (int a = 0; a < added; ++a) stamp[list[count + a]] = -1;
          break;
        }
        count += added;
        ++r1;
        if (!ok) break;                             /* a single row beyond the capacity: block of its own */
      }
      if (j->fill) {
        j->blk_row[bi] = r;
        j->src_ptr[bi] = so;
        qsort(list, (size_t)count, 4, cmp_i32);
        for (int a = 0; a < count; ++a) { where[list[a]] = (uint16_t)a; j->src[so + a] = list[a]; }
        for (int64_t e = j->indptr[r]; e < j->indptr[r1]; ++e) {
          const int32_t c = j->indices[e];
          j->slot[e] = stamp[c] == epoch ? where[c] : (uint16_t)0xFFFF;
        }
        ++bi;
        so += count;
      }
      ++nb;
      ns += count;
      r = r1;
    }
    if (!j->fill) { j->cnt_blocks[s] = nb; j->cnt_src[s] = ns; }
  }
  free(stamp); free(where); free(list);
  return NULL;
}

int64_t micro_walk_blocks(int64_t n_local, int64_t n_cols, const int64_t* indptr, const int32_t* indices, int bmax,
                             int cap, int super, int nthreads, int64_t* blk_row, int64_t* src_ptr, int32_t* src,
                             uint16_t* slot) {
  if (n_local <= 0) { blk_row[0] = 0; src_ptr[0] = 0; return 0; }
  if (bmax < 1) bmax = 1;
  if (cap < 1) cap = 1;
  if (cap > 0xFFFE) cap = 0xFFFE;
  if (super < bmax) super = bmax;
  const int64_t nsuper = (n_local + super - 1) / super;
  if (nthreads > 64) nthreads = 64;
  if ((int64_t)nthreads > nsuper) nthreads = (int)nsuper;
  if (nthreads < 1) nthreads = 1;
  int64_t* cb = (int64_t*)malloc(8 * (size_t)(nsuper + 1));
  int64_t* cs = (int64_t*)malloc(8 * (size_t)(nsuper + 1));
  if (!cb || !cs) { free(cb); free(cs); return -1; }
  struct wb_job jobs[64];
  pthread_t th[64];
  int started[64];
  int64_t nblocks = -1;
  for (int fill = 0; fill < 2; ++fill) {
    for (int t = 0; t < nthreads; ++t) {
      jobs[t] = (struct wb_job){n_local, n_cols, indptr, indices, bmax, cap, super, cb, cs, blk_row, src_ptr, src, slot,
                                nsuper, t, nthreads, fill, 0};
      started[t] = 0;
    }
    for (int t = 1; t < nthreads; ++t) started[t] = pthread_create(&th[t], NULL, wb_worker, &jobs[t]) == 0;
    wb_worker(&jobs[0]);
    for (int t = 1; t < nthreads; ++t) {
      if (started[t]) pthread_join(th[t], NULL);
      else wb_worker(&jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) if (jobs[t].failed) { free(cb); free(cs); return -1; }
    if (!fill) {                                    /* counts -> offsets */
      int64_t b = 0, s = 0;
      for (int64_t i = 0; i < nsuper; ++i) { const int64_t nb = cb[i], ns = cs[i]; cb[i] = b; cs[i] = s; b += nb; s += ns; }
      nblocks = b;
      blk_row[b] = n_local;
      src_ptr[b] = s;
    }
  }
  free(cb); free(cs);
  return nblocks;
}

/* ------------------------------------------------------------------------------------------------
 * Tile program of the LDS-tiled walk step (csrc/walk_lds.hip: k_walk_tiles).
 *
 * A workgroup owns a block of B = nw * rpw consecutive device rows (wave w: rows w, w + nw, w + 2 nw ... of the block) and
 * keeps their sums in registers.  The distinct neighbour rows ("sources") of the block are visited in
 * tiles of `S` sources IN ASCENDING ORDER OF THE CALLER'S COLUMN INDEX (key[c] = caller's index of device
 * column c).  When every CSR row lists its columns in ascending caller's index -- scipy's canonical form --
 * each row then meets its edges in its own CSR order, so the sums of _nam.py:33 are added in the reference's
 * sequence whatever the tiling.  Returns -2 when a row is not sorted that way (the caller uses the
 * row-gather kernel instead), -1 when out of memory.
 *
 * Outputs (nb = ceil(n_local / B) blocks).  Call with rec_pos == NULL first: only blk_tile is filled and
 * the number of tiles returned, which sizes tile_src0 and seg for the second call.
 *   blk_tile  int64[nb + 1]            first tile of every block
 *   tile_src0 int64[ntiles + 1]        start of a tile's source list in tile_src
 *   tile_src  int32[<= nnz]            device row of every source, tile after tile
 *   seg       int64[ntiles * nw + 1]   start of the records of (tile, wave); they end at the next entry
 *   rec_pos   int64[nnz]               position of CSR entry e in the record order (tile, wave, row, CSR order)
 *   rec_slot  uint16[nnz]              per record: index of its source inside its tile
 *   rec_row   uint8[nnz]               per record: row inside its wave
 * The records of a block occupy the positions of the block's CSR entries; the caller scatters the edge
 * weights through rec_pos. */
struct wt_job {
  int64_t n_local; const int64_t* indptr; const int32_t* indices; const int64_t* key; int nw, rpw, S;
  int64_t* blk_tile; int64_t* blk_src;            /* per block: tiles / sources (pass 1: counts, pass 2: offsets) */
  int64_t* tile_src0; int32_t* tile_src; int64_t* seg; int64_t* rec_pos; uint16_t* rec_slot; uint8_t* rec_row;
  int64_t nb; int tid, nthreads, fill, status;
};

struct wt_pair { int64_t key; int32_t col; };
static int cmp_pair(const void* a, const void* b) {
  const int64_t x = ((const struct wt_pair*)a)->key, y = ((const struct wt_pair*)b)->key;
  return (x > y) - (x < y);
}

static void* wt_worker(void* arg) {
  struct wt_job* j = (struct wt_job*)arg;
  const int B = j->nw * j->rpw;
  size_t cap = 1 << 14;
  struct wt_pair* pairs = (struct wt_pair*)malloc(cap * sizeof(struct wt_pair));
  int32_t* rank = (int32_t*)malloc(cap * 4);       /* per edge of the block: rank of its source */
  int64_t* cur = (int64_t*)malloc(8 * (size_t)B);
  if (!pairs || !rank || !cur) { j->status = -1; free(pairs); free(rank); free(cur); return NULL; }
  for (int64_t b = j->tid; b < j->nb && j->status == 0; b += j->nthreads) {
    const int64_t r0 = b * B, r1 = (b + 1) * B < j->n_local ? (b + 1) * B : j->n_local;
    const int64_t e0 = j->indptr[r0], e1 = j->indptr[r1];
    const size_t ne = (size_t)(e1 - e0);
    if (ne > cap) {
      cap = ne * 2;
      pairs = (struct wt_pair*)realloc(pairs, cap * sizeof(struct wt_pair));
      rank = (int32_t*)realloc(rank, cap * 4);
      if (!pairs || !rank) { j->status = -1; break; }
    }
    for (int64_t r = r0; r < r1; ++r) {            /* rows sorted by the caller's column index? */
      int64_t prev = -1;
      for (int64_t e = j->indptr[r]; e < j->indptr[r + 1]; ++e) {
        const int64_t k = j->key[j->indices[e]];
        if (k <= prev) { j->status = -2; break; }
        prev = k;
        pairs[e - e0].key = k;
        pairs[e - e0].col = (int32_t)(e - e0);     /* position of the edge, for the rank scatter below */
      }
      if (j->status) break;
    }
    if (j->status) break;
    qsort(pairs, ne, sizeof(struct wt_pair), cmp_pair);
    int64_t nsrc = 0, last = -1;
    for (size_t i = 0; i < ne; ++i) {
      if (pairs[i].key != last) { last = pairs[i].key; ++nsrc; }
      rank[pairs[i].col] = (int32_t)(nsrc - 1);
    }
    const int64_t nt = (nsrc + j->S - 1) / j->S;
    if (!j->fill) { j->blk_tile[b] = nt; j->blk_src[b] = nsrc; continue; }
    const int64_t T0 = j->blk_tile[b], S0 = j->blk_src[b];
    /* source lists: rank order = ascending caller's index */
    last = -1;
    int64_t s = 0;
    for (size_t i = 0; i < ne; ++i) {
      if (pairs[i].key != last) {
        last = pairs[i].key;
        j->tile_src[S0 + s] = j->indices[e0 + pairs[i].col];
        ++s;
      }
    }
    for (int64_t t = 0; t < nt; ++t) j->tile_src0[T0 + t] = S0 + t * j->S;
    /* records: (tile, wave, row, CSR order) */
    for (int64_t r = r0; r < r1; ++r) cur[r - r0] = j->indptr[r];
    int64_t pos = e0;
    for (int64_t t = 0; t < nt; ++t) {
      const int32_t hi = (int32_t)((t + 1) * j->S);
      for (int w = 0; w < j->nw; ++w) {
        j->seg[(T0 + t) * j->nw + w] = pos;
        for (int i = 0; i < j->rpw; ++i) {
          const int64_t r = r0 + (int64_t)i * j->nw + w;            /* rows dealt round-robin over the waves: neighbouring rows share
                                                                       neighbours, so their edges fall into the same tiles */
          if (r >= r1) break;
          int64_t e = cur[r - r0];
          const int64_t end = j->indptr[r + 1];
          while (e < end && rank[e - e0] < hi) {
            j->rec_pos[e] = pos;
            j->rec_slot[pos] = (uint16_t)(rank[e - e0] - (int32_t)(t * j->S));
            j->rec_row[pos] = (uint8_t)i;
            ++pos; ++e;
          }
          cur[r - r0] = e;
        }
      }
    }
  }
  free(pairs); free(rank); free(cur);
  return NULL;
}

int64_t micro_walk_tiles(int64_t n_local, const int64_t* indptr, const int32_t* indices, const int64_t* key, int nw,
                            int rpw, int S, int nthreads, int64_t* blk_tile, int64_t* tile_src0, int32_t* tile_src,
                            int64_t* seg, int64_t* rec_pos, uint16_t* rec_slot, uint8_t* rec_row) {
  if (nw < 1 || rpw < 1 || rpw > 255 || S < 1 || S > 65535) return -1;
  const int B = nw * rpw;
  const int64_t nb = n_local > 0 ? (n_local + B - 1) / B : 0;
  if (nb == 0) { blk_tile[0] = 0; return 0; }
  if (nthreads > 64) nthreads = 64;
  if ((int64_t)nthreads > nb) nthreads = (int)nb;
  if (nthreads < 1) nthreads = 1;
  int64_t* blk_src = (int64_t*)malloc(8 * (size_t)(nb + 1));
  if (!blk_src) return -1;
  struct wt_job jobs[64];
  pthread_t th[64];
  int started[64];
  int64_t ntiles = 0;
  const int passes = rec_pos ? 2 : 1;
  for (int fill = 0; fill < passes; ++fill) {
    for (int t = 0; t < nthreads; ++t) {
      jobs[t] = (struct wt_job){n_local, indptr, indices, key, nw, rpw, S, blk_tile, blk_src, tile_src0, tile_src, seg,
                                rec_pos, rec_slot, rec_row, nb, t, nthreads, fill, 0};
      started[t] = 0;
    }
    for (int t = 1; t < nthreads; ++t) started[t] = pthread_create(&th[t], NULL, wt_worker, &jobs[t]) == 0;
    wt_worker(&jobs[0]);
    for (int t = 1; t < nthreads; ++t) {
      if (started[t]) pthread_join(th[t], NULL);
      else wt_worker(&jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) if (jobs[t].status) { const int64_t st = jobs[t].status; free(blk_src); return st; }
    if (!fill) {
      int64_t tt = 0, ss = 0;
      for (int64_t b = 0; b < nb; ++b) { const int64_t a = blk_tile[b], c = blk_src[b]; blk_tile[b] = tt; blk_src[b] = ss; tt += a; ss += c; }
      blk_tile[nb] = tt;
      ntiles = tt;
      if (passes == 2) { tile_src0[tt] = ss; seg[tt * nw] = indptr[n_local]; }
    }
  }
  free(blk_src);
  return ntiles;
}

