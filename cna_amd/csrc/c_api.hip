// extern "C" entry points of libcna_hip.so (see include/cna_hip.h for the contract).
#include "common.h"
#include <sched.h>
#include <chrono>
#include <cmath>
#include <cstring>

int comm_destroy(cna_ctx* c);

static thread_local std::string g_err;
void cna_set_error(const std::string& msg) { g_err = msg; }

// ------------------------------------------------------------------ memory / profiling
// Two host threads may use one context at a time (the association's helper thread conditions the
// phenotypes -- its own buffers, the second stream -- while the main thread launches kernels): the
// allocator's bookkeeping and the alloc / free calls themselves are serialised by c->alloc_mu.  Each
// buffer still has one owner: the helper only ever (re)allocates zc and gt.
int dev_alloc(cna_ctx* c, void** p, size_t bytes) {
  std::lock_guard<std::recursive_mutex> lk(c->alloc_mu);
  *p = nullptr;
  if (bytes == 0) bytes = 256;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    cna_set_error(std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    return CNA_ENOMEM;
  }
  c->dev_bytes += (int64_t)bytes;
  return 0;
}
int dev_free(cna_ctx* c, void* p, size_t bytes) {
  std::lock_guard<std::recursive_mutex> lk(c->alloc_mu);
  if (p) {
    (void)hipFree(p);
    c->dev_bytes -= (int64_t)(bytes ? bytes : 256);
  }
  return 0;
}
static void halo_clear(cna_ctx* c);

int dev_reserve(cna_ctx* c, void** p, int64_t* cap, int64_t need) {
  std::lock_guard<std::recursive_mutex> lk(c->alloc_mu);
  if (need <= *cap && *p) return 0;
  if (*p) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    dev_free(c, *p, (size_t)*cap);
    *p = nullptr;
    *cap = 0;
  }
  if (need < 256) need = 256;
  CNA_TRY(dev_alloc(c, p, (size_t)need));
  *cap = need;
  return 0;
}

static hipEvent_t ev_get(cna_ctx* c) {
  if (!c->ev_pool.empty()) {
    hipEvent_t e = c->ev_pool.back();
    c->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
void prof_begin(cna_ctx* c, int kid, hipStream_t st) {
  std::lock_guard<std::mutex> lock(c->prof_mu);
  ProfSpan s;
  s.kid = kid;
  s.a = ev_get(c);
  s.b = ev_get(c);
  (void)hipEventRecord(s.a, st);
  c->prof_pending.push_back(s);
}
void prof_end(cna_ctx* c, int kid, hipStream_t st) {
  std::lock_guard<std::mutex> lock(c->prof_mu);
  for (size_t i = c->prof_pending.size(); i-- > 0;) {
    if (c->prof_pending[i].kid == kid) {
      (void)hipEventRecord(c->prof_pending[i].b, st);
      return;
    }
  }
}
static void prof_flush(cna_ctx* c) {
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->copy_stream);
  if (c->coef_stream) (void)hipStreamSynchronize(c->coef_stream);
  if (c->gram_stream) (void)hipStreamSynchronize(c->gram_stream);
  if (c->halo_stream) (void)hipStreamSynchronize(c->halo_stream);
  std::lock_guard<std::mutex> lock(c->prof_mu);
  if (c->prof_pending.empty()) return;
  for (auto& s : c->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      c->prof_ms[s.kid] += ms;
      c->prof_n[s.kid] += 1;
    }
    c->ev_pool.push_back(s.a);
    c->ev_pool.push_back(s.b);
  }
  c->prof_pending.clear();
}

static const char* kKernelNames[CNA_K_COUNT] = {
    "colsum", "nam_first", "nam_step", "batch_kurtosis", "zero_variance", "select", "resid_xb",
    "standardize", "gram", "gram_reduce", "ncorrs", "null_local", "obs_counts", "percell_fdr",
    "project_xb", "transpose", "rccl", "condition", "global_test", "nam_step_sparse", "halo_exchange", "halo_wait"};

#define CHECK_CTX(c)                                        \
  do {                                                      \
    if (!(c)) CNA_FAIL(CNA_EINVAL, "null context");         \
    HIP_TRY(hipSetDevice((c)->device));                     \
  } while (0)

// entry points that read or replace the NAM collect a pending walk's verdict first
#define AUTO_FINISH(c)                                                   \
  do {                                                                   \
    if ((c)->auto_pending) CNA_TRY(cna_nam_auto_finish((c), nullptr, nullptr)); \
  } while (0)

// scratch layout helper: carve 256-byte aligned pieces out of c->scratch
struct Carver {
  char* base;
  int64_t off = 0;
  explicit Carver(void* p) : base((char*)p) {}
  template <typename T>
  T* take(int64_t count) {
    T* r = (T*)(base + off);
    off += round_up64((int64_t)sizeof(T) * count, 256);
    return r;
  }
};
static int64_t carve_bytes(std::initializer_list<int64_t> sizes) {
  int64_t t = 0;
  for (auto s : sizes) t += round_up64(s, 256);
  return t;
}


// concatenate count_local doubles from every rank (rank order) into a host buffer
static int ragged_gather(cna_ctx* c, const double* src_dev, int64_t count_local, double* out, int64_t n_expected) {
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, 256 * 2 + 8 * c->nranks));
  int64_t* cnt_dev = (int64_t*)c->scratch;
  std::vector<int64_t> cnt(c->nranks, 0);
  cnt[c->rank] = count_local;
  HIP_TRY(hipMemcpyAsync(cnt_dev, cnt.data(), 8 * c->nranks, hipMemcpyHostToDevice, c->stream));
  CNA_TRY(comm_allreduce_i64_sum(c, cnt_dev, c->nranks));
  HIP_TRY(hipMemcpyAsync(cnt.data(), cnt_dev, 8 * c->nranks, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  int64_t mx = 1, tot = 0;
  for (auto v : cnt) { mx = std::max(mx, v); tot += v; }
  if (tot != n_expected) CNA_FAIL(CNA_EINVAL, "ragged gather: expected total does not match the ranks' counts");
  CNA_TRY(dev_reserve(c, &c->scratch2, &c->scratch2_cap, 8 * mx * c->nranks));
  double* all = (double*)c->scratch2;
  if (count_local > 0)
    HIP_TRY(hipMemcpyAsync(all + mx * c->rank, src_dev, 8 * count_local, hipMemcpyDeviceToDevice, c->stream));
  CNA_TRY(comm_allgather_bytes(c, all + mx * c->rank, all, 8 * mx));
  std::vector<double> host(mx * c->nranks);
  HIP_TRY(hipMemcpyAsync(host.data(), all, 8 * mx * c->nranks, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  int64_t o = 0;
  for (int r = 0; r < c->nranks; ++r) {
    std::memcpy(out + o, host.data() + mx * r, 8 * cnt[r]);
    o += cnt[r];
  }
  return 0;
}

// the main stream waits for an exchange still in flight on the halo stream (see cna_nam_step); no-op when none is
int halo_settle(cna_ctx* c) {
  if (!c->halo_wait_pending) return 0;
  c->halo_wait_pending = false;
  // (profiling: how long the main stream stands still here -- the part of the exchange the walk did not hide)
  ProfScope ps(c, CNA_K_HALO_WAIT);
  HIP_TRY(hipStreamWaitEvent(c->stream, c->halo_e2, 0));
  return 0;
}

extern "C" {

const char* cna_last_error(void) { return g_err.c_str(); }
int cna_abi_version(void) { return 1; }
const char* cna_kernel_name(int k) { return (k >= 0 && k < CNA_K_COUNT) ? kKernelNames[k] : "?"; }

int cna_device_count(int* count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    cna_set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return (int)e;
  }
  *count = n;
  return 0;
}

int cna_ctx_create(int device, cna_ctx** out) {
  if (!out) CNA_FAIL(CNA_EINVAL, "null out pointer");
  *out = nullptr;
  int n = 0;
  CNA_TRY(cna_device_count(&n));
  if (n <= 0) CNA_FAIL(CNA_EINVAL, "no HIP device visible: cna_amd has no CPU fallback");
  if (device < 0 || device >= n) CNA_FAIL(CNA_EINVAL, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  cna_ctx* c = new cna_ctx();
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    // the second stream carries short sample-space kernels (F-tests, conditioning) that should slip in
    // between the workgroup rounds of a long kernel on the main stream: highest priority
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    e = hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, hi);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->coef_stream, hipStreamNonBlocking);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->gram_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->coef_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->gt_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->scal_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->stage_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->coef_copied, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->null_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->bins_copied, hipEventDisableTiming);
  if (e != hipSuccess) {
    delete c;
    cna_set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e));
    return (int)e;
  }
  *out = c;
  return 0;
}

int cna_ctx_destroy(cna_ctx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->gram_stream) (void)hipStreamSynchronize(c->gram_stream);
  if (c->auto_state) { (void)hipFree(c->auto_state); c->auto_state = nullptr; }
  if (c->byp_buf) { (void)hipFree(c->byp_buf); c->byp_buf = nullptr; }
  if (c->pair_buf) { (void)hipFree(c->pair_buf); c->pair_buf = nullptr; }
  prof_flush(c);
  comm_destroy(c);
  void* bufs[] = {c->halo_rows_safe, c->halo_rows_need, c->idx_t, c->i8_buf, c->xq, c->xq_scale, c->coef_dev, c->proj, c->sp_pair, c->sp_cnt, c->null_part, c->rp16_buf, c->halo_send_idx, c->halo_recv_idx, c->halo_rows_b, c->halo_rows_i, c->halo_sbuf, c->halo_rbuf, c->orig_idx, c->indptr, c->indices, c->data, c->colsum, c->sid, c->counts, c->T[0], c->T[1], c->dense_s,
                  c->nam, c->X, c->X2, c->resid_f, c->keep_store, c->stat, c->ncorrs, c->scratch, c->scratch2, c->cellinfo, c->zc, c->gt, c->gram_tiles_ptr, c->gram_buf, c->gram_part, c->bins_dev};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(c->stream);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->coef_stream) (void)hipStreamDestroy(c->coef_stream);
  if (c->gram_done) (void)hipEventDestroy(c->gram_done);
  if (c->coef_ready) (void)hipEventDestroy(c->coef_ready);
  if (c->gt_done) (void)hipEventDestroy(c->gt_done);
  if (c->scal_ready) (void)hipEventDestroy(c->scal_ready);
  if (c->stage_done) (void)hipEventDestroy(c->stage_done);
  if (c->h_gt) (void)hipHostFree(c->h_gt);
  if (c->coef_copied) (void)hipEventDestroy(c->coef_copied);
  if (c->null_done) (void)hipEventDestroy(c->null_done);
  if (c->bins_copied) (void)hipEventDestroy(c->bins_copied);
  if (c->gram_stream) (void)hipStreamDestroy(c->gram_stream);
  for (hipEvent_t e : {c->gram_pre_done, c->range_done})
    if (e) (void)hipEventDestroy(e);
  if (c->halo_stream) (void)hipStreamDestroy(c->halo_stream);
  if (c->halo_e1) (void)hipEventDestroy(c->halo_e1);
  if (c->halo_e2) (void)hipEventDestroy(c->halo_e2);
  if (c->h_bins) (void)hipHostFree(c->h_bins);
  if (c->h_tab) (void)hipHostFree(c->h_tab);
  if (c->h_res) (void)hipHostFree(c->h_res);
  if (c->h_scal) (void)hipHostFree(c->h_scal);
  if (c->h_gram) (void)hipHostFree(c->h_gram);
  if (c->h_hint) (void)hipHostFree(c->h_hint);
  if (c->h_cell) (void)hipHostFree(c->h_cell);
  delete c;
  return 0;
}

int cna_ctx_sync(cna_ctx* c) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  CNA_TRY(halo_settle(c));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->gram_stream && c->gram_pre_pending) HIP_TRY(hipStreamSynchronize(c->gram_stream));
  return 0;
}

int cna_ctx_device_bytes(cna_ctx* c, int64_t* bytes) {
  if (!c || !bytes) CNA_FAIL(CNA_EINVAL, "null argument");
  *bytes = c->dev_bytes.load();
  return 0;
}

int cna_set_state_f32(cna_ctx* c, int on) {
  CHECK_CTX(c);
  // takes effect at the next step that writes the state: every buffer carries its own format (t_f32), so a walk in
  // progress stays consistent; the caller decides what a NAM already on the device is worth (engine.set_state_f32)
  c->state_f32_mode = on != 0;
  return 0;
}

// ------------------------------------------------------------------------------ graph
int cna_graph_upload(cna_ctx* c, int64_t n_global, int64_t row0, int64_t n_local, const int64_t* indptr,
                     const int32_t* indices, const void* data, int data_is_f64) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (n_global <= 0 || n_local < 0 || row0 < 0 || row0 + n_local > n_global || !indptr)
    CNA_FAIL(CNA_EINVAL, "cna_graph_upload: bad shape");
  const int64_t rpr = (n_global + c->nranks - 1) / c->nranks;
  if (row0 != (int64_t)c->rank * rpr && !(n_local == 0))
    CNA_FAIL(CNA_EINVAL, "cna_graph_upload: row0 must be rank*ceil(n/nranks)");
  const int64_t want_local = std::max<int64_t>(0, std::min(rpr, n_global - (int64_t)c->rank * rpr));
  if (n_local != want_local) CNA_FAIL(CNA_EINVAL, "cna_graph_upload: n_local must be the rank's block size");
  if (indptr[0] != 0) CNA_FAIL(CNA_EINVAL, "cna_graph_upload: indptr must be rebased to start at 0");
  const int64_t nnz = indptr[n_local];
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->indptr) dev_free(c, c->indptr, sizeof(int64_t) * (c->n_local + 1));
  if (c->indices) dev_free(c, c->indices, sizeof(int32_t) * c->nnz);
  if (c->data) dev_free(c, c->data, (c->data_f64 ? 8 : 4) * c->nnz);
  if (c->colsum) dev_free(c, c->colsum, sizeof(double) * c->n_pad);
  if (c->stat) dev_free(c, c->stat, sizeof(double) * c->n_pad);
  if (c->orig_idx) dev_free(c, c->orig_idx, sizeof(int64_t) * c->n_local);
  c->indptr = nullptr; c->indices = nullptr; c->data = nullptr; c->colsum = nullptr; c->stat = nullptr;
  c->orig_idx = nullptr;
  halo_clear(c);
  c->n_global = n_global;
  c->row0 = (int64_t)c->rank * rpr;
  c->n_local = n_local;
  c->rows_per_rank = rpr;
  c->n_pad = rpr * c->nranks;
  c->nnz = nnz;
  c->data_f64 = data_is_f64 ? 1 : 0;
  const size_t vb = data_is_f64 ? 8 : 4;
  CNA_TRY(dev_alloc(c, (void**)&c->indptr, sizeof(int64_t) * (n_local + 1)));
  CNA_TRY(dev_alloc(c, (void**)&c->indices, sizeof(int32_t) * nnz));
  CNA_TRY(dev_alloc(c, (void**)&c->data, vb * nnz));
  CNA_TRY(dev_alloc(c, (void**)&c->colsum, sizeof(double) * c->n_pad));
  CNA_TRY(dev_alloc(c, (void**)&c->stat, sizeof(double) * c->n_pad));
  HIP_TRY(hipMemcpy(c->indptr, indptr, sizeof(int64_t) * (n_local + 1), hipMemcpyHostToDevice));
  if (nnz > 0) {
    HIP_TRY(hipMemcpy(c->indices, indices, sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->data, data, vb * nnz, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemset(c->stat, 0, sizeof(double) * c->n_pad));
  c->have_colsum = false;
  c->cellinfo_valid = false;
  c->t_valid = false;
  c->nam_valid = false; c->nam_lazy = false;
  c->x_valid = false;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  c->ncorrs_valid = false;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  c->coef_early = false;
  c->fdr_inline = false;
  c->steps_done = 0;
  return 0;
}

int cna_set_cell_order(cna_ctx* c, const int64_t* orig_index) {
  CHECK_CTX(c);
  if (!c->indptr) CNA_FAIL(CNA_ESTATE, "cna_set_cell_order before cna_graph_upload");
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (!orig_index) {
    if (c->orig_idx) dev_free(c, c->orig_idx, sizeof(int64_t) * c->n_local);
    c->orig_idx = nullptr;
    return 0;
  }
  const int64_t lim = c->local_view ? c->n_local : c->n_global;
  for (int64_t i = 0; i < c->n_local; ++i)
    if (orig_index[i] < 0 || orig_index[i] >= lim)
      CNA_FAIL(CNA_EINVAL, "cna_set_cell_order: index out of range");
  if (!c->orig_idx && c->n_local > 0) CNA_TRY(dev_alloc(c, (void**)&c->orig_idx, sizeof(int64_t) * c->n_local));
  if (c->n_local > 0)
    HIP_TRY(hipMemcpy(c->orig_idx, orig_index, sizeof(int64_t) * c->n_local, hipMemcpyHostToDevice));
  return 0;
}

int cna_graph_reorder(cna_ctx* c, const int64_t* perm) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (!c->indptr || !perm) CNA_FAIL(CNA_ESTATE, "cna_graph_reorder before cna_graph_upload");
  if (c->nranks != 1 || c->halo_on || c->local_view || c->n_local != c->n_global || c->n_pad != c->n_global)
    CNA_FAIL(CNA_ESTATE, "cna_graph_reorder: one rank holding the whole graph only");
  if (c->orig_idx) CNA_FAIL(CNA_ESTATE, "cna_graph_reorder: the resident graph already has a device cell order");
  CNA_TRY(halo_settle(c));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->gram_stream && c->gram_pre_pending) HIP_TRY(hipStreamSynchronize(c->gram_stream));
  const int64_t n = c->n_global;
  int64_t* pd = nullptr;
  CNA_TRY(dev_alloc(c, (void**)&pd, sizeof(int64_t) * n));
  hipError_t e = hipMemcpyAsync(pd, perm, sizeof(int64_t) * n, hipMemcpyHostToDevice, c->stream);
  int rc = e == hipSuccess ? graph_reorder_device(c, pd) : (int)e;
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    dev_free(c, pd, sizeof(int64_t) * n);
    if (e != hipSuccess) cna_set_error(hipGetErrorString(e));
    return rc;
  }
  c->orig_idx = pd;                       // (what cna_set_cell_order would have uploaded)
  HIP_TRY(hipMemsetAsync(c->stat, 0, sizeof(double) * c->n_pad, c->stream));
  // everything that hangs on the order of the cells goes, as after an upload; the column sums and the sample codes were
  // permuted with the graph and stay
  c->cellinfo_valid = false;
  c->t_valid = false;
  c->nam_valid = false; c->nam_lazy = false;
  c->x_valid = false;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  c->ncorrs_valid = false;
  c->coef_early = false;
  c->fdr_inline = false;
  c->steps_done = 0;
  c->t_f32[0] = c->t_f32[1] = false;
  return 0;
}

int cna_set_local_view(cna_ctx* c, int on) {
  CHECK_CTX(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  if ((on != 0) != c->local_view && c->orig_idx) {      // a cell order belongs to the view it was given in
    dev_free(c, c->orig_idx, sizeof(int64_t) * c->n_local);
    c->orig_idx = nullptr;
  }
  c->local_view = on != 0;
  return 0;
}

int cna_colsums(cna_ctx* c, double self_weight) {
  CHECK_CTX(c);
  if (!c->indptr) CNA_FAIL(CNA_ESTATE, "cna_colsums before cna_graph_upload");
  c->self_weight = self_weight;
  CNA_TRY(launch_colsum(c));
  CNA_TRY(comm_allreduce_f64_sum(c, c->colsum, (size_t)c->n_pad));
  CNA_TRY(launch_add_scalar(c, c->colsum, c->n_global, self_weight));
  c->have_colsum = true;
  c->cellinfo_valid = false;
  return 0;
}

int cna_fetch_colsums(cna_ctx* c, double* out) {
  CHECK_CTX(c);
  if (!c->have_colsum) CNA_FAIL(CNA_ESTATE, "colsums not computed");
  HIP_TRY(hipMemcpyAsync(out, c->colsum, sizeof(double) * c->n_global, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

static int64_t state_rows(const cna_ctx* c) { return c->t_compact ? c->t_rows : c->n_pad; }

static int ensure_T(cna_ctx* c, int ld) {
  // + 64 doubles: the gather kernel lets lanes past the row width read into the following row
  const int64_t need = (int64_t)sizeof(double) * (state_rows(c) * ld + 64);
  // (a buffer left over from a much larger row space -- the graph's halo plan came after a first allocation -- goes:
  // the point of the compact row space is the memory)
  if (need > c->t_cap || !c->T[0] || (c->t_compact && c->t_cap > need + need / 2 + (1 << 20))) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->halo_wait_pending && c->halo_stream) { HIP_TRY(hipStreamSynchronize(c->halo_stream)); c->halo_wait_pending = false; }
    for (int i = 0; i < 2; ++i) {
      if (c->T[i]) dev_free(c, c->T[i], (size_t)c->t_cap);
      c->T[i] = nullptr;
    }
    for (int i = 0; i < 2; ++i) {
      CNA_TRY(dev_alloc(c, (void**)&c->T[i], (size_t)need));
      HIP_TRY(hipMemsetAsync(c->T[i], 0, (size_t)need, c->stream));
    }
    c->t_cap = need;
  }
  return 0;
}

// The second walk step can gather a compressed copy of the state (diffuse.hip: k_nam_step_sparse).
// Worth it when rows are mostly zeros after one step, i.e. many more samples than neighbours.  Sharded over
// ranks (round 3): the exchange between ranks still carries dense rows -- the first step then writes the dense row
// of every cell as well, and the rows that arrive from other ranks are marked "dense" in sp_cnt once (this rank's
// steps only ever write the marks of its own rows), so the second step takes its local neighbours -- nine in ten
// at eight ranks -- from their pairs and the others from the dense rows the halo exchange delivered, exactly as it
// does for a row that overflowed its pairs.  Needs the halo exchange (with the all-gather fallback every row would
// be foreign).  CNA_SPARSE_MIN_N moves the switch-over (0 = never).
static int ensure_sparse_state(cna_ctx* c) {
  int min_n = 96;
  if (const char* e = getenv("CNA_SPARSE_MIN_N")) min_n = atoi(e);
  const bool multi = c->nranks > 1 || comm_active(c);
  const bool want = min_n > 0 && c->N >= min_n && (!multi || c->halo_on);
  if (!want) {
    if (c->sp_cnt) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      dev_free(c, c->sp_pair, 16 * 64 * (size_t)c->sp_pair_rows);
      dev_free(c, c->sp_cnt, (size_t)c->sp_rows);
      c->sp_pair = c->sp_cnt = nullptr;
      c->sp_rows = c->sp_pair_rows = 0;
    }
    return 0;
  }
  // compact row space: counts for every row of the state (halo rows stay "dense"), pairs for this rank's own rows only
  const int64_t cnt_rows = state_rows(c), pair_rows = c->t_compact ? std::max<int64_t>(c->n_local, 1) : c->n_pad;
  if (c->sp_cnt && c->sp_rows == cnt_rows && c->sp_pair_rows == pair_rows) return 0;
  if (c->sp_cnt) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    dev_free(c, c->sp_pair, 16 * 64 * (size_t)c->sp_pair_rows);
    dev_free(c, c->sp_cnt, (size_t)c->sp_rows);
    c->sp_pair = c->sp_cnt = nullptr;
  }
  c->sp_rows = cnt_rows;
  c->sp_pair_rows = pair_rows;
  CNA_TRY(dev_alloc(c, &c->sp_pair, 16 * 64 * (size_t)c->sp_pair_rows));   // 64 = SP_CAP records of 16 bytes (diffuse.hip)
  CNA_TRY(dev_alloc(c, &c->sp_cnt, (size_t)c->sp_rows));
  // every row "dense" (255 = SP_DENSE) until a first step of this rank says otherwise: rows of other ranks stay so
  HIP_TRY(hipMemsetAsync(c->sp_cnt, 0xff, (size_t)c->sp_rows, c->stream));
  return 0;
}

// -------------------------------------------------------------------------------- NAM
int cna_set_samples(cna_ctx* c, const int32_t* codes, int n_samples, const double* counts) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (!c->indptr) CNA_FAIL(CNA_ESTATE, "cna_set_samples before cna_graph_upload");
  if (n_samples < 1 || n_samples > 1024) CNA_FAIL(CNA_EINVAL, "n_samples must be in [1, 1024]");
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (!c->sid || !c->counts || c->N != n_samples || c->sid_n != c->n_global) {
    if (c->sid) dev_free(c, c->sid, sizeof(int32_t) * c->sid_n);
    if (c->counts) dev_free(c, c->counts, sizeof(double) * c->N);
    c->sid = nullptr; c->counts = nullptr;
    CNA_TRY(dev_alloc(c, (void**)&c->sid, sizeof(int32_t) * c->n_global));
    CNA_TRY(dev_alloc(c, (void**)&c->counts, sizeof(double) * n_samples));
    c->sid_n = c->n_global;
  }
  c->N = n_samples;
  int ld_align = 4;
  c->ld = round_up(n_samples, ld_align);
  HIP_TRY(hipMemcpyAsync(c->sid, codes, sizeof(int32_t) * c->n_global, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->counts, counts, sizeof(double) * n_samples, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->cellinfo_valid = false;
  CNA_TRY(ensure_T(c, c->ld));
  void* nm = c->nam;
  CNA_TRY(dev_reserve(c, &nm, &c->nam_cap, (int64_t)sizeof(double) * std::max<int64_t>(c->n_local, 1) * c->ld));
  c->nam = (double*)nm;
  c->t_width = c->N;
  c->t_ld = c->ld;
  c->t_cur = 0;
  c->t_f32[0] = c->t_f32[1] = false;
  c->steps_done = 0;
  c->t_valid = false;
  c->nam_valid = false; c->nam_lazy = false;
  c->x_valid = false;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  CNA_TRY(ensure_sparse_state(c));
  return 0;
}

int cna_restart_nam(cna_ctx* c) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (!c->sid || !c->counts) CNA_FAIL(CNA_ESTATE, "cna_restart_nam needs cna_set_samples first");
  // (no wait for the stream here: the new walk is queued behind whatever still runs on it, and ensure_T waits by itself
  // in the one case that needs it, a reallocation -- the wait cost 20-40 us at the front of every call)
  CNA_TRY(ensure_T(c, c->ld));
  c->t_width = c->N;
  c->t_ld = c->ld;
  c->t_cur = 0;
  c->t_f32[0] = c->t_f32[1] = false;
  c->steps_done = 0;
  c->t_valid = false;
  c->nam_valid = false; c->nam_lazy = false;
  c->x_valid = false;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  return 0;
}

static void halo_clear(cna_ctx* c) {
  if (c->halo_wait_pending && c->halo_stream) (void)hipStreamSynchronize(c->halo_stream);     // (nothing may still write into buffers about to go)
  c->halo_wait_pending = false;
  if (c->halo_rows_safe) dev_free(c, c->halo_rows_safe, sizeof(int32_t) * std::max<int64_t>(c->halo_nsafe, 1));
  if (c->halo_rows_need) dev_free(c, c->halo_rows_need, sizeof(int32_t) * std::max<int64_t>(c->halo_nneed, 1));
  c->halo_rows_safe = c->halo_rows_need = nullptr;
  c->halo_nsafe = c->halo_nneed = 0;
  if (c->idx_t) dev_free(c, c->idx_t, sizeof(int32_t) * std::max<int64_t>(c->idx_t_n, 1));
  c->idx_t = nullptr;
  c->idx_t_n = 0;
  if (c->t_compact) {          // the state's row space changes: whatever is in it is void
    c->t_compact = false;
    c->t_rows = 0;
    c->t_valid = false;
    c->cellinfo_valid = false;
  }
  if (c->halo_send_idx) dev_free(c, c->halo_send_idx, sizeof(int64_t) * std::max<int64_t>(c->halo_ns, 1));
  if (c->halo_recv_idx) dev_free(c, c->halo_recv_idx, sizeof(int64_t) * std::max<int64_t>(c->halo_nr, 1));
  c->halo_send_idx = c->halo_recv_idx = nullptr;
  c->halo_ns = c->halo_nr = 0;
  c->halo_on = false;
  if (c->halo_rows_b) dev_free(c, c->halo_rows_b, sizeof(int32_t) * std::max<int64_t>(c->halo_nb, 1));
  if (c->halo_rows_i) dev_free(c, c->halo_rows_i, sizeof(int32_t) * std::max<int64_t>(c->halo_ni, 1));
  c->halo_rows_b = c->halo_rows_i = nullptr;
  c->halo_nb = c->halo_ni = 0;
}

int cna_set_halo(cna_ctx* c, const int64_t* send_rows, const int64_t* send_counts, const int64_t* recv_rows,
                 const int64_t* recv_counts) {
  CHECK_CTX(c);
  if (!c->indptr) CNA_FAIL(CNA_ESTATE, "cna_set_halo before cna_graph_upload");
  HIP_TRY(hipStreamSynchronize(c->stream));
  halo_clear(c);
  if (!send_counts || !recv_counts) return 0;
  if (!comm_active(c)) CNA_FAIL(CNA_ESTATE, "cna_set_halo needs cna_comm_init");
  int64_t ns = 0, nr = 0;
  for (int p = 0; p < c->nranks; ++p) {
    if (send_counts[p] < 0 || recv_counts[p] < 0) CNA_FAIL(CNA_EINVAL, "cna_set_halo: negative count");
    ns += send_counts[p];
    nr += recv_counts[p];
  }
  if ((ns > 0 && !send_rows) || (nr > 0 && !recv_rows)) CNA_FAIL(CNA_EINVAL, "cna_set_halo: missing row list");
  for (int64_t k = 0; k < ns; ++k)
    if (send_rows[k] < 0 || send_rows[k] >= c->n_local) CNA_FAIL(CNA_EINVAL, "cna_set_halo: send row outside the local block");
  for (int64_t k = 0; k < nr; ++k)
    if (recv_rows[k] < 0 || recv_rows[k] >= c->n_global) CNA_FAIL(CNA_EINVAL, "cna_set_halo: recv row out of range");
  CNA_TRY(dev_alloc(c, (void**)&c->halo_send_idx, sizeof(int64_t) * std::max<int64_t>(ns, 1)));
  c->halo_ns = ns;
  CNA_TRY(dev_alloc(c, (void**)&c->halo_recv_idx, sizeof(int64_t) * std::max<int64_t>(nr, 1)));
  c->halo_nr = nr;
  if (ns) HIP_TRY(hipMemcpy(c->halo_send_idx, send_rows, sizeof(int64_t) * ns, hipMemcpyHostToDevice));
  if (nr) HIP_TRY(hipMemcpy(c->halo_recv_idx, recv_rows, sizeof(int64_t) * nr, hipMemcpyHostToDevice));
  c->halo_send_cnt.assign(send_counts, send_counts + c->nranks);
  c->halo_recv_cnt.assign(recv_counts, recv_counts + c->nranks);
  c->halo_on = true;
  // the block's rows in two lists -- those some other rank has asked for, and the rest: a step that is followed by an
  // exchange walks the first list, hands its rows to the exchange and walks the second meanwhile (cna_nam_step)
  {
    std::vector<char> wanted((size_t)std::max<int64_t>(c->n_local, 1), 0);
    for (int64_t k = 0; k < ns; ++k) wanted[(size_t)send_rows[k]] = 1;
    std::vector<int32_t> rb, ri;
    for (int64_t r = 0; r < c->n_local; ++r) (wanted[(size_t)r] ? rb : ri).push_back((int32_t)r);
    c->halo_nb = (int64_t)rb.size();
    c->halo_ni = (int64_t)ri.size();
    CNA_TRY(dev_alloc(c, (void**)&c->halo_rows_b, sizeof(int32_t) * std::max<int64_t>(c->halo_nb, 1)));
    CNA_TRY(dev_alloc(c, (void**)&c->halo_rows_i, sizeof(int32_t) * std::max<int64_t>(c->halo_ni, 1)));
    if (c->halo_nb) HIP_TRY(hipMemcpy(c->halo_rows_b, rb.data(), sizeof(int32_t) * c->halo_nb, hipMemcpyHostToDevice));
    if (c->halo_ni) HIP_TRY(hipMemcpy(c->halo_rows_i, ri.data(), sizeof(int32_t) * c->halo_ni, hipMemcpyHostToDevice));
    // (the stream may exist already: cna_comm_selftest makes it for its own exchange)
    if (!c->halo_stream) HIP_TRY(hipStreamCreateWithFlags(&c->halo_stream, hipStreamNonBlocking));
    if (!c->halo_e1) HIP_TRY(hipEventCreateWithFlags(&c->halo_e1, hipEventDisableTiming));
    if (!c->halo_e2) HIP_TRY(hipEventCreateWithFlags(&c->halo_e2, hipEventDisableTiming));
  }
  // the rows that read no foreign row, and the rest (from the graph block itself: no symmetry assumed)
  if (c->n_local > 0) {
    unsigned char* fl = nullptr;
    HIP_TRY(hipMalloc(&fl, (size_t)c->n_local));
    struct Free { unsigned char* p; ~Free() { (void)hipFree(p); } } free_fl{fl};
    CNA_TRY(launch_rows_need_halo(c, fl));
    std::vector<unsigned char> hf((size_t)c->n_local);
    HIP_TRY(hipMemcpyAsync(hf.data(), fl, (size_t)c->n_local, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<int32_t> rs, rn;
    for (int64_t r = 0; r < c->n_local; ++r) (hf[(size_t)r] ? rn : rs).push_back((int32_t)r);
    c->halo_nsafe = (int64_t)rs.size();
    c->halo_nneed = (int64_t)rn.size();
    CNA_TRY(dev_alloc(c, (void**)&c->halo_rows_safe, sizeof(int32_t) * std::max<int64_t>(c->halo_nsafe, 1)));
    CNA_TRY(dev_alloc(c, (void**)&c->halo_rows_need, sizeof(int32_t) * std::max<int64_t>(c->halo_nneed, 1)));
    if (c->halo_nsafe) HIP_TRY(hipMemcpy(c->halo_rows_safe, rs.data(), sizeof(int32_t) * c->halo_nsafe, hipMemcpyHostToDevice));
    if (c->halo_nneed) HIP_TRY(hipMemcpy(c->halo_rows_need, rn.data(), sizeof(int32_t) * c->halo_nneed, hipMemcpyHostToDevice));
  }
  // State for local + halo rows only: the rows this rank receives are the ONLY foreign rows its graph block
  // references (that is what the plan is), so the state needs n_local + nr rows, not n_global -- and what arrives can land
  // in the tail directly, in the order it is sent.  The walk steps read the graph through column indices renumbered
  // into that row space (one pass, here).  CNA_COMPACT_STATE=0 keeps the global row space (A/B, tests).
  {
    const char* e = getenv("CNA_COMPACT_STATE");
    bool ascending = true;
    for (int64_t k = 1; k < nr && ascending; ++k) ascending = recv_rows[k] > recv_rows[k - 1];
    for (int64_t k = 0; k < nr && ascending; ++k) ascending = recv_rows[k] < c->row0 || recv_rows[k] >= c->row0 + c->n_local;
    if (!(e && atoi(e) == 0) && ascending && c->n_local + nr < ((int64_t)1 << 31)) {
      CNA_TRY(dev_alloc(c, (void**)&c->idx_t, sizeof(int32_t) * std::max<int64_t>(c->nnz, 1)));
      c->idx_t_n = c->nnz;
      int bad = 0;
      CNA_TRY(launch_remap_indices(c, c->halo_recv_idx, nr, c->idx_t, &bad));
      if (bad) {
        dev_free(c, c->idx_t, sizeof(int32_t) * std::max<int64_t>(c->idx_t_n, 1));
        c->idx_t = nullptr;
        c->idx_t_n = 0;
        CNA_FAIL(CNA_EINVAL, "cna_set_halo: the graph block references rows that are neither local nor in the receive list");
      }
      c->t_compact = true;
      c->t_rows = c->n_local + nr;
      c->t_valid = false;
      c->cellinfo_valid = false;
      if (c->sid) {                      // (a plan that arrives after cna_set_samples: size the buffers for the new row space)
        CNA_TRY(ensure_T(c, c->ld));
        CNA_TRY(ensure_sparse_state(c));
      }
    }
  }
  return 0;
}

// Between diffusion steps every rank needs the state rows of its cells' neighbours.  Default: ring
// all-gather of the row blocks (every rank ends up with everything).  With cna_set_halo: pack the
// rows other ranks asked for, one grouped send/recv per peer, scatter what arrives -- with a banded
// cell order that is a few per cent of the all-gather volume.
static int exchange_state(cna_ctx* c, double* T, hipStream_t st = nullptr) {
  if (c->halo_on) {
    const int ld = c->t_ld;
    CNA_TRY(dev_reserve(c, &c->halo_sbuf, &c->halo_sbuf_cap, 8 * std::max<int64_t>(c->halo_ns, 1) * ld));
    ProfScope ps(c, CNA_K_HALO_EXCHANGE, st);              // pack + send / receive + unpack, on the stream that carries them
    if (c->t_compact) {
      // the rows that arrive ARE the tail of the state, in the order of the receive list: no staging, no scatter
      CNA_TRY(launch_pack_rows(c, T, c->halo_send_idx, c->halo_ns, ld, (double*)c->halo_sbuf, st));
      if (c->halo_ns + c->halo_nr > 0)
        CNA_TRY(comm_halo_exchange(c, (const double*)c->halo_sbuf, T + c->n_local * (int64_t)ld, ld, st));
      return 0;
    }
    CNA_TRY(dev_reserve(c, &c->halo_rbuf, &c->halo_rbuf_cap, 8 * std::max<int64_t>(c->halo_nr, 1) * ld));
    CNA_TRY(launch_pack_rows(c, T + c->row0 * ld, c->halo_send_idx, c->halo_ns, ld, (double*)c->halo_sbuf, st));
    if (c->halo_ns + c->halo_nr > 0)
      CNA_TRY(comm_halo_exchange(c, (const double*)c->halo_sbuf, (double*)c->halo_rbuf, ld, st));
    CNA_TRY(launch_unpack_rows(c, (const double*)c->halo_rbuf, c->halo_recv_idx, c->halo_nr, ld, T, st));
    return 0;
  }
  if (c->nranks == 1) return 0;
  const size_t block = sizeof(double) * (size_t)c->rows_per_rank * c->t_ld;
  return comm_allgather_bytes(c, (char*)T + block * c->rank, T, block);
}
static int exchange_stat(cna_ctx* c) {
  if (c->nranks == 1) return 0;
  const size_t block = sizeof(double) * (size_t)c->rows_per_rank;
  return comm_allgather_bytes(c, (char*)c->stat + block * c->rank, c->stat, block);
}

// y: the standardised phenotype the analysis will pass to cna_select_standardized[_fused] (n = number of samples);
// the next walk step that is the last of its walk then also leaves that call's results (see common.h: byp_*).
// y = NULL clears.  A hint, never a promise: a selection call that asks for anything else runs its own pass.
int cna_nam_select_hint(cna_ctx* c, const double* y, int n) {
  CHECK_CTX(c);
  c->byp_hint.clear();
  if (!y || n < 2 || n > 1024) return 0;
  if (!c->byp_buf) HIP_TRY(hipMalloc(&c->byp_buf, 16 + 8 * 1024));
  // Through pinned staging, queued on the MAIN stream behind the steps already there: asynchronous (this call does not wait
  // for them) and in stream order in front of the step that reads y.  (Round 5 copied from the caller's pageable array on
  // the copy stream and waited for it: ~15 us of the host in front of the walk's last launch.)  The staging is free again:
  // the previous analysis that used it has been collected.
  if (!c->h_hint) HIP_TRY(hipHostMalloc(&c->h_hint, 8 * 1024, hipHostMallocDefault));
  std::memcpy(c->h_hint, y, 8 * (size_t)n);
  HIP_TRY(hipMemcpyAsync((char*)c->byp_buf + 16, c->h_hint, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
  c->byp_hint.assign(y, y + n);
  return 0;
}

static int x_ld(int Nx);
// The NAM on the device, for whoever reads it.  A last step that left the selection pass's results instead (see
// cna_nam_select_hint) is run once more, now for the NAM: its input state is still in T[t_cur] (a step that ends a walk
// writes no state), so are the pairs of a two-step walk; same kernel, same inputs, same bits as a first run would give.
static int need_nam(cna_ctx* c) {
  if (!c->nam_valid && c->nam_lazy) {
    const int done = c->steps_done;
    c->steps_done = c->lazy_steps_before;                 // launch_nam_step picks the compressed second step by it
    const int rc = launch_nam_step(c, false, false, false, true, false);
    c->steps_done = done;
    CNA_TRY(rc);
    c->nam_valid = true;
    c->nam_lazy = false;
  }
  if (!c->nam_valid) CNA_FAIL(CNA_ESTATE, "NAM not available");
  return 0;
}
// 1: the launch that follows is to produce the by-product (buffers sized, y and counters on the device); 0: no
static int arm_select_byproduct(cna_ctx* c) {
  const char* sw = getenv("CNA_WALK_SELECT");              // A/B switch, read at every walk (tests flip it)
  const bool off = sw && atoi(sw) == 0;
  std::vector<double> y;
  y.swap(c->byp_hint);                                  // one-shot
  if (off || (int)y.size() != c->N || c->t_ld != c->ld || c->t_ld <= 64 || c->N < 2 || c->n_local < 1) return 0;
  const int64_t nx = c->n_local;
  const int Nx = c->N;
  void* xp = c->X;
  if (dev_reserve(c, &xp, &c->x_cap, (int64_t)sizeof(double) * nx * x_ld(Nx))) return 0;
  c->X = (double*)xp;
  void* np = c->ncorrs;
  if (dev_reserve(c, &np, &c->ncorrs_cap, 8 * nx)) return 0;
  c->ncorrs = (double*)np;
  c->nx = nx;
  c->Nx = Nx;
  c->ldx = x_ld(Nx);
  c->keep_idx = nullptr;
  c->x_valid = false;
  c->ncorrs_valid = false;
  c->xq_valid = false;
  c->coef_early = false;
  c->fdr_inline = false;
  c->byp_with_q = Nx <= 256 && null_i8_enabled();
  if (c->byp_with_q && ensure_xq(c, (Nx + 31) / 32)) return 0;
  if (!c->byp_buf) return 0;
  c->byp_y.swap(y);
  if (hipMemsetAsync(c->byp_buf, 0, 16, c->stream) != hipSuccess) return 0;
  return 1;
}

// ---- the Gram matrix under the walk's last step (see common.h: gram_pre) -------------------------------------
// CNA_GRAM_OVERLAP = K: the step that leaves the selection by-product runs in K row ranges (0 / 1: one launch, the
// Gram kernel afterwards as before).  Measured on C4 (profiles/r04_ab_gram_overlap.txt): without stream priorities the
// Gram ranges take the chip whenever they become ready and the step grows by what they take; with priorities, or with
// the two sides confined to disjoint CUs (hipExtStreamCreateWithCUMask), nothing is gained either -- the gather is bound
// per CU and a Gram workgroup displaces the walk waves of the CU it lands on.  Off by default; the switch stays for the
// bit-identity test of the ranged product.
static int gram_overlap_ranges() {
  const char* e = getenv("CNA_GRAM_OVERLAP");
  const int k = e ? atoi(e) : 0;
  return k < 0 ? 0 : (k > 64 ? 64 : k);
}
static int ensure_gram_stream(cna_ctx* c) {
  if (c->gram_stream_state) return c->gram_stream_state;
  c->gram_stream_state = -1;
  hipError_t err = hipStreamCreateWithFlags(&c->gram_stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipEventCreateWithFlags(&c->gram_pre_done, hipEventDisableTiming);
  if (err == hipSuccess) err = hipEventCreateWithFlags(&c->range_done, hipEventDisableTiming);
  if (err != hipSuccess) { (void)hipGetLastError(); return -1; }
  c->gram_stream_state = 1;
  return 1;
}
// whoever is about to write c->gram_buf / c->gram_part on another stream, or to start the next ranged product
// The Gram matrix reaches the host without a copy engine (round 6): its reduction kernel writes it a second time into
// pinned memory (one rank), or a copy kernel behind the ranks' sum does.  An asynchronous copy from another stream sat
// in the engine's queue behind the per-cell columns of the same analysis: 2 ms at 2M cells (cna_assoc_out.t_ms[9]).
static int gram_host_reserve(cna_ctx* c, int Nx) {
  const int64_t bytes = (int64_t)sizeof(double) * Nx * Nx;
  if (bytes > c->h_gram_cap) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->h_gram) HIP_TRY(hipHostFree(c->h_gram));
    c->h_gram = nullptr;
    c->h_gram_cap = 0;
    HIP_TRY(hipHostMalloc(&c->h_gram, (size_t)bytes, hipHostMallocDefault));
    c->h_gram_cap = bytes;
  }
  return 0;
}
static bool gram_solo(cna_ctx* c) { return !(c->nranks > 1 || comm_active(c)); }
static int gram_host_finish(cna_ctx* c, int Nx) {      // behind the reduction (and the ranks' sum): G on the host when gram_done fires
  if (!c->gram_mirrored) CNA_TRY(launch_copy_f64(c, c->gram_buf, (double*)c->h_gram, (int64_t)Nx * Nx));
  HIP_TRY(hipEventRecord(c->gram_done, c->stream));
  c->gram_n = Nx;
  return 0;
}
static int gram_pre_settle(cna_ctx* c) {
  if (c->gram_pre_pending) {
    HIP_TRY(hipStreamWaitEvent(c->stream, c->gram_pre_done, 0));
    c->gram_pre_pending = false;
  }
  return 0;
}
static int64_t lcm64(int64_t a, int64_t b) {
  int64_t x = a, y = b;
  while (y) { const int64_t t = x % y; x = y; y = t; }
  return a / x * b;
}
// The last step of a walk that leaves the selection by-product, in K row ranges with the Gram kernel of every range
// behind it on gram_stream.  1: queued that way (c->gram_pre set), 0: not eligible (the caller launches the step as
// one kernel), < 0 never; errors are returned as positive codes through *rc.
static int ranged_last_step(cna_ctx* c, bool first, bool want_kurt, bool may_stop, int* rc) {
  *rc = 0;
  const int K = gram_overlap_ranges();
  if (K < 2 || c->nx != c->n_local || c->Nx < 2 || c->Nx > 1024) return 0;
  if (ensure_gram_stream(c) != 1) return 0;
  const int64_t turn = nam_step_turn_rows(c, c->n_local);
  if (turn <= 0) return 0;
  if ((*rc = gram_pre_settle(c)) != 0) return 1;
  int64_t unit_g = 0;
  if ((*rc = gram_pre_begin(c, &unit_g)) != 0) return 1;
  const int64_t unit = lcm64(turn, unit_g);
  const int64_t nunits = c->n_local / unit;
  if (nunits < 2 * K || nam_step_turn_rows(c, unit) != turn) return 0;
  void* g = c->gram_buf;
  if ((*rc = dev_reserve(c, &g, &c->gram_cap, (int64_t)sizeof(double) * c->Nx * c->Nx)) != 0) return 1;
  c->gram_buf = (double*)g;
  hipStream_t W = c->stream;
  // an error from here on: close the profiling span that was opened and wait for what already sits on the Gram stream
  // (nobody would otherwise: gram_pre_pending stays false, and the next writer of X or of the partial tiles must not
  // find those kernels still reading them)
  bool span_open = false;
  int kid_of_span = CNA_K_NAM_STEP;
  auto fail = [&]() { if (span_open) prof_end(c, kid_of_span, c->stream); (void)hipStreamSynchronize(c->gram_stream); return 1; };
#define RL_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cna_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); *rc = (int)e_; return fail(); } } while (0)
  const bool sparse_step = c->sp_cnt && c->steps_done == 1;
  const int kid = sparse_step ? CNA_K_NAM_STEP_SPARSE : CNA_K_NAM_STEP;
  kid_of_span = kid;
  if (c->prof) { prof_begin(c, kid, W); span_open = true; }   // one span over all ranges: the step as the other launches report it
  int64_t b0 = 0;
  for (int r = 0; r < K; ++r) {
    const int64_t b1 = r + 1 == K ? c->n_local : (nunits * (r + 1) / K) * unit;
    if ((*rc = launch_nam_step(c, first, want_kurt, false, may_stop, false, nullptr, 0, b0, b1 - b0, W, false)) != 0) break;
    if (r + 1 == K && span_open) { prof_end(c, kid, W); span_open = false; }
    RL_TRY(hipEventRecord(c->range_done, W));
    RL_TRY(hipStreamWaitEvent(c->gram_stream, c->range_done, 0));
    if ((*rc = gram_pre_range(c, b0, b1, c->gram_stream)) != 0) break;
    b0 = b1;
  }
  if (*rc) return fail();
  if ((*rc = gram_pre_finish(c, c->gram_buf, c->gram_stream)) != 0) return fail();
  RL_TRY(hipEventRecord(c->gram_pre_done, c->gram_stream));
#undef RL_TRY
  c->gram_pre = true;
  c->gram_pre_pending = true;
  return 1;
}

int cna_nam_step(cna_ctx* c, int want_kurt, int may_continue, int may_stop) {
  CHECK_CTX(c);
  if (!c->sid || !c->have_colsum) CNA_FAIL(CNA_ESTATE, "cna_nam_step needs cna_colsums and cna_set_samples");
  if (c->t_ld != c->ld) CNA_FAIL(CNA_ESTATE, "state buffers hold a dense diffusion; call cna_set_samples again");
  const bool first = c->steps_done == 0;
  if (!first && !c->t_valid) CNA_FAIL(CNA_ESTATE, "previous step did not keep its state (may_continue=0)");
  c->byp_valid = false;
  c->x_ident = false;
  c->nam_lazy = false;
  c->gram_pre = false;
  const bool arm = !first && !may_continue && may_stop && !c->auto_stop && arm_select_byproduct(c) == 1;
  // ... and then the NAM itself is not written: the analysis reads X, and whoever does ask for the NAM (res.nam, a later
  // call with other covariates) gets it from a second run of this step (need_nam), whose input state stays where it is
  const bool skip_nam = arm;
  c->byp_arm = arm;
  c->byp_skip_nam = skip_nam;
  // A step whose state other ranks need (halo exchange): the rows they asked for first, their exchange on its own
  // stream while the rest of the block is walked; the next step waits for both.  (The buffers are sized before anything
  // is queued: a reallocation would wait for the device.)
  const char* ov = getenv("CNA_HALO_OVERLAP");
  // (RCCL: only with a communicator of the halo stream's own, cna_comm_init; the shared-memory test backend stages
  // through the host and has no such constraint)
  const bool overlap_ok = c->halo_on && c->halo_stream && !(ov && atoi(ov) == 0) && (c->shm || c->comm_halo);
  // (NOT a function of this rank's own row counts: a rank nobody asks for rows, or one without interior rows, takes the
  // same path with an empty launch -- which communicator carries the exchange must be the same decision on every rank)
  const bool overlap = may_continue && overlap_ok;
  // Round 5: the main stream no longer waits for an exchange where it is queued (halo_wait_pending) but where its rows
  // are first needed.  A step that reads a state walks the rows WITHOUT a foreign neighbour first -- under the exchange that
  // brings the foreign rows -- then waits, then walks the rest; its own exchange starts when both are done and is in turn
  // hidden under the next step's safe rows (the last step included: row list and selection by-product combine).  The
  // first step reads no state: the rows other ranks asked for first, their exchange beside the interior.
  const bool two_lists = overlap_ok && c->halo_nsafe > 0 && c->halo_nneed > 0;
  auto walk_safe_then_rest = [&](bool wt) -> int {
    c->halo_safe_launch = true;
    int rc2 = launch_nam_step(c, first, want_kurt != 0, wt, may_stop != 0, false, c->halo_rows_safe, c->halo_nsafe);
    c->halo_safe_launch = false;
    if (rc2 == 0) rc2 = launch_nam_step(c, first, want_kurt != 0, wt, may_stop != 0, false, c->halo_rows_need, c->halo_nneed);
    return rc2;
  };
  if (overlap) {
    c->byp_arm = false;
    c->byp_skip_nam = false;
    const int ld = c->t_ld;
    double* Tn = c->T[c->t_cur ^ 1];
    CNA_TRY(dev_reserve(c, &c->halo_sbuf, &c->halo_sbuf_cap, 8 * std::max<int64_t>(c->halo_ns, 1) * ld));
    if (!c->t_compact) CNA_TRY(dev_reserve(c, &c->halo_rbuf, &c->halo_rbuf_cap, 8 * std::max<int64_t>(c->halo_nr, 1) * ld));
    if (first || !two_lists) {
      CNA_TRY(launch_nam_step(c, first, want_kurt != 0, true, may_stop != 0, false, c->halo_rows_b, c->halo_nb));
      HIP_TRY(hipEventRecord(c->halo_e1, c->stream));
      // (the interior rows: nobody else reads their state after the FIRST step -- this rank's second step takes its local
      // neighbours from their pairs -- so their dense rows, 8N bytes each, are not written: only rows that overflow the pairs)
      c->sp_dense_interior_off = first;
      const int rc_int = launch_nam_step(c, first, want_kurt != 0, true, may_stop != 0, false, c->halo_rows_i, c->halo_ni);
      c->sp_dense_interior_off = false;
      CNA_TRY(rc_int);
    } else {
      CNA_TRY(walk_safe_then_rest(true));
      HIP_TRY(hipEventRecord(c->halo_e1, c->stream));
    }
    HIP_TRY(hipStreamWaitEvent(c->halo_stream, c->halo_e1, 0));
    CNA_TRY(exchange_state(c, Tn, c->halo_stream));
    HIP_TRY(hipEventRecord(c->halo_e2, c->halo_stream));
    c->halo_wait_pending = true;                 // (halo_settle: before the first launch that reads those rows, before any collective)
    c->t_cur ^= 1;
    c->lazy_steps_before = c->steps_done;
  } else {
    int rc_step = 0;
    if (!first && !may_continue && c->halo_wait_pending && two_lists && !c->auto_stop && gram_overlap_ranges() < 2) {
      rc_step = walk_safe_then_rest(false);     // the walk's last step: safe rows under the last exchange, then the rest
    } else if (!(arm && ranged_last_step(c, first, want_kurt != 0, may_stop != 0, &rc_step) == 1)) {
      rc_step = launch_nam_step(c, first, want_kurt != 0, may_continue != 0, may_stop != 0, false);
    }
    c->byp_arm = false;
    c->byp_skip_nam = false;
    CNA_TRY(rc_step);
    c->byp_valid = arm;
    c->lazy_steps_before = c->steps_done;
    if (may_continue) {
      CNA_TRY(exchange_state(c, c->T[c->t_cur ^ 1]));
      c->t_cur ^= 1;
    }
  }
  c->t_valid = may_continue != 0;
  c->nam_valid = may_stop != 0 && !skip_nam;
  c->nam_lazy = skip_nam;
  c->steps_done += 1;
  if (want_kurt) {
    c->stat_space = CNA_MAT_NAM;
    CNA_TRY(exchange_stat(c));
  }
  return 0;
}

int cna_nam_steps(cna_ctx* c, int nsteps) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (nsteps < 1) CNA_FAIL(CNA_EINVAL, "nsteps < 1");
  for (int i = 0; i < nsteps; ++i) CNA_TRY(cna_nam_step(c, 0, i + 1 < nsteps, i + 1 == nsteps));
  return 0;
}

// The walk of _nam.py:57-70 with nsteps=None: steps until the median kurtosis stops falling by 3 or more (checked from
// the third step on, at most maxnsteps).  Steps, medians and the rule are queued without waiting for a verdict --
// four steps by cna_nam_auto_launch, which returns at once, then two at a time by cna_nam_auto_finish, which reads
// one word per batch; the steps queued behind the one that met the rule return at once (StepArgs::stop).  The first
// step's kurtosis is never looked at by the rule (_nam.py:65: i + 1 >= 3 compares steps 2 and 3) and is not
// computed.  Same steps, same NAM as cna_nam_step + cna_stat_median in a host loop.
static int ensure_auto_state(cna_ctx* c, unsigned long long** hist);
static int auto_queue(cna_ctx* c, int upto) {
  const size_t sb = (auto_state_bytes() + 255) & ~(size_t)255;
  unsigned long long* hist = (unsigned long long*)((char*)c->auto_state + sb);
  struct Guard { cna_ctx* c; ~Guard() { c->auto_stop = nullptr; } } guard{c};
  c->auto_stop = (const int*)((const char*)c->auto_state + auto_state_stopped_offset());
  for (int i = c->auto_queued; i < upto; ++i) {
    const bool last = i + 1 == c->auto_max;
    CNA_TRY(cna_nam_step(c, i >= 1, !last, last || i + 1 >= 3));
    if (i >= 1) CNA_TRY(launch_auto_median(c, c->stat, c->n_global, c->auto_state, hist, i, 3));
    c->auto_queued = i + 1;
  }
  return 0;
}

int cna_nam_auto_launch(cna_ctx* c, int maxnsteps) {
  CHECK_CTX(c);
  if (maxnsteps < 1 || maxnsteps > 16) CNA_FAIL(CNA_EINVAL, "cna_nam_auto: 1 <= maxnsteps <= 16");
  if (c->steps_done != 0) CNA_FAIL(CNA_ESTATE, "cna_nam_auto starts a walk: call cna_set_samples / cna_restart_nam first");
  const size_t sb = (auto_state_bytes() + 255) & ~(size_t)255;
  CNA_TRY(ensure_auto_state(c, nullptr));
  HIP_TRY(hipMemsetAsync(c->auto_state, 0, sb, c->stream));
  c->auto_max = maxnsteps;
  c->auto_queued = 0;
  c->auto_pending = true;
  const int rc = auto_queue(c, std::min(4, maxnsteps));
  if (rc) c->auto_pending = false;
  return rc;
}

int cna_nam_auto_finish(cna_ctx* c, int* steps_out, double* medkurt_out) {
  CHECK_CTX(c);
  if (!c->auto_pending) CNA_FAIL(CNA_ESTATE, "cna_nam_auto_finish without cna_nam_auto_launch");
  c->auto_pending = false;
  const size_t sb = (auto_state_bytes() + 255) & ~(size_t)255;
  std::vector<char> host(sb);
  int taken = 0;
  for (;;) {
    HIP_TRY(hipMemcpyAsync(host.data(), c->auto_state, sb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int stopped;
    std::memcpy(&stopped, host.data() + auto_state_stopped_offset(), 4);
    taken = stopped ? stopped : c->auto_queued;
    if (stopped || c->auto_queued >= c->auto_max) break;
    CNA_TRY(auto_queue(c, std::min(c->auto_queued + 2, c->auto_max)));
  }
  // the steps queued behind the last one taken did nothing: the NAM on the device is that of step `taken`
  c->steps_done = taken;
  c->nam_valid = true;
  c->t_valid = false;
  c->stat_space = CNA_MAT_NAM;
  if (steps_out) *steps_out = taken;
  if (medkurt_out) {
    const size_t off = 2 * 8 + 2 * 8 + 2 * 8;             // AutoState: prefix[2], k[2], n_tot, n_nan, then med[16]
    std::memcpy(medkurt_out, host.data() + off, 8 * (size_t)taken);
    medkurt_out[0] = __builtin_nan("");                    // (not computed: the rule never reads it)
  }
  return 0;
}

int cna_nam_auto(cna_ctx* c, int maxnsteps, int* steps_out, double* medkurt_out) {
  CNA_TRY(cna_nam_auto_launch(c, maxnsteps));
  return cna_nam_auto_finish(c, steps_out, medkurt_out);
}

int cna_fetch_cell_stat(cna_ctx* c, double* out, int64_t n_expected) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (c->stat_space == CNA_MAT_NAM && c->local_view) {
    if (n_expected != c->n_local) CNA_FAIL(CNA_EINVAL, "cna_fetch_cell_stat: expected n_local entries");
    HIP_TRY(hipMemcpyAsync(out, c->stat + c->row0, sizeof(double) * c->n_local, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
  }
  if (c->stat_space == CNA_MAT_NAM) {
    if (n_expected != c->n_global) CNA_FAIL(CNA_EINVAL, "cna_fetch_cell_stat: expected n_global entries");
    HIP_TRY(hipMemcpyAsync(out, c->stat, sizeof(double) * c->n_global, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
  }
  if (c->stat_space != CNA_MAT_X) CNA_FAIL(CNA_ESTATE, "no per-cell statistic available");
  if (c->nranks == 1 || c->local_view) {
    if (n_expected != c->nx) CNA_FAIL(CNA_EINVAL, "cna_fetch_cell_stat: expected nx entries");
    HIP_TRY(hipMemcpyAsync(out, c->stat, sizeof(double) * c->nx, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
  }
  return ragged_gather(c, c->stat, c->nx, out, n_expected);
}

// Exact median of the per-cell statistic (np.median semantics: NaN if any entry is NaN, mean of the
// two middle values for an even count) by radix select on the device: 8 passes of an 8-bit digit
// histogram per order statistic, a 2 KB read-back each -- ~0.3 ms instead of the 1-6 ms numpy's
// partition takes on 200k-1M values, and no cells-sized transfer.  NAM-space statistics are
// replicated on every rank (all n_global values); X-space ones are the local rows, so the
// histograms are summed over ranks.
static int stat_select(cna_ctx* c, const double* v, int64_t n_loc, bool sharded, int64_t k, unsigned long long* hist_dev,
                       double* value, int64_t* n_nan, int64_t* n_total) {
  unsigned long long prefix = 0;
  std::vector<unsigned long long> h(257);
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    CNA_TRY(launch_digit_hist(c, v, n_loc, prefix, shift, hist_dev));
    if (sharded) CNA_TRY(comm_allreduce_i64_sum(c, (int64_t*)hist_dev, 257));
    HIP_TRY(hipMemcpyAsync(h.data(), hist_dev, 8 * 257, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (pass == 0) {
      int64_t tot = 0;
      for (int d = 0; d < 256; ++d) tot += (int64_t)h[d];
      *n_nan = (int64_t)h[256];
      *n_total = tot + *n_nan;
      if (*n_nan > 0 || tot == 0) return 0;
      if (k < 0) k = (tot - 1) / 2;                        // caller asked for the lower middle
      if (k >= tot) k = tot - 1;
    }
    int d = 0;
    int64_t before = 0;
    while (d < 255 && before + (int64_t)h[d] <= k) before += (int64_t)h[d++];
    k -= before;
    prefix = (prefix << 8) | (unsigned long long)d;
  }
  const unsigned long long bits = (prefix >> 63) ? (prefix & 0x7fffffffffffffffull) : ~prefix;
  std::memcpy(value, &bits, 8);
  return 0;
}

// the device-side bookkeeping block of the medians (rows.hip:AutoState) + 2 x 257 histogram words behind it
static int ensure_auto_state(cna_ctx* c, unsigned long long** hist) {
  const size_t sb = (auto_state_bytes() + 255) & ~(size_t)255;
  if (!c->auto_state) HIP_TRY(hipMalloc(&c->auto_state, sb + 8 * 2 * 257));
  if (hist) *hist = (unsigned long long*)((char*)c->auto_state + sb);
  return 0;
}

// median (and, with qc, threshold and count of _qc_nam) of the current per-cell statistic; everything is queued, the
// host waits once.  out3 = {median, threshold, count}
static int stat_median_device(cna_ctx* c, bool qc, double* out3) {
  const double* v = c->stat;
  int64_t n_loc;
  bool sharded;
  if (c->stat_space == CNA_MAT_NAM) { n_loc = c->n_global; sharded = false; }
  else if (c->stat_space == CNA_MAT_X) { n_loc = c->nx; sharded = c->nranks > 1 || comm_active(c); }
  else CNA_FAIL(CNA_ESTATE, "no per-cell statistic available");
  unsigned long long* hist = nullptr;
  CNA_TRY(ensure_auto_state(c, &hist));
  const size_t sb = (auto_state_bytes() + 255) & ~(size_t)255;
  HIP_TRY(hipMemsetAsync(c->auto_state, 0, sb, c->stream));
  CNA_TRY(launch_auto_median(c, v, n_loc, c->auto_state, hist, -1, 0, sharded));
  if (qc) {
    CNA_TRY(launch_qc_count(c, v, n_loc, c->auto_state));
    if (sharded) CNA_FAIL(CNA_ESTATE, "cna_stat_qc is for the NAM-space statistic");
  }
  struct { double median, threshold; unsigned long long count; } res;
  HIP_TRY(hipMemcpyAsync(&res, (const char*)c->auto_state + auto_state_result_offset(), sizeof(res), hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  out3[0] = res.median;
  out3[1] = res.threshold;
  out3[2] = (double)res.count;
  return 0;
}

int cna_stat_median(cna_ctx* c, double* median_out) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (!median_out) CNA_FAIL(CNA_EINVAL, "cna_stat_median: null output");
  if (getenv("CNA_MEDIAN_HOST")) {            // the first implementation: digit choice on the host, 16 round trips
    const double* v = c->stat;
    int64_t n_loc;
    bool sharded;
    if (c->stat_space == CNA_MAT_NAM) { n_loc = c->n_global; sharded = false; }
    else if (c->stat_space == CNA_MAT_X) { n_loc = c->nx; sharded = c->nranks > 1 || comm_active(c); }
    else CNA_FAIL(CNA_ESTATE, "no per-cell statistic available");
    CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, 8 * 257 + 64));   // main-stream scratch (not c->gt:
    unsigned long long* hist = (unsigned long long*)c->scratch;            // the helper thread may be conditioning)
    double lo = 0.0, hi = 0.0;
    int64_t n_nan = 0, n_tot = 0;
    CNA_TRY(stat_select(c, v, n_loc, sharded, -1, hist, &lo, &n_nan, &n_tot));
    if (n_tot == 0 || n_nan > 0) { *median_out = __builtin_nan(""); return 0; }
    if (n_tot & 1) { *median_out = lo; return 0; }
    CNA_TRY(stat_select(c, v, n_loc, sharded, n_tot / 2, hist, &hi, &n_nan, &n_tot));
    *median_out = (lo + hi) / 2.0;
    return 0;
  }
  double out3[3];
  CNA_TRY(stat_median_device(c, false, out3));
  *median_out = out3[0];
  return 0;
}

// _qc_nam's decision (_nam.py:94-96) on the batch kurtosis left by cna_batch_kurtosis(CNA_MAT_NAM, ..): the median,
// threshold = max(6, 2 median) and the number of cells that fail `kurtosis < threshold` -- zero means "every cell
// stays", and then nothing cells-sized has to travel to the host (cna_fetch_cell_stat otherwise).  One wait.
int cna_stat_qc(cna_ctx* c, double* median_out, double* threshold_out, int64_t* n_dropped_out) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (c->stat_space != CNA_MAT_NAM) CNA_FAIL(CNA_ESTATE, "cna_stat_qc needs the NAM-space statistic of cna_batch_kurtosis");
  double out3[3];
  CNA_TRY(stat_median_device(c, true, out3));
  if (median_out) *median_out = out3[0];
  if (threshold_out) *threshold_out = out3[1];
  if (n_dropped_out) *n_dropped_out = (int64_t)out3[2];
  return 0;
}

// --------------------------------------------------------------------- dense diffusion
int cna_dense_load(cna_ctx* c, const double* s_local, int m) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (!c->have_colsum) CNA_FAIL(CNA_ESTATE, "cna_dense_load needs cna_colsums");
  if (m < 1 || m > 1024) CNA_FAIL(CNA_EINVAL, "dense state must have 1..1024 columns");
  const int ld = round_up(m, 4);
  CNA_TRY(ensure_T(c, ld));
  void* ds = c->dense_s;
  CNA_TRY(dev_reserve(c, &ds, &c->dense_cap, (int64_t)sizeof(double) * std::max<int64_t>(c->n_local, 1) * ld));
  c->dense_s = (double*)ds;
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, (int64_t)sizeof(double) * std::max<int64_t>(c->n_local, 1) * m));
  HIP_TRY(hipMemcpyAsync(c->scratch, s_local, sizeof(double) * c->n_local * m, hipMemcpyHostToDevice, c->stream));
  c->t_width = m;
  c->t_ld = ld;
  c->t_cur = 0;
  c->t_f32[0] = c->t_f32[1] = false;
  CNA_TRY(launch_scale_rows(c, (const double*)c->scratch, c->T[0], m, ld));
  CNA_TRY(exchange_state(c, c->T[0]));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->t_valid = true;
  c->nam_valid = false; c->nam_lazy = false;
  c->steps_done = 1;  // never take the one-hot path
  return 0;
}

int cna_dense_step(cna_ctx* c) {
  CHECK_CTX(c);
  if (!c->t_valid || !c->dense_s) CNA_FAIL(CNA_ESTATE, "cna_dense_step before cna_dense_load");
  CNA_TRY(launch_nam_step(c, false, false, true, false, true));
  CNA_TRY(exchange_state(c, c->T[c->t_cur ^ 1]));
  c->t_cur ^= 1;
  return 0;
}

int cna_dense_fetch(cna_ctx* c, double* out) {
  CHECK_CTX(c);
  if (!c->dense_s) CNA_FAIL(CNA_ESTATE, "no dense state");
  if (c->n_local > 0)
    HIP_TRY(hipMemcpy2DAsync(out, sizeof(double) * c->t_width, c->dense_s, sizeof(double) * c->t_ld,
                             sizeof(double) * c->t_width, c->n_local, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// ------------------------------------------------------------------------ QC / select
int cna_batch_kurtosis(cna_ctx* c, int which, const int32_t* batch_codes, int n_batches) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  const double* mat;
  int64_t rows;
  int ncols, ld;
  double* out;
  if (which == CNA_MAT_NAM) {
    CNA_TRY(need_nam(c));
    mat = c->nam; rows = c->n_local; ncols = c->N; ld = c->ld; out = c->stat + c->row0;
  } else if (which == CNA_MAT_X) {
    if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
    mat = c->X; rows = c->nx; ncols = c->Nx; ld = c->ldx; out = c->stat;
    if (c->nx > c->n_pad) CNA_FAIL(CNA_EINVAL, "X larger than the stat buffer");
  } else {
    CNA_FAIL(CNA_EINVAL, "bad matrix selector");
  }
  if (n_batches < 1) CNA_FAIL(CNA_EINVAL, "n_batches < 1");
  std::vector<int32_t> order, boff(n_batches + 1, 0);
  for (int s = 0; s < ncols; ++s)
    if (batch_codes[s] >= 0 && batch_codes[s] < n_batches) boff[batch_codes[s] + 1]++;
  for (int b = 0; b < n_batches; ++b) boff[b + 1] += boff[b];
  order.resize(std::max(boff[n_batches], 1));
  std::vector<int32_t> cur(boff.begin(), boff.end() - 1);
  for (int s = 0; s < ncols; ++s)
    if (batch_codes[s] >= 0 && batch_codes[s] < n_batches) order[cur[batch_codes[s]]++] = s;
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, carve_bytes({(int64_t)4 * (int64_t)order.size(), 4 * (n_batches + 1)})));
  Carver cv(c->scratch);
  int32_t* order_dev = cv.take<int32_t>(order.size());
  int32_t* boff_dev = cv.take<int32_t>(n_batches + 1);
  HIP_TRY(hipMemcpyAsync(order_dev, order.data(), 4 * order.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(boff_dev, boff.data(), 4 * (n_batches + 1), hipMemcpyHostToDevice, c->stream));
  CNA_TRY(launch_batch_kurtosis(c, mat, rows, ncols, ld, order_dev, boff_dev, n_batches, out));
  c->stat_space = which;
  if (which == CNA_MAT_NAM) CNA_TRY(exchange_stat(c));
  HIP_TRY(hipStreamSynchronize(c->stream));  // host vectors go out of scope
  return 0;
}

int cna_zero_variance(cna_ctx* c, const int32_t* colmap, int n_sel, uint8_t* flags_out, int64_t* n_zero_out) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  CNA_TRY(need_nam(c));
  if (!colmap) n_sel = c->N;
  if (n_sel < 1) CNA_FAIL(CNA_EINVAL, "no samples selected");
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, carve_bytes({4 * (int64_t)n_sel, c->n_pad, 8})));
  Carver cv(c->scratch);
  int32_t* cm = cv.take<int32_t>(n_sel);
  uint8_t* flags = cv.take<uint8_t>(c->n_pad);        // all cells; this rank fills [row0, row0+n_local)
  unsigned long long* cnt = cv.take<unsigned long long>(1);
  if (colmap) HIP_TRY(hipMemcpyAsync(cm, colmap, 4 * n_sel, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(cnt, 0, 8, c->stream));
  HIP_TRY(hipMemsetAsync(flags, 0, c->n_pad, c->stream));
  CNA_TRY(launch_zero_variance(c, colmap ? cm : nullptr, n_sel, flags + c->row0, cnt));
  CNA_TRY(comm_allreduce_i64_sum(c, (int64_t*)cnt, 1));
  unsigned long long h = 0;
  HIP_TRY(hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (n_zero_out) *n_zero_out = (int64_t)h;
  if (flags_out) {
    if (c->local_view) {
      HIP_TRY(hipMemcpyAsync(flags_out, flags + c->row0, c->n_local, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    } else if (h == 0) {
      std::memset(flags_out, 0, c->n_global);
    } else {
      if (c->nranks > 1)
        CNA_TRY(comm_allgather_bytes(c, flags + c->rows_per_rank * c->rank, flags, (size_t)c->rows_per_rank));
      HIP_TRY(hipMemcpyAsync(flags_out, flags, c->n_global, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
  }
  return 0;
}

// Row stride of the working matrix X: a multiple of 4 (MFMA k-depth).  Beyond 128 samples a stride
// that is a multiple of 256 bytes sends the 16 rows of every MFMA A tile to the same memory channels
// (local null at N = 160 / 192 / 224: 37.6 TFLOP/s; with one more quad of zero columns 51 / 56 / 57);
// up to 128 samples the plain stride is as fast or faster (measured at 64, 96, 128).  N = 256 stays
// as it is: the MFMA kernels' instantiations end at 64 quads.
static int x_ld(int Nx) {
  int ld = round_up(Nx, 4);
  if (ld > 128 && (ld * 8) % 256 == 0 && ld + 4 <= 256) ld += 4;
  return ld;
}

int cna_select(cna_ctx* c, const int64_t* keep_idx, int64_t n_keep, const int32_t* colmap, int n_sel) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  CNA_TRY(need_nam(c));
  const int64_t nx = keep_idx ? n_keep : c->n_local;
  const int Nx = colmap ? n_sel : c->N;
  if (nx < 0 || nx > c->n_local || Nx < 1) CNA_FAIL(CNA_EINVAL, "cna_select: bad sizes");
  c->nx = nx;
  c->Nx = Nx;
  c->ldx = x_ld(Nx);
  void* xp = c->X;
  CNA_TRY(dev_reserve(c, &xp, &c->x_cap, (int64_t)sizeof(double) * std::max<int64_t>(nx, 1) * c->ldx));
  c->X = (double*)xp;
  if (keep_idx) {
    void* kp = c->keep_store;
    CNA_TRY(dev_reserve(c, &kp, &c->keep_cap, 8 * std::max<int64_t>(nx, 1)));
    c->keep_store = (int64_t*)kp;
    HIP_TRY(hipMemcpyAsync(c->keep_store, keep_idx, 8 * nx, hipMemcpyHostToDevice, c->stream));
    c->keep_idx = c->keep_store;
  } else {
    c->keep_idx = nullptr;   // identity: every local NAM row
  }
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, 4 * (int64_t)Nx + 256));
  int32_t* cm = (int32_t*)c->scratch;
  if (colmap) HIP_TRY(hipMemcpyAsync(cm, colmap, 4 * Nx, hipMemcpyHostToDevice, c->stream));
  int r = launch_select(c, colmap ? cm : nullptr);
  CNA_TRY(r);
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->x_valid = true;
  c->x_from_nam = true;
  c->ncorrs_valid = false;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  c->coef_early = false;
  c->fdr_inline = false;
  return 0;
}

// Factors of a projector M = I - C.W (C: N x r, W: r x N, row-major; see cna_resid_lowrank) for the NEXT
// cna_select_standardized[_fused] call over N selected samples: that pass then leaves X residualised as well
// (select + zero-variance count + centre + M + /std [+ coefficients] in one read of the NAM).  r = 0 clears.
int cna_set_resid_factors(cna_ctx* c, const double* C, const double* W, int r, int N) {
  CHECK_CTX(c);
  if (r < 0 || N < 1 || (r > 0 && (!C || !W))) CNA_FAIL(CNA_EINVAL, "cna_set_resid_factors: bad arguments");
  c->resid_rk = 0;
  if (r == 0) return 0;
  if ((int64_t)2 * r * N * 8 > 128 * 1024) CNA_FAIL(CNA_EINVAL, "cna_set_resid_factors: factors exceed 128 KB");
  void* p = c->resid_f;
  CNA_TRY(dev_reserve(c, &p, &c->resid_f_cap, 16 * (int64_t)r * N));
  c->resid_f = (double*)p;
  std::vector<double> buf((size_t)2 * r * N);
  std::memcpy(buf.data(), W, 8 * (size_t)r * N);
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < r; ++k) buf[(size_t)r * N + (size_t)k * N + i] = C[(size_t)i * r + k];
  // on the copy stream: the main stream is busy with the walk kernels and this call must not wait for them
  HIP_TRY(hipMemcpyAsync(c->resid_f, buf.data(), 16 * (size_t)r * N, hipMemcpyHostToDevice, c->copy_stream));
  HIP_TRY(hipStreamSynchronize(c->copy_stream));                 // buf is a local; the kernels that read resid_f are issued later
  c->resid_rk = r;
  c->resid_n = N;
  return 0;
}

// cna_select that also counts the selected cells with zero variance over the selected samples
// (_association.py:182) in the same pass; *n_zero_out > 0: redo with cna_zero_variance + cna_select
int cna_select_checked(cna_ctx* c, const int64_t* keep_idx, int64_t n_keep, const int32_t* colmap, int n_sel,
                       int64_t* n_zero_out) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  CNA_TRY(need_nam(c));
  const int64_t nx = keep_idx ? n_keep : c->n_local;
  const int Nx = colmap ? n_sel : c->N;
  if (nx < 0 || nx > c->n_local || Nx < 1) CNA_FAIL(CNA_EINVAL, "cna_select_checked: bad sizes");
  c->nx = nx;
  c->Nx = Nx;
  c->ldx = x_ld(Nx);
  void* xp = c->X;
  CNA_TRY(dev_reserve(c, &xp, &c->x_cap, (int64_t)sizeof(double) * std::max<int64_t>(nx, 1) * c->ldx));
  c->X = (double*)xp;
  if (keep_idx) {
    void* kp = c->keep_store;
    CNA_TRY(dev_reserve(c, &kp, &c->keep_cap, 8 * std::max<int64_t>(nx, 1)));
    c->keep_store = (int64_t*)kp;
    HIP_TRY(hipMemcpyAsync(c->keep_store, keep_idx, 8 * nx, hipMemcpyHostToDevice, c->stream));
    c->keep_idx = c->keep_store;
  } else {
    c->keep_idx = nullptr;
  }
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, carve_bytes({4 * (int64_t)Nx, 8})));
  Carver cv(c->scratch);
  int32_t* cm = cv.take<int32_t>(Nx);
  unsigned long long* cnt = cv.take<unsigned long long>(1);
  if (colmap) HIP_TRY(hipMemcpyAsync(cm, colmap, 4 * Nx, hipMemcpyHostToDevice, c->stream));
  CNA_TRY(launch_select_zv(c, colmap ? cm : nullptr, cnt));
  CNA_TRY(comm_allreduce_i64_sum(c, (int64_t*)cnt, 1));
  unsigned long long h = 0;
  HIP_TRY(hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (n_zero_out) *n_zero_out = (int64_t)h;
  c->x_valid = true;
  c->x_from_nam = true;
  c->ncorrs_valid = false;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  c->coef_early = false;
  c->fdr_inline = false;
  return 0;
}

static int select_standardized_impl(cna_ctx* c, const int64_t* keep_idx, int64_t n_keep, const int32_t* colmap, int n_sel,
                                    int64_t* n_zero_out, const double* y, double* max_abs_out, bool* gram_too);
static int gram_launch_now(cna_ctx* c);

int cna_select_standardized(cna_ctx* c, const int64_t* keep_idx, int64_t n_keep, const int32_t* colmap, int n_sel,
                            int64_t* n_zero_out, const double* y, double* max_abs_out) {
  return select_standardized_impl(c, keep_idx, n_keep, colmap, n_sel, n_zero_out, y, max_abs_out, nullptr);
}

// gram_too != nullptr: the caller will want the Gram matrix of the result next.  When the selection is "all cells,
// samples in place, no projector" and the shape suits it, selection and Gram kernels are ONE launch
// (mfma.hip:k_selgram_blk) and *gram_too comes back true: the matrix is in c->gram_buf, summed over the ranks, as
// after cna_gram_launch (meaningless, like X, when *n_zero_out != 0).
static int select_standardized_impl(cna_ctx* c, const int64_t* keep_idx, int64_t n_keep, const int32_t* colmap, int n_sel,
                                    int64_t* n_zero_out, const double* y, double* max_abs_out, bool* gram_too) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (gram_too) *gram_too = false;
  if (!c->nam_valid && !c->nam_lazy) CNA_FAIL(CNA_ESTATE, "NAM not available");
  const bool byp_was = c->byp_valid;
  const bool gram_pre_was = c->gram_pre;
  c->byp_valid = false;
  c->gram_pre = false;
  const int64_t nx = keep_idx ? n_keep : c->n_local;
  const int Nx = colmap ? n_sel : c->N;
  if (nx < 0 || nx > c->n_local || Nx < 2) CNA_FAIL(CNA_EINVAL, "cna_select_standardized: bad sizes");
  c->nx = nx;
  c->Nx = Nx;
  c->ldx = x_ld(Nx);
  void* xp = c->X;
  CNA_TRY(dev_reserve(c, &xp, &c->x_cap, (int64_t)sizeof(double) * std::max<int64_t>(nx, 1) * c->ldx));
  c->X = (double*)xp;
  if (keep_idx) {
    void* kp = c->keep_store;
    CNA_TRY(dev_reserve(c, &kp, &c->keep_cap, 8 * std::max<int64_t>(nx, 1)));
    c->keep_store = (int64_t*)kp;
    HIP_TRY(hipMemcpyAsync(c->keep_store, keep_idx, 8 * nx, hipMemcpyHostToDevice, c->stream));
    c->keep_idx = c->keep_store;
  } else {
    c->keep_idx = nullptr;
  }
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, carve_bytes({4 * (int64_t)Nx, 8 * (int64_t)Nx, 8 * 4099})));
  Carver cv(c->scratch);
  int32_t* cm = cv.take<int32_t>(Nx);
  double* yd = cv.take<double>(Nx);
  unsigned long long* nz = cv.take<unsigned long long>(4099);      // {zero-variance rows, bits of max |coefficient|, per-workgroup maxima}:
  unsigned long long* mb = nz + 1;                                  // the two results adjacent, one copy to the host
  if (colmap) HIP_TRY(hipMemcpyAsync(cm, colmap, 4 * Nx, hipMemcpyHostToDevice, c->stream));
  if (y) {
    void* np = c->ncorrs;
    CNA_TRY(dev_reserve(c, &np, &c->ncorrs_cap, 8 * std::max<int64_t>(nx, 1)));
    c->ncorrs = (double*)np;
    HIP_TRY(hipMemcpyAsync(yd, y, 8 * Nx, hipMemcpyHostToDevice, c->stream));
  }
  const int rk = (c->resid_rk > 0 && c->resid_n == Nx) ? c->resid_rk : 0;     // one-shot: cna_set_resid_factors
  // the rows leave this pass final: their fixed-point digit planes for the integer local null go out with them
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false;
  const bool with_q = y != nullptr && Nx <= 256 && nx > 0 && null_i8_enabled();
  const int KSq = (Nx + 31) / 32;
  if (with_q) CNA_TRY(ensure_xq(c, KSq));
  bool in_place = colmap == nullptr || n_sel == c->N;
  for (int i = 0; colmap && in_place && i < n_sel; ++i) in_place = colmap[i] == i;
  // the walk's last step may have done this pass already (cna_nam_select_hint): same cells, samples in place, nothing
  // regressed out, the same phenotype bit for bit -- then X, planes, coefficients and the two counters are there
  const bool byp = byp_was && y && !keep_idx && in_place && rk == 0 && Nx == c->N && nx == c->n_local &&
                   with_q == c->byp_with_q && (int)c->byp_y.size() == Nx && std::memcmp(y, c->byp_y.data(), 8 * (size_t)Nx) == 0;
  if (byp) {
    nz = (unsigned long long*)c->byp_buf;
    mb = nz + 1;
  }
  if (!byp) CNA_TRY(need_nam(c));           // (a last step that left X instead of the NAM is run again for the NAM)
  if (!byp) CNA_TRY(gram_pre_settle(c));    // (... and a Gram matrix taken under it is dropped; its kernels read the X this pass rewrites)
  const bool fused = !byp && gram_too && y && !keep_idx && in_place && rk == 0 && gram_fused_ok(c, Nx, c->ldx, 32 * KSq);
  if (byp) {
  } else if (fused) {
    void* g = c->gram_buf;
    CNA_TRY(dev_reserve(c, &g, &c->gram_cap, (int64_t)sizeof(double) * Nx * Nx));
    c->gram_buf = (double*)g;
    CNA_TRY(gram_host_reserve(c, Nx));
    c->gram_mirrored = false;
    c->gram_mirror = gram_solo(c) ? (double*)c->h_gram : nullptr;
    const int rc_sg = launch_selgram(c, c->gram_buf, nz, yd, mb, with_q ? (unsigned char*)c->xq : nullptr,
                                     with_q ? c->xq_scale : nullptr, 32 * KSq);
    c->gram_mirror = nullptr;
    CNA_TRY(rc_sg);
  } else {
    CNA_TRY(launch_select_std(c, (colmap && !in_place) ? cm : nullptr, nz, y ? yd : nullptr, y ? mb : nullptr, c->resid_f,
                              c->resid_f ? c->resid_f + (size_t)rk * Nx : nullptr, rk,
                              with_q ? (unsigned char*)c->xq : nullptr, with_q ? c->xq_scale : nullptr, 32 * KSq));
  }
  c->resid_rk = 0;
  if (y && comm_active(c) && c->nranks > 1 && c->nranks <= 64) {
    // the two counters of this pass in ONE collective: every rank's {zero-variance rows, bits of max |coefficient|}
    // gathered (16 bytes per rank), summed / maximised by a one-wave kernel -- the same numbers as an integer sum and
    // a floating-point maximum over the ranks (non-negative doubles and the NaN pattern order like their bit patterns)
    if (!c->pair_buf) HIP_TRY(hipMalloc(&c->pair_buf, 16 * 65));
    unsigned long long* pb = (unsigned long long*)c->pair_buf;
    CNA_TRY(launch_pair_pack(c, nz, mb, pb + 2 * (size_t)c->rank));
    CNA_TRY(comm_allgather_bytes(c, pb + 2 * (size_t)c->rank, pb, 16));
    CNA_TRY(launch_pair_fold(c, pb, c->nranks, nz, mb));
  } else {
    CNA_TRY(comm_allreduce_i64_sum(c, (int64_t*)nz, 1));
    if (y) CNA_TRY(comm_allreduce_f64_max(c, (double*)mb, 1));
  }
  if (!c->h_scal) HIP_TRY(hipHostMalloc(&c->h_scal, 256, hipHostMallocDefault));
  volatile unsigned long long* hs = (volatile unsigned long long*)c->h_scal;
  hs[0] = 0;
  hs[1] = 0;
  HIP_TRY(hipMemcpyAsync((void*)&hs[0], nz, y ? 16 : 8, hipMemcpyDeviceToHost, c->stream));
  // The caller wants the Gram matrix of this X next: its kernels are queued NOW, behind the two scalars on their way to the
  // host, and the host waits for the scalars only -- the device goes from the selection pass (or from the walk's last step
  // that left X as a by-product) straight into the product instead of idling through the host's round trip (30-35 us,
  // profiles/r06_timeline_C2.txt).  Should the scalars say "rows of zero variance", the matrix is dropped with X.
  const bool gram_early = gram_too && y && !fused && !(byp && gram_pre_was);
  if (gram_early) {
    HIP_TRY(hipEventRecord(c->scal_ready, c->stream));
    CNA_TRY(gram_launch_now(c));
    *gram_too = true;
    HIP_TRY(hipEventSynchronize(c->scal_ready));
  } else {
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  const unsigned long long h = hs[0];
  double m = 0.0;
  {
    const unsigned long long bits = hs[1];
    std::memcpy(&m, &bits, 8);
  }
  if (n_zero_out) *n_zero_out = (int64_t)h;
  if (max_abs_out) *max_abs_out = m;
  c->x_valid = true;
  c->x_from_nam = true;
  c->ncorrs_valid = y != nullptr;     // meaningful only when no cell had zero variance (the caller checks)
  c->xq_valid = with_q;
  // X is now the standardised NAM of every cell with the samples in place and nothing regressed out: a function of
  // the NAM alone, so a further analysis of the resident dataset that asks for the same selection can keep it
  // (cna_x_identity) and take only its coefficients (cna_ncorrs)
  c->x_ident = !keep_idx && in_place && rk == 0 && h == 0 && Nx == c->N && nx == c->n_local;
  c->coef_early = false;
  c->fdr_inline = false;
  if (fused) {                                   // what cna_gram_launch does after its kernels
    CNA_TRY(comm_allreduce_f64_sum(c, c->gram_buf, (size_t)Nx * Nx));
    CNA_TRY(gram_host_finish(c, Nx));
    *gram_too = true;
  } else if (byp && gram_pre_was) {
    c->gram_pre = true;                          // X^T X of this X was taken under the walk: cna_gram_launch finds it
    if (gram_too) {
      CNA_TRY(cna_gram_launch(c));
      *gram_too = true;
    }
  }
  return 0;
}

// thresholds = np.arange(maxcorr/4, maxcorr, maxcorr/400) and edges = thr**2 - 1e-8 - 1e-5*thr**2
// exactly as numpy evaluates them (_association.py:101-103, _stats.py:47): arange's length is
// ceil((stop - start) / step) and its values start + i*delta with delta = (start + step) - start;
// the edge expression rounds after every operation, left to right.  Returns T (0: out of range).
static int null_local_prepare(cna_ctx* c, int P, const double* edges, int T, int want_tails, const double* thr);
static int null_local_go(cna_ctx* c, int col0);

int cna_reference_thresholds(double maxabs, int cap, double* thr, double* edges) {
#pragma clang fp contract(off)
  const double maxcorr = maxabs > 0.001 ? maxabs : 0.001;
  if (!(maxcorr < 1e300)) return 0;
  const double start = maxcorr / 4, stop = maxcorr, step = maxcorr / 400;
  const double len = std::ceil((stop - start) / step);
  if (!(len >= 1) || len > cap) return 0;
  const int T = (int)len;
  const double next = start + step;
  const double delta = next - start;
  for (int i = 0; i < T; ++i) {
    volatile double t = i == 0 ? start : (i == 1 ? next : start + (double)i * delta);
    thr[i] = t;
    volatile double z2 = t * t;
    volatile double a = z2 - 1e-8;
    volatile double b = 1e-5 * z2;
    edges[i] = a - b;
  }
  return T;
}

// select + standardise + coefficients (cna_select_standardized with y) and, when no selected cell has
// zero variance, everything the host would issue next from values it has to wait for anyway -- the
// Gram kernels, the thresholds of the local null from max|ncorrs|, the threshold-only half of the
// local-null pass (cna_null_local_prepare) and the early coefficient column -- in the same call, so
// that none of it waits for the interpreter.  *T_out = 0: nothing beyond the selection was issued.
int cna_select_standardized_fused(cna_ctx* c, const int64_t* keep_idx, int64_t n_keep, const int32_t* colmap, int n_sel,
                                  int64_t* n_zero_out, const double* y, double* max_abs_out, int null_P, int* T_out,
                                  double* thr_out, int* gram_queued, int* coef_queued, int null_col0, const int* null_flag,
                                  int* null_launched) {
  if (T_out) *T_out = 0;
  if (gram_queued) *gram_queued = 0;
  if (coef_queued) *coef_queued = 0;
  if (null_launched) *null_launched = 0;
  int64_t nz = 0;
  double m = 0.0;
  bool gram_done = false;
  CNA_TRY(select_standardized_impl(c, keep_idx, n_keep, colmap, n_sel, &nz, y, &m, &gram_done));
  if (n_zero_out) *n_zero_out = nz;
  if (max_abs_out) *max_abs_out = m;
  if (nz != 0 || !y) return 0;
  double edges[512];
  int T = 0;
  if (null_P >= 1 && T_out && thr_out && !c->null_pending) T = cna_reference_thresholds(m, 512, thr_out, edges);
  // The Gram kernels first (round 6): every call issued here costs the host ~5 us and the device has nothing to do
  // until the first kernel arrives -- 0.18 ms between the selection pass and the Gram kernel at 200 000 cells when the
  // coefficient column's five calls went first (rocprofv3 --kernel-trace, profiles/r06_timeline_C2_*.txt).  Whoever
  // consumes the Gram matrix (the eigenpairs) is the longer chain; the coefficient column (short kernels, a copy on
  // its own stream) follows and still reaches the host under the local null.
  if (!gram_done) CNA_TRY(cna_gram_launch(c));
  if (gram_queued) *gram_queued = 1;
  if (T >= 1 && coef_queued && !((c->nranks > 1 || comm_active(c)) && !c->local_view)) {
    CNA_TRY(cna_percell_coef_launch(c));
    *coef_queued = 1;
  }
  if (T < 1) return 0;
  CNA_TRY(null_local_prepare(c, null_P, edges, T, 0, thr_out));
  *T_out = T;
  // ... and, when the conditioned phenotypes of THIS analysis are already resident (columns null_col0 .. null_col0 +
  // null_P of Zc: the draw and cna_condition_phenotypes ran beside the walk -- the caller vouches for it, or *null_flag,
  // set by the library's draw thread when its conditioning has returned, says so at this very moment), the local-null
  // pass itself: it then starts right behind the Gram kernel instead of after the interpreter's next few statements
  if (null_col0 >= 0 && null_launched && (!null_flag || __atomic_load_n(null_flag, __ATOMIC_ACQUIRE) == 1) && c->zc &&
      null_col0 + null_P <= c->zc_cols && c->zc_rows == c->Nx) {
    CNA_TRY(null_local_go(c, null_col0));
    *null_launched = 1;
  }
  return 0;
}

int cna_upload_x(cna_ctx* c, const double* x_local, int64_t n_rows, int n_cols) {
  CHECK_CTX(c);
  if (n_rows < 0 || n_cols < 1 || n_cols > 1024) CNA_FAIL(CNA_EINVAL, "cna_upload_x: bad shape");
  c->nx = n_rows;
  c->Nx = n_cols;
  c->ldx = x_ld(n_cols);
  void* xp = c->X;
  CNA_TRY(dev_reserve(c, &xp, &c->x_cap, (int64_t)sizeof(double) * std::max<int64_t>(n_rows, 1) * c->ldx));
  c->X = (double*)xp;
  HIP_TRY(hipMemsetAsync(c->X, 0, sizeof(double) * std::max<int64_t>(n_rows, 1) * c->ldx, c->stream));
  if (n_rows > 0)
    HIP_TRY(hipMemcpy2DAsync(c->X, sizeof(double) * c->ldx, x_local, sizeof(double) * n_cols,
                             sizeof(double) * n_cols, n_rows, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->keep_idx = nullptr;
  c->x_valid = true;
  c->x_from_nam = false;
  c->ncorrs_valid = false;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  c->coef_early = false;
  c->fdr_inline = false;
  return 0;
}

// ------------------------------------------------------------------- residualise + PCA
int cna_resid_apply(cna_ctx* c, const double* M, int center) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  const int Nx = c->Nx, ldx = c->ldx;
  const int ldb = round_up(Nx, 16);
  std::vector<double> B((size_t)ldx * ldb, 0.0);
  for (int k = 0; k < Nx; ++k)
    for (int j = 0; j < Nx; ++j) B[(size_t)k * ldb + j] = M ? M[(size_t)j * Nx + k] : (j == k ? 1.0 : 0.0);
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, (int64_t)sizeof(double) * ldx * ldb));
  HIP_TRY(hipMemcpyAsync(c->scratch, B.data(), sizeof(double) * ldx * ldb, hipMemcpyHostToDevice, c->stream));
  CNA_TRY(launch_xb(c, (const double*)c->scratch, ldb, Nx, center != 0, c->X, ldx));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->ncorrs_valid = false;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  c->coef_early = false;
  c->fdr_inline = false;
  return 0;
}

// X <- (X - mean).M^T [/ std] [and ncorrs = X.y/N] for M = I - C.W given by its factors (C: N x r, W: r x N,
// both row-major): one row-local pass (rows.hip:k_resid_lowrank) instead of cna_resid_apply + cna_standardize
// + cna_ncorrs.  y == NULL: no coefficients.  r = 0: centring (and standardisation) only.
int cna_resid_lowrank(cna_ctx* c, const double* C, const double* W, int r, int center, int standardize, const double* y,
                      double* max_abs_out) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  if (r < 0 || (r > 0 && (!C || !W))) CNA_FAIL(CNA_EINVAL, "cna_resid_lowrank: bad factors");
  const int Nx = c->Nx;
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, carve_bytes({8 * (int64_t)std::max(r, 1) * Nx, 8 * (int64_t)std::max(r, 1) * Nx, 8 * (int64_t)Nx, 8 * 2049})));
  Carver cv(c->scratch);
  double* Wd = cv.take<double>((int64_t)std::max(r, 1) * Nx);
  double* Ctd = cv.take<double>((int64_t)std::max(r, 1) * Nx);
  double* yd = cv.take<double>(Nx);
  unsigned long long* mb = cv.take<unsigned long long>(2049);
  std::vector<double> Ct((size_t)std::max(r, 1) * Nx, 0.0);
  for (int i = 0; i < Nx; ++i)
    for (int k = 0; k < r; ++k) Ct[(size_t)k * Nx + i] = C[(size_t)i * r + k];
  if (r > 0) {
    HIP_TRY(hipMemcpyAsync(Wd, W, 8 * (size_t)r * Nx, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(Ctd, Ct.data(), 8 * (size_t)r * Nx, hipMemcpyHostToDevice, c->stream));
  }
  if (y) {
    void* np = c->ncorrs;
    CNA_TRY(dev_reserve(c, &np, &c->ncorrs_cap, 8 * std::max<int64_t>(c->nx, 1)));
    c->ncorrs = (double*)np;
    HIP_TRY(hipMemcpyAsync(yd, y, 8 * Nx, hipMemcpyHostToDevice, c->stream));
  }
  CNA_TRY(launch_resid_lowrank(c, Wd, Ctd, r, center, standardize, y ? yd : nullptr, mb));
  double m = 0.0;
  if (y) {
    CNA_TRY(comm_allreduce_f64_max(c, (double*)mb, 1));
    HIP_TRY(hipMemcpyAsync(&m, mb, 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));        // Ct is a local
  if (max_abs_out) *max_abs_out = m;
  c->ncorrs_valid = y != nullptr;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  c->coef_early = false;
  c->fdr_inline = false;
  return 0;
}

// One ridge of the schedule of _nam.py:142-156 in ONE pass over X and one wait: X <- (X - mean).M^T for M = I - C.W, the
// batch kurtosis of the result (_nam.py:150: its median decides whether the schedule ends here), then -- optimistically --
// the division by the std (_nam.py:159) and the coefficients X.y/N (_association.py:77).  *median_out: np.median of the
// batch kurtosis (formed on the device).  When it is <= 6 the schedule is over and X is final; otherwise the caller
// restores X (selection from the NAM) and continues with cna_resid_lowrank + cna_batch_kurtosis ridge by ridge.
int cna_resid_lowrank_bk(cna_ctx* c, const double* C, const double* W, int r, const double* y, double* max_abs_out,
                         const int32_t* batch_codes, int n_batches, double* median_out) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  if (r < 0 || (r > 0 && (!C || !W)) || !y || !batch_codes || n_batches < 1 || n_batches > 256 || !median_out)
    CNA_FAIL(CNA_EINVAL, "cna_resid_lowrank_bk: bad arguments");
  if (c->nx > c->n_pad) CNA_FAIL(CNA_EINVAL, "X larger than the stat buffer");
  const int Nx = c->Nx;
  std::vector<int32_t> order, boff(n_batches + 1, 0);
  for (int s = 0; s < Nx; ++s)
    if (batch_codes[s] >= 0 && batch_codes[s] < n_batches) boff[batch_codes[s] + 1]++;
  for (int b = 0; b < n_batches; ++b) boff[b + 1] += boff[b];
  order.resize(std::max(boff[n_batches], 1));
  std::vector<int32_t> cur(boff.begin(), boff.end() - 1);
  for (int s = 0; s < Nx; ++s)
    if (batch_codes[s] >= 0 && batch_codes[s] < n_batches) order[cur[batch_codes[s]]++] = s;
  const int64_t rn = 8 * (int64_t)std::max(r, 1) * Nx;
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap,
                      carve_bytes({rn, rn, 8 * (int64_t)Nx, 8 * 2049, 4 * (int64_t)order.size(), 4 * (n_batches + 1)})));
  Carver cv(c->scratch);
  double* Wd = cv.take<double>((int64_t)std::max(r, 1) * Nx);
  double* Ctd = cv.take<double>((int64_t)std::max(r, 1) * Nx);
  double* yd = cv.take<double>(Nx);
  unsigned long long* mb = cv.take<unsigned long long>(2049);
  int32_t* order_dev = cv.take<int32_t>(order.size());
  int32_t* boff_dev = cv.take<int32_t>(n_batches + 1);
  std::vector<double> Ct((size_t)std::max(r, 1) * Nx, 0.0);
  for (int i = 0; i < Nx; ++i)
    for (int k = 0; k < r; ++k) Ct[(size_t)k * Nx + i] = C[(size_t)i * r + k];
  if (r > 0) {
    HIP_TRY(hipMemcpyAsync(Wd, W, 8 * (size_t)r * Nx, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(Ctd, Ct.data(), 8 * (size_t)r * Nx, hipMemcpyHostToDevice, c->stream));
  }
  void* np = c->ncorrs;
  CNA_TRY(dev_reserve(c, &np, &c->ncorrs_cap, 8 * std::max<int64_t>(c->nx, 1)));
  c->ncorrs = (double*)np;
  HIP_TRY(hipMemcpyAsync(yd, y, 8 * Nx, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(order_dev, order.data(), 4 * order.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(boff_dev, boff.data(), 4 * (n_batches + 1), hipMemcpyHostToDevice, c->stream));
  CNA_TRY(launch_resid_lowrank(c, Wd, Ctd, r, 1, 1, yd, mb, order_dev, boff_dev, n_batches, c->stat));
  c->stat_space = CNA_MAT_X;
  CNA_TRY(comm_allreduce_f64_max(c, (double*)mb, 1));
  // the median of the batch kurtosis, on the device; its result and max |coefficient| come back with one wait
  const bool sharded = c->nranks > 1 || comm_active(c);
  unsigned long long* hist = nullptr;
  CNA_TRY(ensure_auto_state(c, &hist));
  const size_t sb = (auto_state_bytes() + 255) & ~(size_t)255;
  HIP_TRY(hipMemsetAsync(c->auto_state, 0, sb, c->stream));
  CNA_TRY(launch_auto_median(c, c->stat, c->nx, c->auto_state, hist, -1, 0, sharded));
  double m = 0.0, med = 0.0;
  HIP_TRY(hipMemcpyAsync(&m, mb, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(&med, (const char*)c->auto_state + auto_state_result_offset(), 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));        // (Ct, order, boff are locals)
  if (max_abs_out) *max_abs_out = m;
  *median_out = med;
  c->ncorrs_valid = true;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  c->coef_early = false;
  c->fdr_inline = false;
  return 0;
}

// Selection + QC + the first ridge in ONE pass over the NAM (round 6).  For the demo's call shape -- covariates AND batches
// (demo/demo.ipynb:149) with every cell and every sample kept in place and at most seven batches -- the three passes
// cna_batch_kurtosis(CNA_MAT_NAM) + cna_stat_qc, cna_select_checked and cna_resid_lowrank_bk read the NAM / X three times
// and write X twice; here the rows go NAM -> registers -> X once (rows16.hip:k_rowpass16<.., QC>), and the answers of all
// three come back with one wait:
//   *n_qc_failed   rows whose batch kurtosis is NaN (with <= 7 batches the only way to fail `kurtosis < max(6, 2 median)`,
//                  _nam.py:94-96: the kurtosis of so few batch means is below 6)
//   *n_zero        rows that are constant over the samples (_association.py:182)
//   *median_out    np.median of the batch kurtosis of the residualised rows (_nam.py:150); *max_abs_out as cna_resid_lowrank_bk
// All of n_qc_failed == 0, n_zero == 0, median <= 6: X is final (what the three calls leave).  Anything else: the caller takes
// the three calls (the NAM is untouched).  *done = 0: shape not covered, nothing was queued.
int cna_select_resid_bk(cna_ctx* c, const double* C, const double* W, int r, const double* y, double* max_abs_out,
                        const int32_t* batch_codes, int n_batches, double* median_out, int64_t* n_qc_failed, int64_t* n_zero,
                        int* done) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  if (!done || !median_out || !n_qc_failed || !n_zero || !y || !batch_codes) CNA_FAIL(CNA_EINVAL, "cna_select_resid_bk: null argument");
  *done = 0;
  if (!c->nam_valid && !c->nam_lazy) CNA_FAIL(CNA_ESTATE, "NAM not available");
  const int Nx = c->N;
  if (r < 1 || r > 16 || !C || !W || n_batches < 2 || n_batches > 7 || Nx < 2 || Nx > 128 || c->n_local < 1) return 0;
  for (int s = 0; s < Nx; ++s)
    if (batch_codes[s] < 0 || batch_codes[s] >= n_batches) return 0;      // (a sample of no batch: the general sequence)
  if (c->n_local > c->n_pad) return 0;
  CNA_TRY(need_nam(c));
  CNA_TRY(gram_pre_settle(c));
  std::vector<int32_t> order(Nx), boff(n_batches + 1, 0);
  for (int s = 0; s < Nx; ++s) boff[batch_codes[s] + 1]++;
  for (int b = 0; b < n_batches; ++b) boff[b + 1] += boff[b];
  std::vector<int32_t> cur(boff.begin(), boff.end() - 1);
  for (int s = 0; s < Nx; ++s) order[cur[batch_codes[s]]++] = s;
  c->byp_valid = false; c->gram_pre = false; c->x_ident = false; c->xq_valid = false;
  c->nx = c->n_local;
  c->Nx = Nx;
  c->ldx = x_ld(Nx);
  c->keep_idx = nullptr;
  void* xp = c->X;
  CNA_TRY(dev_reserve(c, &xp, &c->x_cap, (int64_t)sizeof(double) * std::max<int64_t>(c->nx, 1) * c->ldx));
  c->X = (double*)xp;
  void* np = c->ncorrs;
  CNA_TRY(dev_reserve(c, &np, &c->ncorrs_cap, 8 * std::max<int64_t>(c->nx, 1)));
  c->ncorrs = (double*)np;
  const int64_t rn = 8 * (int64_t)r * Nx;
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, carve_bytes({rn, rn, 8 * (int64_t)Nx, 8 * 2049, 4 * (int64_t)Nx, 4 * (n_batches + 1), 16})));
  Carver cv(c->scratch);
  double* Wd = cv.take<double>((int64_t)r * Nx);
  double* Ctd = cv.take<double>((int64_t)r * Nx);
  double* yd = cv.take<double>(Nx);
  unsigned long long* mb = cv.take<unsigned long long>(2049);
  int32_t* order_dev = cv.take<int32_t>(Nx);
  int32_t* boff_dev = cv.take<int32_t>(n_batches + 1);
  unsigned long long* counters = cv.take<unsigned long long>(2);
  std::vector<double> Ct((size_t)r * Nx, 0.0);
  for (int i = 0; i < Nx; ++i)
    for (int k = 0; k < r; ++k) Ct[(size_t)k * Nx + i] = C[(size_t)i * r + k];
  HIP_TRY(hipMemcpyAsync(Wd, W, 8 * (size_t)r * Nx, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(Ctd, Ct.data(), 8 * (size_t)r * Nx, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(yd, y, 8 * Nx, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(order_dev, order.data(), 4 * (size_t)Nx, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(boff_dev, boff.data(), 4 * (n_batches + 1), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(counters, 0, 16, c->stream));
  int rc;
  {
    ProfScope ps(c, CNA_K_RESID);
    rc = launch_rowpass16(c, c->nam, c->ld, c->X, c->ldx, c->nx, Nx, Wd, Ctd, r, 1, 1, 1, yd, mb, order_dev, boff_dev, n_batches,
                          c->stat, counters);
  }
  if (rc < 0) return rc;
  if (rc == 0) {                                   // (CNA_ROWPASS16=0 or a shape the sixteen-row pass does not take)
    HIP_TRY(hipStreamSynchronize(c->stream));      // (the uploads read locals)
    c->x_valid = false;
    return 0;
  }
  // every cell is a row of X, also where the NAM has padding columns beyond ldx
  c->stat_space = CNA_MAT_X;
  CNA_TRY(comm_allreduce_f64_max(c, (double*)mb, 1));
  CNA_TRY(comm_allreduce_i64_sum(c, (int64_t*)counters, 2));
  const bool sharded = c->nranks > 1 || comm_active(c);
  unsigned long long* hist = nullptr;
  CNA_TRY(ensure_auto_state(c, &hist));
  const size_t sb = (auto_state_bytes() + 255) & ~(size_t)255;
  HIP_TRY(hipMemsetAsync(c->auto_state, 0, sb, c->stream));
  CNA_TRY(launch_auto_median(c, c->stat, c->nx, c->auto_state, hist, -1, 0, sharded));
  double m = 0.0, med = 0.0;
  unsigned long long cnt[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(&m, mb, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(&med, (const char*)c->auto_state + auto_state_result_offset(), 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(cnt, counters, 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));        // (Ct, order, boff are locals)
  if (max_abs_out) *max_abs_out = m;
  *median_out = med;
  *n_qc_failed = (int64_t)cnt[0];
  *n_zero = (int64_t)cnt[1];
  c->x_valid = true;
  c->x_from_nam = true;
  c->ncorrs_valid = true;
  c->coef_early = false;
  c->fdr_inline = false;
  *done = 1;
  return 0;
}

int cna_standardize(cna_ctx* c, int center) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  CNA_TRY(launch_standardize(c, center));
  c->ncorrs_valid = false;
  c->xq_valid = false; c->byp_valid = false; c->x_ident = false; c->gram_pre = false;
  c->coef_early = false;
  c->fdr_inline = false;
  return 0;
}

int cna_gram_launch(cna_ctx* c) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  return gram_launch_now(c);
}
static int gram_launch_now(cna_ctx* c) {
  const int Nx = c->Nx;
  const bool pre = c->gram_pre && c->gram_pre_pending;     // taken under the walk's last step (ranged_last_step), X untouched since
  c->gram_pre = false;
  CNA_TRY(gram_pre_settle(c));
  CNA_TRY(gram_host_reserve(c, Nx));
  c->gram_mirrored = false;
  if (!pre) {
    void* g = c->gram_buf;
    CNA_TRY(dev_reserve(c, &g, &c->gram_cap, (int64_t)sizeof(double) * Nx * Nx));
    c->gram_buf = (double*)g;
    c->gram_mirror = gram_solo(c) ? (double*)c->h_gram : nullptr;
    const int rc = launch_gram(c, c->gram_buf);
    c->gram_mirror = nullptr;
    CNA_TRY(rc);
  }
  CNA_TRY(comm_allreduce_f64_sum(c, c->gram_buf, (size_t)Nx * Nx));
  return gram_host_finish(c, Nx);
}

// Only waits for the event and reads the pinned copy: safe to call while another host thread is blocked inside a
// long entry point of the same context (cna_null_local_resident).
int cna_gram_fetch(cna_ctx* c, double* G_out) {
  CHECK_CTX(c);
  if (c->gram_n < 1) CNA_FAIL(CNA_ESTATE, "cna_gram_fetch before cna_gram_launch");
  const int64_t bytes = (int64_t)sizeof(double) * c->gram_n * c->gram_n;
  HIP_TRY(hipEventSynchronize(c->gram_done));
  std::memcpy(G_out, c->h_gram, (size_t)bytes);
  return 0;
}

// Gram matrix -> leading eigenpairs -> F-tests queued, without the interpreter in between (round 5): what follows the
// Gram kernels on the critical path of a small block is sample-space work of ~0.5 ms, and three trips through Python
// between its pieces cost a fifth of that again.  The acceptance rule of the native pairs is tools/_nam.py's
// (_top_pcs_native): residual and orthogonality at rounding level, every leading gap wide enough for the individual
// vectors to be defined; otherwise *accepted = 0 and the caller takes LAPACK (G_out is valid either way).
int cna_gram_pcs_tests(cna_ctx* c, int kmax, const int32_t* ks, int K, int r, int use_native, double resid_tol, double gap_tol,
                       double* G_out, double* U_out, int* accepted) {
  CHECK_CTX(c);
  if (!G_out || !U_out || !accepted || !ks) CNA_FAIL(CNA_EINVAL, "cna_gram_pcs_tests: null argument");
  *accepted = 0;
  CNA_TRY(cna_gram_fetch(c, G_out));
  c->t_gram_fetched = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  const int n = c->gram_n;
  if (!use_native || n < 8 || kmax < 1 || 4 * kmax > n || kmax + 1 > 256) return 0;
  for (int64_t i = 0; i < (int64_t)n * n; ++i)
    if (!std::isfinite(G_out[i])) return 0;
  std::vector<double> lam((size_t)kmax + 1);
  double resid = 0.0, ortho = 0.0;
  if (cna_host_top_eig(G_out, n, kmax, U_out, lam.data(), &resid, &ortho) != 0) return 0;
  const double top = lam[0];
  if (!(top > 0.0)) return 0;
  for (int t = 0; t <= kmax; ++t)
    if (!std::isfinite(lam[t])) return 0;
  if (!(resid <= resid_tol * top && ortho <= resid_tol)) return 0;
  for (int t = 0; t < kmax; ++t)
    if (!(lam[t] - lam[t + 1] > gap_tol * top)) return 0;
  CNA_TRY(cna_global_test_launch(c, U_out, kmax, ks, K, r));
  *accepted = 1;
  return 0;
}

int cna_gram(cna_ctx* c, double* G_out) {
  CNA_TRY(cna_gram_launch(c));
  return cna_gram_fetch(c, G_out);
}

int cna_project(cna_ctx* c, const double* W, int n_w, double* out_local) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  if (n_w < 1) CNA_FAIL(CNA_EINVAL, "n_w < 1");
  const int Nx = c->Nx, ldx = c->ldx;
  const int ldb = round_up(n_w, 16);
  std::vector<double> B((size_t)ldx * ldb, 0.0);
  for (int k = 0; k < Nx; ++k)
    for (int j = 0; j < n_w; ++j) B[(size_t)k * ldb + j] = W[(size_t)k * n_w + j];
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, (int64_t)sizeof(double) * ldx * ldb));
  HIP_TRY(hipMemcpyAsync(c->scratch, B.data(), sizeof(double) * ldx * ldb, hipMemcpyHostToDevice, c->stream));
  CNA_TRY(dev_reserve(c, &c->scratch2, &c->scratch2_cap, (int64_t)sizeof(double) * std::max<int64_t>(c->nx, 1) * ldb));
  double* out = (double*)c->scratch2;
  CNA_TRY(launch_xb(c, (const double*)c->scratch, ldb, n_w, false, out, ldb));
  if (c->nx > 0)
    HIP_TRY(hipMemcpy2DAsync(out_local, sizeof(double) * n_w, out, sizeof(double) * ldb, sizeof(double) * n_w,
                             c->nx, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// cna_project with the result left on the device (rows of X x n_w), to be read with
// cna_fetch_rows(CNA_MAT_PROJ, ...) in whatever row order the caller wants
int cna_project_keep(cna_ctx* c, const double* W, int n_w) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  if (n_w < 1) CNA_FAIL(CNA_EINVAL, "n_w < 1");
  const int Nx = c->Nx, ldx = c->ldx;
  const int ldb = round_up(n_w, 16);
  std::vector<double> B((size_t)ldx * ldb, 0.0);
  for (int k = 0; k < Nx; ++k)
    for (int j = 0; j < n_w; ++j) B[(size_t)k * ldb + j] = W[(size_t)k * n_w + j];
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, (int64_t)sizeof(double) * ldx * ldb));
  HIP_TRY(hipMemcpyAsync(c->scratch, B.data(), sizeof(double) * ldx * ldb, hipMemcpyHostToDevice, c->stream));
  CNA_TRY(dev_reserve(c, &c->proj, &c->proj_cap, (int64_t)sizeof(double) * std::max<int64_t>(c->nx, 1) * ldb));
  CNA_TRY(launch_xb(c, (const double*)c->scratch, ldb, n_w, false, (double*)c->proj, ldb));
  HIP_TRY(hipStreamSynchronize(c->stream));               // B is a local
  c->proj_ld = ldb;
  c->proj_cols = n_w;
  c->proj_rows = c->nx;
  c->proj_valid = true;
  return 0;
}

// ------------------------------------------------------------------------- association
int cna_x_identity(cna_ctx* c, int* yes) {
  CHECK_CTX(c);
  if (!yes) CNA_FAIL(CNA_EINVAL, "cna_x_identity: yes is required");
  *yes = (c->x_valid && c->x_ident && !c->auto_pending) ? 1 : 0;
  return 0;
}

int cna_ncorrs(cna_ctx* c, const double* y, double* out_local, double* max_abs) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  void* np = c->ncorrs;
  CNA_TRY(dev_reserve(c, &np, &c->ncorrs_cap, 8 * std::max<int64_t>(c->nx, 1)));
  c->ncorrs = (double*)np;
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, carve_bytes({8 * (int64_t)c->Nx, 8 * 2049})));
  Carver cv(c->scratch);
  double* yd = cv.take<double>(c->Nx);
  unsigned long long* mb = cv.take<unsigned long long>(2049);
  HIP_TRY(hipMemcpyAsync(yd, y, 8 * c->Nx, hipMemcpyHostToDevice, c->stream));
  CNA_TRY(launch_ncorrs(c, yd, mb));
  CNA_TRY(comm_allreduce_f64_max(c, (double*)mb, 1));
  double m = 0.0;
  HIP_TRY(hipMemcpyAsync(&m, mb, 8, hipMemcpyDeviceToHost, c->stream));
  if (out_local && c->nx > 0)
    HIP_TRY(hipMemcpyAsync(out_local, c->ncorrs, 8 * c->nx, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (max_abs) *max_abs = m;
  c->ncorrs_valid = true;
  c->coef_early = false;
  c->fdr_inline = false;
  return 0;
}

static void guess_from_thr(const double* thr, int T, double* thr0, double* inv_step) {
  *thr0 = T > 0 ? thr[0] : 0.0;
  const double step = T > 1 ? thr[1] - thr[0] : 0.0;
  *inv_step = step > 0 ? 1.0 / step : 0.0;
}

// cut[t] = smallest double a >= 0 with fl(fl(a/N)^2) >= edges[t]: the reference's test on
// z^2 = (|x.yc|/N)^2 (_association.py:99, _stats.py:47-54) moved onto the raw dot product.  Both
// roundings are monotone, so a bisection over the bit patterns of the positive doubles finds the
// exact switch point.  The cuts are (nearly) an arithmetic progression cut0 + t*step; eps bounds,
// in steps, how far any cut is from that line (the kernel trusts floor((x-cut0)/step) outside
// +-eps of a cut and walks the table otherwise).
static void exact_cuts(const double* edges, int T, int Nx, std::vector<double>& cuts, double* cut0,
                       double* inv_step, double* eps) {
  cuts.resize(T);
  const double dn = (double)Nx;
  for (int t = 0; t < T; ++t) {
    const double e = edges[t];
    auto ok = [&](double a) { volatile double z = a / dn; volatile double z2 = z * z; return z2 >= e; };
    if (e <= 0.0 || ok(0.0)) { cuts[t] = 0.0; continue; }
    // bracket the switch point around N*sqrt(e) (a few ulps wide), then bisect the bit patterns
    uint64_t lo = 0, hi = 0x7ff0000000000000ull;          // lo fails, hi (+inf) passes
    const double guess = dn * std::sqrt(e);
    uint64_t g;
    std::memcpy(&g, &guess, 8);
    if (guess > 0.0 && g > 64 && g < hi - 64) {
      double a;
      uint64_t b = g - 64;
      std::memcpy(&a, &b, 8);
      if (!ok(a)) lo = b;
      b = g + 64;
      std::memcpy(&a, &b, 8);
      if (ok(a)) hi = b;
    }
    while (hi - lo > 1) {
      const uint64_t mid = lo + (hi - lo) / 2;
      double a;
      std::memcpy(&a, &mid, 8);
      if (ok(a)) hi = mid; else lo = mid;
    }
    std::memcpy(&cuts[t], &hi, 8);
  }
  *cut0 = cuts[0];
  *inv_step = 0.0;
  *eps = 2.0;                                             // eps >= 1: always walk the table
  if (T >= 3 && cuts[T - 1] > cuts[0]) {
    const double step = (cuts[T - 1] - cuts[0]) / (T - 1);
    double dev = 0.0;
    for (int t = 0; t < T; ++t) dev = std::max(dev, std::fabs(cuts[t] - (*cut0 + t * step)) / step);
    *inv_step = 1.0 / step;
    *eps = 2.0 * dev + 1e-9;
    if (!(*eps < 0.25)) *eps = 2.0;
  }
}

static int ensure_zc(cna_ctx* c, int N, int P, hipStream_t st) {
  const int ldy = round_up(P, 64) + 64;   // one spare tile: a resident read may start at any column
  // = ldx of a working matrix with N samples, INCLUDING the bank-spreading pad quad x_ld() adds at
  // N = 157...160 / 189...192 / 221...224: the local-null kernel loads ldx rows of Zc (the matching
  // pad columns of X are zero, but 0 * whatever-lies-past-the-buffer is only 0 while that is finite)
  const int rows = x_ld(N);
  void* p = c->zc;
  CNA_TRY(dev_reserve(c, &p, &c->zc_cap, (int64_t)sizeof(double) * rows * ldy));
  c->zc = (double*)p;
  HIP_TRY(hipMemsetAsync(c->zc, 0, sizeof(double) * rows * ldy, st));   // zero pads (rows >= N, cols >= P)
  c->zc_ld = ldy;
  c->zc_cols = P;
  c->zc_rows = N;
  return 0;
}

// One local-null pass in two halves.  prepare: everything that needs only the thresholds (exact cuts,
// their upload, the threshold counts of the observed coefficients) -- the caller can issue it while
// the permuted phenotypes are still on their way.  go: the kernel and its reductions; results land in
// the pinned buffer h_res ([T sums][P*T tails if requested][2T observed counts]) and null_done fires
// when they are there.  Nothing else may use c->scratch between the two.
static int null_local_prepare(cna_ctx* c, int P, const double* edges, int T, int want_tails, const double* thr) {
  if (c->bins_pending) {          // the per-cell counts of the previous pass read its thresholds out of c->scratch (coef_stream)
    HIP_TRY(hipStreamWaitEvent(c->stream, c->bins_copied, 0));
    c->bins_pending = false;
  }
  if (P < 1 || T < 1) CNA_FAIL(CNA_EINVAL, "cna_null_local: P and T must be positive");
  if (c->null_pending) CNA_FAIL(CNA_ESTATE, "a local-null pass is still pending: fetch it first");
  for (int t = 1; t < T; ++t)
    if (!(edges[t] >= edges[t - 1])) CNA_FAIL(CNA_EINVAL, "cna_null_local: edges must ascend");
  c->null_prepared = 0;
  c->fdr_inline = false;
  std::vector<double> cuts;
  exact_cuts(edges, T, c->Nx, cuts, &c->null_cut0, &c->null_inv_step, &c->null_eps);
  const int64_t obs_off = 8 * (int64_t)T + (want_tails ? 8 * (int64_t)P * T : 0);
  const int64_t stage_off = obs_off + 16 * (int64_t)T;      // pinned copies of cuts | edges | thr: uploads need no wait
  const int64_t hbytes = stage_off + 24 * (int64_t)T + 16;  // (+ the integer pass's status word, see null_local_go)
  HIP_TRY(hipEventSynchronize(c->stage_done));            // uploads of the previous pass out of the staging area (long done)
  if (hbytes > c->h_res_cap) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->h_res) HIP_TRY(hipHostFree(c->h_res));
    c->h_res = nullptr;
    HIP_TRY(hipHostMalloc(&c->h_res, (size_t)hbytes, hipHostMallocDefault));
    c->h_res_cap = hbytes;
  }
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap,
                      carve_bytes({8 * (int64_t)T, 8 * (int64_t)P * T, 8 * (int64_t)P * T, 8 * (int64_t)T, 8 * (int64_t)T,
                                   8 * (int64_t)T, 16 * (int64_t)T, 16 * (int64_t)T})));
  Carver cv(c->scratch);
  double* ed = cv.take<double>(T);
  cv.take<unsigned long long>((int64_t)P * T);
  cv.take<int64_t>((int64_t)P * T);
  cv.take<int64_t>(T);
  double* oed = cv.take<double>(T);
  double* otd = cv.take<double>(T);
  unsigned long long* ohist = cv.take<unsigned long long>(2 * (int64_t)T);
  int64_t* otails = cv.take<int64_t>(2 * (int64_t)T);
  c->null_has_obs = 0;
  if (thr) {
    // threshold counts of the observed coefficients (cna_obs_counts) ride along: tiny kernels in front
    // of the long one, results in the same pinned block
    if (!c->ncorrs_valid) CNA_FAIL(CNA_ESTATE, "threshold counts need cna_ncorrs");
    double thr0, ostep;
    guess_from_thr(thr, T, &thr0, &ostep);
    c->null_thr0 = thr0;
    c->null_thr_step = ostep;
    double* st = (double*)((char*)c->h_res + stage_off);
    std::memcpy(st + T, edges, 8 * (size_t)T);
    std::memcpy(st + 2 * T, thr, 8 * (size_t)T);
    HIP_TRY(hipMemcpyAsync(oed, st + T, 8 * T, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(otd, st + 2 * T, 8 * T, hipMemcpyHostToDevice, c->stream));
    CNA_TRY(launch_obs_counts(c, oed, otd, T, thr0, ostep, ohist));
    CNA_TRY(comm_allreduce_i64_sum(c, (int64_t*)ohist, (size_t)2 * T));
    CNA_TRY(launch_suffix_sum(c, ohist, 2, T, otails));
    HIP_TRY(hipMemcpyAsync((char*)c->h_res + obs_off, otails, 16 * (size_t)T, hipMemcpyDeviceToHost, c->stream));
    c->null_has_obs = 1;
    c->null_obs_off = obs_off;
  }
  std::memcpy((char*)c->h_res + stage_off, cuts.data(), 8 * (size_t)T);
  HIP_TRY(hipMemcpyAsync(ed, (char*)c->h_res + stage_off, 8 * T, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(c->stage_done, c->stream));
  c->null_P = P;
  c->null_T = T;
  c->null_has_tails = want_tails;
  c->null_prepared = 1;
  return 0;
}

static int null_local_go(cna_ctx* c, int col0) {
  if (!c->null_prepared) CNA_FAIL(CNA_ESTATE, "local-null pass not prepared");
  c->null_prepared = 0;
  const int P = c->null_P, T = c->null_T;
  if (!c->zc || col0 < 0 || col0 + P > c->zc_cols || c->zc_rows != c->Nx ||
      c->zc_cap < (int64_t)sizeof(double) * c->ldx * c->zc_ld)
    CNA_FAIL(CNA_ESTATE, "no conditioned phenotypes resident for these columns");
  Carver cv(c->scratch);                                   // same carve as in null_local_prepare
  double* ed = cv.take<double>(T);
  unsigned long long* hist = cv.take<unsigned long long>((int64_t)P * T);
  int64_t* tails = cv.take<int64_t>((int64_t)P * T);
  int64_t* sums = cv.take<int64_t>(T);
  // columns beyond col0+P inside the last 64-wide tile are other phenotypes: the kernel only
  // flushes counters of p < P, and reads stay inside the zero-padded leading dimension
  // Only the sums over permutations wanted (the analysis): the integer matrix cores do the products
  // (null_i8.hip: exact counts, outputs near a cut rechecked in f64).
  // The per-cell half of the FDR lookup -- how many thresholds lie at or below |coef_i| -- needs nothing from the null: its
  // kernel goes IN FRONT of the null on the main stream (beside it, on the coefficient stream, it was starved for the
  // whole pass: 2.1 ms at 2M cells, profiles/r06_kernel_stats_C4.csv), its 2 bytes per cell cross PCIe under the null
  // from the coefficient stream, and the host puts table and counts together (cna_percell_fdr_copy_early).
  const bool inline_fdr = c->coef_early && c->null_has_obs && T <= 512;
  if (inline_fdr) {
    Carver cw(c->scratch);
    cw.take<double>(T);
    cw.take<unsigned long long>((int64_t)P * T);
    cw.take<int64_t>((int64_t)P * T);
    cw.take<int64_t>(T);
    cw.take<double>(T);                                    // oed
    double* otd = cw.take<double>(T);
    const int64_t n_out = c->local_view ? c->n_local : c->n_global;
    void* bp = c->bins_dev;
    CNA_TRY(dev_reserve(c, &bp, &c->bins_cap, 2 * std::max<int64_t>(std::max(c->n_pad, n_out), 1)));
    c->bins_dev = (unsigned short*)bp;
    if (2 * n_out > c->h_bins_cap) {
      if (c->h_bins) HIP_TRY(hipHostFree(c->h_bins));
      c->h_bins = nullptr;
      HIP_TRY(hipHostMalloc((void**)&c->h_bins, (size_t)std::max<int64_t>(2 * n_out, 64), hipHostMallocDefault));
      c->h_bins_cap = 2 * n_out;
    }
    hipStream_t cs = c->coef_stream;
    CNA_TRY(launch_percell_bins(c, c->stream, c->coef_dev, otd, T, c->null_thr0, c->null_thr_step, c->bins_dev));
    HIP_TRY(hipEventRecord(c->stage_done, c->stream));      // (a later point of the stream than the one prepare recorded)
    HIP_TRY(hipStreamWaitEvent(cs, c->stage_done, 0));
    if (n_out > 0) HIP_TRY(hipMemcpyAsync(c->h_bins, c->bins_dev, 2 * (size_t)n_out, hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipEventRecord(c->bins_copied, cs));
    c->bins_pending = true;
  }
  int* i8_status = nullptr;
  int64_t* i8_sums = nullptr;
  c->i8_last = false;
  if (!c->null_has_tails && null_i8_eligible(c, P, T, c->null_cut0, c->null_inv_step, c->null_eps))
    CNA_TRY(launch_null_local_i8(c, c->zc + col0, c->zc_ld, P, ed, T, c->null_cut0, c->null_inv_step, c->null_eps,
                                 &i8_sums, &i8_status));
  c->i8_last = i8_status != nullptr;
  c->null_col0 = col0;
  c->null_status_off = -1;
  if (i8_status) {
    // Round 6: the f64 kernel is no longer queued behind the integer pass as a stand-by (two guarded launches, the
    // reductions of an empty histogram and the pick: 45 us of device time and seven launches on the tail of EVERY call,
    // rocprofv3 timeline profiles/r06_timeline_C2.txt).  The pass's status word travels with its sums; should it be
    // raised (recheck queue overflow: never seen outside the test that forces it) whoever collects the pass runs the f64
    // kernel then (null_local_collect).  Several ranks: the word is summed over the ranks, so that all of them decide alike.
    sums = i8_sums;
    CNA_TRY(comm_allreduce_i64_sum(c, sums, (size_t)T));
    int64_t* stw = (int64_t*)i8_status;                    // (the word's upper half is zero: launch_null_local_i8 clears the block)
    CNA_TRY(comm_allreduce_i64_sum(c, stw, 1));
    c->null_status_off = 8 * (int64_t)T + 16 * (int64_t)T + 24 * (int64_t)T;      // behind sums | observed counts | staging (no tails here)
    HIP_TRY(hipMemcpyAsync((char*)c->h_res + c->null_status_off, stw, 8, hipMemcpyDeviceToHost, c->stream));
  } else {
    CNA_TRY(launch_null_local(c, c->zc + col0, c->zc_ld, P, ed, T, c->null_cut0, c->null_inv_step, c->null_eps, hist, nullptr));
    // suffix sums and the sum over permutations are linear: when only the sums are wanted the ranks
    // exchange T integers instead of the P x T histogram
    if (c->null_has_tails) CNA_TRY(comm_allreduce_i64_sum(c, (int64_t*)hist, (size_t)P * T));
    CNA_TRY(launch_suffix_sum(c, hist, P, T, tails));
    CNA_TRY(launch_tail_sums(c, tails, P, T, sums));
    if (!c->null_has_tails) CNA_TRY(comm_allreduce_i64_sum(c, sums, (size_t)T));
  }
  HIP_TRY(hipMemcpyAsync(c->h_res, sums, 8 * (size_t)T, hipMemcpyDeviceToHost, c->stream));
  if (inline_fdr) {
    // the caller already has the coefficient column (cna_percell_coef_launch): the FDR column can follow the null without
    // the host in between -- behind the null only the FDR table is formed from the tail sums and the observed counts
    // (still in the scratch carve of the prepare half) and sent (2.4 KB).  (Round 2 stored the finished 8-byte column
    // from a kernel behind the null: 16 MB over PCIe on the critical path at 2M cells.)
    cv.take<double>(T);                                    // oed
    cv.take<double>(T);                                    // otd
    cv.take<unsigned long long>(2 * (int64_t)T);           // ohist
    int64_t* otails = cv.take<int64_t>(2 * (int64_t)T);    // [ranks | num_detected]
    double* tab = c->coef_dev + 4 * c->n_pad;
    CNA_TRY(launch_fdr_table(c, sums, otails, T, P, tab, tab + 512));
    if (!c->h_tab) HIP_TRY(hipHostMalloc((void**)&c->h_tab, 8 * 512, hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(c->h_tab, tab + 512, 8 * (size_t)T, hipMemcpyDeviceToHost, c->stream));
    c->fdr_inline = true;
  }
  if (c->null_has_tails)
    HIP_TRY(hipMemcpyAsync((char*)c->h_res + 8 * (size_t)T, tails, 8 * (size_t)P * T, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipEventRecord(c->null_done, c->stream));
  c->fdr_early_copied = false;
  c->fdr_early_dst = nullptr;
  c->fdr_early_served = false;
  c->null_pending = 1;
  return 0;
}

static int null_local_queue(cna_ctx* c, int col0, int P, const double* edges, int T, int want_tails,
                            const double* thr = nullptr) {
  CNA_TRY(null_local_prepare(c, P, edges, T, want_tails, thr));
  return null_local_go(c, col0);
}

static int null_local_collect(cna_ctx* c, int64_t* tails_out, int64_t* sums_out, int64_t* ranks_out = nullptr,
                              int64_t* numdet_out = nullptr) {
  if (!c->null_pending) CNA_FAIL(CNA_ESTATE, "no local-null pass pending");
  c->null_pending = 0;
  HIP_TRY(hipEventSynchronize(c->null_done));
  if (c->null_status_off >= 0 && *(volatile int64_t*)((char*)c->h_res + c->null_status_off) != 0) {
    // the integer pass gave up (on some rank): the same counts from the f64 kernel, now (the thresholds of the prepare
    // half are still in the scratch carve: nothing that uses it may run between launch and fetch); the FDR table that
    // followed the pass on the device was made of the wrong sums -- the per-cell column is looked up again
    c->null_status_off = -1;
    const int P = c->null_P, T = c->null_T;
    Carver cv(c->scratch);
    double* ed = cv.take<double>(T);
    unsigned long long* hist = cv.take<unsigned long long>((int64_t)P * T);
    int64_t* tails = cv.take<int64_t>((int64_t)P * T);
    int64_t* sums = cv.take<int64_t>(T);
    CNA_TRY(launch_null_local(c, c->zc + c->null_col0, c->zc_ld, P, ed, T, c->null_cut0, c->null_inv_step, c->null_eps, hist, nullptr));
    CNA_TRY(launch_suffix_sum(c, hist, P, T, tails));
    CNA_TRY(launch_tail_sums(c, tails, P, T, sums));
    CNA_TRY(comm_allreduce_i64_sum(c, sums, (size_t)T));
    HIP_TRY(hipMemcpyAsync(c->h_res, sums, 8 * (size_t)T, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->fdr_inline = false;
  }
  c->null_status_off = -1;
  if (ranks_out || numdet_out) {
    if (!c->null_has_obs) CNA_FAIL(CNA_EINVAL, "the pending pass was launched without thresholds");
    const char* o = (const char*)c->h_res + c->null_obs_off;
    if (ranks_out) std::memcpy(ranks_out, o, 8 * (size_t)c->null_T);
    if (numdet_out) std::memcpy(numdet_out, o + 8 * (size_t)c->null_T, 8 * (size_t)c->null_T);
  }
  if (sums_out) std::memcpy(sums_out, c->h_res, 8 * (size_t)c->null_T);
  if (tails_out) {
    if (!c->null_has_tails) CNA_FAIL(CNA_EINVAL, "the pending pass was launched without want_tails");
    std::memcpy(tails_out, (char*)c->h_res + 8 * (size_t)c->null_T, 8 * (size_t)c->null_P * c->null_T);
  }
  return 0;
}

static int null_local_on_resident(cna_ctx* c, int col0, int P, const double* edges, int T, int64_t* tails_out,
                                  int64_t* sums_out) {
  CNA_TRY(null_local_queue(c, col0, P, edges, T, tails_out != nullptr));
  return null_local_collect(c, tails_out, sums_out);
}

int cna_null_local_prepare(cna_ctx* c, int P, const double* edges, int T, int want_tails, const double* thr) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  if (!edges) CNA_FAIL(CNA_EINVAL, "cna_null_local_prepare: edges required");
  return null_local_prepare(c, P, edges, T, want_tails, thr);
}

int cna_null_local_launch(cna_ctx* c, int col0, int P, const double* edges, int T, int want_tails, const double* thr) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  if (!edges) {                                            // second half of a prepared pass
    if (!c->null_prepared || P != c->null_P || T != c->null_T)
      CNA_FAIL(CNA_ESTATE, "cna_null_local_launch without edges needs a matching cna_null_local_prepare");
    return null_local_go(c, col0);
  }
  return null_local_queue(c, col0, P, edges, T, want_tails, thr);
}

int cna_null_local_fetch(cna_ctx* c, int64_t* tails_out, int64_t* tail_sums_out, int64_t* ranks_out,
                         int64_t* num_detected_out) {
  CHECK_CTX(c);
  return null_local_collect(c, tails_out, tail_sums_out, ranks_out, num_detected_out);
}

// A pass that was launched and never collected (the caller raised between launch and fetch: a bad `ks`, a failed
// draw, Ctrl-C): wait for its kernels, drop its results and the prepared half, so that the next analysis on this
// context starts clean instead of failing with "still pending".  A no-op when nothing is pending.
int cna_null_local_discard(cna_ctx* c) {
  CHECK_CTX(c);
  c->null_prepared = 0;
  if (!c->null_pending) return 0;
  c->null_pending = 0;
  HIP_TRY(hipEventSynchronize(c->null_done));
  return 0;
}

int cna_null_local_i8_stats(cna_ctx* c, int* used_out, int64_t* rechecked_out, int* fallback_out) {
  CHECK_CTX(c);
  unsigned long long v[2] = {0, 0};
  if (c->i8_last) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(v, c->i8_qcount, 16, hipMemcpyDeviceToHost));
  }
  if (used_out) *used_out = c->i8_last ? 1 : 0;
  if (rechecked_out) *rechecked_out = (int64_t)v[0];
  if (fallback_out) *fallback_out = (int)(v[1] & 0xffffffffull) != 0;
  return 0;
}

int cna_null_local(cna_ctx* c, const double* Yc, int P, const double* edges, int T, int64_t* tails_out) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  if (P < 1) CNA_FAIL(CNA_EINVAL, "cna_null_local: P must be positive");
  CNA_TRY(ensure_zc(c, c->Nx, P, c->stream));
  HIP_TRY(hipMemcpy2DAsync(c->zc, sizeof(double) * c->zc_ld, Yc, sizeof(double) * P, sizeof(double) * P, c->Nx,
                           hipMemcpyHostToDevice, c->stream));
  return null_local_on_resident(c, 0, P, edges, T, tails_out, nullptr);
}

int cna_null_local_resident(cna_ctx* c, int col0, int P, const double* edges, int T, int64_t* tails_out,
                            int64_t* tail_sums_out) {
  CHECK_CTX(c);
  if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
  return null_local_on_resident(c, col0, P, edges, T, tails_out, tail_sums_out);
}

int cna_condition_phenotypes(cna_ctx* c, const double* M, const double* Y, int N, int P) {
  CHECK_CTX(c);
  if (P < 1) CNA_FAIL(CNA_EINVAL, "P must be positive");
  if (N < 2) CNA_FAIL(CNA_EINVAL, "need at least two samples");
  if (c->null_pending) CNA_FAIL(CNA_ESTATE, "a local-null pass is still pending: fetch it first");
  // Sample-space only, so it runs on the second stream: the caller may issue it while the diffusion
  // kernels of the same analysis are still executing on the main stream.  Synchronised before
  // returning, hence complete for every later consumer on either stream.
  hipStream_t st = c->copy_stream;
  CNA_TRY(ensure_zc(c, N, P, st));
  void* g = c->gt;
  CNA_TRY(dev_reserve(c, &g, &c->gt_cap, carve_bytes({8 * (int64_t)N * N, 8 * (int64_t)N * P})));
  c->gt = g;
  Carver cv(c->gt);
  double* Md = cv.take<double>((int64_t)N * N);
  double* Yd = cv.take<double>((int64_t)N * P);
  HIP_TRY(hipMemcpyAsync(Md, M, 8 * (size_t)N * N, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(Yd, Y, 8 * (size_t)N * P, hipMemcpyHostToDevice, st));
  CNA_TRY(launch_condition(c, st, Md, Yd, N, P, c->zc, c->zc_ld));
  HIP_TRY(hipStreamSynchronize(st));               // host buffers may be released
  return 0;
}

// The global F-tests in two halves: launch stages U and ks in pinned memory, queues upload, kernels
// and the copy of the three result vectors back into the pinned block on the second stream, and
// returns; fetch waits for them.  The caller can do host work (write the coefficient column) between.
int cna_global_test_launch(cna_ctx* c, const double* U, int kmax, const int32_t* ks, int K, int r) {
  CHECK_CTX(c);
  if (!c->zc || c->zc_cols < 1 || c->zc_rows != c->Nx) CNA_FAIL(CNA_ESTATE, "cna_global_test needs cna_condition_phenotypes");
  const int N = c->Nx, P = c->zc_cols;
  if (kmax < 1 || kmax > N || K < 1) CNA_FAIL(CNA_EINVAL, "cna_global_test: bad kmax / K");
  for (int a = 0; a < K; ++a)
    if (ks[a] < 1 || ks[a] > kmax) CNA_FAIL(CNA_EINVAL, "cna_global_test: ks must lie in [1, kmax]");
  if (c->gt_pending_P) CNA_FAIL(CNA_ESTATE, "a global test is still pending: fetch it first");
  void* g = c->gt;
  const int64_t nwork = global_test_scratch_doubles(P, kmax, K);
  CNA_TRY(dev_reserve(c, &g, &c->gt_cap, carve_bytes({8 * (int64_t)N * kmax, 4 * (int64_t)K, 8 * (int64_t)P, 8 * (int64_t)P, 4 * (int64_t)P, 8 * nwork})));
  c->gt = g;
  Carver cv(c->gt);
  double* Ud = cv.take<double>((int64_t)N * kmax);
  int32_t* ksd = cv.take<int32_t>(K);
  double* mp = cv.take<double>(P);
  double* r2 = cv.take<double>(P);
  int32_t* ki = cv.take<int32_t>(P);
  double* work = cv.take<double>(nwork);
  const int64_t off_ks = 8 * (int64_t)N * kmax;
  const int64_t off_out = round_up(off_ks + 4 * (int64_t)K, 64);
  const int64_t need = off_out + 20 * (int64_t)P + 64;
  if (need > c->h_gt_cap) {
    if (c->h_gt) HIP_TRY(hipHostFree(c->h_gt));
    c->h_gt = nullptr;
    HIP_TRY(hipHostMalloc(&c->h_gt, (size_t)need, hipHostMallocDefault));
    c->h_gt_cap = need;
  }
  char* h = (char*)c->h_gt;
  std::memcpy(h, U, 8 * (size_t)N * kmax);
  std::memcpy(h + off_ks, ks, 4 * (size_t)K);
  // Runs on the copy stream: Zc is complete (cna_condition_phenotypes synchronised) and only read, so
  // these tiny kernels need not queue behind a local-null pass in flight on the main stream.
  hipStream_t st = c->copy_stream;
  HIP_TRY(hipMemcpyAsync(Ud, h, 8 * (size_t)N * kmax, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(ksd, h + off_ks, 4 * (size_t)K, hipMemcpyHostToDevice, st));
  CNA_TRY(launch_global_test(c, st, c->zc, c->zc_ld, N, P, Ud, kmax, ksd, K, r, work, mp, r2, ki));
  HIP_TRY(hipMemcpyAsync(h + off_out, mp, 8 * (size_t)P, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(h + off_out + 8 * (size_t)P, r2, 8 * (size_t)P, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(h + off_out + 16 * (size_t)P, ki, 4 * (size_t)P, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipEventRecord(c->gt_done, st));
  c->gt_pending_P = P;
  c->gt_off_out = off_out;
  return 0;
}

int cna_global_test_fetch(cna_ctx* c, double* minp_out, double* r2_out, int32_t* kidx_out) {
  CHECK_CTX(c);
  if (!c->gt_pending_P) CNA_FAIL(CNA_ESTATE, "no global test pending");
  const size_t P = (size_t)c->gt_pending_P;
  c->gt_pending_P = 0;
  HIP_TRY(hipEventSynchronize(c->gt_done));
  const char* o = (const char*)c->h_gt + c->gt_off_out;
  if (minp_out) std::memcpy(minp_out, o, 8 * P);
  if (r2_out) std::memcpy(r2_out, o + 8 * P, 8 * P);
  if (kidx_out) std::memcpy(kidx_out, o + 16 * P, 4 * P);
  return 0;
}

int cna_global_test(cna_ctx* c, const double* U, int kmax, const int32_t* ks, int K, int r, double* minp_out,
                    double* r2_out, int32_t* kidx_out) {
  CNA_TRY(cna_global_test_launch(c, U, kmax, ks, K, r));
  return cna_global_test_fetch(c, minp_out, r2_out, kidx_out);
}

int cna_obs_counts(cna_ctx* c, const double* edges, const double* thr, int T, int64_t* ranks_out,
                   int64_t* num_detected_out) {
  CHECK_CTX(c);
  if (!c->ncorrs_valid) CNA_FAIL(CNA_ESTATE, "cna_obs_counts needs cna_ncorrs");
  if (T < 1) CNA_FAIL(CNA_EINVAL, "T < 1");
  double thr0, inv_step;
  guess_from_thr(thr, T, &thr0, &inv_step);
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap, carve_bytes({8 * (int64_t)T, 8 * (int64_t)T, 16 * (int64_t)T, 16 * (int64_t)T})));
  Carver cv(c->scratch);
  double* ed = cv.take<double>(T);
  double* td = cv.take<double>(T);
  unsigned long long* hist = cv.take<unsigned long long>(2 * T);
  int64_t* tails = cv.take<int64_t>(2 * T);
  HIP_TRY(hipMemcpyAsync(ed, edges, 8 * T, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(td, thr, 8 * T, hipMemcpyHostToDevice, c->stream));
  CNA_TRY(launch_obs_counts(c, ed, td, T, thr0, inv_step, hist));
  CNA_TRY(comm_allreduce_i64_sum(c, (int64_t*)hist, (size_t)2 * T));
  CNA_TRY(launch_suffix_sum(c, hist, 2, T, tails));
  std::vector<int64_t> h(2 * T);
  HIP_TRY(hipMemcpyAsync(h.data(), tails, 16 * T, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (ranks_out) std::memcpy(ranks_out, h.data(), 8 * T);
  if (num_detected_out) std::memcpy(num_detected_out, h.data() + T, 8 * T);
  return 0;
}

static int ensure_cell_pinned(cna_ctx* c, int64_t n_out) {
  const int64_t need = 16 * std::max<int64_t>(n_out, 1);
  if (need > c->h_cell_cap) {
    if (c->h_cell) HIP_TRY(hipHostFree(c->h_cell));
    c->h_cell = nullptr;
    c->coef_early = false;
  c->fdr_inline = false;
    HIP_TRY(hipHostMalloc(&c->h_cell, (size_t)need, hipHostMallocDefault));
    c->h_cell_cap = need;
  }
  return 0;
}

// The coefficient column of the result (data.obs[key_added], _association.py:230-233) depends on
// the observed phenotype only, not on the permutation null: queued here -- ahead of the local-null
// kernel in stream order, copied out on the second stream -- it reaches the host while that kernel
// runs, and the caller can write the column before the null is even finished.
int cna_percell_coef_launch(cna_ctx* c) {
  CHECK_CTX(c);
  if (!c->ncorrs_valid || !c->x_from_nam) CNA_FAIL(CNA_ESTATE, "cna_percell_coef_launch needs cna_select + cna_ncorrs");
  if ((c->nranks > 1 || comm_active(c)) && !c->local_view)
    CNA_FAIL(CNA_ESTATE, "cna_percell_coef_launch: replicated multi-rank outputs are assembled by cna_percell_fdr");
  const int64_t n_out = c->local_view ? c->n_local : c->n_global;
  CNA_TRY(ensure_cell_pinned(c, n_out));
  void* p = c->coef_dev;
  CNA_TRY(dev_reserve(c, &p, &c->coef_dev_cap, 32 * std::max<int64_t>(c->n_pad, 1) + 16 * 512));   // coef, coef_u, fdr, fdr_u, FDR table
  c->coef_dev = (double*)p;
  double* tmp = c->coef_dev;
  double* out = tmp;
  CNA_TRY(launch_percell_fdr(c, nullptr, nullptr, 0, 0.0, 0.0, tmp, nullptr));
  if (c->orig_idx) {
    out = c->coef_dev + c->n_pad;
    CNA_TRY(launch_unpermute2(c, tmp, nullptr, c->orig_idx, c->n_local, out, nullptr));
  }
  HIP_TRY(hipEventRecord(c->coef_ready, c->stream));
  hipStream_t cs = c->coef_stream;   // not copy_stream: the helper thread's conditioning call waits on that one
  HIP_TRY(hipStreamWaitEvent(cs, c->coef_ready, 0));
  if (n_out > 0)
    HIP_TRY(hipMemcpyAsync(c->h_cell, out, 8 * n_out, hipMemcpyDeviceToHost, cs));
  HIP_TRY(hipEventRecord(c->coef_copied, cs));
  c->coef_early = true;
  return 0;
}

int cna_percell_coef_wait(cna_ctx* c, double** coef_ptr) {
  CHECK_CTX(c);
  if (!coef_ptr) CNA_FAIL(CNA_EINVAL, "cna_percell_coef_wait: coef_ptr is required");
  if (!c->coef_early) CNA_FAIL(CNA_ESTATE, "cna_percell_coef_wait without cna_percell_coef_launch");
  HIP_TRY(hipEventSynchronize(c->coef_copied));
  *coef_ptr = (double*)c->h_cell;
  return 0;
}

// The FDR column of the pending local-null pass, copied into the caller's own storage as soon as the device has
// stored it in the pinned block -- meant for a helper thread while the main thread is busy on the host (the
// samples x samples SVD outlasts the local null by ~0.5 ms at 2M x 200 and the copy of 16 MB takes 0.4 ms).
// Touches nothing but the event, the pinned block and one flag.  *done = 0: not applicable (the column does not
// follow this pass on the device), nothing was copied.
int cna_percell_fdr_copy_early(cna_ctx* c, double* dst, int64_t n, int nthreads, int* done) {
  CHECK_CTX(c);
  if (!dst || !done) CNA_FAIL(CNA_EINVAL, "cna_percell_fdr_copy_early: dst and done are required");
  *done = 0;
  const int64_t n_out = c->local_view ? c->n_local : c->n_global;
  if (!c->coef_early || !c->fdr_inline || !c->null_pending || !c->h_cell || n != n_out) return 0;
  // (the main thread's cna_percell_fdr_pinned waits while this one is at work instead of doing the same work again)
  struct Flight { std::atomic<int>& f; Flight(std::atomic<int>& f_) : f(f_) { f.store(1); } ~Flight() { f.store(0); } } flight(c->fdr_early_inflight);
  HIP_TRY(hipEventSynchronize(c->bins_copied));
  HIP_TRY(hipEventSynchronize(c->null_done));
  if (c->null_status_off >= 0 && *(volatile int64_t*)((char*)c->h_res + c->null_status_off) != 0)
    return 0;                                   // the integer pass gave up: its table is void (cna_null_local_fetch reruns in f64)
  if (cna_host_expand_u16(dst, c->h_bins, n, c->h_tab, c->null_T, nthreads) != 0)
    CNA_FAIL(CNA_ESTATE, "cna_percell_fdr_copy_early: expansion failed");
  c->fdr_early_dst = dst;
  c->fdr_early_copied = true;
  *done = 1;
  return 0;
}

// 1 when the FDR column cna_percell_fdr_pinned last returned is the one cna_percell_fdr_copy_early copied
int cna_percell_fdr_copied_early(cna_ctx* c, int* yes) {
  CHECK_CTX(c);
  if (!yes) CNA_FAIL(CNA_EINVAL, "cna_percell_fdr_copied_early: yes is required");
  *yes = c->fdr_early_copied && c->fdr_inline && c->fdr_early_served;
  return 0;
}

int cna_percell_fdr_pinned(cna_ctx* c, const double* thr, const double* runmin_fdr, int T, double** coef_ptr,
                           double** fdr_ptr) {
  CHECK_CTX(c);
  if (!coef_ptr) CNA_FAIL(CNA_EINVAL, "cna_percell_fdr_pinned: coef_ptr is required");
  const int64_t n_out = c->local_view ? c->n_local : c->n_global;
  CNA_TRY(ensure_cell_pinned(c, n_out));
  double* hc = (double*)c->h_cell;
  const bool want_fdr = fdr_ptr && thr && runmin_fdr && T > 0;
  if (c->coef_early) HIP_TRY(hipEventSynchronize(c->coef_copied));     // coefficients already on the host
  if (c->coef_early && c->fdr_inline && want_fdr && T == c->null_T && !c->null_pending) {
    // the coefficient column is in the pinned block; the FDR column is the table that followed the local null looked
    // up with the per-cell counts that left before it -- already put together in the caller's own storage by
    // cna_percell_fdr_copy_early (then that is what *fdr_ptr names), else put together here
    HIP_TRY(hipEventSynchronize(c->bins_copied));
    HIP_TRY(hipEventSynchronize(c->null_done));
    while (c->fdr_early_inflight.load()) sched_yield();
    *coef_ptr = hc;
    if (c->fdr_early_copied && c->fdr_early_dst) {
      *fdr_ptr = c->fdr_early_dst;
    } else {
      if (cna_host_expand_u16(hc + n_out, c->h_bins, n_out, c->h_tab, T, 4) != 0)
        CNA_FAIL(CNA_ESTATE, "cna_percell_fdr_pinned: expansion failed");
      *fdr_ptr = hc + n_out;
    }
    c->fdr_early_served = true;
    return 0;
  }
  c->fdr_early_served = false;
  CNA_TRY(cna_percell_fdr(c, thr, runmin_fdr, T, c->coef_early ? nullptr : hc, want_fdr ? hc + n_out : nullptr));
  *coef_ptr = hc;
  if (fdr_ptr) *fdr_ptr = want_fdr ? hc + n_out : nullptr;
  return 0;
}

int cna_percell_fdr(cna_ctx* c, const double* thr, const double* runmin_fdr, int T, double* coef_out,
                    double* fdr_out) {
  CHECK_CTX(c);
  if (!c->ncorrs_valid || !c->x_from_nam) CNA_FAIL(CNA_ESTATE, "cna_percell_fdr needs cna_select + cna_ncorrs");
  const bool want_fdr = fdr_out && thr && runmin_fdr && T > 0;
  double thr0 = 0, inv_step = 0;
  if (want_fdr) guess_from_thr(thr, T, &thr0, &inv_step);
  const int64_t Tn = want_fdr ? T : 1;
  CNA_TRY(dev_reserve(c, &c->scratch, &c->scratch_cap,
                      carve_bytes({8 * Tn, 8 * Tn, 8 * c->n_pad, 8 * c->n_pad, 8 * c->n_pad, 8 * c->n_pad})));
  Carver cv(c->scratch);
  double* td = cv.take<double>(Tn);
  double* rd = cv.take<double>(Tn);
  double* coef = cv.take<double>(c->n_pad);
  double* fdr = cv.take<double>(c->n_pad);
  double* coef_u = cv.take<double>(c->n_pad);
  double* fdr_u = cv.take<double>(c->n_pad);
  if (want_fdr) {
    HIP_TRY(hipMemcpyAsync(td, thr, 8 * T, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(rd, runmin_fdr, 8 * T, hipMemcpyHostToDevice, c->stream));
  }
  CNA_TRY(launch_percell_fdr(c, td, rd, want_fdr ? T : 0, thr0, inv_step, coef + c->row0, want_fdr ? fdr + c->row0 : nullptr));
  const bool sharded = (c->nranks > 1 || comm_active(c)) && !c->local_view;
  int64_t n_out = c->n_global;
  if (c->local_view) {
    // this rank's rows only, in the caller's (local) order: nothing crosses the fabric
    n_out = c->n_local;
    if (c->orig_idx) {
      CNA_TRY(launch_unpermute2(c, coef + c->row0, want_fdr ? fdr + c->row0 : nullptr, c->orig_idx, c->n_local, coef_u,
                                fdr_u));
      coef = coef_u;
      fdr = fdr_u;
    } else {
      coef += c->row0;
      fdr += c->row0;
    }
  } else if (c->orig_idx) {
    // back to the caller's numbering: every rank scatters its rows into a zeroed vector, the sum
    // over ranks (x + 0 keeps NaNs and bit patterns) is the full answer
    if (sharded) {
      HIP_TRY(hipMemsetAsync(coef_u, 0, 8 * c->n_pad, c->stream));
      if (want_fdr) HIP_TRY(hipMemsetAsync(fdr_u, 0, 8 * c->n_pad, c->stream));
    }
    CNA_TRY(launch_unpermute2(c, coef + c->row0, want_fdr ? fdr + c->row0 : nullptr, c->orig_idx, c->n_local, coef_u,
                              fdr_u));
    if (sharded) {
      CNA_TRY(comm_allreduce_f64_sum(c, coef_u, (size_t)c->n_global));
      if (want_fdr) CNA_TRY(comm_allreduce_f64_sum(c, fdr_u, (size_t)c->n_global));
    }
    coef = coef_u;
    fdr = fdr_u;
  } else if (c->nranks > 1) {
    const size_t block = 8 * (size_t)c->rows_per_rank;
    CNA_TRY(comm_allgather_bytes(c, (char*)coef + block * c->rank, coef, block));
    if (want_fdr) CNA_TRY(comm_allgather_bytes(c, (char*)fdr + block * c->rank, fdr, block));
  }
  if (coef_out && n_out > 0) HIP_TRY(hipMemcpyAsync(coef_out, coef, 8 * n_out, hipMemcpyDeviceToHost, c->stream));
  if (want_fdr && n_out > 0) HIP_TRY(hipMemcpyAsync(fdr_out, fdr, 8 * n_out, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// -------------------------------------------------------------------------------- D2H
int cna_matrix_shape(cna_ctx* c, int which, int64_t* n_rows_local, int* n_cols) {
  CHECK_CTX(c);                                  // (the NAM may be materialised here: kernels on this context's device)
  if (which == CNA_MAT_NAM) AUTO_FINISH(c);
  if (which == CNA_MAT_NAM) {
    CNA_TRY(need_nam(c));
    *n_rows_local = c->n_local; *n_cols = c->N;
  } else if (which == CNA_MAT_X) {
    if (!c->x_valid) CNA_FAIL(CNA_ESTATE, "X not available");
    *n_rows_local = c->nx; *n_cols = c->Nx;
  } else {
    CNA_FAIL(CNA_EINVAL, "bad matrix selector");
  }
  return 0;
}

int cna_fetch_matrix(cna_ctx* c, int which, double* out, int transposed) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  int64_t rows;
  int cols;
  CNA_TRY(cna_matrix_shape(c, which, &rows, &cols));
  const double* src = which == CNA_MAT_NAM ? c->nam : c->X;
  const int ld = which == CNA_MAT_NAM ? c->ld : c->ldx;
  if (rows == 0) return 0;
  if (transposed) {
    CNA_TRY(dev_reserve(c, &c->scratch2, &c->scratch2_cap, 8 * rows * cols));
    CNA_TRY(launch_transpose(c, src, rows, cols, ld, (double*)c->scratch2));
    HIP_TRY(hipMemcpyAsync(out, c->scratch2, 8 * rows * cols, hipMemcpyDeviceToHost, c->stream));
  } else {
    HIP_TRY(hipMemcpy2DAsync(out, 8 * (size_t)cols, src, 8 * (size_t)ld, 8 * (size_t)cols, rows,
                             hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// cna_fetch_matrix with the rows (and columns) picked and ordered on the device: out[i][j] =
// matrix[rows[i]][cols[j]], or its transpose.  which = CNA_MAT_NAM / CNA_MAT_X, or CNA_MAT_PROJ for the
// result of the last cna_project_keep.  One gather kernel and one contiguous copy instead of a
// strided copy plus cells x samples sized reshuffles on the host.
int cna_fetch_rows(cna_ctx* c, int which, const int64_t* rows, int64_t n_out, const int32_t* cols, int n_cols,
                   double* out, int transposed) {
  CHECK_CTX(c);
  AUTO_FINISH(c);
  const double* src;
  int ld, width;
  int64_t have;
  if (which == CNA_MAT_PROJ) {
    if (!c->proj_valid) CNA_FAIL(CNA_ESTATE, "no projection resident (cna_project_keep)");
    src = (const double*)c->proj; ld = c->proj_ld; width = c->proj_cols; have = c->proj_rows;
  } else {
    int cc;
    CNA_TRY(cna_matrix_shape(c, which, &have, &cc));
    width = cc;
    src = which == CNA_MAT_NAM ? c->nam : c->X;
    ld = which == CNA_MAT_NAM ? c->ld : c->ldx;
  }
  if (!rows) n_out = have;
  if (!cols) n_cols = width;
  if (n_out < 0 || n_cols < 0) CNA_FAIL(CNA_EINVAL, "cna_fetch_rows: bad sizes");
  if (n_out == 0 || n_cols == 0) return 0;
  if (rows)
    for (int64_t i = 0; i < n_out; ++i)
      if (rows[i] < 0 || rows[i] >= have) CNA_FAIL(CNA_EINVAL, "cna_fetch_rows: row index out of range");
  if (cols)
    for (int j = 0; j < n_cols; ++j)
      if (cols[j] < 0 || cols[j] >= width) CNA_FAIL(CNA_EINVAL, "cna_fetch_rows: column index out of range");
  CNA_TRY(dev_reserve(c, &c->scratch2, &c->scratch2_cap, carve_bytes({8 * n_out * (int64_t)n_cols, 8 * n_out, 4 * (int64_t)n_cols})));
  Carver cv(c->scratch2);
  double* dst = cv.take<double>(n_out * (int64_t)n_cols);
  int64_t* rd = cv.take<int64_t>(n_out);
  int32_t* cd = cv.take<int32_t>(n_cols);
  if (rows) HIP_TRY(hipMemcpyAsync(rd, rows, 8 * n_out, hipMemcpyHostToDevice, c->stream));
  if (cols) HIP_TRY(hipMemcpyAsync(cd, cols, 4 * (size_t)n_cols, hipMemcpyHostToDevice, c->stream));
  CNA_TRY(launch_gather_rows(c, src, ld, rows ? rd : nullptr, n_out, cols ? cd : nullptr, n_cols, dst, transposed));
  HIP_TRY(hipMemcpyAsync(out, dst, 8 * (size_t)n_out * n_cols, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int cna_allgather_host(cna_ctx* c, const double* local, int64_t count_local, double* out_all, int64_t count_total) {
  CHECK_CTX(c);
  if (count_local < 0 || count_total < count_local) CNA_FAIL(CNA_EINVAL, "cna_allgather_host: bad counts");
  if (c->nranks == 1) {
    if (count_total != count_local) CNA_FAIL(CNA_EINVAL, "cna_allgather_host: single rank but totals differ");
    std::memcpy(out_all, local, 8 * count_local);
    return 0;
  }
  void* tmp = nullptr;
  CNA_TRY(dev_alloc(c, &tmp, 8 * std::max<int64_t>(count_local, 1)));
  int r = 0;
  if (count_local > 0) {
    hipError_t e = hipMemcpyAsync(tmp, local, 8 * count_local, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { cna_set_error(hipGetErrorString(e)); r = (int)e; }
  }
  if (r == 0) r = ragged_gather(c, (const double*)tmp, count_local, out_all, count_total);
  (void)hipStreamSynchronize(c->stream);
  dev_free(c, tmp, 8 * std::max<int64_t>(count_local, 1));
  return r;
}

// -------------------------------------------------------------------------- profiling
int cna_prof_enable(cna_ctx* c, int on) {
  if (!c) CNA_FAIL(CNA_EINVAL, "null context");
  if (!on) prof_flush(c);
  c->prof = on != 0;
  static_assert(CNA_K_COUNT <= 64, "prof_mask is one word");
  c->prof_mask = on == 2 ? (1ull << CNA_K_NAM_FIRST | 1ull << CNA_K_NAM_STEP | 1ull << CNA_K_NAM_STEP_SPARSE | 1ull << CNA_K_ALLGATHER |
                            1ull << CNA_K_HALO_EXCHANGE | 1ull << CNA_K_HALO_WAIT)
                         : ~0ull;
  return 0;
}
int cna_prof_reset(cna_ctx* c) {
  if (!c) CNA_FAIL(CNA_EINVAL, "null context");
  prof_flush(c);
  for (int i = 0; i < CNA_K_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
  return 0;
}
int cna_prof_get(cna_ctx* c, int k, double* total_ms, int64_t* launches) {
  if (!c || k < 0 || k >= CNA_K_COUNT) CNA_FAIL(CNA_EINVAL, "bad kernel id");
  prof_flush(c);
  if (total_ms) *total_ms = c->prof_ms[k];
  if (launches) *launches = c->prof_n[k];
  return 0;
}

}  // extern "C"
