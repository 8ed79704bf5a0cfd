// RCCL (xGMI) collectives used by the sharded path.  librccl.so is opened lazily so that the
// single-GPU path and the CPU-side "does the library load" check do not depend on it.
// Exchanges: column sums (all-reduce), scaled diffusion state between steps
// (all-gather), Gram matrix / tail histograms / threshold counts (all-reduce), per-cell vectors
// for exact medians and data.obs columns (all-gather).
#include "common.h"
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.h) return 0;
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) CNA_FAIL(CNA_ERCCL, std::string("cannot load librccl.so: ") + dlerror());
#define SYM(name)                                                              \
  g_rccl.name = (decltype(g_rccl.name))dlsym(h, "nccl" #name);                 \
  if (!g_rccl.name) CNA_FAIL(CNA_ERCCL, "librccl.so lacks nccl" #name)
  SYM(GetUniqueId);
  SYM(CommInitRank);
  SYM(CommDestroy);
  SYM(AllReduce);
  SYM(AllGather);
  SYM(Send);
  SYM(Recv);
  SYM(GroupStart);
  SYM(GroupEnd);
  SYM(GetErrorString);
#undef SYM
  g_rccl.h = h;
  return 0;
}

#define NCCL_TRY(expr)                                                                       \
  do {                                                                                       \
    ncclResult_t _r = (expr);                                                                \
    if (_r != ncclSuccess) CNA_FAIL(CNA_ERCCL, std::string(#expr) + ": " + g_rccl.GetErrorString(_r)); \
  } while (0)

}  // namespace

extern "C" int cna_comm_unique_id(void* id128) {
  CNA_TRY(rccl_load());
  ncclUniqueId id;
  NCCL_TRY(g_rccl.GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int cna_comm_init(cna_ctx* c, int rank, int nranks, const void* id128) {
  if (!c || nranks < 1 || rank < 0 || rank >= nranks) CNA_FAIL(CNA_EINVAL, "cna_comm_init: bad rank/nranks");
  c->rank = rank;
  c->nranks = nranks;
  if (nranks == 1 && !id128) return 0;
  CNA_TRY(rccl_load());
  HIP_TRY(hipSetDevice(c->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t comm;
  NCCL_TRY(g_rccl.CommInitRank(&comm, nranks, id, rank));
  c->comm = (void*)comm;
  return 0;
}

int comm_destroy(cna_ctx* c) {
  if (c->comm) {
    g_rccl.CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
  }
  return 0;
}

static int allreduce(cna_ctx* c, void* buf, size_t count, ncclDataType_t dt, ncclRedOp_t op) {
  if (c->nranks == 1 && !c->comm) return 0;
  if (!c->comm) CNA_FAIL(CNA_ESTATE, "multi-rank context without cna_comm_init");
  ProfScope ps(c, CNA_K_ALLGATHER);
  NCCL_TRY(g_rccl.AllReduce(buf, buf, count, dt, op, (ncclComm_t)c->comm, c->stream));
  return 0;
}

int comm_allreduce_f64_sum(cna_ctx* c, double* buf, size_t count) { return allreduce(c, buf, count, ncclFloat64, ncclSum); }
int comm_allreduce_f64_max(cna_ctx* c, double* buf, size_t count) { return allreduce(c, buf, count, ncclFloat64, ncclMax); }
int comm_allreduce_i64_sum(cna_ctx* c, int64_t* buf, size_t count) { return allreduce(c, buf, count, ncclInt64, ncclSum); }

// recv holds nranks blocks of bytes_per_rank; send may alias recv + rank*bytes_per_rank (in place)
int comm_allgather_bytes(cna_ctx* c, const void* send, void* recv, size_t bytes_per_rank) {
  if (c->nranks == 1 && !c->comm) {
    if (send != recv) HIP_TRY(hipMemcpyAsync(recv, send, bytes_per_rank, hipMemcpyDeviceToDevice, c->stream));
    return 0;
  }
  if (!c->comm) CNA_FAIL(CNA_ESTATE, "multi-rank context without cna_comm_init");
  ProfScope ps(c, CNA_K_ALLGATHER);
  NCCL_TRY(g_rccl.AllGather(send, recv, bytes_per_rank, ncclInt8, (ncclComm_t)c->comm, c->stream));
  return 0;
}

// Point-to-point exchange of packed state rows: rank p receives halo_send_cnt[p] rows from us and
// sends us halo_recv_cnt[p]; one grouped launch, so the xGMI links to all peers run concurrently.
int comm_halo_exchange(cna_ctx* c, const double* sendbuf, double* recvbuf, int64_t doubles_per_row) {
  if (!c->comm) CNA_FAIL(CNA_ESTATE, "halo exchange without cna_comm_init");
  ProfScope ps(c, CNA_K_ALLGATHER);
  NCCL_TRY(g_rccl.GroupStart());
  int64_t so = 0, ro = 0;
  ncclResult_t bad = ncclSuccess;
  for (int p = 0; p < c->nranks; ++p) {
    const int64_t ns = c->halo_send_cnt[p], nr = c->halo_recv_cnt[p];
    if (ns > 0 && bad == ncclSuccess)
      bad = g_rccl.Send(sendbuf + so * doubles_per_row, (size_t)(ns * doubles_per_row), ncclFloat64, p,
                        (ncclComm_t)c->comm, c->stream);
    if (nr > 0 && bad == ncclSuccess)
      bad = g_rccl.Recv(recvbuf + ro * doubles_per_row, (size_t)(nr * doubles_per_row), ncclFloat64, p,
                        (ncclComm_t)c->comm, c->stream);
    so += ns;
    ro += nr;
  }
  ncclResult_t end = g_rccl.GroupEnd();
  if (bad != ncclSuccess) CNA_FAIL(CNA_ERCCL, std::string("ncclSend/ncclRecv: ") + g_rccl.GetErrorString(bad));
  NCCL_TRY(end);
  return 0;
}
