// rccl_comm.hip -- gh_comm_ops bound natively to RCCL (xGMI): no collective of the stepped loop crosses the host language.
//
// Replaces the reference's MPI calls of the hot path one for one (src/Mpi/MpiControl.cpp: MPI_Allgather :329-337,
// MPI_Alltoallv :1073-1150; src/Tree/HydroTree.cpp:1044-1230 MPI_Isend / MPI_Irecv of the pruned trees):
//   allgather  = ncclAllGather on the context's HIP stream
//   alltoallv  = one ncclGroupStart ... ncclGroupEnd of ncclSend / ncclRecv pairs - on xGMI every pair of GPUs has its
//                own direct link, so the ragged exchange is per-link bound, not ring bound
// RCCL is resolved with dlopen at the first use (the copy already mapped into the process - PyTorch ships one - or
// /opt/rocm/lib/librccl.so.1), so that single-GPU users of libgandalf_hip.so never load it.
//
// Two ways to make a communicator:
//   gh_rccl_unique_id + gh_rccl_create      one process per GPU (bench.py under torch.distributed.run: the 128-byte id
//                                           travels through the launcher's store; gandalf_hip -g under mpirun-less hosts
//                                           through a file)
//   gh_rccl_create_all                      one process, one thread per GPU (ncclCommInitAll; gandalf_hip -g N)
#include "gh_internal.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace {

struct RcclApi {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  char err[256] = {0};
};

RcclApi g_api;
std::once_flag g_once;

void load_api()
{
  RcclApi &a = g_api;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) { a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (a.lib) break; }     // the copy already in the process
  if (!a.lib) for (const char *n : names) { a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (a.lib) break; }
  if (!a.lib) { snprintf(a.err, sizeof a.err, "RCCL not found: %s", dlerror()); return; }
#define SYM(field, name) a.field = (decltype(a.field)) dlsym(a.lib, name); if (!a.field) { snprintf(a.err, sizeof a.err, "RCCL symbol missing: %s", name); return; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommInitAll, "ncclCommInitAll")
  SYM(CommDestroy, "ncclCommDestroy") SYM(AllGather, "ncclAllGather") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
  SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
}

const RcclApi *api()
{
  std::call_once(g_once, load_api);
  return g_api.err[0] ? nullptr : &g_api;
}

}  // namespace

struct gh_rccl {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1, device = 0;
  gh_comm_ops ops;
  long long n_allgather = 0, n_alltoallv = 0, bytes = 0;
  char err[256] = {0};
};

static int rccl_fail(gh_rccl *c, const char *what, ncclResult_t r)
{
  snprintf(c->err, sizeof c->err, "%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "RCCL error");
  return 1;
}

static int rccl_allgather(void *user, const void *send, void *recv, int64_t bytes, void *stream)
{
  gh_rccl *c = (gh_rccl*) user;
  c->n_allgather++; c->bytes += bytes*c->nranks;
  if (bytes <= 0) return 0;
  const ncclResult_t r = g_api.AllGather(send, recv, (size_t) bytes, ncclInt8, c->comm, (hipStream_t) stream);
  return r == ncclSuccess ? 0 : rccl_fail(c, "ncclAllGather", r);
}

static int rccl_alltoallv(void *user, const void *send, const int64_t *sb, void *recv, const int64_t *rb, void *stream)
{
  gh_rccl *c = (gh_rccl*) user;
  c->n_alltoallv++;
  const char *s = (const char*) send;
  char *d = (char*) recv;
  hipStream_t st = (hipStream_t) stream;
  ncclResult_t r = g_api.GroupStart();
  if (r != ncclSuccess) return rccl_fail(c, "ncclGroupStart", r);
  size_t so = 0, ro = 0;
  for (int p = 0; p < c->nranks; p++) {
    if (p == c->rank) {
      if (sb[p] != rb[p]) { g_api.GroupEnd(); snprintf(c->err, sizeof c->err, "alltoallv: own block sizes differ"); return 1; }
      if (sb[p] > 0 && hipMemcpyAsync(d + ro, s + so, (size_t) sb[p], hipMemcpyDeviceToDevice, st) != hipSuccess) { g_api.GroupEnd(); return 1; }
    }
    else {
      if (sb[p] > 0 && (r = g_api.Send(s + so, (size_t) sb[p], ncclInt8, p, c->comm, st)) != ncclSuccess) { g_api.GroupEnd(); return rccl_fail(c, "ncclSend", r); }
      if (rb[p] > 0 && (r = g_api.Recv(d + ro, (size_t) rb[p], ncclInt8, p, c->comm, st)) != ncclSuccess) { g_api.GroupEnd(); return rccl_fail(c, "ncclRecv", r); }
    }
    c->bytes += sb[p];
    so += (size_t) sb[p]; ro += (size_t) rb[p];
  }
  r = g_api.GroupEnd();
  return r == ncclSuccess ? 0 : rccl_fail(c, "ncclGroupEnd", r);
}

static gh_rccl *make(ncclComm_t comm, int rank, int nranks, int device)
{
  gh_rccl *c = new gh_rccl();
  c->comm = comm; c->rank = rank; c->nranks = nranks; c->device = device;
  c->ops.user = c; c->ops.allgather = rccl_allgather; c->ops.alltoallv = rccl_alltoallv;
  return c;
}

extern "C" const char *gh_rccl_load_error(void) { return api() ? "" : g_api.err; }

extern "C" int gh_rccl_unique_id(void *id128)
{
  const RcclApi *a = api();
  if (!a || !id128) return GH_ERR_INVALID;
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (a->GetUniqueId(&id) != ncclSuccess) return GH_ERR_HIP;
  memcpy(id128, &id, sizeof id);
  return GH_OK;
}

extern "C" int gh_rccl_create(gh_rccl **out, int rank, int nranks, const void *id128, int device)
{
  const RcclApi *a = api();
  if (!a || !out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return GH_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return GH_ERR_HIP;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclComm_t comm = nullptr;
  const ncclResult_t r = a->CommInitRank(&comm, nranks, id, rank);
  if (r != ncclSuccess) { fprintf(stderr, "gh_rccl_create: ncclCommInitRank: %s\n", a->GetErrorString(r)); return GH_ERR_HIP; }
  *out = make(comm, rank, nranks, device);
  return GH_OK;
}

extern "C" int gh_rccl_create_all(gh_rccl **out, int ndev, const int *devices)
{
  const RcclApi *a = api();
  if (!a || !out || ndev < 1 || ndev > GH_MAX_RANKS) return GH_ERR_INVALID;
  ncclComm_t comms[GH_MAX_RANKS];
  const ncclResult_t r = a->CommInitAll(comms, ndev, devices);
  if (r != ncclSuccess) { fprintf(stderr, "gh_rccl_create_all: ncclCommInitAll: %s\n", a->GetErrorString(r)); return GH_ERR_HIP; }
  for (int i = 0; i < ndev; i++) out[i] = make(comms[i], i, ndev, devices ? devices[i] : i);
  return GH_OK;
}

extern "C" const gh_comm_ops *gh_rccl_ops(gh_rccl *c) { return c ? &c->ops : nullptr; }
extern "C" const char *gh_rccl_last_error(gh_rccl *c) { return c ? c->err : "no communicator"; }

extern "C" int gh_rccl_counters(gh_rccl *c, int64_t *n_allgather, int64_t *n_alltoallv, int64_t *bytes, int reset)
{
  if (!c) return GH_ERR_INVALID;
  if (n_allgather) *n_allgather = c->n_allgather;
  if (n_alltoallv) *n_alltoallv = c->n_alltoallv;
  if (bytes) *bytes = c->bytes;
  if (reset) { c->n_allgather = 0; c->n_alltoallv = 0; c->bytes = 0; }
  return GH_OK;
}

extern "C" void gh_rccl_destroy(gh_rccl *c)
{
  if (!c) return;
  if (c->comm && g_api.CommDestroy) g_api.CommDestroy(c->comm);
  delete c;
}
