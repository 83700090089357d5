// comm.hip — the row-sharded lookup's collectives issued straight onto the compute stream.
//
// torch.distributed runs every RCCL collective on the process group's own stream: each call costs two cross-stream event
// hand-offs, 10-13 us of idle on either side of the collective (rocprof, round 1) — ~60 us of a 0.40 ms sharded step at
// one rank.  Here the library owns a second communicator (same RCCL, its own ncclComm_t) and enqueues
//   all-to-all  = ncclGroupStart; world x (ncclSend, ncclRecv) of equal byte counts; ncclGroupEnd
//   all-reduce  = ncclAllReduce(sum, f32)
// on the stream the caller passes — the one its kernels run on — so stream order is the only synchronisation.
// xGMI is a point-to-point mesh: the grouped send/recv is one direct hop per peer pair.
//
// RCCL is resolved at run time (dlopen of the librccl the process already has), so libmi355x_recsys.so keeps loading on
// a box without it; every entry point returns MI_ERR_UNSUPPORTED then.  No reference counterpart (the reference has no
// multi-GPU path).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.hpp"

namespace {
struct Rccl {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
  void *h = nullptr;
  for (const char *name : {"librccl.so.1", "librccl.so"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) return;
#define SYM(field, sym) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, #sym))
  SYM(GetUniqueId, ncclGetUniqueId);
  SYM(CommInitRank, ncclCommInitRank);
  SYM(CommDestroy, ncclCommDestroy);
  SYM(CommAbort, ncclCommAbort);
  SYM(GroupStart, ncclGroupStart);
  SYM(GroupEnd, ncclGroupEnd);
  SYM(Send, ncclSend);
  SYM(Recv, ncclRecv);
  SYM(AllReduce, ncclAllReduce);
#undef SYM
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.CommAbort && g_rccl.GroupStart && g_rccl.GroupEnd &&
              g_rccl.Send && g_rccl.Recv && g_rccl.AllReduce;
}
const Rccl *rccl() {
  std::call_once(g_once, load_rccl);
  return g_rccl.ok ? &g_rccl : nullptr;
}
struct Comm {
  ncclComm_t c;
  int world, rank;
};
}  // namespace

extern "C" {

// MI_OK when a librccl could be loaded in THIS process (every rank checks for itself and the caller agrees on the answer
// over its own process group BEFORE anyone enters mi_comm_init, which would otherwise wait forever for the rank that cannot)
int mi_comm_available(void) { return rccl() ? MI_OK : MI_ERR_UNSUPPORTED; }

// 128 bytes of rendezvous id, created on ONE rank and handed to the others by the caller (torch.distributed broadcast)
int mi_comm_unique_id(char *id128) {
  const Rccl *r = rccl();
  if (!r) return MI_ERR_UNSUPPORTED;
  if (!id128) return MI_ERR_INVALID_ARG;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != ncclSuccess) return MI_ERR_LAUNCH;
  std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return MI_OK;
}

// collective over all `world` ranks (blocks until every rank has called it); the current HIP device is the rank's GPU
int mi_comm_init(const char *id128, int32_t world, int32_t rank, void **comm_out) {
  const Rccl *r = rccl();
  if (!r) return MI_ERR_UNSUPPORTED;
  if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return MI_ERR_INVALID_ARG;
  ncclUniqueId id;
  std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  Comm *c = new Comm{nullptr, world, rank};
  if (r->CommInitRank(&c->c, world, id, rank) != ncclSuccess) {
    delete c;
    return MI_ERR_LAUNCH;
  }
  *comm_out = c;
  return MI_OK;
}

int mi_comm_destroy(void *comm) {
  const Rccl *r = rccl();
  if (!r) return MI_ERR_UNSUPPORTED;
  if (!comm) return MI_OK;
  Comm *c = static_cast<Comm *>(comm);
  const bool ok = r->CommDestroy(c->c) == ncclSuccess;
  delete c;
  return ok ? MI_OK : MI_ERR_LAUNCH;
}

// gives up a communicator whose enqueued work never completes (a peer died, a link is down): RCCL stops its kernels and
// frees it without waiting for them.  The caller's stream becomes usable again; `comm` is invalid afterwards.
int mi_comm_abort(void *comm) {
  const Rccl *r = rccl();
  if (!r) return MI_ERR_UNSUPPORTED;
  if (!comm) return MI_OK;
  Comm *c = static_cast<Comm *>(comm);
  const bool ok = r->CommAbort(c->c) == ncclSuccess;
  delete c;
  return ok ? MI_OK : MI_ERR_LAUNCH;
}

// recv[p * bytes_per_peer ..) <- rank p's send[me * bytes_per_peer ..), all peers, on `stream`
int mi_comm_all_to_all(void *comm, const void *send, void *recv, int64_t bytes_per_peer, void *stream) {
  const Rccl *r = rccl();
  if (!r) return MI_ERR_UNSUPPORTED;
  if (!comm || bytes_per_peer < 0) return MI_ERR_INVALID_ARG;
  if (bytes_per_peer == 0) return MI_OK;
  if (!send || !recv) return MI_ERR_INVALID_ARG;
  Comm *c = static_cast<Comm *>(comm);
  const char *s = static_cast<const char *>(send);
  char *d = static_cast<char *>(recv);
  if (r->GroupStart() != ncclSuccess) return MI_ERR_LAUNCH;
  bool ok = true;
  for (int p = 0; p < c->world; ++p) {
    ok = ok && r->Send(s + (int64_t)p * bytes_per_peer, (size_t)bytes_per_peer, ncclInt8, p, c->c, (hipStream_t)stream) == ncclSuccess;
    ok = ok && r->Recv(d + (int64_t)p * bytes_per_peer, (size_t)bytes_per_peer, ncclInt8, p, c->c, (hipStream_t)stream) == ncclSuccess;
  }
  if (r->GroupEnd() != ncclSuccess) return MI_ERR_LAUNCH;
  return ok ? MI_OK : MI_ERR_LAUNCH;
}

// buf[i] <- sum over ranks of buf[i]  (fp32, in place), on `stream`
int mi_comm_all_reduce_sum_f32(void *comm, float *buf, int64_t count, void *stream) {
  const Rccl *r = rccl();
  if (!r) return MI_ERR_UNSUPPORTED;
  if (!comm || count < 0) return MI_ERR_INVALID_ARG;
  if (count == 0) return MI_OK;
  if (!buf) return MI_ERR_INVALID_ARG;
  Comm *c = static_cast<Comm *>(comm);
  return r->AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, c->c, (hipStream_t)stream) == ncclSuccess ? MI_OK : MI_ERR_LAUNCH;
}

}  // extern "C"
