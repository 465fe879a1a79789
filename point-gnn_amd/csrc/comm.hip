// comm.hip -- the training step's collectives on RCCL.
//
// What the reference does with in-graph towers -- `average_gradients`
// (util/tf_util.py:3-43: stack the towers' gradients, reduce_mean) behind
// `unify_copies` (train.py:264-288: every tower's loss re-weighted by the
// GLOBAL endpoint counts) and one `apply_gradients` (train.py:397-405) -- is,
// with one process per GPU, ONE all-reduce(sum) of the flat fp32 gradient
// buffer plus an all-reduce of the two endpoint counts / the loss sums.  The
// communicator lives behind the C ABI: the caller moves 128 id bytes between
// its ranks by whatever it has (a file, a torch.distributed store, MPI) and
// every collective is enqueued on the caller's stream like any other entry.
//
// RCCL is bound at run time (dlopen), not at link time: a process that already
// holds an RCCL image (PyTorch ships one under the same soname) must keep
// exactly one -- two copies of the library in one process each build their own
// topology / IPC state -- and a process that never creates a communicator
// (inference) never pays for loading a 500 MB library.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "pgnn_common.h"

namespace pgnn {
namespace {

struct Rccl {
  void *handle = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int *) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t,
                            ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int,
                            ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  std::string path, error;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

template <typename F>
bool bind(void *h, const char *name, F &fn, std::string &err) {
  fn = (F)dlsym(h, name);
  if (!fn) {
    err = std::string("librccl: symbol ") + name + " not found";
    return false;
  }
  return true;
}

void load_rccl() {
  Rccl &r = g_rccl;
  // an image the process already holds first (RTLD_NOLOAD), under either name
  // it may have been opened by; then the system's
  const char *names[] = {"librccl.so.1", "librccl.so"};
  for (const char *n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (r.handle) { r.path = std::string(n) + " (already loaded)"; break; }
  }
  if (!r.handle) {
    const char *env = getenv("PGNN_RCCL_LIB");
    const char *cands[] = {env, "librccl.so.1", "librccl.so",
                           "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : cands) {
      if (!n || !*n) continue;
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) { r.path = n; break; }
    }
  }
  if (!r.handle) {
    const char *e = dlerror();
    r.error = std::string("librccl not found: ") + (e ? e : "?");
    return;
  }
  void *h = r.handle;
  bool ok = bind(h, "ncclGetErrorString", r.GetErrorString, r.error) &&
            bind(h, "ncclGetVersion", r.GetVersion, r.error) &&
            bind(h, "ncclGetUniqueId", r.GetUniqueId, r.error) &&
            bind(h, "ncclCommInitRank", r.CommInitRank, r.error) &&
            bind(h, "ncclCommDestroy", r.CommDestroy, r.error) &&
            bind(h, "ncclCommAbort", r.CommAbort, r.error) &&
            bind(h, "ncclCommGetAsyncError", r.CommGetAsyncError, r.error) &&
            bind(h, "ncclAllReduce", r.AllReduce, r.error) &&
            bind(h, "ncclBroadcast", r.Broadcast, r.error) &&
            bind(h, "ncclGroupStart", r.GroupStart, r.error) &&
            bind(h, "ncclGroupEnd", r.GroupEnd, r.error);
  if (!ok) r.handle = nullptr;
}

Rccl *rccl() {
  std::call_once(g_rccl_once, load_rccl);
  return g_rccl.handle ? &g_rccl : nullptr;
}

#define PGNN_RCCL(r, expr)                                                    \
  do {                                                                        \
    ncclResult_t e_ = (expr);                                                 \
    if (e_ != ncclSuccess) {                                                  \
      char buf_[320];                                                         \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr,             \
               (r)->GetErrorString(e_), __FILE__, __LINE__);                  \
      pgnn::last_error() = buf_;                                              \
      return PGNN_E_COMM;                                                     \
    }                                                                         \
  } while (0)

struct Comm {
  uint32_t magic;
  ncclComm_t comm;
  int world, rank, device;
};
constexpr uint32_t kCommMagic = 0x50474e43u;  // 'PGNC'

Comm *as_comm(void *p) {
  Comm *c = (Comm *)p;
  return c && c->magic == kCommMagic ? c : nullptr;
}

}  // namespace
}  // namespace pgnn

using namespace pgnn;

static_assert(sizeof(ncclUniqueId) == PGNN_COMM_ID_BYTES,
              "PGNN_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

extern "C" int pgnn_comm_unique_id(void *id_host) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(id_host, PGNN_E_INVALID, "comm_unique_id: null id buffer");
  Rccl *r = rccl();
  PGNN_REQUIRE(r, PGNN_E_COMM, g_rccl.error.c_str());
  ncclUniqueId id;
  PGNN_RCCL(r, r->GetUniqueId(&id));
  memcpy(id_host, &id, sizeof id);
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_comm_init_rank(const void *id_host, int32_t world,
                                   int32_t rank, void **comm_out) {
  PGNN_GUARD_BEGIN
  PGNN_REQUIRE(id_host && comm_out, PGNN_E_INVALID,
               "comm_init_rank: null argument");
  PGNN_REQUIRE(world >= 1 && rank >= 0 && rank < world, PGNN_E_INVALID,
               "comm_init_rank: need 0 <= rank < world");
  *comm_out = nullptr;
  Rccl *r = rccl();
  PGNN_REQUIRE(r, PGNN_E_COMM, g_rccl.error.c_str());
  ncclUniqueId id;
  memcpy(&id, id_host, sizeof id);
  int dev = 0;
  PGNN_HIP(hipGetDevice(&dev));
  ncclComm_t nc = nullptr;
  PGNN_RCCL(r, r->CommInitRank(&nc, world, id, rank));
  Comm *c = new Comm{kCommMagic, nc, world, rank, dev};
  *comm_out = c;
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_comm_info(void *comm, int32_t *world, int32_t *rank,
                              int32_t *rccl_version) {
  PGNN_GUARD_BEGIN
  if (rccl_version) {
    Rccl *r = rccl();
    PGNN_REQUIRE(r, PGNN_E_COMM, g_rccl.error.c_str());
    int v = 0;
    PGNN_RCCL(r, r->GetVersion(&v));
    *rccl_version = v;
  }
  if (comm) {
    Comm *c = as_comm(comm);
    PGNN_REQUIRE(c, PGNN_E_INVALID, "comm_info: not a communicator");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
  }
  return 0;
  PGNN_GUARD_END
}

extern "C" const char *pgnn_comm_library(void) {
  return rccl() ? g_rccl.path.c_str() : g_rccl.error.c_str();
}

extern "C" int pgnn_comm_destroy(void *comm) {
  PGNN_GUARD_BEGIN
  if (!comm) return 0;
  Comm *c = as_comm(comm);
  PGNN_REQUIRE(c, PGNN_E_INVALID, "comm_destroy: not a communicator");
  Rccl *r = rccl();
  PGNN_REQUIRE(r, PGNN_E_COMM, g_rccl.error.c_str());
  c->magic = 0;
  ncclResult_t e = r->CommDestroy(c->comm);
  delete c;
  PGNN_RCCL(r, e);
  return 0;
  PGNN_GUARD_END
}

namespace {
int allreduce(void *comm, void *buf, int64_t n, ncclDataType_t dt,
              void *stream, const char *what) {
  Comm *c = as_comm(comm);
  PGNN_REQUIRE(c, PGNN_E_INVALID, what);
  PGNN_REQUIRE(n >= 0 && (buf || n == 0), PGNN_E_INVALID, what);
  if (n == 0) return 0;
  Rccl *r = rccl();
  PGNN_RCCL(r, r->AllReduce(buf, buf, (size_t)n, dt, ncclSum, c->comm,
                            (hipStream_t)stream));
  return 0;
}
}  // namespace

extern "C" int pgnn_allreduce_sum_f32(void *comm, float *buf, int64_t n,
                                      void *stream) {
  PGNN_GUARD_BEGIN
  return allreduce(comm, buf, n, ncclFloat32, stream,
                   "allreduce_sum_f32: bad communicator / buffer");
  PGNN_GUARD_END
}

extern "C" int pgnn_allreduce_sum_f64(void *comm, double *buf, int64_t n,
                                      void *stream) {
  PGNN_GUARD_BEGIN
  return allreduce(comm, buf, n, ncclFloat64, stream,
                   "allreduce_sum_f64: bad communicator / buffer");
  PGNN_GUARD_END
}

extern "C" int pgnn_allreduce_step(void *comm, float *grads, int64_t n_grads,
                                   double *sums, int64_t n_sums, void *stream) {
  PGNN_GUARD_BEGIN
  Comm *c = as_comm(comm);
  PGNN_REQUIRE(c, PGNN_E_INVALID, "allreduce_step: not a communicator");
  PGNN_REQUIRE(n_grads >= 0 && n_sums >= 0 && (grads || !n_grads) &&
                   (sums || !n_sums),
               PGNN_E_INVALID, "allreduce_step: bad buffer");
  Rccl *r = rccl();
  // one group: RCCL fuses the two reductions into one launch
  PGNN_RCCL(r, r->GroupStart());
  ncclResult_t e0 = ncclSuccess, e1 = ncclSuccess;
  if (n_grads)
    e0 = r->AllReduce(grads, grads, (size_t)n_grads, ncclFloat32, ncclSum,
                      c->comm, (hipStream_t)stream);
  if (n_sums)
    e1 = r->AllReduce(sums, sums, (size_t)n_sums, ncclFloat64, ncclSum,
                      c->comm, (hipStream_t)stream);
  ncclResult_t e2 = r->GroupEnd();
  PGNN_RCCL(r, e0);
  PGNN_RCCL(r, e1);
  PGNN_RCCL(r, e2);
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_broadcast_f32(void *comm, float *buf, int64_t n,
                                  int32_t root, void *stream) {
  PGNN_GUARD_BEGIN
  Comm *c = as_comm(comm);
  PGNN_REQUIRE(c, PGNN_E_INVALID, "broadcast_f32: not a communicator");
  PGNN_REQUIRE(n >= 0 && (buf || n == 0) && root >= 0 && root < c->world,
               PGNN_E_INVALID, "broadcast_f32: bad buffer / root");
  if (n == 0) return 0;
  Rccl *r = rccl();
  PGNN_RCCL(r, r->Broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->comm,
                            (hipStream_t)stream));
  return 0;
  PGNN_GUARD_END
}

extern "C" int pgnn_comm_async_error(void *comm) {
  PGNN_GUARD_BEGIN
  Comm *c = as_comm(comm);
  PGNN_REQUIRE(c, PGNN_E_INVALID, "comm_async_error: not a communicator");
  Rccl *r = rccl();
  ncclResult_t st = ncclSuccess;
  PGNN_RCCL(r, r->CommGetAsyncError(c->comm, &st));
  PGNN_RCCL(r, st);
  return 0;
  PGNN_GUARD_END
}
