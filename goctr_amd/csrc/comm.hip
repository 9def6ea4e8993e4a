// comm.hip -- data-parallel communicator: RCCL over xGMI, one process per GPU.
//
// No reference counterpart (go-ctr is single-process).  The path shards by rows:
// every rank runs the step on its own batch shard and the flat gradient buffer (+ the BCE sum) is
// summed with ONE ncclAllReduce per step (the buffer is ~175 KB: latency-bound, so one call).
// RCCL is loaded lazily (dlopen) the first time a communicator with world > 1 is created: a
// single-GPU process never touches it.
#include <dlfcn.h>
#include <cstdlib>
#include <rccl/rccl.h>

#include "common.h"

namespace goctr {
namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.h) return 0;
  const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.h) break;
  }
  GOCTR_CHECK(g_rccl.h, "cannot load RCCL: %s", dlerror());
#define GOCTR_SYM(field, name)                                                      \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.h, name));   \
  GOCTR_CHECK(g_rccl.field, "RCCL symbol %s missing", name)
  GOCTR_SYM(GetUniqueId, "ncclGetUniqueId");
  GOCTR_SYM(CommInitRank, "ncclCommInitRank");
  GOCTR_SYM(AllReduce, "ncclAllReduce");
  GOCTR_SYM(AllGather, "ncclAllGather");
  GOCTR_SYM(Send, "ncclSend");
  GOCTR_SYM(Recv, "ncclRecv");
  GOCTR_SYM(GroupStart, "ncclGroupStart");
  GOCTR_SYM(GroupEnd, "ncclGroupEnd");
  GOCTR_SYM(CommDestroy, "ncclCommDestroy");
  GOCTR_SYM(GetErrorString, "ncclGetErrorString");
#undef GOCTR_SYM
  g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(g_rccl.h, "ncclCommAbort"));
  return 0;
}

#define GOCTR_NCCL(call)                                                                          \
  do {                                                                                            \
    ncclResult_t _r = (call);                                                                     \
    if (_r != ncclSuccess) {                                                                      \
      set_error("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
      return -1;                                                                                  \
    }                                                                                             \
  } while (0)

}  // namespace

int comm_allreduce_f32(float* dev, size_t n) {
  Engine& e = engine();
  if (!e.comm_active()) return 0;
  GOCTR_NCCL(g_rccl.AllReduce(dev, dev, n, ncclFloat32, ncclSum, (ncclComm_t)e.nccl_comm, e.stream));
  return 0;
}
// all-gather of n int32 per rank (recv = [world][n])
int comm_allgather_i32(const int* send, int* recv, size_t n) {
  Engine& e = engine();
  GOCTR_CHECK(e.comm_active(), "comm_allgather_i32: no communicator");
  GOCTR_NCCL(g_rccl.AllGather(send, recv, n, ncclInt32, (ncclComm_t)e.nccl_comm, e.stream));
  return 0;
}
// One grouped exchange: for every peer p, send_cnt[p] elements from send + send_off[p] and recv_cnt[p] elements into
// recv + recv_off[p] (offsets and counts in elements of `bytes_per_elem` bytes, which must be 4 or 8) -- an all-to-all-v
// built from ncclSend / ncclRecv pairs inside one group (point-to-point over xGMI, including the self pair).
int comm_alltoallv(const void* send, const size_t* send_off, const size_t* send_cnt, void* recv, const size_t* recv_off,
                   const size_t* recv_cnt, int bytes_per_elem) {
  Engine& e = engine();
  GOCTR_CHECK(e.comm_active(), "comm_alltoallv: no communicator");
  GOCTR_CHECK(bytes_per_elem == 4 || bytes_per_elem == 8, "comm_alltoallv: element size %d", bytes_per_elem);
  const ncclDataType_t ty = bytes_per_elem == 4 ? ncclInt32 : ncclInt64;
  GOCTR_NCCL(g_rccl.GroupStart());
  for (int p = 0; p < e.world; ++p) {
    if (send_cnt[p]) GOCTR_NCCL(g_rccl.Send(static_cast<const char*>(send) + send_off[p] * bytes_per_elem, send_cnt[p], ty, p, (ncclComm_t)e.nccl_comm, e.stream));
    if (recv_cnt[p]) GOCTR_NCCL(g_rccl.Recv(static_cast<char*>(recv) + recv_off[p] * bytes_per_elem, recv_cnt[p], ty, p, (ncclComm_t)e.nccl_comm, e.stream));
  }
  GOCTR_NCCL(g_rccl.GroupEnd());
  return 0;
}

// A data-parallel call that fails on ONE rank between two collectives (a HIP error, a failed allocation) would leave its
// peers blocked inside the next collective for ever.  The failing rank aborts the communicator instead: the peers' pending
// and later RCCL calls return an error, every rank's call fails, nobody hangs.  (The communicator is unusable afterwards;
// goctr_comm_init builds a new one.)
void comm_abort_on_failure() {
  Engine& e = engine();
  if (!e.nccl_comm) return;
  if (g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)e.nccl_comm);
  else if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)e.nccl_comm);
  e.nccl_comm = nullptr; e.rank = 0; e.world = 1;
}

int comm_allreduce_f64_dev(double* dev, size_t n) {
  Engine& e = engine();
  if (!e.comm_active()) return 0;
  GOCTR_NCCL(g_rccl.AllReduce(dev, dev, n, ncclFloat64, ncclSum, (ncclComm_t)e.nccl_comm, e.stream));
  return 0;
}

}  // namespace goctr

using namespace goctr;

extern "C" {

int goctr_comm_unique_id(uint8_t id[128]) {
  GOCTR_ENTER();
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (load_rccl()) return -1;
  ncclUniqueId u;
  GOCTR_NCCL(g_rccl.GetUniqueId(&u));
  memcpy(id, &u, 128);
  return 0;
}

int goctr_comm_init(int rank, int world, const uint8_t id[128]) {
  GOCTR_ENTER();
  GOCTR_CHECK(world >= 1 && rank >= 0 && rank < world, "goctr_comm_init: bad rank %d / world %d", rank, world);
  Engine& e = engine();
  if (e.nccl_comm) goctr_comm_destroy();
  e.rank = rank; e.world = world;
  const char* force = getenv("GOCTR_FORCE_COMM");
  if (world == 1 && !(force && *force && *force != '0')) return 0;
  if (load_rccl()) return -1;
  ncclUniqueId u;
  memcpy(&u, id, 128);
  ncclComm_t c = nullptr;
  GOCTR_NCCL(g_rccl.CommInitRank(&c, world, u, rank));
  e.nccl_comm = c;
  return 0;
}

int goctr_comm_world(int* rank, int* world) {
  if (rank) *rank = engine().rank;
  if (world) *world = engine().world;
  return 0;
}

int goctr_comm_allreduce_f64(double* v, int n) {
  GOCTR_ENTER();
  Engine& e = engine();
  if (!e.comm_active()) return 0;
  DevBuf<double> d;
  if (d.alloc(n, false) || d.upload(v, n)) return -1;
  if (comm_allreduce_f64_dev(d.p, n)) return -1;
  return d.download(v, n);
}

int goctr_comm_destroy(void) {
  std::lock_guard<std::recursive_mutex> lk(engine_mutex());
  Engine& e = engine();
  if (e.nccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)e.nccl_comm);
  e.nccl_comm = nullptr; e.rank = 0; e.world = 1;
  return 0;
}

}  // extern "C"
