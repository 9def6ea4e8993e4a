// comm.hip -- data-parallel communicators of libgoctr_hip.so.
//
// No reference counterpart (go-ctr is single-process).  The path shards by rows: every rank runs the step
// on its own batch shard and the flat gradient buffer (+ the BCE sum) is summed with ONE all-reduce per step (the buffer is
// ~175 KB: latency-bound, so one call).  Three ways a rank gets its communicator:
//   * one process per GPU (the launcher path, goctr_comm_init): RCCL over xGMI, unique id handed out by the host;
//   * one process, one engine per DISTINCT device (goctr_init_devices): RCCL, ncclCommInitAll, every rank a host thread;
//   * one process, several logical ranks on the SAME device (goctr_init_devices with a repeated id) or GOCTR_COMM=loopback:
//     the LOOP-BACK communicator below -- the collectives are kernels / copies that read the peers' buffers directly, ordered
//     by events and a host barrier between the rank threads.  RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so
//     this is what runs the W > 1 device code (bucket packing, owner scans, padded exchange layouts, the split step graphs) on
//     a one-GPU box; it sums in fixed rank order, so every replica gets the same bits.
// RCCL is loaded lazily (dlopen) the first time a communicator needs it: a single-GPU process never touches it.
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <rccl/rccl.h>

#include "common.h"

namespace goctr {
namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;      // optional
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;   // optional
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return 0;
  const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  GOCTR_CHECK(h, "cannot load RCCL: %s", dlerror());
#define GOCTR_SYM(field, name)                                             \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
  GOCTR_CHECK(g_rccl.field, "RCCL symbol %s missing", name)
  GOCTR_SYM(GetUniqueId, "ncclGetUniqueId");
  GOCTR_SYM(CommInitRank, "ncclCommInitRank");
  GOCTR_SYM(CommInitAll, "ncclCommInitAll");
  GOCTR_SYM(AllReduce, "ncclAllReduce");
  GOCTR_SYM(AllGather, "ncclAllGather");
  GOCTR_SYM(Broadcast, "ncclBroadcast");
  GOCTR_SYM(Send, "ncclSend");
  GOCTR_SYM(Recv, "ncclRecv");
  GOCTR_SYM(GroupStart, "ncclGroupStart");
  GOCTR_SYM(GroupEnd, "ncclGroupEnd");
  GOCTR_SYM(CommDestroy, "ncclCommDestroy");
  GOCTR_SYM(GetErrorString, "ncclGetErrorString");
#undef GOCTR_SYM
  g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(h, "ncclCommAbort"));
  g_rccl.CommGetAsyncError = reinterpret_cast<decltype(g_rccl.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
  g_rccl.h = h;
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

int env_int_comm(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// ---------------------------------------------------------------- loop-back kernels
constexpr int kLoopMaxWorld = 64;
struct LoopPtrs { const void* p[kLoopMaxWorld]; };

// out[i] = src[0][i] + src[1][i] + ... in FIXED rank order: every rank computes the same bits
template <typename T>
__global__ __launch_bounds__(256) void loop_sum_kernel(LoopPtrs src, int world, T* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    T acc = static_cast<const T*>(src.p[0])[i];
    for (int r = 1; r < world; ++r) acc += static_cast<const T*>(src.p[r])[i];
    out[i] = acc;
  }
}

}  // namespace

// ---------------------------------------------------------------- loop-back communicator
// One LoopGroup is shared by the W engines of a goctr_init_devices group.  Every collective is called by all W rank threads
// (each on its own engine / stream) and follows one protocol:
//   publish my buffer, record `ready` on my stream | HOST BARRIER | my stream waits for every peer's `ready`, then reads the
//   peers' buffers (sum kernel / device copies) into memory only I write, record `done` | HOST BARRIER | my stream waits for
//   every peer's `done` (nobody still reads what I am about to overwrite), then finishes in place.
// The barrier carries an abort flag and a timeout (GOCTR_LOOP_TIMEOUT_S, default 120): a rank that fails between two
// collectives aborts the group and its peers' calls return an error instead of waiting for ever.
struct LoopGroup {
  int world = 0;
  std::vector<Engine*> members;
  std::mutex mu; std::condition_variable cv;
  int arrived = 0; uint64_t generation = 0; bool aborted = false;
  struct Slot {
    const void* pub = nullptr;                 // the buffer this rank published for the collective in flight
    const size_t* off = nullptr; const size_t* cnt = nullptr;   // all-to-all-v: this rank's send offsets / counts
    hipEvent_t ready = nullptr, done = nullptr;
    DevBuf<char> scratch;
  };
  std::vector<Slot> slot;

  int barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) { set_error("loop-back communicator: aborted by a failing rank"); return -1; }
    const uint64_t gen = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return 0; }
    const int timeout_s = std::max(1, env_int_comm("GOCTR_LOOP_TIMEOUT_S", 120));
    const bool ok = cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return generation != gen || aborted; });
    if (!ok) { aborted = true; cv.notify_all(); set_error("loop-back communicator: a rank did not reach the collective within %d s", timeout_s); return -1; }
    if (aborted && generation == gen) { set_error("loop-back communicator: aborted by a failing rank"); return -1; }
    return 0;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = true;
    cv.notify_all();
  }
};

namespace {

// the two-barrier protocol around `body` (which queues this rank's reads of the peers' published buffers on its stream)
template <class Body>
int loop_collective(Engine& e, const void* pub, const size_t* off, const size_t* cnt, Body body) {
  LoopGroup& g = *e.loop;
  LoopGroup::Slot& me = g.slot[e.rank];
  me.pub = pub; me.off = off; me.cnt = cnt;
  GOCTR_HIP(hipEventRecord(me.ready, e.stream));
  if (g.barrier()) return -1;
  for (int p = 0; p < g.world; ++p)
    if (p != e.rank) GOCTR_HIP(hipStreamWaitEvent(e.stream, g.slot[p].ready, 0));
  const int rc = body(g);
  if (rc) { g.abort(); return -1; }
  GOCTR_HIP(hipEventRecord(me.done, e.stream));
  if (g.barrier()) return -1;
  for (int p = 0; p < g.world; ++p)
    if (p != e.rank) GOCTR_HIP(hipStreamWaitEvent(e.stream, g.slot[p].done, 0));
  return 0;
}

template <typename T>
int loop_allreduce(Engine& e, T* dev, size_t n) {
  LoopGroup::Slot& me = e.loop->slot[e.rank];
  if (me.scratch.ensure(n * sizeof(T), false)) { e.loop->abort(); return -1; }
  T* out = reinterpret_cast<T*>(me.scratch.p);
  const int rc = loop_collective(e, dev, nullptr, nullptr, [&](LoopGroup& g) -> int {
    LoopPtrs ptrs{};
    for (int p = 0; p < g.world; ++p) ptrs.p[p] = g.slot[p].pub;
    const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((n + 255) / 256, 1), 1024);
    hipLaunchKernelGGL((loop_sum_kernel<T>), dim3(blocks), dim3(256), 0, e.stream, ptrs, g.world, out, n);
    GOCTR_HIP(hipGetLastError());
    return 0;
  });
  if (rc) return -1;
  GOCTR_HIP(hipMemcpyAsync(dev, out, n * sizeof(T), hipMemcpyDeviceToDevice, e.stream));
  return 0;
}

int loop_allgather(Engine& e, const void* send, void* recv, size_t bytes) {
  return loop_collective(e, send, nullptr, nullptr, [&](LoopGroup& g) -> int {
    for (int p = 0; p < g.world; ++p)
      GOCTR_HIP(hipMemcpyAsync(static_cast<char*>(recv) + (size_t)p * bytes, g.slot[p].pub, bytes, hipMemcpyDeviceToDevice, e.stream));
    return 0;
  });
}

int loop_alltoallv(Engine& e, const void* send, const size_t* send_off, const size_t* send_cnt, void* recv, const size_t* recv_off,
                   const size_t* recv_cnt, int bytes_per_elem) {
  return loop_collective(e, send, send_off, send_cnt, [&](LoopGroup& g) -> int {
    for (int p = 0; p < g.world; ++p) {
      const size_t n = g.slot[p].cnt[e.rank];
      GOCTR_CHECK(n == recv_cnt[p], "loop-back all-to-all: rank %d sends %zu elements to rank %d, which expects %zu", p, n, e.rank, recv_cnt[p]);
      if (!n) continue;
      GOCTR_HIP(hipMemcpyAsync(static_cast<char*>(recv) + recv_off[p] * bytes_per_elem,
                               static_cast<const char*>(g.slot[p].pub) + g.slot[p].off[e.rank] * bytes_per_elem, n * bytes_per_elem,
                               hipMemcpyDeviceToDevice, e.stream));
    }
    return 0;
  });
}

int loop_broadcast(Engine& e, void* dev, size_t bytes, int root) {
  return loop_collective(e, dev, nullptr, nullptr, [&](LoopGroup& g) -> int {
    if (e.rank != root) GOCTR_HIP(hipMemcpyAsync(dev, g.slot[root].pub, bytes, hipMemcpyDeviceToDevice, e.stream));
    return 0;
  });
}

}  // namespace

int comm_allreduce_f32(float* dev, size_t n) {
  Engine& e = engine();
  if (!e.comm_active()) return 0;
  if (e.loop) return loop_allreduce<float>(e, dev, n);
  GOCTR_NCCL(g_rccl.AllReduce(dev, dev, n, ncclFloat32, ncclSum, (ncclComm_t)e.nccl_comm, e.stream));
  return 0;
}
int comm_allreduce_f64_dev(double* dev, size_t n) {
  Engine& e = engine();
  if (!e.comm_active()) return 0;
  if (e.loop) return loop_allreduce<double>(e, dev, n);
  GOCTR_NCCL(g_rccl.AllReduce(dev, dev, n, ncclFloat64, ncclSum, (ncclComm_t)e.nccl_comm, e.stream));
  return 0;
}
// all-gather of n int32 per rank (recv = [world][n])
int comm_allgather_i32(const int* send, int* recv, size_t n) {
  Engine& e = engine();
  GOCTR_CHECK(e.comm_active(), "comm_allgather_i32: no communicator");
  if (e.loop) return loop_allgather(e, send, recv, n * sizeof(int));
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
  if (e.loop) return loop_alltoallv(e, send, send_off, send_cnt, recv, recv_off, recv_cnt, bytes_per_elem);
  const ncclDataType_t ty = bytes_per_elem == 4 ? ncclInt32 : ncclInt64;
  GOCTR_NCCL(g_rccl.GroupStart());
  for (int p = 0; p < e.world; ++p) {
    if (send_cnt[p]) GOCTR_NCCL(g_rccl.Send(static_cast<const char*>(send) + send_off[p] * bytes_per_elem, send_cnt[p], ty, p, (ncclComm_t)e.nccl_comm, e.stream));
    if (recv_cnt[p]) GOCTR_NCCL(g_rccl.Recv(static_cast<char*>(recv) + recv_off[p] * bytes_per_elem, recv_cnt[p], ty, p, (ncclComm_t)e.nccl_comm, e.stream));
  }
  GOCTR_NCCL(g_rccl.GroupEnd());
  return 0;
}
int comm_broadcast(void* dev, size_t bytes, int root) {
  Engine& e = engine();
  if (!e.comm_active()) return 0;
  if (e.loop) return loop_broadcast(e, dev, bytes, root);
  GOCTR_NCCL(g_rccl.Broadcast(dev, dev, bytes, ncclChar, root, (ncclComm_t)e.nccl_comm, e.stream));
  return 0;
}
namespace {
__global__ void capture_test_fill_kernel(float* buf, int n, int rank, int round) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) buf[i] = (float)((i * 31 + rank * 7 + round) % 97) * 0.125f - 3.0f;
}
__global__ void capture_test_cmp_kernel(const float* a, const float* b, int n, float* mismatches) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && __float_as_uint(a[i]) != __float_as_uint(b[i])) atomicAdd(mismatches, 1.0f);
}
}  // namespace

// Collective: every rank of the communicator calls it at the same point (goctr_comm_init; the start of the first multi-device
// call of a goctr_init_devices group).  Two rounds of [fill | all-reduce] as a replayed graph against the same eagerly; the mismatch counts (and a
// failed capture, counted as one) are summed over the ranks with an EAGER all-reduce, so all ranks reach the same verdict.
// Waits go through comm_watch_stream: a replay that never completes fails the call on a timeout instead of hanging it.
// Returns 1 = captured collectives work, 0 = they do not (the communicator is healthy: the collective stays between graph
// launches), -1 = the COMMUNICATOR was lost while probing (a wait timed out or RCCL reported an error, so comm_watch_stream
// aborted it; or this rank could not take part at all) -- the error text is set and the caller must fail its call: a rank that
// carried on without a communicator would train alone on its shard, silently.  Every path either takes part in
// all of the probe's collectives or aborts the communicator, so that no rank is left inside a collective its peers skipped.
int comm_capture_selftest() {
  Engine& e = engine();
  if (e.capture_state != 0) return e.capture_state == 1 ? 1 : 0;
  if (!comm_capturable()) { e.capture_state = -1; return 0; }
  const int n = 4096;
  // (generous: the first collective of a fresh 8-rank communicator builds its connections -- and a timeout here is fatal for the rank)
  const int wait_s = 180;
  auto lost = [&](const char* what) {
    // (comm_watch_stream has already aborted on a timeout / asynchronous error; abort here for the synchronous failures, which
    // would otherwise leave the peers waiting inside the next collective of the probe)
    std::string why = goctr_last_error() ? goctr_last_error() : "";
    comm_abort_on_failure();
    e.capture_state = -1;
    set_error("captured-collective self-test: %s%s%s -- the communicator was aborted; this rank cannot continue the job", what,
              why.empty() ? "" : ": ", why.c_str());
    return -1;
  };
  DevBuf<float> cap, eag, bad;
  if (cap.alloc(n, false) || eag.alloc(n, false) || bad.alloc(1)) return lost("device allocation failed");
  hipStream_t s = e.stream;
  // one EAGER collective first: a communicator sets up its connections (buffers, IPC handles, proxy threads) inside its first
  // collective, and none of that may happen inside a stream capture
  hipLaunchKernelGGL(capture_test_fill_kernel, dim3(n / 256), dim3(256), 0, s, eag.p, n, e.rank, 7);
  if (g_rccl.AllReduce(eag.p, eag.p, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)e.nccl_comm, s) != ncclSuccess)
    return lost("the first all-reduce could not be enqueued");
  if (comm_watch_stream(wait_s)) return lost("the first all-reduce did not complete");
  hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
  bool captured = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
  if (captured) {
    const ncclResult_t r = g_rccl.AllReduce(cap.p, cap.p, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)e.nccl_comm, s);
    const hipError_t ce = hipStreamEndCapture(s, &g);
    captured = r == ncclSuccess && ce == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess;
  }
  (void)hipGetLastError();
  float local_bad = captured ? 0.f : 1.f;
  const char* fail = nullptr;
  for (int round = 0; round < 2 && !fail; ++round) {
    hipLaunchKernelGGL(capture_test_fill_kernel, dim3(n / 256), dim3(256), 0, s, cap.p, n, e.rank, round);
    hipLaunchKernelGGL(capture_test_fill_kernel, dim3(n / 256), dim3(256), 0, s, eag.p, n, e.rank, round);
    if (captured && hipGraphLaunch(ge, s) != hipSuccess) { captured = false; local_bad = 1.f; (void)hipGetLastError(); }
    if (g_rccl.AllReduce(eag.p, eag.p, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)e.nccl_comm, s) != ncclSuccess) fail = "an eager all-reduce could not be enqueued";
    if (!fail && captured) hipLaunchKernelGGL(capture_test_cmp_kernel, dim3(n / 256), dim3(256), 0, s, cap.p, eag.p, n, bad.p);
  }
  if (!fail && local_bad != 0.f && hipMemcpyAsync(bad.p, &local_bad, sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) fail = "host-to-device copy failed";
  if (!fail && g_rccl.AllReduce(bad.p, bad.p, 1, ncclFloat32, ncclSum, (ncclComm_t)e.nccl_comm, s) != ncclSuccess) fail = "the verdict all-reduce could not be enqueued";
  float total_bad = 1.f;
  // (a replay that never completes: the watchdog aborts the communicator -- the stream behind a hung collective cannot be
  // used again, so this IS fatal for the rank, and said so)
  if (!fail && comm_watch_stream(wait_s)) fail = "a replayed or eager collective of the probe did not complete";
  if (!fail && bad.download(&total_bad, 1)) fail = "device-to-host copy failed";
  if (ge) (void)hipGraphExecDestroy(ge);
  if (g) (void)hipGraphDestroy(g);
  if (fail) return lost(fail);
  e.capture_state = total_bad == 0.f ? 1 : -1;
  if (e.capture_state != 1 && e.rank == 0)
    fprintf(stderr, "goctr: captured RCCL all-reduce failed its self-test (results differ from the eager collective, or the capture "
                    "failed on a rank): the data-parallel step keeps the collective between graph launches\n");
  return e.capture_state == 1 ? 1 : 0;
}

int comm_group_reset() {
  Engine& e = *engine_at(0);
  if (e.loop) {
    std::lock_guard<std::mutex> lk(e.loop->mu);
    e.loop->aborted = false; e.loop->arrived = 0;
    return 0;
  }
  GOCTR_CHECK(e.nccl_comm, "the group's RCCL communicator was aborted by a failed call; restart the process");
  return 0;
}
bool comm_capturable() { return engine().comm_active() && engine().nccl_comm != nullptr && engine().loop == nullptr; }

// A data-parallel call that fails on ONE rank between two collectives (a HIP error, a failed allocation) must not leave its
// peers blocked inside the next collective.  Loop-back: the group's abort flag wakes every rank waiting in the barrier (and a
// rank that never arrives trips the barrier's timeout).  RCCL: ncclCommAbort is LOCAL -- it frees this rank; peers already
// spinning inside a collective kernel only stop when THEY poll ncclCommGetAsyncError and abort, which comm_watch() below does
// for a rank that waits on its stream; a peer blocked elsewhere is the launcher's job (goctr_amd/launch.py stops every rank
// when one exits non-zero).  The communicator is unusable afterwards; goctr_comm_init builds a new one, and a repeated
// goctr_init_devices notices the missing half (comm_group_live), drops the rest (comm_group_drop) and builds the group anew.
void comm_abort_on_failure() {
  Engine& e = engine();
  if (e.loop) { e.loop->abort(); return; }
  if (!e.nccl_comm) return;
  if (g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)e.nccl_comm);
  else if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)e.nccl_comm);
  e.nccl_comm = nullptr; e.rank = 0; e.world = 1;
}

// Host-side watchdog for a rank that waits for its stream behind RCCL collectives: polls the stream and the communicator's
// asynchronous error state; on an error (a peer aborted or died) or after GOCTR_COMM_TIMEOUT_S (default 300) it aborts the
// local communicator and fails the call instead of blocking in hipStreamSynchronize for ever.
int comm_watch_stream(int timeout_override_s) {
  Engine& e = engine();
  if (!e.nccl_comm || !g_rccl.CommGetAsyncError) { GOCTR_HIP(hipStreamSynchronize(e.stream)); return 0; }
  const auto t0 = std::chrono::steady_clock::now();
  const int timeout_s = timeout_override_s > 0 ? timeout_override_s : std::max(1, env_int_comm("GOCTR_COMM_TIMEOUT_S", 300));
  for (unsigned spin = 0;; ++spin) {
    const hipError_t q = hipStreamQuery(e.stream);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) { set_error("stream failed behind a collective: %s", hipGetErrorString(q)); comm_abort_on_failure(); return -1; }
    if ((spin & 1023) == 1023) {
      ncclResult_t ar = ncclSuccess;
      if (g_rccl.CommGetAsyncError((ncclComm_t)e.nccl_comm, &ar) != ncclSuccess || (ar != ncclSuccess && ar != ncclInProgress)) {
        set_error("RCCL reported an asynchronous error (%s): a peer failed; communicator aborted", g_rccl.GetErrorString ? g_rccl.GetErrorString(ar) : "?");
        comm_abort_on_failure();
        return -1;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) {
        set_error("collective did not complete within %d s (GOCTR_COMM_TIMEOUT_S); communicator aborted", timeout_s);
        comm_abort_on_failure();
        return -1;
      }
    }
  }
}

// What is left of a goctr_init_devices group's RCCL communicators after a rank aborted its own (comm_abort_on_failure is local):
// the other ranks' halves are useless without it.  goctr_init_devices calls this before it builds the group anew.
void comm_group_drop(int n) {
  for (int k = 0; k < n; ++k) {
    Engine* e = engine_at(k);
    if (!e) continue;
    if (e->nccl_comm) {
      EngineScope on(e);
      if (g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)e->nccl_comm);
      else if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)e->nccl_comm);
      e->nccl_comm = nullptr;
    }
    e->capture_state = 0; e->comm_enabled = false;
  }
}
// is the group's communicator still there on every rank?  (n == 1 without GOCTR_FORCE_COMM has none by design)
bool comm_group_live(int n) {
  const char* force = getenv("GOCTR_FORCE_COMM");
  if (n == 1 && !(force && *force && *force != '0')) return true;
  for (int k = 0; k < n; ++k) {
    Engine* e = engine_at(k);
    if (!e || (!e->nccl_comm && !e->loop)) return false;
  }
  return true;
}

// The communicator of a goctr_init_devices group: RCCL when every engine has its own device (GOCTR_COMM=loopback forces the
// other), the loop-back communicator when a device repeats (RCCL rejects duplicate GPUs).  n == 1: none, unless
// GOCTR_FORCE_COMM=1 (a one-rank communicator exercises the split step path on a single GPU).
int comm_group_init(int n) {
  const char* force = getenv("GOCTR_FORCE_COMM");
  const bool forced = force && *force && *force != '0';
  if (n == 1 && !forced) return 0;
  GOCTR_CHECK(n <= kLoopMaxWorld, "goctr_init_devices: more than %d ranks", kLoopMaxWorld);
  bool distinct = true;
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b) distinct = distinct && engine_at(a)->device != engine_at(b)->device;
  const char* mode = getenv("GOCTR_COMM");
  const bool want_loop = mode && (mode[0] == 'l' || mode[0] == 'L');
  GOCTR_CHECK(distinct || !(mode && (mode[0] == 'r' || mode[0] == 'R')),
              "goctr_init_devices: GOCTR_COMM=rccl needs distinct devices (RCCL rejects duplicate GPUs)");
  if (distinct && !want_loop) {
    if (load_rccl()) return -1;
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    std::vector<int> devs((size_t)n);
    for (int k = 0; k < n; ++k) devs[k] = engine_at(k)->device;
    GOCTR_NCCL(g_rccl.CommInitAll(comms.data(), n, devs.data()));
    for (int k = 0; k < n; ++k) { Engine* e = engine_at(k); e->nccl_comm = comms[k]; e->rank = k; e->world = n; e->comm_enabled = false; }
    return 0;
  }
  // loop-back: kernels of one rank read the other ranks' buffers directly -- peer access between the distinct devices
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      const int da = engine_at(a)->device, db = engine_at(b)->device;
      if (da == db) continue;
      GOCTR_HIP(hipSetDevice(da));
      const hipError_t pe = hipDeviceEnablePeerAccess(db, 0);
      if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();
        set_error("loop-back communicator: device %d cannot access device %d (%s)", da, db, hipGetErrorString(pe));
        return -1;
      }
      (void)hipGetLastError();
    }
  LoopGroup* g = new LoopGroup;      // (lives as long as the process, like the engines)
  g->world = n;
  g->slot = std::vector<LoopGroup::Slot>((size_t)n);
  for (int k = 0; k < n; ++k) {
    Engine* e = engine_at(k);
    g->members.push_back(e);
    EngineScope on(e);
    GOCTR_HIP(hipEventCreateWithFlags(&g->slot[k].ready, hipEventDisableTiming));
    GOCTR_HIP(hipEventCreateWithFlags(&g->slot[k].done, hipEventDisableTiming));
  }
  for (int k = 0; k < n; ++k) { Engine* e = engine_at(k); e->loop = g; e->rank = k; e->world = n; e->comm_enabled = false; }
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
  GOCTR_CHECK(!e.loop, "goctr_comm_init: this engine belongs to a goctr_init_devices group (loop-back communicator)");
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
  e.capture_state = 0;
  // may the data-parallel step graphs hold the all-reduce?  Decided HERE, where every rank is (comm_capture_selftest is a collective)
  if (env_int_comm("GOCTR_DP_CAPTURE_COMM", 1) == 1 && comm_capture_selftest() < 0) return -1;   // (communicator lost: say so)
  GOCTR_CHECK(e.nccl_comm, "goctr_comm_init: the communicator did not survive its first collectives");
  return 0;
}

int goctr_comm_group_enable(int on) {
  GOCTR_ENTER();
  Engine& e = engine();
  GOCTR_CHECK(e.loop || e.nccl_comm, "goctr_comm_group_enable: this engine has no communicator (goctr_init_devices with n > 1)");
  e.comm_enabled = on != 0;
  return 0;
}

int goctr_comm_capture_mode(int* mode) {
  GOCTR_CHECK(mode, "goctr_comm_capture_mode: null argument");
  const char* v = getenv("GOCTR_DP_CAPTURE_COMM");
  const Engine& e = engine();
  *mode = (v && *v == '0') || e.loop ? -1 : e.capture_state;
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
  Engine& e = engine();
  std::lock_guard<std::recursive_mutex> lk(e.mu);
  if (e.loop) return 0;             // (a goctr_init_devices group lives as long as the process)
  if (e.nccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)e.nccl_comm);
  e.nccl_comm = nullptr; e.rank = 0; e.world = 1;
  return 0;
}

}  // extern "C"
