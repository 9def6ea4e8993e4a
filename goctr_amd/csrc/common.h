// common.h -- engine-wide plumbing for libgoctr_hip.so (gfx950 only): error strings, the bound
// device + stream, device buffers, the optional RCCL communicator, per-kernel hipEvent timers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/goctr.h"

namespace goctr {

void set_error(const char* fmt, ...);

#define GOCTR_HIP(call)                                                                       \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::goctr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return -1;                                                                              \
    }                                                                                         \
  } while (0)

#define GOCTR_CHECK(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ::goctr::set_error(__VA_ARGS__);    \
      return -1;                          \
    }                                     \
  } while (0)

struct LoopGroup;   // loop-back communicator shared by the engines of one goctr_init_devices group (comm.hip)

// One ENGINE = one logical rank: a HIP device binding with its own streams, device arena, lock, profiler and communicator
// slot.  A process holds one engine (goctr_init: one process per GPU, the launcher path) or several (goctr_init_devices:
// one process drives n ranks from n host threads -- distinct devices over RCCL, or several logical ranks on ONE device over
// the loop-back communicator).  Every handle (goctr_model, goctr_emb, ...) remembers the engine it was created on; an entry
// point that takes a handle runs on the handle's engine whatever thread calls it (EngineScope below).
struct Engine {
  int index = 0;                  // position in the process' engine list (= the rank goctr_init_devices gave it)
  bool inited = false;
  int device = -1;
  std::recursive_mutex mu;        // GOCTR_ENTER: calls that queue work on this engine's main stream are serialised
  // device arena (engine.hip): ONE large hipMalloc per engine, first-fit free list
  struct Arena {
    char* base = nullptr;
    size_t size = 0;
    std::map<size_t, size_t> free_blocks;   // offset -> length
    std::map<size_t, size_t> used;          // offset -> length
    std::mutex mu;
  } arena;
  bool kernel_attrs_done = false, mlp_attrs_done = false;   // hipFuncSetAttribute is per device: once per engine
  void* serve_pool = nullptr;     // ctr.hip: this engine's serving slots (streams + pinned staging live on its device)
  int compute_units = 0;
  bool large_bar = false;      // the host can write device memory through the PCIe BAR (hipDeviceProp_t::isLargeBar)
  hipStream_t stream = nullptr;   // the engine's main stream
  hipStream_t side = nullptr;     // forked inside captured step graphs for independent kernels
  // stream the launch helpers of THE CALLING THREAD currently target: the main stream unless a StreamScope switched it
  // (a serving slot's stream, ctr.hip).  Thread-local: serving calls of several host threads run beside each other
  struct ActiveStream {
    operator hipStream_t() const;
    ActiveStream& operator=(hipStream_t s);
  } active;
  hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join = nullptr;
  // data-parallel communicator (comm.hip)
  int rank = 0, world = 1;
  void* nccl_comm = nullptr;
  LoopGroup* loop = nullptr;      // loop-back communicator (several logical ranks of one process, comm.hip)
  // true when the step must take the split path (reduce -> all-reduce -> Adam): world > 1, or a one-rank
  // communicator forced with GOCTR_FORCE_COMM=1 (exercises the RCCL path on a single-GPU box)
  // A goctr_init_devices group's communicator takes part only in calls that run on ALL its ranks (the single-call multi-device
  // entries switch it on for their duration, goctr_comm_group_enable for explicit per-rank threads): a plain call on one engine
  // of the group is a single-device call.  goctr_comm_init's communicator (one process per GPU) is always on.
  bool comm_enabled = true;
  int capture_state = 0;          // comm_capture_selftest: 0 not tested, 1 captured collectives work, -1 they do not
  // goctr_engine_call_ms: events around this rank's part of the last multi-device training call (train_multi)
  hipEvent_t call_begin = nullptr, call_end = nullptr; bool call_timed = false;
  bool comm_active() const { return (nccl_comm != nullptr || loop != nullptr) && comm_enabled; }
  // rank / world of the CALL in flight: a group engine running a single-device call is rank 0 of 1
  int eff_rank() const { return comm_active() ? rank : 0; }
  int eff_world() const { return comm_active() ? world : 1; }
  // profiling
  bool prof = false;
  double prof_ms[GOCTR_K_COUNT] = {0};
  int64_t prof_n[GOCTR_K_COUNT] = {0};
  const char* prof_kernel[GOCTR_K_COUNT] = {nullptr};   // symbol of the kernel a family's last launch ran (goctr_prof_kernel)
  struct Pending { int id; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
};
// the engine the CALLING THREAD is bound to: the one an EngineScope switched to, else the thread's selection
// (goctr_engine_select), else the process' first engine
Engine& engine();
Engine* engine_at(int k);     // k-th engine of the process, or nullptr
int engine_count();           // engines created so far (>= 1)
Engine* engine_create();      // appends an (un-initialised) engine; engine.hip
int engine_bind(Engine& e, int device_ordinal);   // hipSetDevice + streams + events (what goctr_init does for engine 0)
int require_engine();  // 0 if the calling thread's engine is bound to a device, else sets the error and returns -1

// Binds the calling thread to `e` (null: keep the thread's current engine) for the scope: engine() returns it and the
// thread's HIP device is e's device (hipSetDevice is per host thread).
struct EngineScope {
  Engine* prev;
  int prev_device = -1;       // the thread's HIP device before the scope switched it (restored by the destructor)
  explicit EngineScope(Engine* e);
  ~EngineScope();
};
template <class H> inline Engine* handle_engine(const H* h) { return h ? h->eng : nullptr; }
// run fn(rank) for rank = 0 .. n-1, each on a host thread bound to engine rank (rank 0 on the calling thread); returns -1
// and the first failing rank's error text if any rank failed.  The ranks may block on each other (collectives).
int run_on_engines(int n, const std::function<int(int)>& fn);

// Every C-ABI entry point that queues work on an engine's MAIN stream (training, uploads, dataset builds, the captured
// step graphs) starts with GOCTR_ENTER() / GOCTR_ENTER_H(handle): those calls are serialised per engine, whatever handles
// they use.  Recursive: entry points call each other (goctr_train_dense -> goctr_dataset_create_dense -> ...).
// The serving entry points (goctr_batch_predict / goctr_rank / goctr_predict_dense: what concurrent gin handler goroutines
// call, recommend/api.go:106-131) do NOT take the lock: each runs on a serving slot with its own stream, staging buffers and
// forward workspace (ctr.hip: ServeSlot) under a shared lock of the model.
#define GOCTR_ENTER_ON(eng_ptr)                                   \
  ::goctr::EngineScope _goctr_engine_scope(eng_ptr);              \
  if (::goctr::require_engine()) return -1;                       \
  std::lock_guard<std::recursive_mutex> _goctr_engine_lock(::goctr::engine().mu)
#define GOCTR_ENTER() GOCTR_ENTER_ON(nullptr)
// entry points that take a handle run on the handle's engine
#define GOCTR_ENTER_H(h) GOCTR_ENTER_ON(::goctr::handle_engine(h))
// two handles of one call must live on the same engine
#define GOCTR_SAME_ENGINE(a, b) \
  GOCTR_CHECK(!(a) || !(b) || (a)->eng == (b)->eng, "%s: the handles were created on different engines (devices)", __func__)

// phase stamps / debug prints of one kernel family: GOCTR_DBG=chain,tn,mlp,knn (any subset; read at call time)
inline bool dbg_on(const char* what) {
  const char* v = getenv("GOCTR_DBG");
  return v && *v && strstr(v, what) != nullptr;
}

// generation ids for handles whose device pointers get baked into captured graphs (a freed handle's host address may be
// handed out again by malloc; its uid never is)
uint64_t next_uid();

// timing scope used around every launch of a kernel family when profiling is on
struct ProfScope {
  int id; bool on; hipEvent_t a = nullptr, b = nullptr;
  explicit ProfScope(int kernel_id);
  ~ProfScope();
};
void prof_flush();

// RAII switch of the stream the launch helpers target
struct StreamScope {
  hipStream_t prev;
  explicit StreamScope(hipStream_t s) : prev(engine().active) { engine().active = s; }
  ~StreamScope() { engine().active = prev; }
};
// remembers which kernel symbol a profiled family's launch ran (bench.py matches it against the committed rocprofv3
// summary before it quotes that summary's counters)
inline void prof_note_kernel(int id, const char* symbol) { engine().prof_kernel[id] = symbol; }

// Device memory comes from ONE large hipMalloc'd arena (first-fit free list) so that every buffer of
// the engine sits in the same 2 MiB-fragment mapping (few TLB entries) instead of dozens of small
// separately-mapped allocations.  Requests that do not fit fall back to a plain hipMalloc.
void* arena_alloc(size_t bytes);              // from the calling thread's engine
// Fine-grained DEVICE memory that the host may store into through the PCIe BAR (a k-NN call's queries, a serving pass's keys:
// search.hip, ctr.hip), or null: no large BAR, the runtime refuses, or the range is not mapped writable into this process
// (checked in /proc/self/maps before anybody stores into it).  Freed with hipFree.
void* bar_alloc(size_t bytes);
void arena_free(Engine* owner, void* p);      // back to the engine it came from (any thread)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  Engine* owner = nullptr;     // the engine whose arena p came from
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) arena_free(owner, p);
    p = nullptr; n = 0;
  }
  int alloc(size_t count, bool zero = true) {
    release();
    if (count == 0) count = 1;
    owner = &engine();
    p = static_cast<T*>(arena_alloc(count * sizeof(T)));
    if (!p) return -1;
    n = count;
    if (zero) GOCTR_HIP(hipMemsetAsync(p, 0, count * sizeof(T), engine().stream));
    return 0;
  }
  int ensure(size_t count, bool zero = true) { return count <= n && p ? 0 : alloc(count, zero); }
  int upload(const T* host, size_t count, size_t dst_off = 0) {
    GOCTR_HIP(hipMemcpyAsync(p + dst_off, host, count * sizeof(T), hipMemcpyHostToDevice, engine().stream));
    GOCTR_HIP(hipStreamSynchronize(engine().stream));  // host buffer may be freed by the caller
    return 0;
  }
  int download(T* host, size_t count, size_t src_off = 0) const {
    GOCTR_HIP(hipMemcpyAsync(host, p + src_off, count * sizeof(T), hipMemcpyDeviceToHost, engine().stream));
    GOCTR_HIP(hipStreamSynchronize(engine().stream));
    return 0;
  }
};

// Capture `body`'s launches on stream s into an executable graph (uploaded, so that its first launch inside a timed call does not).
// A device-wide wait from ANOTHER host thread invalidates an open capture, ThreadLocal mode notwithstanding (seen once in 14 runs
// of tests/test_gpu_multi.py::test_missing_rank_fails_instead_of_hanging: a peer rank's thread dropping its handles --
// goctr_model_destroy used hipDeviceSynchronize -- while this one captured its step graphs).  The library no longer makes such
// calls, but the host program may: when hipStreamEndCapture reports hipErrorStreamCaptureInvalidated the capture is taken again
// (the collision is transient); restore() puts back whatever host state body() changed.
template <class Body, class Restore>
inline int capture_graph(hipStream_t s, hipGraphExec_t* exec, Body body, Restore restore) {
  for (int attempt = 0;; ++attempt) {
    hipGraph_t g = nullptr;
    GOCTR_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = body();
    const hipError_t ce = hipStreamEndCapture(s, &g);
    // (an invalidated capture shows up as hipErrorStreamCaptureInvalidated, as a launch error inside body(), as a null graph or as
    // a graph that does not instantiate -- all seen)
    hipError_t ie = hipSuccess;
    if (ce == hipSuccess && g && !rc) ie = hipGraphInstantiate(exec, g, nullptr, nullptr, 0);
    if ((ce != hipSuccess || !g || ie != hipSuccess) && attempt < 4) {
      (void)hipGetLastError();
      if (g) (void)hipGraphDestroy(g);
      restore();
      std::this_thread::sleep_for(std::chrono::milliseconds(1 << attempt));
      continue;
    }
    if (rc) { if (g) (void)hipGraphDestroy(g); return -1; }
    if (ce != hipSuccess || ie != hipSuccess) {      // the last retry failed too: drop the graph before reporting
      if (g) (void)hipGraphDestroy(g);
      GOCTR_HIP(ce);
      GOCTR_HIP(ie);
    }
    GOCTR_CHECK(g, "stream capture produced no graph");
    (void)hipGraphUpload(*exec, s);
    (void)hipGraphDestroy(g);
    return 0;
  }
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// collective hooks (comm.hip), on the calling thread's engine: RCCL (one process per GPU, or one process with one engine
// per distinct device) or the loop-back communicator (several logical ranks of one process; any device list).
// in-place sum over ranks on the engine stream; no-op when world == 1
int comm_allreduce_f32(float* dev, size_t n);
int comm_allreduce_f64_dev(double* dev, size_t n);
// a rank whose data-parallel call failed between collectives aborts the communicator so that its peers fail too instead of hanging
void comm_abort_on_failure();
// sparse embedding update (bucketed exchange, emb_train.h): counts all-gather + all-to-all-v of ids / rows
int comm_allgather_i32(const int* send, int* recv, size_t n);
int comm_alltoallv(const void* send, const size_t* send_off, const size_t* send_cnt, void* recv, const size_t* recv_off,
                   const size_t* recv_cnt, int bytes_per_elem);
// root's buffer to every rank (replica set-up of the single-call multi-device entry)
int comm_broadcast(void* dev, size_t bytes, int root);
// waits for the engine stream like hipStreamSynchronize, but behind RCCL collectives it polls the communicator's asynchronous
// error state and a timeout: a peer that failed or died makes this rank's call fail instead of hanging
int comm_watch_stream(int timeout_override_s = 0);
// a goctr_init_devices group before a new multi-rank call: clears the loop-back barrier's abort flag (every rank of the
// previous call has returned -- run_on_engines serialises the calls); fails if an RCCL group communicator was aborted
int comm_group_reset();
// 1: a captured all-reduce replays correctly on this engine's communicator (tested once per communicator, every rank takes part
// and all agree), 0: not (or not capturable at all)
int comm_capture_selftest();
// true when a captured graph may hold this engine's collectives (RCCL: yes; loop-back: host barriers, never)
bool comm_capturable();

}  // namespace goctr
