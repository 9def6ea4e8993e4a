// common.h -- engine-wide plumbing for libgoctr_hip.so (gfx950 only): error strings, the bound
// device + stream, device buffers, the optional RCCL communicator, per-kernel hipEvent timers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
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

struct Engine {
  bool inited = false;
  int device = -1;
  int compute_units = 0;
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
  // true when the step must take the split path (reduce -> all-reduce -> Adam): world > 1, or a one-rank
  // communicator forced with GOCTR_FORCE_COMM=1 (exercises the RCCL path on a single-GPU box)
  bool comm_active() const { return nccl_comm != nullptr; }
  // profiling
  bool prof = false;
  double prof_ms[GOCTR_K_COUNT] = {0};
  int64_t prof_n[GOCTR_K_COUNT] = {0};
  const char* prof_kernel[GOCTR_K_COUNT] = {nullptr};   // symbol of the kernel a family's last launch ran (goctr_prof_kernel)
  struct Pending { int id; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
};
Engine& engine();
int require_engine();  // 0 if goctr_init succeeded, else sets the error and returns -1

// Every C-ABI entry point that queues work on the engine's MAIN stream (training, uploads, dataset builds, the captured
// step graphs) starts with GOCTR_ENTER(): those calls are serialised engine-wide, whatever handles they use.  Recursive:
// entry points call each other (goctr_train_dense -> goctr_dataset_create_dense -> ...).
// The serving entry points (goctr_batch_predict / goctr_rank / goctr_predict_dense: what concurrent gin handler goroutines
// call, recommend/api.go:106-131) do NOT take it: each runs on a serving slot with its own stream, staging buffers and
// forward workspace (ctr.hip: ServeSlot) under a shared lock of the model.
std::recursive_mutex& engine_mutex();
#define GOCTR_ENTER()                              \
  if (::goctr::require_engine()) return -1;        \
  std::lock_guard<std::recursive_mutex> _goctr_engine_lock(::goctr::engine_mutex())

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
void* arena_alloc(size_t bytes);
void arena_free(void* p);

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) arena_free(p);
    p = nullptr; n = 0;
  }
  int alloc(size_t count, bool zero = true) {
    release();
    if (count == 0) count = 1;
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

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// collective hook (comm.hip): in-place sum over ranks on the engine stream; no-op when world == 1
int comm_allreduce_f32(float* dev, size_t n);
int comm_allreduce_f64_dev(double* dev, size_t n);
// a rank whose data-parallel call failed between collectives aborts the communicator so that its peers fail too instead of hanging
void comm_abort_on_failure();
// sparse embedding update (bucketed exchange, emb_train.h): counts all-gather + all-to-all-v of ids / rows
int comm_allgather_i32(const int* send, int* recv, size_t n);
int comm_alltoallv(const void* send, const size_t* send_off, const size_t* send_cnt, void* recv, const size_t* recv_off,
                   const size_t* recv_cnt, int bytes_per_elem);

}  // namespace goctr
