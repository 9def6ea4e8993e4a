// engine.hip -- runtime plumbing of libgoctr_hip.so: device binding, error strings, hipEvent
// per-kernel timers.  (No reference counterpart: go-ctr has no device runtime)
#include "common.h"

#include <atomic>
#include <cstdlib>
#include <map>

namespace goctr {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

Engine& engine() {
  static Engine e;
  return e;
}

// the calling thread's target stream (null = the engine's main stream)
static thread_local hipStream_t t_active = nullptr;
Engine::ActiveStream::operator hipStream_t() const { return t_active ? t_active : engine().stream; }
Engine::ActiveStream& Engine::ActiveStream::operator=(hipStream_t s) {
  t_active = (s == engine().stream) ? nullptr : s;
  return *this;
}

std::recursive_mutex& engine_mutex() {
  static std::recursive_mutex mu;
  return mu;
}

uint64_t next_uid() {
  static std::atomic<uint64_t> n{1};
  return n.fetch_add(1);
}

int require_engine() {
  if (!engine().inited) {
    set_error("goctr: no HIP device bound -- call goctr_init() first (there is no CPU fallback)");
    return -1;
  }
  return 0;
}

static const char* kNames[GOCTR_K_COUNT] = {
    "attn_fwd", "gemm_fwd0", "gemm_fwd1", "gemm_out", "bwd_dz1", "bwd_dz0", "bwd_dp",
    "attn_bwd", "dW0", "dW1", "dW2", "reduce", "allreduce", "adam", "chain", "emb_train", "emb_grad"};

// (main-stream launches only: the event pool is not shared with the serving slots' threads)
ProfScope::ProfScope(int kernel_id) : id(kernel_id), on(engine().prof && t_active == nullptr) {
  if (!on) return;
  Engine& e = engine();
  auto get = [&]() {
    hipEvent_t ev = nullptr;
    if (!e.event_pool.empty()) { ev = e.event_pool.back(); e.event_pool.pop_back(); }
    else (void)hipEventCreate(&ev);
    return ev;
  };
  a = get(); b = get();
  (void)hipEventRecord(a, e.active);
}
ProfScope::~ProfScope() {
  if (!on) return;
  Engine& e = engine();
  (void)hipEventRecord(b, e.active);
  e.pending.push_back({id, a, b});
  if (e.pending.size() > 4096) prof_flush();
}
void prof_flush() {
  Engine& e = engine();
  if (e.pending.empty()) return;
  (void)hipStreamSynchronize(e.stream);
  for (auto& p : e.pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { e.prof_ms[p.id] += ms; e.prof_n[p.id] += 1; }
    e.event_pool.push_back(p.a); e.event_pool.push_back(p.b);
  }
  e.pending.clear();
}

// ---------------------------------------------------------------- device arena
namespace {
struct Arena {
  char* base = nullptr;
  size_t size = 0;
  std::map<size_t, size_t> free_blocks;   // offset -> length
  std::map<size_t, size_t> used;          // offset -> length
  std::mutex mu;
};
Arena g_arena;
}  // namespace

void* arena_alloc(size_t bytes) {
  std::lock_guard<std::mutex> lk(g_arena.mu);
  Arena& a = g_arena;
  const size_t need = (bytes + 255) / 256 * 256;
  if (!a.base) {
    const char* ev = getenv("GOCTR_ARENA_MB");
    size_t mb = ev && *ev ? (size_t)atoll(ev) : 2048;
    if (mb > 0 && hipMalloc((void**)&a.base, mb << 20) == hipSuccess) {
      a.size = mb << 20;
      a.free_blocks[0] = a.size;
    } else {
      (void)hipGetLastError();
      a.base = nullptr; a.size = 0;
    }
  }
  if (a.base) {
    for (auto it = a.free_blocks.begin(); it != a.free_blocks.end(); ++it) {
      if (it->second >= need) {
        const size_t off = it->first, len = it->second;
        a.free_blocks.erase(it);
        if (len > need) a.free_blocks[off + need] = len - need;
        a.used[off] = need;
        return a.base + off;
      }
    }
  }
  void* p = nullptr;
  if (hipMalloc(&p, need) != hipSuccess) {
    set_error("device allocation of %zu bytes failed", need);
    return nullptr;
  }
  return p;
}

void arena_free(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_arena.mu);
  Arena& a = g_arena;
  char* c = static_cast<char*>(p);
  if (a.base && c >= a.base && c < a.base + a.size) {
    size_t off = (size_t)(c - a.base);
    auto u = a.used.find(off);
    if (u == a.used.end()) return;
    size_t len = u->second;
    a.used.erase(u);
    auto nxt = a.free_blocks.lower_bound(off);
    if (nxt != a.free_blocks.end() && off + len == nxt->first) { len += nxt->second; nxt = a.free_blocks.erase(nxt); }
    if (nxt != a.free_blocks.begin()) {
      auto prv = std::prev(nxt);
      if (prv->first + prv->second == off) { off = prv->first; len += prv->second; a.free_blocks.erase(prv); }
    }
    a.free_blocks[off] = len;
    return;
  }
  (void)hipFree(p);
}

}  // namespace goctr

using namespace goctr;

extern "C" {

const char* goctr_last_error(void) { return g_err.c_str(); }
const char* goctr_version(void) { return "goctr-hip 0.1 (gfx950)"; }

int goctr_device_count(int* n) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { c = 0; (void)hipGetLastError(); }
  *n = c;
  return 0;
}

int goctr_init(int device_ordinal) {
  std::lock_guard<std::recursive_mutex> lk(engine_mutex());
  Engine& e = engine();
  int n = 0;
  goctr_device_count(&n);
  GOCTR_CHECK(n > 0, "goctr_init: no HIP device visible (this engine has no CPU fallback)");
  GOCTR_CHECK(device_ordinal >= 0 && device_ordinal < n, "goctr_init: device %d out of range (have %d)", device_ordinal, n);
  if (e.inited && e.device == device_ordinal) return 0;
  GOCTR_HIP(hipSetDevice(device_ordinal));
  {
    // GOCTR_SYNC=spin|yield|block (experiments): how the host waits in goctr_sync / blocking copies
    const char* sm = getenv("GOCTR_SYNC");
    if (sm && *sm) {
      const unsigned f = sm[0] == 's' ? hipDeviceScheduleSpin : (sm[0] == 'y' ? hipDeviceScheduleYield : hipDeviceScheduleBlockingSync);
      (void)hipSetDeviceFlags(f);
      (void)hipGetLastError();
    }
  }
  hipDeviceProp_t prop;
  GOCTR_HIP(hipGetDeviceProperties(&prop, device_ordinal));
  GOCTR_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
              "goctr_init: device is %s; this library is built for gfx950 (MI355X) only", prop.gcnArchName);
  e.compute_units = prop.multiProcessorCount;
  if (!e.stream) {
    GOCTR_HIP(hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking));
    GOCTR_HIP(hipStreamCreateWithFlags(&e.side, hipStreamNonBlocking));
    for (auto& ev : e.ev_fork) GOCTR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    GOCTR_HIP(hipEventCreateWithFlags(&e.ev_join, hipEventDisableTiming));
  }
  e.device = device_ordinal;
  e.inited = true;
  return 0;
}

int goctr_sync(void) {
  GOCTR_ENTER();
  GOCTR_HIP(hipStreamSynchronize(engine().stream));
  GOCTR_HIP(hipDeviceSynchronize());
  return 0;
}

int goctr_device_info(char* name, size_t cap, int* cus, int64_t* hbm) {
  GOCTR_ENTER();
  hipDeviceProp_t prop;
  GOCTR_HIP(hipGetDeviceProperties(&prop, engine().device));
  if (name && cap) snprintf(name, cap, "%s (%s)", prop.name, prop.gcnArchName);
  if (cus) *cus = prop.multiProcessorCount;
  if (hbm) *hbm = (int64_t)prop.totalGlobalMem;
  return 0;
}

int goctr_prof_enable(int on) {
  GOCTR_ENTER();
  prof_flush();
  engine().prof = on != 0;
  return 0;
}
int goctr_prof_reset(void) {
  GOCTR_ENTER();
  prof_flush();
  for (int i = 0; i < GOCTR_K_COUNT; ++i) { engine().prof_ms[i] = 0; engine().prof_n[i] = 0; }
  return 0;
}
int goctr_prof_get(int id, double* ms, int64_t* n) {
  GOCTR_ENTER();
  GOCTR_CHECK(id >= 0 && id < GOCTR_K_COUNT, "goctr_prof_get: bad kernel id %d", id);
  prof_flush();
  if (ms) *ms = engine().prof_ms[id];
  if (n) *n = engine().prof_n[id];
  return 0;
}
const char* goctr_prof_name(int id) { return id >= 0 && id < GOCTR_K_COUNT ? kNames[id] : "?"; }
const char* goctr_prof_kernel(int id) {
  const char* k = id >= 0 && id < GOCTR_K_COUNT ? engine().prof_kernel[id] : nullptr;
  return k ? k : "";
}

}  // extern "C"
