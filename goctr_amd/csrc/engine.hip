// engine.hip -- runtime plumbing of libgoctr_hip.so: device binding, error strings, hipEvent
// per-kernel timers.  (No reference counterpart: go-ctr has no device runtime)
#include "common.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <thread>

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

// ---------------------------------------------------------------- engines of the process
// (never destroyed: a static destructor would release streams after the HIP runtime has shut down)
namespace {
constexpr int kMaxEngines = 64;
std::atomic<Engine*> g_engines[kMaxEngines];
std::atomic<int> g_nengines{0};
std::mutex g_engines_mu;
thread_local Engine* t_scope = nullptr;      // EngineScope
thread_local Engine* t_selected = nullptr;   // goctr_engine_select / a rank worker's own engine
}  // namespace

Engine* engine_create() {
  std::lock_guard<std::mutex> lk(g_engines_mu);
  const int n = g_nengines.load();
  if (n >= kMaxEngines) { set_error("more than %d engines", kMaxEngines); return nullptr; }
  Engine* e = new Engine;
  e->index = n;
  g_engines[n].store(e);
  g_nengines.store(n + 1);
  return e;
}
Engine* engine_at(int k) {
  if (k == 0 && g_nengines.load() == 0) (void)engine_create();
  return k >= 0 && k < g_nengines.load() ? g_engines[k].load() : nullptr;
}
int engine_count() { return std::max(1, g_nengines.load()); }

Engine& engine() {
  if (t_scope) return *t_scope;
  if (t_selected) return *t_selected;
  return *engine_at(0);
}

EngineScope::EngineScope(Engine* e) : prev(t_scope) {
  if (e) t_scope = e;
  Engine& cur = engine();
  if (cur.inited) {     // hipSetDevice is per host thread: any thread may call any entry point on any handle
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != cur.device) {
      if (dev >= 0) prev_device = dev;
      (void)hipSetDevice(cur.device);
    }
  }
}
// the thread's HIP device goes back with the engine: after `for k: EngineScope on(engine k)` the thread would otherwise stay on
// the last device while engine() is engine 0 again, and a first-touch arena / fallback hipMalloc for an engine-0 buffer would
// land on the wrong GPU
EngineScope::~EngineScope() {
  t_scope = prev;
  if (prev_device >= 0) (void)hipSetDevice(prev_device);
}

// the calling thread's target stream (null = the engine's main stream)
static thread_local hipStream_t t_active = nullptr;
Engine::ActiveStream::operator hipStream_t() const { return t_active ? t_active : engine().stream; }
Engine::ActiveStream& Engine::ActiveStream::operator=(hipStream_t s) {
  t_active = (s == engine().stream) ? nullptr : s;
  return *this;
}

uint64_t next_uid() {
  static std::atomic<uint64_t> n{1};
  return n.fetch_add(1);
}

int require_engine() {
  if (!engine().inited) {
    set_error("goctr: no HIP device bound -- call goctr_init() / goctr_init_devices() first (there is no CPU fallback)");
    return -1;
  }
  return 0;
}

// ---------------------------------------------------------------- rank workers (one persistent host thread per engine >= 1)
namespace {
struct Worker {
  std::thread th;
  std::mutex mu; std::condition_variable cv;
  const std::function<int(int)>* job = nullptr;
  bool done = false; int rc = 0; std::string err;
};
Worker* g_workers[kMaxEngines] = {nullptr};
std::mutex g_run_mu;      // one multi-engine call at a time

void worker_main(int k) {
  Worker& w = *g_workers[k];
  t_selected = engine_at(k);
  for (;;) {
    const std::function<int(int)>* job;
    {
      std::unique_lock<std::mutex> lk(w.mu);
      w.cv.wait(lk, [&] { return w.job != nullptr; });
      job = w.job;
    }
    int rc;
    {
      EngineScope on(engine_at(k));
      rc = (*job)(k);
    }
    {
      std::lock_guard<std::mutex> lk(w.mu);
      w.rc = rc; w.err = rc ? g_err : std::string(); w.job = nullptr; w.done = true;
    }
    w.cv.notify_all();
  }
}
}  // namespace

int run_on_engines(int n, const std::function<int(int)>& fn) {
  GOCTR_CHECK(n >= 1 && n <= engine_count(), "run_on_engines: %d ranks but %d engines (goctr_init_devices)", n, engine_count());
  std::lock_guard<std::mutex> run_lk(g_run_mu);
  for (int k = 1; k < n; ++k) {
    if (!g_workers[k]) {
      g_workers[k] = new Worker;
      g_workers[k]->th = std::thread(worker_main, k);
      g_workers[k]->th.detach();
    }
    Worker& w = *g_workers[k];
    { std::lock_guard<std::mutex> lk(w.mu); w.done = false; w.job = &fn; }
    w.cv.notify_all();
  }
  int rc0;
  {
    EngineScope on(engine_at(0));
    rc0 = fn(0);
  }
  std::string first = rc0 ? ("rank 0: " + g_err) : std::string();
  int rc = rc0;
  for (int k = 1; k < n; ++k) {
    Worker& w = *g_workers[k];
    std::unique_lock<std::mutex> lk(w.mu);
    w.cv.wait(lk, [&] { return w.done; });
    if (w.rc && first.empty()) first = "rank " + std::to_string(k) + ": " + w.err;
    if (w.rc) rc = -1;
  }
  if (rc) set_error("%s", first.c_str());
  return rc ? -1 : 0;
}

static const char* kNames[GOCTR_K_COUNT] = {
    "attn_fwd", "gemm_fwd0", "gemm_fwd1", "gemm_out", "bwd_dz1", "bwd_dz0", "bwd_dp",
    "attn_bwd", "dW0", "dW1", "dW2", "reduce", "allreduce", "adam", "chain", "emb_train", "emb_grad", "emb_plan"};

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

// ---------------------------------------------------------------- device arena (one per engine)
void* arena_alloc(size_t bytes) {
  Engine& e = engine();
  Engine::Arena& a = e.arena;
  std::lock_guard<std::mutex> lk(a.mu);
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
    (void)hipGetLastError();
    set_error("device allocation of %zu bytes failed", need);
    return nullptr;
  }
  return p;
}

void arena_free(Engine* owner, void* p) {
  if (!p) return;
  if (owner) {
    Engine::Arena& a = owner->arena;
    std::lock_guard<std::mutex> lk(a.mu);
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
  }
  (void)hipFree(p);
}

int engine_bind(Engine& e, int device_ordinal) {
  int n = 0;
  goctr_device_count(&n);
  GOCTR_CHECK(n > 0, "goctr_init: no HIP device visible (this engine has no CPU fallback)");
  GOCTR_CHECK(device_ordinal >= 0 && device_ordinal < n, "goctr_init: device %d out of range (have %d)", device_ordinal, n);
  if (e.inited && e.device == device_ordinal) return 0;
  GOCTR_CHECK(!e.inited, "goctr_init: engine %d is already bound to device %d (asked for %d); one engine, one device", e.index,
              e.device, device_ordinal);
  GOCTR_HIP(hipSetDevice(device_ordinal));
  hipDeviceProp_t prop;
  GOCTR_HIP(hipGetDeviceProperties(&prop, device_ordinal));
  GOCTR_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
              "goctr_init: device is %s; this library is built for gfx950 (MI355X) only", prop.gcnArchName);
  e.compute_units = prop.multiProcessorCount;
  e.large_bar = prop.isLargeBar != 0;
  GOCTR_HIP(hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking));
  GOCTR_HIP(hipStreamCreateWithFlags(&e.side, hipStreamNonBlocking));
  for (auto& ev : e.ev_fork) GOCTR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  GOCTR_HIP(hipEventCreateWithFlags(&e.ev_join, hipEventDisableTiming));
  e.device = device_ordinal;
  e.inited = true;
  return 0;
}

void* bar_alloc(size_t bytes) {
  if (!engine().large_bar || bytes == 0) return nullptr;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
  // the whole range must lie in mappings this process may write (the runtime maps host-accessible device memory at its device
  // address; a device-only allocation is a reserved, inaccessible range or no mapping at all)
  const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
  uintptr_t covered = lo;
  if (FILE* f = fopen("/proc/self/maps", "r")) {
    char line[512];
    while (covered < hi && fgets(line, sizeof line, f)) {
      unsigned long long a = 0, b = 0; char perm[8] = {0};
      if (sscanf(line, "%llx-%llx %7s", &a, &b, perm) != 3) continue;
      if (a <= covered && covered < b) {                       // (the file is sorted by address)
        if (perm[1] != 'w') break;
        covered = (uintptr_t)b;
      }
    }
    fclose(f);
  }
  if (covered < hi) { (void)hipFree(p); return nullptr; }
  return p;
}

// comm.hip: builds the communicator of a goctr_init_devices group (RCCL over the distinct devices, else loop-back)
int comm_group_init(int n);
void comm_group_drop(int n);
bool comm_group_live(int n);

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
  Engine& e = *engine_at(0);
  std::lock_guard<std::recursive_mutex> lk(e.mu);
  EngineScope on(&e);
  return engine_bind(e, device_ordinal);
}

int goctr_init_devices(int n, const int* device_ids) {
  GOCTR_CHECK(n >= 1 && n <= kMaxEngines && device_ids, "goctr_init_devices: bad arguments (1 <= n <= %d)", kMaxEngines);
  int have = 0;
  goctr_device_count(&have);
  GOCTR_CHECK(have > 0, "goctr_init_devices: no HIP device visible (this engine has no CPU fallback)");
  for (int k = 0; k < n; ++k)
    GOCTR_CHECK(device_ids[k] >= 0 && device_ids[k] < have, "goctr_init_devices: device %d out of range (have %d)", device_ids[k], have);
  static std::mutex mu;
  static bool group_ready = false;   // engines bound AND the group's communicator built: only then is a repeat a no-op
  std::lock_guard<std::mutex> once(mu);
  const int existing = g_nengines.load();
  if (existing > 1 || (existing == 1 && (engine_at(0)->nccl_comm || engine_at(0)->loop))) {
    // idempotent for the same list; a different one would need every handle of the old engines gone
    bool same = existing == n;
    for (int k = 0; same && k < n; ++k) same = engine_at(k)->inited && engine_at(k)->device == device_ids[k];
    GOCTR_CHECK(same, "goctr_init_devices: the process already runs %d engine(s); the device list cannot change", existing);
    // ready = built once AND still there on every rank (a rank that aborted its communicator -- a timeout, a failed
    // captured-collective self-test -- left the static flag set, and the next training call failed with "restart the process")
    if (group_ready && comm_group_live(n)) return 0;
    // an earlier call bound the engines and then failed to build the communicator (RCCL would not load, peer access refused),
    // or the group lost a rank's half since: drop what is left and try again instead of reporting success without one
    group_ready = false;
    comm_group_drop(n);
    if (comm_group_init(n)) return -1;
    group_ready = true;
    return 0;
  }
  for (int k = 0; k < n; ++k) {
    Engine* e = k == 0 ? engine_at(0) : (engine_at(k) ? engine_at(k) : engine_create());
    if (!e) return -1;
    std::lock_guard<std::recursive_mutex> lk(e->mu);
    EngineScope on(e);
    if (engine_bind(*e, device_ids[k])) return -1;
    e->rank = k; e->world = n;
  }
  if (comm_group_init(n)) return -1;
  group_ready = true;
  return 0;
}

int goctr_engine_count(int* n) {
  GOCTR_CHECK(n, "goctr_engine_count: null argument");
  *n = engine_count();
  return 0;
}

int goctr_engine_call_ms(int k, double* ms) {
  Engine* e = engine_at(k);
  GOCTR_CHECK(e && e->inited && ms, "goctr_engine_call_ms: engine %d does not exist (goctr_init_devices made %d)", k, engine_count());
  GOCTR_CHECK(e->call_timed, "goctr_engine_call_ms: engine %d has not taken part in a multi-device training call", k);
  EngineScope on(e);
  GOCTR_HIP(hipEventSynchronize(e->call_end));
  float f = 0.f;
  GOCTR_HIP(hipEventElapsedTime(&f, e->call_begin, e->call_end));
  *ms = (double)f;
  return 0;
}

int goctr_engine_select(int k) {
  Engine* e = engine_at(k);
  GOCTR_CHECK(e && e->inited, "goctr_engine_select: engine %d does not exist (goctr_init_devices made %d)", k, engine_count());
  t_selected = e;
  return 0;
}

int goctr_sync(void) {
  if (require_engine()) return -1;
  for (int k = 0; k < engine_count(); ++k) {     // every engine of the process
    Engine* e = engine_at(k);
    if (!e || !e->inited) continue;
    EngineScope on(e);
    std::lock_guard<std::recursive_mutex> lk(e->mu);
    // (the engine's streams, never hipDeviceSynchronize: a device-wide wait invalidates the stream capture of another thread that
    // is building step graphs on the same device; serving slots' streams are idle between calls)
    GOCTR_HIP(hipStreamSynchronize(e->stream));
    GOCTR_HIP(hipStreamSynchronize(e->side));
  }
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
