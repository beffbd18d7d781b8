// rrtmg_abi.hip -- extern "C" surface of librrtmg_hip.so (declared in include/rrtmg_hip.h).
#include <dlfcn.h>
#include <pthread.h>

#include <cstring>
#include <mutex>

#include <atomic>
#include <condition_variable>
#include <memory>

#include <cstddef>

#include "rrtmg_ctx.h"

namespace rrtmg {

const char *status_message(int code) {
  switch (code) {
    case RRTMG_OK: return "ok";
    case RRTMG_ERR_HIP: return "HIP runtime error";
    case RRTMG_ERR_NOT_INITIALISED: return "not initialised";
    case RRTMG_ERR_TABLES: return "table blob error";
    case RRTMG_ERR_ARG: return "bad argument";
    case RRTMG_ERR_PARTIAL_CLOUD: return "PARTIAL CLOUD NOT ALLOWED";
    case RRTMG_ERR_ICE_RADIUS: return "ICE RADIUS OUT OF BOUNDS";
    case RRTMG_ERR_LIQ_RADIUS: return "LIQUID EFFECTIVE RADIUS OUT OF BOUNDS";
    case RRTMG_ERR_KISS_PRESSURE: return "MCICA_SUBCOL: KISSVEC SEED GENERATOR REQUIRES PMID FROM BOTTOM FOUR LAYERS.";
    case RRTMG_ERR_ICLD: return "MCICA_SUBCOL: INVALID ICLD";
    case RRTMG_ERR_INFLAG1_MCICA: return "INFLAG = 1 OPTION NOT AVAILABLE WITH MCICA";
    case RRTMG_ERR_ICE_GEN_SIZE: return "ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS";
    case RRTMG_ERR_ICE_RADIUS_SMALL: return "ICE RADIUS TOO SMALL";
    case RRTMG_ERR_UNSUPPORTED: return "option not supported by RRTMG";
    case RRTMG_ERR_ICE_EXT_NEG: return "ICE EXTINCTION LESS THAN 0.0";
    case RRTMG_ERR_ICE_SSA_GT1: return "ICE SSA GRTR THAN 1.0";
    case RRTMG_ERR_ICE_SSA_NEG: return "ICE SSA LESS THAN 0.0";
    case RRTMG_ERR_ICE_ASYM_GT1: return "ICE ASYM GRTR THAN 1.0";
    case RRTMG_ERR_ICE_ASYM_NEG: return "ICE ASYM LESS THAN 0.0";
    case RRTMG_ERR_FDELTA_NEG: return "FDELTA LESS THAN 0.0";
    case RRTMG_ERR_FDELTA_GT1: return "FDELTA GT THAN 1.0";
    case RRTMG_ERR_LIQ_EXT_NEG: return "LIQUID EXTINCTION LESS THAN 0.0";
    case RRTMG_ERR_LIQ_SSA_GT1: return "LIQUID SSA GRTR THAN 1.0";
    case RRTMG_ERR_LIQ_SSA_NEG: return "LIQUID SSA LESS THAN 0.0";
    case RRTMG_ERR_LIQ_ASYM_GT1: return "LIQUID ASYM GRTR THAN 1.0";
    case RRTMG_ERR_LIQ_ASYM_NEG: return "LIQUID ASYM LESS THAN 0.0";
    default: return "unknown error";
  }
}

int ctx_prepare_device(rrtmg_ctx *ctx) {
  RRTMG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!ctx->stream) RRTMG_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  if (!ctx->stream_lw) RRTMG_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->stream_lw, hipStreamNonBlocking));
  if (!ctx->hint) {
    RRTMG_HIP_CHECK(ctx, hipHostMalloc((void **)&ctx->hint, 2 * sizeof(rrtmg_ctx::CallHint), hipHostMallocDefault));
    for (int w = 0; w < 2; ++w) { ctx->hint[w].ntile = -1; ctx->hint[w].nlay = -1; ctx->hint[w].ncloudy = -1; }
    RRTMG_HIP_CHECK(ctx, hipMalloc((void **)&ctx->ncloudy_dev, 2 * sizeof(int)));
    RRTMG_HIP_CHECK(ctx, hipMemset(ctx->ncloudy_dev, 0, 2 * sizeof(int)));
  }
  if (!ctx->err_dev) {
    RRTMG_HIP_CHECK(ctx, hipMalloc((void **)&ctx->err_dev, 64));
    RRTMG_HIP_CHECK(ctx, hipMemset(ctx->err_dev, 0, 64));
  }
  for (int w = 0; w < 2; ++w)
    for (int k = 0; k < 2; ++k)
      if (!ctx->kiss_ev[w][k]) RRTMG_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->kiss_ev[w][k], hipEventDisableTiming));
  for (int w = 0; w < 2; ++w)
    if (!ctx->sync_ev[w]) RRTMG_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->sync_ev[w], hipEventDisableTiming));
  return RRTMG_OK;
}

int copy_out(rrtmg_ctx *ctx, hipStream_t s, const OutCopy *o, int count, int *herr_dev, int *herr_host) {
  // A destination that is page-locked memory the runtime knows (hipHostMalloc / hipHostRegister: the components' output pool
  // hands such arrays out) takes its copy directly; the others go through the staging buffer.
  constexpr int kMaxOut = 16;
  if (count > kMaxOut) return ctx->fail(RRTMG_ERR_ARG, "copy_out: %d output arrays (at most %d)", count, kMaxOut);
  bool direct[kMaxOut];
  size_t total = 0;
  for (int i = 0; i < count; ++i) {
    hipPointerAttribute_t at;
    direct[i] = hipPointerGetAttributes(&at, o[i].host) == hipSuccess && at.type == hipMemoryTypeHost;
    if (!direct[i]) total += o[i].n * sizeof(double);
  }
  (void)hipGetLastError();   // (an unknown pointer is an error code of the query, not of this call)
  if (total > ctx->pinned_cap) {
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr; ctx->pinned_cap = 0;
    RRTMG_HIP_CHECK(ctx, hipHostMalloc(&ctx->pinned, total, hipHostMallocDefault));
    ctx->pinned_cap = total;
  }
  char *p = (char *)ctx->pinned;
  if (herr_dev) RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(herr_host, herr_dev, sizeof(int), hipMemcpyDeviceToHost, s));
  size_t off = 0;
  for (int i = 0; i < count; ++i) {
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(direct[i] ? (char *)o[i].host : p + off, o[i].dev, o[i].n * sizeof(double), hipMemcpyDeviceToHost, s));
    if (!direct[i]) off += o[i].n * sizeof(double);
  }
  RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));
  if (total == 0) return RRTMG_OK;
  // pinned staging -> the caller's arrays, in slices on a few host threads (first touch of fresh pages dominates)
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt == 0 ? 1 : (nt > 8 ? 8 : nt);
  if (total < (size_t)4 << 20) nt = 1;
  std::vector<std::thread> th;
  auto work = [&](unsigned t) {
    size_t off2 = 0;
    for (int i = 0; i < count; ++i) {
      if (direct[i]) continue;
      const size_t bytes = o[i].n * sizeof(double), per = (bytes / nt + 4095) & ~(size_t)4095;
      const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
      if (lo < bytes) memcpy((char *)o[i].host + lo, p + off2 + lo, hi - lo);
      off2 += bytes;
    }
  };
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto &x : th) x.join();
  return RRTMG_OK;
}

// ---- host-pointer inputs (rrtmg_host_inputs.h) ---------------------------------------------------------------------------
namespace {
// A few persistent host threads that scan input arrays (created on first use, never joined: they sleep between calls and
// end with the process).  The caller of wait() works its OWN batch's queue too, so a batch completes even in a process whose
// pool has no workers, and never pays for another context's batch.  fork(): the child gets a fresh pool (pthread_atfork child
// handler: new mutex, empty queue, workers started on first use) -- the parent's may have been locked by a thread that does
// not exist in the child.
struct ScanJob {
  const uint64_t *q = nullptr;
  size_t n = 0;
  uint64_t first = 0;
  std::atomic<bool> differs{false};
};
constexpr size_t kSliceWords = (size_t)1 << 16;   // 512 KB per task
class HostPool {
 public:
  struct Task { ScanJob *job; size_t lo; };
  // A batch owns its slices: `next` is a cursor into `tasks` (both under the pool's mutex), so the waiter takes from its OWN
  // batch and a worker from the newest batch that has any left in O(1) -- no scan of, and no erase from, a queue shared by
  // every context while the workers need the same lock.
  struct Batch { std::atomic<long> open{0}; std::vector<Task> tasks; size_t next = 0; };
  static HostPool &get() {
    HostPool *p = g_pool.load(std::memory_order_acquire);
    if (p) return *p;
    static std::once_flag atfork_once;
    std::call_once(atfork_once, [] { pthread_atfork(nullptr, nullptr, [] { g_pool.store(nullptr, std::memory_order_release); }); });
    HostPool *fresh = new HostPool();   // (no threads yet: a pool that loses the race below is simply deleted)
    HostPool *expected = nullptr;
    if (g_pool.compare_exchange_strong(expected, fresh, std::memory_order_acq_rel)) { fresh->start_workers(); return *fresh; }
    delete fresh;
    return *expected;
  }
  void start(ScanJob *jobs, int njobs, Batch &b) {
    std::lock_guard<std::mutex> lk(m_);
    for (int j = 0; j < njobs; ++j)
      for (size_t lo = 0; lo < jobs[j].n; lo += kSliceWords) b.tasks.push_back(Task{&jobs[j], lo});
    b.open.store((long)b.tasks.size(), std::memory_order_release);
    if (!b.tasks.empty()) { active_.push_back(&b); cv_.notify_all(); }
  }
  void wait(Batch &b) {
    for (;;) {   // help with THIS batch's slices until none is left to hand out ...
      Task t;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (!take(&b, t)) break;
      }
      run(t, b);
    }
    std::unique_lock<std::mutex> lk(m_);   // ... then sleep until the workers have handed in the slices they still hold
    done_.wait(lk, [&b] { return b.open.load(std::memory_order_acquire) <= 0; });
  }

 private:
  HostPool() = default;
  // (m_ held) the batch's next slice; a batch that has handed out its last one leaves active_ at once -- it is in there only
  // while it has slices (active_ holds one entry per context that is scanning right now)
  bool take(Batch *b, Task &t) {
    if (b->next >= b->tasks.size()) return false;
    t = b->tasks[b->next++];
    if (b->next >= b->tasks.size())
      for (size_t i = active_.size(); i-- > 0;)
        if (active_[i] == b) { active_.erase(active_.begin() + (long)i); break; }
    return true;
  }
  void start_workers() {
    unsigned n = std::thread::hardware_concurrency();
    n = n <= 2 ? 1 : (n / 2 > 16 ? 16 : n / 2);
    if (const char *env = getenv("RRTMG_HIP_HOST_THREADS")) { const int v = atoi(env); if (v >= 0 && v <= 64) n = (unsigned)v; }
    for (unsigned i = 0; i < n; ++i) std::thread([this] { loop(); }).detach();
  }
  void loop() {
    for (;;) {
      Task t;
      Batch *b;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !active_.empty(); });
        b = active_.back();
        if (!take(b, t)) continue;   // (cannot happen: a batch is in active_ only while it has slices)
      }
      run(t, *b);
    }
  }
  void run(const Task &t, Batch &b) {
    ScanJob &j = *t.job;
    const size_t hi = t.lo + kSliceWords < j.n ? t.lo + kSliceWords : j.n;
    for (size_t i = t.lo; i < hi && !j.differs.load(std::memory_order_relaxed); i += 4096) {
      const size_t e = i + 4096 < hi ? i + 4096 : hi;
      uint64_t acc = 0;
      for (size_t k = i; k < e; ++k) acc |= j.q[k] ^ j.first;
      if (acc) j.differs.store(true, std::memory_order_relaxed);
    }
    if (b.open.fetch_sub(1, std::memory_order_acq_rel) == 1) {   // the batch's last slice: wake its waiter (b is not touched again)
      std::lock_guard<std::mutex> lk(m_);
      done_.notify_all();
    }
  }
  static std::atomic<HostPool *> g_pool;
  std::mutex m_;
  std::condition_variable cv_, done_;
  std::vector<Batch *> active_;
};
std::atomic<HostPool *> HostPool::g_pool{nullptr};
}  // namespace

void HostInputs::add(const double **slot, const double *host, size_t n, const char *name, bool required, InPolicy policy, double mul, double div) {
  *slot = nullptr;
  if (!host) {
    if (required) { ctx_->fail(RRTMG_ERR_ARG, "required array '%s' is NULL", name); ok_ = false; }
    return;
  }
  if (memspace_ == 1) { *slot = host; return; }   // device pointers are used as they are
  Entry e{slot, host, n, name, policy, mul, div};
  entries_.push_back(e);
}

bool HostInputs::upload(const Entry &e) {
  const std::string key = std::string(prefix_) + e.name;
  double *dp = (double *)ctx_->buf(key, e.n * sizeof(double));
  if (!dp) return false;
  ctx_->bufs[key].uniform = false;
  if (hipMemcpyAsync(dp, e.host, e.n * sizeof(double), hipMemcpyHostToDevice, s_) != hipSuccess) { ctx_->fail(RRTMG_ERR_HIP, "H2D copy of '%s' failed", e.name); return false; }
  if (e.mul != 0.0) launch_scale(s_, dp, e.n, e.mul, e.div);
  *e.slot = dp;
  return true;
}

bool HostInputs::fill(const Entry &e, double host_value) {
  // the value numpy would have formed on the host: one rounding per operation
  volatile double v = host_value;
  if (e.mul != 0.0) { v = v * e.mul; if (e.div != 0.0) v = v / e.div; }
  const double value = v;
  const std::string key = std::string(prefix_) + e.name;
  double *dp = (double *)ctx_->buf(key, e.n * sizeof(double));
  if (!dp) return false;
  DevBuf &b = ctx_->bufs[key];
  if (!(b.uniform && b.uni_n == e.n && memcmp(&b.uni_value, &value, sizeof value) == 0)) {
    launch_fill(s_, dp, e.n, value);
    b.uniform = true; b.uni_value = value; b.uni_n = e.n;
  }
  *e.slot = dp;
  return true;
}

bool HostInputs::finish() {
  if (!ok_) return false;
  std::unique_ptr<ScanJob[]> jobs(new ScanJob[entries_.size() + 1]);
  int njobs = 0;
  for (Entry &e : entries_) {
    if (e.n < kScanMin) continue;
    const uint64_t *q = (const uint64_t *)e.host;
    const uint64_t w0 = q[0];
    uint64_t acc = 0;
    for (size_t i = 1; i < 2048; ++i) acc |= q[i] ^ w0;                       // the head ...
    for (size_t k = 1; k <= 16; ++k) acc |= q[(e.n - 1) / 16 * k] ^ w0;        // ... and sixteen places further on
    if (acc) continue;                                                         // certainly not uniform: goes up at once
    e.job = njobs;
    jobs[njobs].q = q; jobs[njobs].n = e.n; jobs[njobs].first = w0;
    ++njobs;
  }
  HostPool::Batch batch;
  if (njobs) HostPool::get().start(jobs.get(), njobs, batch);
  for (const Entry &e : entries_)
    if (e.job < 0 && !upload(e)) ok_ = false;
  if (njobs) HostPool::get().wait(batch);
  for (const Entry &e : entries_) {
    if (e.job < 0) continue;
    const ScanJob &j = jobs[e.job];
    if (j.differs.load()) { if (!upload(e)) ok_ = false; continue; }
    if (e.policy == InPolicy::ZeroAbsent && j.first == 0) { *e.slot = nullptr; continue; }   // all +0.0: the "array absent" path adds the same
    double v;
    memcpy(&v, &j.first, sizeof v);
    if (!fill(e, v)) ok_ = false;
  }
  return ok_;
}

std::string default_blob_path(const char *which) {
  if (const char *env = getenv(strcmp(which, "sw") == 0 ? "RRTMG_HIP_SW_DATA" : "RRTMG_HIP_LW_DATA")) return env;
  Dl_info info;
  std::string dir = ".";
  if (dladdr((void *)&default_blob_path, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    size_t k = p.rfind('/');
    dir = (k == std::string::npos) ? "." : p.substr(0, k);
  }
  return dir + "/../data/rrtmg_" + which + "_data.bin";
}

}  // namespace rrtmg

using namespace rrtmg;

// The caller's struct is copied into one of the library's own.  struct_size must be sizeof(Args) of THIS header; anything else
// -- 0 included: that slot was `reserved0` in two earlier layouts, one that ended with the outputs and one that already carried
// the unit factors, and the library cannot tell which of them a zero comes from -- was built against another header and is
// refused, so that no caller ever gets RRTMG_OK with fields dropped or read past its struct.  Unit factors: host arrays only.
template <class Args, class Impl>
static int checked_call(rrtmg_ctx *ctx, const Args *a, const char *what, Impl impl) {
  if (!ctx) return RRTMG_ERR_ARG;
  if (!a) return ctx->fail(RRTMG_ERR_ARG, "%s: NULL argument struct", what);
  if ((size_t)a->struct_size != sizeof(Args))
    return ctx->fail(RRTMG_ERR_ARG, "%s: struct_size %d is not sizeof(%s_args) = %zu of this library (ABI version %d): set it to sizeof of the struct and rebuild the caller against include/rrtmg_hip.h",
                     what, (int)a->struct_size, what, sizeof(Args), RRTMG_HIP_ABI_VERSION);
  Args own = *a;
  if (own.memspace == 1 && (own.pressure_scale != 0.0 || own.water_path_scale != 0.0 || own.h2o_mul != 0.0 || own.h2o_div != 0.0))
    return ctx->fail(RRTMG_ERR_ARG, "%s: unit factors (pressure_scale, water_path_scale, h2o_mul, h2o_div) apply to host arrays only (memspace 0)", what);
  return impl(ctx, &own);
}

extern "C" {

#ifndef RRTMG_SRC_HASH
#define RRTMG_SRC_HASH "unknown"
#endif
// "... src:<hash>": sha256 over climt_amd/csrc + include/rrtmg_hip.h at build time (climt_amd/build.py::source_hash)
const char *rrtmg_hip_version(void) { return "rrtmg-hip 0.2 (gfx950) src:" RRTMG_SRC_HASH; }

int rrtmg_hip_create(rrtmg_ctx **out, int device_ordinal) {
  if (!out) return RRTMG_ERR_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  rrtmg_ctx *c = new rrtmg_ctx();
  c->device = device_ordinal;
  if (const char *env = getenv("RRTMG_HIP_CHUNK_TILES")) { const int v = atoi(env); if (v > 0) { c->chunk_tiles = v; c->chunk_auto = false; } }
  if (const char *env = getenv("RRTMG_HIP_SORT_COLUMNS")) c->sort_columns = atoi(env) != 0;
  if (const char *env = getenv("RRTMG_HIP_MAX_SCRATCH_BYTES")) { const long long v = atoll(env); if (v > 0) c->max_scratch_bytes = (size_t)v; }
  *out = c;
  if (e != hipSuccess || n <= 0)
    return c->fail(RRTMG_ERR_HIP, "no HIP device available (%s): librrtmg_hip has no CPU path", hipGetErrorString(e));
  if (device_ordinal < 0 || device_ordinal >= n) return c->fail(RRTMG_ERR_ARG, "device ordinal %d out of range (%d devices)", device_ordinal, n);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) c->device_mem = prop.totalGlobalMem;
  }
  return ctx_prepare_device(c);
}

void rrtmg_hip_destroy(rrtmg_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (auto &kv : ctx->bufs)
    if (kv.second.p) (void)hipFree(kv.second.p);
  if (ctx->sw_tab_dev) (void)hipFree(ctx->sw_tab_dev);
  if (ctx->lw_tab_dev) (void)hipFree(ctx->lw_tab_dev);
  if (ctx->err_dev) (void)hipFree(ctx->err_dev);
  if (ctx->ncloudy_dev) (void)hipFree(ctx->ncloudy_dev);
  if (ctx->hint) (void)hipHostFree((void *)ctx->hint);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  for (int w = 0; w < 4; ++w)
    for (hipEvent_t e : ctx->ev[w])
      if (e) (void)hipEventDestroy(e);
  for (int w = 0; w < 2; ++w)
    for (int k = 0; k < 2; ++k)
      if (ctx->kiss_ev[w][k]) (void)hipEventDestroy(ctx->kiss_ev[w][k]);
  for (int w = 0; w < 2; ++w)
    if (ctx->sync_ev[w]) (void)hipEventDestroy(ctx->sync_ev[w]);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream_lw) (void)hipStreamDestroy(ctx->stream_lw);
  rrtmg::free_sw_desc(ctx);
  rrtmg::free_lw_desc(ctx);
  delete ctx;
}

const char *rrtmg_hip_last_error(const rrtmg_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
void *rrtmg_hip_stream(rrtmg_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int rrtmg_hip_kernel_ms(rrtmg_ctx *ctx, int which, double *ms) {
  if (!ctx || which < 0 || which > 3 || !ms || ctx->ev_chunks[which] <= 0) return RRTMG_ERR_ARG;
  double sum = 0.0;
  for (int c = 0; c < ctx->ev_chunks[which]; ++c) {
    float f = 0.f;
    RRTMG_HIP_CHECK(ctx, hipEventElapsedTime(&f, ctx->ev[which][2 * c], ctx->ev[which][2 * c + 1]));
    sum += (double)f;
  }
  *ms = sum;
  return RRTMG_OK;
}
int rrtmg_hip_kernel_launches(rrtmg_ctx *ctx, int which) { return (!ctx || which < 0 || which > 3) ? -1 : ctx->ev_chunks[which]; }
int rrtmg_hip_synchronize(rrtmg_ctx *ctx) {
  if (!ctx) return RRTMG_ERR_ARG;
  if (ctx->stream) RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->stream_lw) RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream_lw));
  // deferred calls: collect the device-side error flags now (the former Fortran `stop` conditions).  Deferred calls do
  // not clear their flag (atomicMax accumulates over every call enqueued since the last collection); it is cleared here.
  if (ctx->pending[0] || ctx->pending[1]) {
    int herr[2] = {0, 0};
    const int zero[2] = {0, 0};
    RRTMG_HIP_CHECK(ctx, hipMemcpy(herr, ctx->err_dev, sizeof herr, hipMemcpyDeviceToHost));
    RRTMG_HIP_CHECK(ctx, hipMemcpy(ctx->err_dev, zero, sizeof zero, hipMemcpyHostToDevice));
    const bool p0 = ctx->pending[0], p1 = ctx->pending[1];
    ctx->pending[0] = ctx->pending[1] = false;
    const int e0 = p0 ? herr[0] : 0, e1 = p1 ? herr[1] : 0;
    if (e0 && e1) return ctx->fail(e0, "shortwave: %s; longwave (status %d): %s", status_message(e0), e1, status_message(e1));
    if (e0) return ctx->fail(e0, "shortwave: %s", status_message(e0));
    if (e1) return ctx->fail(e1, "longwave: %s", status_message(e1));
  }
  return RRTMG_OK;
}
int rrtmg_hip_stream_wait(rrtmg_ctx *ctx, void *other_stream) {
  if (!ctx || !ctx->stream || !ctx->stream_lw) return RRTMG_ERR_ARG;
  hipStream_t o = (hipStream_t)other_stream;
  RRTMG_HIP_CHECK(ctx, hipEventRecord(ctx->sync_ev[0], ctx->stream));
  RRTMG_HIP_CHECK(ctx, hipEventRecord(ctx->sync_ev[1], ctx->stream_lw));
  RRTMG_HIP_CHECK(ctx, hipStreamWaitEvent(o, ctx->sync_ev[0], 0));
  RRTMG_HIP_CHECK(ctx, hipStreamWaitEvent(o, ctx->sync_ev[1], 0));
  return RRTMG_OK;
}
int rrtmg_hip_set_deferred(rrtmg_ctx *ctx, int on) {
  if (!ctx) return RRTMG_ERR_ARG;
  int rc = rrtmg_hip_synchronize(ctx);
  ctx->deferred = on != 0;
  return rc;
}

int rrtmg_hip_set_column_sort(rrtmg_ctx *ctx, int on) {
  if (!ctx) return RRTMG_ERR_ARG;
  ctx->sort_columns = on != 0;
  return RRTMG_OK;
}

int rrtmg_hip_set_constants(rrtmg_ctx *ctx, double pi, double grav, double planck, double boltz, double clight,
                            double avogad, double alosmt, double gascon, double sbcnst, double secdy) {
  if (!ctx) return RRTMG_ERR_ARG;
  Constants &k = ctx->k;
  k.pi = pi; k.grav = grav; k.planck = planck; k.boltz = boltz; k.clight = clight; k.avogad = avogad;
  k.alosmt = alosmt; k.gascon = gascon; k.sbcnst = sbcnst; k.secdy = secdy;
  k.radcn1 = 2. * planck * clight * clight * 1.e-07;
  k.radcn2 = planck * clight / boltz;
  ctx->have_constants = true;
  return RRTMG_OK;
}

int rrtmg_hip_sw_init(rrtmg_ctx *ctx, double cpdair, const char *blob_path) { return ctx ? sw_init_impl(ctx, cpdair, blob_path) : RRTMG_ERR_ARG; }
int rrtmg_hip_lw_init(rrtmg_ctx *ctx, double cpdair, const char *blob_path) { return ctx ? lw_init_impl(ctx, cpdair, blob_path) : RRTMG_ERR_ARG; }
int rrtmg_hip_lw_tables_synthetic(const rrtmg_ctx *ctx) { return ctx && ctx->lw_ts.synthetic ? 1 : 0; }

long rrtmg_hip_get_table(rrtmg_ctx *ctx, const char *name, double *out, long capacity) {
  if (!ctx || !name) return -1;
  const TableSet &ts = (strncmp(name, "sw/", 3) == 0) ? ctx->sw_ts : ctx->lw_ts;
  auto it = ts.reg.find(name);
  if (it == ts.reg.end()) return -1;
  const long n = it->second.n;
  if (out) {
    if (capacity < n) return -2;
    memcpy(out, ts.flat.data() + it->second.off, (size_t)n * sizeof(double));
  }
  return n;
}

int rrtmg_hip_sw_fluxes(rrtmg_ctx *ctx, const rrtmg_sw_args *a) { return checked_call(ctx, a, "rrtmg_sw", sw_fluxes_impl); }
int rrtmg_hip_lw_fluxes(rrtmg_ctx *ctx, const rrtmg_lw_args *a) { return checked_call(ctx, a, "rrtmg_lw", lw_fluxes_impl); }
int rrtmg_hip_abi_version(void) { return RRTMG_HIP_ABI_VERSION; }

int rrtmg_hip_mcica_mask(rrtmg_ctx *ctx, int which, int ncol, int nlay, int icld, int permuteseed, int irng,
                         const double *play, const double *cldfrac, double *cldfmcl) {
  return ctx ? mcica_mask_impl(ctx, which, ncol, nlay, icld, permuteseed, irng, play, cldfrac, cldfmcl) : RRTMG_ERR_ARG;
}

}  // extern "C"
