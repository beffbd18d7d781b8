// rrtmg_abi.hip -- extern "C" surface of librrtmg_hip.so (declared in include/rrtmg_hip.h).
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include <atomic>

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
    case RRTMG_ERR_CLOUD_OPTICS: return "CLOUD OPTICAL PROPERTY OUT OF RANGE";
    case RRTMG_ERR_KISS_PRESSURE: return "MCICA_SUBCOL: KISSVEC SEED GENERATOR REQUIRES PMID FROM BOTTOM FOUR LAYERS.";
    case RRTMG_ERR_ICLD: return "MCICA_SUBCOL: INVALID ICLD";
    case RRTMG_ERR_UNSUPPORTED: return "option not supported by this build";
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

// true when every one of the n doubles at p is +0.0 (all bits clear).  The head is checked first (an array with data in it
// is recognised in microseconds); an array that passes that is scanned in slices on a few host threads, each giving up as
// soon as any of them has found a set bit.
bool host_all_zero(const double *p, size_t n) {
  const uint64_t *q = (const uint64_t *)p;
  const size_t head = n < 4096 ? n : 4096;
  for (size_t i = 0; i < head; ++i) if (q[i]) return false;
  if (n == head) return true;
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt == 0 ? 1 : (nt > 8 ? 8 : nt);
  if (n < ((size_t)1 << 19)) nt = 1;
  std::atomic<bool> found(false);
  auto work = [&](unsigned t) {
    const size_t per = (n + nt - 1) / nt, lo = (size_t)t * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo; i < hi && !found.load(std::memory_order_relaxed); i += 4096) {
      const size_t e = i + 4096 < hi ? i + 4096 : hi;
      uint64_t acc = 0;
      for (size_t j = i; j < e; ++j) acc |= q[j];
      if (acc) found.store(true, std::memory_order_relaxed);
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
  work(0);
  for (auto &x : th) x.join();
  return !found.load();
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

extern "C" {

const char *rrtmg_hip_version(void) { return "rrtmg-hip 0.1 (gfx950)"; }

int rrtmg_hip_create(rrtmg_ctx **out, int device_ordinal) {
  if (!out) return RRTMG_ERR_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  rrtmg_ctx *c = new rrtmg_ctx();
  c->device = device_ordinal;
  if (const char *env = getenv("RRTMG_HIP_CHUNK_TILES")) { const int v = atoi(env); if (v > 0) { c->chunk_tiles = v; c->chunk_auto = false; } }
  *out = c;
  if (e != hipSuccess || n <= 0)
    return c->fail(RRTMG_ERR_HIP, "no HIP device available (%s): librrtmg_hip has no CPU path", hipGetErrorString(e));
  if (device_ordinal < 0 || device_ordinal >= n) return c->fail(RRTMG_ERR_ARG, "device ordinal %d out of range (%d devices)", device_ordinal, n);
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

int rrtmg_hip_sw_fluxes(rrtmg_ctx *ctx, const rrtmg_sw_args *a) { return ctx ? sw_fluxes_impl(ctx, a) : RRTMG_ERR_ARG; }
int rrtmg_hip_lw_fluxes(rrtmg_ctx *ctx, const rrtmg_lw_args *a) { return ctx ? lw_fluxes_impl(ctx, a) : RRTMG_ERR_ARG; }

int rrtmg_hip_mcica_mask(rrtmg_ctx *ctx, int which, int ncol, int nlay, int icld, int permuteseed, int irng,
                         const double *play, const double *cldfrac, double *cldfmcl) {
  return ctx ? mcica_mask_impl(ctx, which, ncol, nlay, icld, permuteseed, irng, play, cldfrac, cldfmcl) : RRTMG_ERR_ARG;
}

}  // extern "C"
