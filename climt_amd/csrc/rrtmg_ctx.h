// rrtmg_ctx.h -- private context of librrtmg_hip.so (one per GPU / component instance).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rrtmg_hip.h"
#include "rrtmg_common.h"
#include "rrtmg_host_inputs.h"
#include "rrtmg_tables.h"

namespace rrtmg {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  // an INPUT buffer that was last filled with one value (rrtmg_host_inputs.h): the next call that finds the same uniform array
  // on the host has nothing to do.  Cleared by every upload into the buffer and when it is re-allocated.
  bool uniform = false;
  double uni_value = 0.0;
  size_t uni_n = 0;
};

}  // namespace rrtmg

struct rrtmg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;      // shortwave (and everything else)
  hipStream_t stream_lw = nullptr;   // longwave in deferred mode, so SW and LW launches overlap on the GPU
  bool deferred = false;             // rrtmg_hip_set_deferred: device-resident calls return after enqueueing
  bool pending[2] = {false, false};  // [sw|lw] enqueued, status not yet collected
  // the solve + flux stages run over chunks of at most this many 64-column tiles, so that the sweep-state scratch
  // and the partial-flux planes stay bounded (~0.23 MB per column and spectrum at 60 layers); env RRTMG_HIP_CHUNK_TILES.
  // 128 tiles = 8192 columns: one full round of the clear-sky shortwave kernel (256 workgroups of 16 tiles x 32 items);
  // measured 1-3 % faster than 512 on grids of 16 384 ... 131 072 columns (64: the launches no longer fill the GPU)
  int chunk_tiles = 128;
  bool chunk_auto = true;            // false: RRTMG_HIP_CHUNK_TILES was given
  size_t device_mem = 0;             // total memory of the device (bounds the work space of the large chunks of mixed grids)
  // Tiles per chunk for a grid that had BOTH kinds of tiles in the previous call (hint_cloudy of ntile cloudy, each kind at
  // least a sixteenth): every launch of a solve variant costs whole rounds of workgroups that hold a CU for ~0.8 ms, so a
  // chunk whose 128 tiles split 96 : 32 between the variants pays two rounds for one and a half rounds of work -- large chunks
  // amortise the rounding (131 072 McICA columns, a quarter of the tiles cloud-free: 55.6 ms at 128 tiles per chunk, 42.0 at
  // 2048; 1 036 800 x 100: 749 -> 572 ms), at the price of work space (bytes_per_tile x tiles).  Grids of one kind keep the
  // small chunks, whose rows stay cached (+2..8 % there).
  // Work space: at most max_scratch_bytes per spectrum (RRTMG_HIP_MAX_SCRATCH_BYTES; default an eighth of the device's memory)
  // and, beyond what the spectrum's scratch buffer already holds, no more than a third of the memory that is free when the
  // plan is made (a model that has filled the device keeps the small chunks instead of failing).  The plan is made ONCE per
  // (spectrum, tiles, layers, kind of grid) and kept: no hipMemGetInfo per call, and a hint that flips back and forth re-uses
  // the two plans it has; ctx->buf never shrinks, so the footprint is the largest plan's (INTEGRATION.md states it).
  size_t max_scratch_bytes = 0;      // 0: device_mem / 8
  struct ChunkPlan { int ntile = -1, nlay = -1, base = -1, chunk = 0; };
  ChunkPlan plans[2][2];             // [sw|lw][grid of one kind | mixed]
  // -> tiles per chunk.  chunk_tiles = the chunk a grid of one kind gets (128, or 64 for a deep cloudy grid)
  int plan_chunks(int which, int chunk_tiles, int ntile, int nlay, int hint_cloudy, size_t bytes_per_tile, const char *scratch) {
    bool mixed = false;
    if (chunk_auto && hint_cloudy >= 0 && ntile > chunk_tiles) {
      const int fewer = hint_cloudy < ntile - hint_cloudy ? hint_cloudy : ntile - hint_cloudy;
      mixed = 16 * fewer >= ntile;
    }
    ChunkPlan &p = plans[which][mixed ? 1 : 0];
    if (p.ntile == ntile && p.nlay == nlay && p.base == chunk_tiles) return p.chunk;
    int chunk = chunk_tiles;
    if (mixed) {
      size_t budget = max_scratch_bytes ? max_scratch_bytes : device_mem / 8, free_b = 0, total_b = 0;
      const auto it = bufs.find(scratch);
      const size_t have = it == bufs.end() ? 0 : it->second.cap;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
      const size_t room = have > free_b / 3 ? have : free_b / 3;
      if (room < budget) budget = room;
      long cap = (long)(budget / (bytes_per_tile ? bytes_per_tile : 1)) / 128 * 128;
      if (cap > 2048) cap = 2048;
      if (cap > chunk_tiles) chunk = (int)cap;
    }
    p.ntile = ntile; p.nlay = nlay; p.base = chunk_tiles; p.chunk = chunk;
    return chunk;
  }
  // What the PREVIOUS call of a spectrum [sw|lw] found -- tiles, layers, tiles with a cloud -- for sizing and ordering the
  // launches of the next one.  The count is left in page-locked memory by the call's last kernel (which also clears the counter) and read
  // WITHOUT waiting when the next call is enqueued (stale, or missing, in a loop that runs ahead of the GPU): a hint.  Every
  // value gives the same results; what it moves is which of the two solve variants is enqueued first (the one expected to
  // find no tile of its kind: its workgroups are then placed while the stream has the GPU, not after the other spectrum has
  // taken the CUs) and the chunk depth of a cloudy grid with more than 80 layers.
  struct CallHint { int ntile, nlay, ncloudy; };
  volatile CallHint *hint = nullptr;   // [2], page-locked
  int *ncloudy_dev = nullptr;          // [2]
  // KISS jump-ahead operators [sw|lw]: host copy, the key they were built for, the device buffer they were uploaded to
  std::vector<uint32_t> kiss_host[2][2];   // two staging copies per spectrum: a rebuild never waits for the previous upload
  hipEvent_t kiss_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // recorded after the upload from kiss_host[w][k]
  int kiss_slot[2] = {0, 0};
  hipEvent_t sync_ev[2] = {nullptr, nullptr};   // rrtmg_hip_stream_wait: "everything enqueued so far" on stream / stream_lw
  int kiss_key[2][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}};
  const void *kiss_dev[2] = {nullptr, nullptr};
  // Mersenne-twister jump polynomials [sw|lw] on the device: the (first draw, stride, runs, piece, pieces) they were built for
  uint64_t mt_key[2][5] = {{~0ull, ~0ull, ~0ull, ~0ull, ~0ull}, {~0ull, ~0ull, ~0ull, ~0ull, ~0ull}};
  const void *mt_dev[2] = {nullptr, nullptr};
  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) holds per DEVICE (a kernel's code object is loaded once per device): a
  // context belongs to one device, so every context asks once -- after ctx_prepare_device has made its device current --
  // and keeps the answer.  [0] lw_prep_fused_kernel, [1] mt_jump_kernel, [2] mt_mask_kernel; -1 = not asked yet
  int big_lds[3] = {-1, -1, -1};
  bool allow_dynamic_lds(int slot, const void *kernel, int bytes) {
    if (big_lds[slot] < 0) big_lds[slot] = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 1 : 0;
    return big_lds[slot] > 0;
  }
  // rrtmg_hip_set_column_sort (rrtmg_sort.h): device-resident calls with clouds run on an internal copy of their inputs, cloud-free
  // columns first; `sorting` = this is the inner call
  bool sort_columns = false, sorting = false;
  std::string err;
  int status = 0;
  rrtmg::Constants k{};
  bool have_constants = false;
  // tables
  rrtmg::TableSet sw_ts, lw_ts;
  bool sw_ready = false, lw_ready = false;
  double *sw_tab_dev = nullptr, *lw_tab_dev = nullptr;
  void *sw_desc = nullptr, *lw_desc = nullptr;   // SwTab / LwTab (host copies, owned)
  // grow-only device work buffers, by name
  std::map<std::string, rrtmg::DevBuf> bufs;
  // grow-only PINNED host staging for the outputs of host-pointer calls (see copy_out)
  void *pinned = nullptr;
  size_t pinned_cap = 0;
  int *err_dev = nullptr;
  // HIP events around the solve launches of EVERY column chunk of the last call: [0] sw clear-sky kernel, [1] lw clear-sky
  // variant, [2] sw cloudy kernel, [3] lw cloudy variant; per chunk a (start, stop) pair, created on demand.
  // rrtmg_hip_kernel_ms adds the chunks' durations up: the time that kernel took for ALL the call's columns.
  std::vector<hipEvent_t> ev[4];
  int ev_chunks[4] = {0, 0, 0, 0};   // chunks bracketed by the last call (0: that kernel was not launched)
  hipEvent_t chunk_event(int which, int chunk, int side) {
    std::vector<hipEvent_t> &v = ev[which];
    while ((int)v.size() < 2 * (chunk + 1)) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) e = nullptr;
      v.push_back(e);
    }
    return v[2 * chunk + side];
  }

  int fail(int code, const char *fmt, ...) {
    char tmp[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    err = tmp;
    status = code;
    return code;
  }
  // returns nullptr on allocation failure (err set)
  void *buf(const std::string &name, size_t bytes) {
    rrtmg::DevBuf &b = bufs[name];
    if (b.cap >= bytes && b.p) return b.p;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    b.uniform = false;
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
      fail(RRTMG_ERR_HIP, "hipMalloc(%zu bytes) for '%s' failed: %s", want, name.c_str(), hipGetErrorString(e));
      b.p = nullptr;
      return nullptr;
    }
    b.cap = want;
    return b.p;
  }
};

namespace rrtmg {
// Outputs of a host-pointer (memspace = 0) call.  The caller's arrays are normally FRESH allocations (climt hands every call
// new zero-filled numpy arrays, lw/component.py:386-399): a device-to-host copy straight into never-touched pageable memory
// runs at < 1 GB/s (the runtime pins it page by page; measured 27 ms for 24 MB), while a copy into pinned staging runs at
// 56 GB/s and a few host threads first-touch and fill the caller's pages at > 10 GB/s (tools/micro/host_path_timing.py).
// interface values of a mid-level quantity (climt/_core/util.py:89-142) on stream s, device pointers
void launch_interface_values(hipStream_t s, int ncol, int nlay, const double *mid, const double *surf, const double *pmid, const double *pint, double *out);
struct OutCopy { double *host; const double *dev; size_t n; };
int copy_out(rrtmg_ctx *ctx, hipStream_t s, const OutCopy *o, int count, int *herr_dev, int *herr_host);
}  // namespace rrtmg

#define RRTMG_HIP_CHECK(ctx, call)                                                                     \
  do {                                                                                                 \
    hipError_t e__ = (call);                                                                           \
    if (e__ != hipSuccess)                                                                             \
      return (ctx)->fail(RRTMG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

namespace rrtmg {
const char *status_message(int code);
int ctx_prepare_device(rrtmg_ctx *ctx);   // hipSetDevice + lazy stream / error-flag creation
std::string default_blob_path(const char *which);
void free_sw_desc(rrtmg_ctx *ctx);
void free_lw_desc(rrtmg_ctx *ctx);
int sw_fluxes_impl(rrtmg_ctx *ctx, const rrtmg_sw_args *a);
int lw_fluxes_impl(rrtmg_ctx *ctx, const rrtmg_lw_args *a);
int sw_init_impl(rrtmg_ctx *ctx, double cpdair, const char *blob);
int lw_init_impl(rrtmg_ctx *ctx, double cpdair, const char *blob);
int mcica_mask_impl(rrtmg_ctx *ctx, int which, int ncol, int nlay, int icld, int permuteseed, int irng,
                    const double *play, const double *cldfrac, double *cldfmcl);
// Mersenne-twister sub-column masks on the device (rrtmg_mt_device.hip): cldfr, mask device pointers, everything on stream s
int mt_mask_device(rrtmg_ctx *ctx, int which, int ncol, int nlay, int nsub, int icld, int seed, const double *cldfr, uint64_t *mask, int nw,
                   int col0, int ncol_total, hipStream_t s);
}  // namespace rrtmg
