// rrtmg_mcica_kernels.h -- McICA sub-column mask kernels shared by the SW and LW translation units
// (static: each TU carries its own device copy; no relocatable device code needed).
#pragma once
#include "rrtmg_ctx.h"
#include "rrtmg_kiss_host.h"
#include "rrtmg_sw_device.h"   // kiss_mask_jump

namespace rrtmg {

// The tiles of one column chunk by solve variant, compacted IN TILE ORDER: list[v * cap + i] = the i-th tile (index within the
// chunk) whose flag is v (0 cloud-free, 1 cloudy), cnt[v] = how many.  One wavefront (ballots + population counts of the lower
// lanes); launched behind the chunk's preparation kernel, which writes the flags.
// (64-bit ballots, a 64-thread block: written for the 64-wide wavefronts of gfx950 and nothing else)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "tile_lists_kernel and the solve kernels assume 64-wide wavefronts (the gfx9 family: gfx950)"
#endif
static __global__ void __launch_bounds__(64) tile_lists_kernel(const int32_t *tile_cld, int ntile, int32_t *list, int32_t *cnt, int cap) {
  const int lane = threadIdx.x;
  if (ntile > cap) ntile = cap;   // (the launch sites size the lists for the chunk: never more tiles than list entries)
  int n0 = 0, n1 = 0;
  for (int t0 = 0; t0 < ntile; t0 += 64) {
    const int t = t0 + lane;
    const bool in = t < ntile;
    const bool cld = in && tile_cld[t] != 0;
    const unsigned long long m1 = __ballot(cld), m0 = __ballot(in && !cld);
    const unsigned long long lower = (1ull << lane) - 1ull;
    if (in) {
      if (cld) list[cap + n1 + __popcll(m1 & lower)] = t;
      else list[n0 + __popcll(m0 & lower)] = t;
    }
    n0 += __popcll(m0); n1 += __popcll(m1);
  }
  if (lane == 0) { cnt[0] = n0; cnt[1] = n1; }
}

// kissvec sub-columns, one thread per (column, sub-column): grid (nsub, tiles), the sub-column index FASTEST -- the nsub
// blocks of a tile run back to back and re-read the tile's 64 x nlay cloud fractions from L2 (with tiles fastest the whole
// cldfr array streamed through once per sub-column: 8 GB per launch at 131072 columns, where it no longer fits the L2s).
// Every thread jumps the column's generator to its sub-column's first draw (kiss_jump) -- nsub times the parallelism of the
// column-sequential reference order, same bits.
static __global__ void __launch_bounds__(64) kiss_mask_kernel(int ncol, int nlay, int icld, const double *play, const double *cldfr,
                                                              uint64_t *mask, int nw, int *err, const uint32_t *jumps) {
  const int col = blockIdx.y * 64 + threadIdx.x;
  if (col < ncol) kiss_mask_jump(ncol, nlay, icld, play, cldfr, mask, nw, err, jumps, col, blockIdx.x);
}

// Jump operators of (nsub, nlay, icld, seed) on the device; rebuilt and uploaded only when the key changes.
// which: 0 shortwave, 1 longwave (separate buffers: the two may be in flight on different streams).
static const uint32_t *kiss_jumps_device(rrtmg_ctx *ctx, int which, int nsub, int nlay, int icld, int seed, hipStream_t s) {
  const int key[4] = {nsub, nlay, icld, seed};
  uint32_t *dev = (uint32_t *)ctx->buf(which == 0 ? "sw.w.kissjump" : "lw.w.kissjump", (size_t)nsub * kKissJumpWords * sizeof(uint32_t));
  if (!dev) return nullptr;
  bool same = ctx->kiss_dev[which] == dev;
  for (int i = 0; i < 4; ++i) same = same && ctx->kiss_key[which][i] == key[i];
  if (!same) {
    // Two host staging copies, used alternately: the one rebuilt now was the source of the upload before the previous
    // one, which has long completed (its event is waited for, without stalling the stream), so a component that redraws
    // its seed on every call does not serialise the SW / LW streams.
    const int slot = ctx->kiss_slot[which] ^= 1;
    std::vector<uint32_t> &host = ctx->kiss_host[which][slot];
    (void)hipEventSynchronize(ctx->kiss_ev[which][slot]);
    kiss_build_jumps(nsub, nlay, icld, seed, host);
    if (hipMemcpyAsync(dev, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s) != hipSuccess) {
      ctx->fail(RRTMG_ERR_HIP, "upload of the KISS jump table failed");
      return nullptr;
    }
    (void)hipEventRecord(ctx->kiss_ev[which][slot], s);
    for (int i = 0; i < 4; ++i) ctx->kiss_key[which][i] = key[i];
    ctx->kiss_dev[which] = dev;
  }
  return dev;
}

// externally supplied cldfmcl [lay][col][nsub] (0/1 doubles) -> bit mask
static __global__ void __launch_bounds__(64) mask_from_cldfmcl_kernel(int ncol, int nlay, int nsub, const double *cldfmcl, uint64_t *mask, int nw) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  const int g = blockIdx.y;
  if (col >= ncol) return;
  for (int w = 0; w < nw; ++w) {
    uint64_t m = 0;
    for (int l = w * 64; l < nlay && l < (w + 1) * 64; ++l)
      if (cldfmcl[((long)l * ncol + col) * nsub + g] > 1.e-12) m |= 1ull << (l & 63);
    mask[((long)g * nw + w) * ncol + col] = m;
  }
}

// bit mask -> cldfmcl doubles (for the stand-alone sub-column generator entry point)
static __global__ void __launch_bounds__(64) cldfmcl_from_mask_kernel(int ncol, int nlay, int nsub, const uint64_t *mask, int nw, double *cldfmcl) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  const int g = blockIdx.y;
  if (col >= ncol) return;
  for (int l = 0; l < nlay; ++l) {
    const uint64_t m = mask[((long)g * nw + (l >> 6)) * ncol + col];
    cldfmcl[((long)l * ncol + col) * nsub + g] = ((m >> (l & 63)) & 1ull) ? 1.0 : 0.0;
  }
}

}  // namespace rrtmg
