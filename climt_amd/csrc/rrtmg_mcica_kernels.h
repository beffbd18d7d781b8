// rrtmg_mcica_kernels.h -- McICA sub-column mask kernels shared by the SW and LW translation units
// (static: each TU carries its own device copy; no relocatable device code needed).
#pragma once
#include "rrtmg_sw_device.h"   // kiss_mask_column

namespace rrtmg {

static __global__ void __launch_bounds__(64) kiss_mask_kernel(int ncol, int nlay, int nsub, int icld, int seed, const double *play,
                                                              const double *cldfr, uint64_t *mask, int nw, int *err) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col < ncol) kiss_mask_column(ncol, nlay, nsub, icld, seed, play, cldfr, mask, nw, err, col);
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
