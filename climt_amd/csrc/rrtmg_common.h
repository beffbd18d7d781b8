// rrtmg_common.h -- shared host/device definitions for librrtmg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#define RRTMG_HD __host__ __device__ __forceinline__
#define RRTMG_WAVE 64

namespace rrtmg {

// physical constants as handed over by the host (rrtmg_sw_set_constants, rrtmg_sw_c_binder.f90:19-46)
struct Constants {
  double pi, grav, planck, boltz, clight, avogad, alosmt, gascon, sbcnst, secdy;
  double radcn1, radcn2;
};

// Report a former Fortran `stop` condition.  The first error code wins (max), checked on the host
// after the launch sequence.
RRTMG_HD void report_error(int *flag, int code) {
#if defined(__HIP_DEVICE_COMPILE__)
  atomicMax(flag, code);
#else
  if (*flag < code) *flag = code;
#endif
}

// Transmittance lookup index: itind = tblint*x/(bpade+x) + 0.5 truncated (rrtmg_sw_reftra.f90:199-203,
// rrtmg_lw_rtrn.f90:426-430).  Index arithmetic stays in fp64 so table entries do not flip.
// bit l of a 4-word (<= 256 layers) cloud mask held in registers: selects instead of dynamic indexing, which
// would push the array into private (scratch) memory
RRTMG_HD bool mask_bit(const uint64_t *w, int l) {
  const uint64_t v = (l < 64) ? w[0] : (l < 128) ? w[1] : (l < 192) ? w[2] : w[3];
  return (v >> (l & 63)) & 1ull;
}

constexpr double kTblInt = 10000.0;
constexpr double kBpade = 1.0 / 0.278;

}  // namespace rrtmg
