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
// Quick fp64 division for the flux arithmetic of the hot loops: v_rcp_f64 + two Newton-Raphson steps
// (<= ~1 ulp) instead of the 11-instruction IEEE sequence (div_scale x2, rcp, 6 fma, div_fmas, div_fixup) --
// divisions were ~60 % of the VALU instructions of the solve kernels.  Operands here are O(1e-20..1e20) and
// never zero/inf/denormal.  NOT used where an integer is derived from the quotient (table indices, specparm ->
// js): those keep the correctly rounded `/`.  On the host (tests/emu) it is the plain division.
RRTMG_HD double qdiv(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RRTMG_EXACT_DIV)
  double r = __builtin_amdgcn_rcp(b);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  return a * r;
#else
  return a / b;
#endif
}
RRTMG_HD double qrcp(double b) { return qdiv(1.0, b); }

// bit l of a 4-word (<= 256 layers) cloud mask held in registers: selects instead of dynamic indexing, which
// would push the array into private (scratch) memory
RRTMG_HD bool mask_bit(const uint64_t *w, int l) {
  const uint64_t v = (l < 64) ? w[0] : (l < 128) ? w[1] : (l < 192) ? w[2] : w[3];
  return (v >> (l & 63)) & 1ull;
}

constexpr double kTblInt = 10000.0;
constexpr double kBpade = 1.0 / 0.278;

}  // namespace rrtmg
