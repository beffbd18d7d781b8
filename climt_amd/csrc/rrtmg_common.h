// rrtmg_common.h -- shared host/device definitions for librrtmg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/rrtmg_hip.h"   // the status codes (RRTMG_ERR_*)
#include "rrtmg_profile.h"

// One code path: the experiments of rounds 1-3 (ablations, occupancy and launch-shape variants, alternative layouts) were
// build switches; their results are in DESIGN.md 5 and the git history, the switches are gone.  The only conditional
// compilation left separates the host emulation of the device functions (tests/emu: no __HIP_DEVICE_COMPILE__) from the
// device build, and RRTMG_PROFILE (rrtmg_profile.h) adds timing diagnostics to a non-product build.
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

// Quick fp64 division for the flux arithmetic of the shortwave hot loops: v_rcp_f64 + ONE Newton-Raphson step
// (relative error <= 2.2e-15, i.e. ~20 ulp; 4 instructions) instead of the 11-instruction IEEE sequence (div_scale x2,
// rcp, 6 fma, div_fmas, div_fixup) -- divisions were ~60 % of the VALU instructions of the solve kernels, which are
// VALU-issue bound.  Fluxes move by ~1e-10 W m-2 (bar: 1e-2); a second step would give 1 ulp.  Operands here are
// O(1e-20..1e20) and never zero/inf/denormal.  NOT used for the quotient of a NEAREST-ENTRY table index in the longwave (LW_TDIV:
// a last-place difference there reads a neighbouring entry 1e-4 away); the shortwave's species parameter (sw_specparm: specparm ->
// js, fs) takes it since round 6 -- between rows js and js + 1 the interpolation is continuous, the integer can only move together
// with fs across 0 / 1 -- the longwave's (lw_spec) keeps the correctly rounded `/`.  On the host (tests/emu) it is the plain division.
RRTMG_HD double qdiv(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(b);                    // measured on gfx950: relative error <= 4.6e-8
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);    // one Newton step: <= 2.2e-15 (tools/micro/rcp_accuracy.hip)
  return a * r;
#else
  return a / b;
#endif
}
RRTMG_HD double qrcp(double b) { return qdiv(1.0, b); }
// Square root of a normal, strictly positive number (the two-stream k = sqrt(gamma1^2 - gamma2^2) > 0): v_rsq_f64 and one
// coupled Goldschmidt iteration -- 4 instructions behind the rsq instead of the 18 of the IEEE expansion (range scaling, class
// fix-up, two corrections).  Measured <= 4.2e-15 relative (tools/micro/rsq_accuracy.hip: x*rsq 5.2e-8, + the iteration 4.1e-15;
// the residual correction that rounds 2-5 had behind it gives sqrt() itself and costs three more instructions per layer operator
// in a kernel whose time follows its VALU instruction count: docs/EXPERIMENTS.md E).  The accuracy class of qdiv; the operand
// itself carries the reference's cancellation noise of ~1e-16 / (1 - w).
RRTMG_HD double qsqrt(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(a);
  const double g = a * y, h = 0.5 * y;
  return __builtin_fma(g, __builtin_fma(-h, g, 0.5), g);
#else
  return sqrt(a);
#endif
}

// ------------------------------------------------------------------------------------------------------
// G consecutive g-points of one band, carried through the k-distribution arithmetic by one thread.  Everything
// that does not depend on the g-point (layer state, species mixtures, interpolation weights, table rows) is
// computed once and the G table entries of a row come from ONE 16-byte-granular load, because the reduced
// tables are stored g-point-fastest ([row][ng]).  Element-wise operators keep the reference's operation order
// per g-point.
// ------------------------------------------------------------------------------------------------------
template <int G> struct V {
  double v[G];
  RRTMG_HD double &operator[](int i) { return v[i]; }
  RRTMG_HD double operator[](int i) const { return v[i]; }
};
#define RRTMG_VOP(OP)                                                                                       \
  template <int G> RRTMG_HD V<G> operator OP(const V<G> &a, const V<G> &b) { V<G> r; _Pragma("unroll") for (int i = 0; i < G; ++i) r.v[i] = a.v[i] OP b.v[i]; return r; } \
  template <int G> RRTMG_HD V<G> operator OP(const V<G> &a, double b) { V<G> r; _Pragma("unroll") for (int i = 0; i < G; ++i) r.v[i] = a.v[i] OP b; return r; }           \
  template <int G> RRTMG_HD V<G> operator OP(double a, const V<G> &b) { V<G> r; _Pragma("unroll") for (int i = 0; i < G; ++i) r.v[i] = a OP b.v[i]; return r; }
RRTMG_VOP(+)
RRTMG_VOP(-)
RRTMG_VOP(*)
#undef RRTMG_VOP
template <int G> RRTMG_HD V<G> vsplat(double x) { V<G> r; _Pragma("unroll") for (int i = 0; i < G; ++i) r.v[i] = x; return r; }

// G consecutive doubles at p; p is 16-byte aligned (tables are 16-byte aligned, ng and the first g-point of a
// work item are even)
template <int G> RRTMG_HD V<G> vload(const double *p) {
  V<G> r;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (G % 2 == 0) {
    _Pragma("unroll") for (int i = 0; i < G; i += 2) {
      const double2 x = *reinterpret_cast<const double2 *>(p + i);
      r.v[i] = x.x; r.v[i + 1] = x.y;
    }
    return r;
  }
#endif
  _Pragma("unroll") for (int i = 0; i < G; ++i) r.v[i] = p[i];
  return r;
}

// G consecutive doubles to p (16-byte aligned), one 16-byte store per pair
template <int G> RRTMG_HD void vstore(double *p, const V<G> &x) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (G % 2 == 0) {
    _Pragma("unroll") for (int i = 0; i < G; i += 2) *reinterpret_cast<double2 *>(p + i) = make_double2(x.v[i], x.v[i + 1]);
    return;
  }
#endif
  _Pragma("unroll") for (int i = 0; i < G; ++i) p[i] = x.v[i];
}

// Scratch-slab rows of the solve kernels: the G values of one (layer, field) for the 64 lanes of a tile.
// Pair-major, [G/2][lane][2]: each 16-byte access of the wave is one contiguous 1 KB run (round 1: [lane][G], 16 bytes every
// 32 per instruction).  p -> this lane's first element (slab row + lane * 2), stride = lanes per row.  Non-temporal
// accesses: the slab is written once and read once, a whole sweep later.
template <int G> RRTMG_HD long scr_lane_offset(int lane) { return (G % 2 == 0) ? (long)lane * 2 : (long)lane * G; }
template <int G> RRTMG_HD V<G> scr_load(const double *p, long stride) {
  V<G> r;
  if constexpr (G % 2 == 0) {
    _Pragma("unroll") for (int i = 0; i < G; i += 2) {
      const double *q = p + (long)(i / 2) * stride * 2;
#if defined(__HIP_DEVICE_COMPILE__)
      typedef double d2 __attribute__((ext_vector_type(2)));
      const d2 x = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(q));
      r.v[i] = x.x; r.v[i + 1] = x.y;
#else
      r.v[i] = q[0]; r.v[i + 1] = q[1];
#endif
    }
  } else {
    _Pragma("unroll") for (int i = 0; i < G; ++i) r.v[i] = p[i];
  }
  return r;
}
template <int G> RRTMG_HD void scr_store(double *p, long stride, const V<G> &x) {
  if constexpr (G % 2 == 0) {
    _Pragma("unroll") for (int i = 0; i < G; i += 2) {
      double *q = p + (long)(i / 2) * stride * 2;
#if defined(__HIP_DEVICE_COMPILE__)
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 v; v.x = x.v[i]; v.y = x.v[i + 1];
      __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(q));
#else
      q[0] = x.v[i]; q[1] = x.v[i + 1];
#endif
    }
  } else {
    _Pragma("unroll") for (int i = 0; i < G; ++i) p[i] = x.v[i];
  }
}

// Partial-flux planes: written once by a solve kernel, read once by the flux kernel: non-temporal accesses.
RRTMG_HD void part_store(double *p, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
RRTMG_HD double part_load(const double *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

// g-point-fastest table view: element (row, ig0 + j) at p[row * NG + j], p already offset by the first g-point
template <int G, int NG> struct KTab {
  const double *p;
  RRTMG_HD V<G> operator[](int row) const { return vload<G>(p + (long)row * NG); }
};

// bit l of a 4-word (<= 256 layers) cloud mask held in registers: selects instead of dynamic indexing, which
// would push the array into private (scratch) memory
RRTMG_HD bool mask_bit(const uint64_t *w, int l) {
  const uint64_t v = (l < 64) ? w[0] : (l < 128) ? w[1] : (l < 192) ? w[2] : w[3];
  return (v >> (l & 63)) & 1ull;
}

// Transmittance lookup index: itind = tblint*x/(bpade+x) + 0.5 truncated (rrtmg_sw_reftra.f90:199-203,
// rrtmg_lw_rtrn.f90:426-430).  Index arithmetic stays in fp64 so table entries do not flip.
constexpr double kTblInt = 10000.0;
constexpr double kBpade = 1.0 / 0.278;

}  // namespace rrtmg
