// rrtmg_profile.h -- timing diagnostics of a NON-PRODUCT build (hipcc -DRRTMG_PROFILE ...); every macro is empty otherwise, so
// the product library contains none of this.  Results of a profile build are WRONG by design where noted.
//
//   env RRTMG_HIP_ONLY_ITEM=k      the solve kernels run position k of their launch order alone (every other workgroup exits):
//                                  kernel_ms is then that work item's duration for all tiles (tools/item_times.py).  Wrong fluxes.
//   longwave phase timers          lane 0 of every wave adds, per phase of the two sweeps, the shader clocks between phase
//                                  boundaries at which the wave waits for everything outstanding (so a latency is booked on the
//                                  phase that issued the access; the overlap between phases is lost: an upper bound of the loop
//                                  time).  rrtmg_hip_lw_fluxes prints the per-layer averages to stderr (tools/gpu_session.sh phases).
#pragma once

#ifdef RRTMG_PROFILE
#include <cstdlib>
#define RRTMG_PROFILE_FIELDS int only_item; unsigned long long *phase;
#define RRTMG_PROFILE_READ_ONLY_ITEM(d) { (d).only_item = -1; if (const char *e__ = getenv("RRTMG_HIP_ONLY_ITEM")) (d).only_item = atoi(e__); }
#define RRTMG_PROFILE_ONLY_ITEM(d, k) if ((d).only_item >= 0 && (k) != (d).only_item) return;
#if defined(__HIP_DEVICE_COMPILE__)
#define RRTMG_PH_DECL unsigned long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_c = __builtin_amdgcn_s_memtime();
#define RRTMG_PH_MARK(k, pin)                                                           \
  {                                                                                     \
    double pin__ = (pin);                                                               \
    asm volatile("s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)" : "+v"(pin__) : : "memory"); \
    const unsigned long long n__ = __builtin_amdgcn_s_memtime();                        \
    ph_t[k] += n__ - ph_c; ph_c = n__;                                                  \
  }
#define RRTMG_PH_FLUSH(d)                                                               \
  if ((threadIdx.x & 63) == 0 && (d).phase) {                                           \
    for (int k__ = 0; k__ < 8; ++k__) atomicAdd((d).phase + k__, ph_t[k__]);            \
    atomicAdd((d).phase + 8, 1ull);                                                     \
  }
#else
#define RRTMG_PH_DECL
#define RRTMG_PH_MARK(k, pin)
#define RRTMG_PH_FLUSH(d)
#endif
#else
#define RRTMG_PROFILE_FIELDS
#define RRTMG_PROFILE_READ_ONLY_ITEM(d)
#define RRTMG_PROFILE_ONLY_ITEM(d, k)
#define RRTMG_PH_DECL
#define RRTMG_PH_MARK(k, pin)
#define RRTMG_PH_FLUSH(d)
#endif
