// rrtmg_kiss_host.h -- host side of the KISS jump-ahead (see kiss_jump in rrtmg_sw_device.h): the operators that
// advance the four component generators by n_g = changeSeed + g * draws_per_subcolumn steps, one record of
// kKissJumpWords 32-bit words per sub-column g.  They depend on (nsub, nlay, icld, changeSeed) only.
#pragma once
#include <cstdint>
#include <vector>

#include "rrtmg_sw_device.h"

namespace rrtmg {

inline uint32_t kiss_powmod(uint32_t a, uint32_t e, uint32_t m) {
  uint64_t r = 1, b = a % m;
  while (e) {
    if (e & 1u) r = r * b % m;
    b = b * b % m;
    e >>= 1;
  }
  return (uint32_t)r;
}

inline void kiss_build_jumps(int nsub, int nlay, int icld, int changeSeed, std::vector<uint32_t> &out) {
  out.assign((size_t)nsub * kKissJumpWords, 0u);
  // one xorshift step as a matrix: column i = step(1 << i)
  uint32_t M1[32];
  for (int i = 0; i < 32; ++i) { uint32_t b = 1u << i; b ^= b << 13; b ^= b >> 17; b ^= b << 5; M1[i] = b; }
  auto apply = [](const uint32_t *M, uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; ++i) if ((v >> i) & 1u) r ^= M[i]; return r; };
  const uint32_t per = (icld == 3) ? 1u : (uint32_t)nlay;
  for (int g = 0; g < nsub; ++g) {
    const uint32_t n = (uint32_t)(changeSeed < 0 ? 0 : changeSeed) + (uint32_t)g * per;
    uint32_t *J = out.data() + (size_t)g * kKissJumpWords;
    J[0] = n;
    // affine map of n congruential steps and M^n, by binary exponentiation
    uint32_t A = 1u, C = 0u, sa = 69069u, sc = 1327217885u;   // (A, C): x -> A x + C ; (sa, sc): current 2^k-step map
    uint32_t R[32], S[32], T[32];
    for (int i = 0; i < 32; ++i) { R[i] = 1u << i; S[i] = M1[i]; }
    for (uint32_t e = n; e; e >>= 1) {
      if (e & 1u) {
        A = sa * A; C = sa * C + sc;                                        // apply the 2^k-step map after (A, C)
        for (int i = 0; i < 32; ++i) T[i] = apply(S, R[i]);
        for (int i = 0; i < 32; ++i) R[i] = T[i];
      }
      sc = sa * sc + sc; sa = sa * sa;                                      // square the 2^k-step map
      for (int i = 0; i < 32; ++i) T[i] = apply(S, S[i]);
      for (int i = 0; i < 32; ++i) S[i] = T[i];
    }
    J[1] = A; J[2] = C;
    J[3] = n > 2u ? kiss_powmod(18000u, n - 2u, kKissM3) : 1u;
    J[4] = n > 2u ? kiss_powmod(30903u, n - 2u, kKissM4) : 1u;
    for (int i = 0; i < 32; ++i) J[8 + i] = R[i];
  }
}

}  // namespace rrtmg
