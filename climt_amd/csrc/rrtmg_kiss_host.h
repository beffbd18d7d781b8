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

// n-step operators of the congruential generator (affine map x -> A x + C) and of the xorshift (matrix R over GF(2), column
// words), by binary exponentiation
struct KissOp { uint32_t A, C, R[32]; };
inline uint32_t kiss_apply(const uint32_t *M, uint32_t v) {
  uint32_t r = 0;
  for (int i = 0; i < 32; ++i) r ^= M[i] & (0u - ((v >> i) & 1u));
  return r;
}
// o <- o followed by p (all operators here are powers of the same one-step maps, so the order does not matter)
inline void kiss_compose(KissOp &o, const KissOp &p) {
  o.C = p.A * o.C + p.C; o.A = p.A * o.A;
  uint32_t T[32];
  for (int i = 0; i < 32; ++i) T[i] = kiss_apply(p.R, o.R[i]);
  for (int i = 0; i < 32; ++i) o.R[i] = T[i];
}
inline KissOp kiss_power(uint32_t n) {
  KissOp r, s;
  r.A = 1u; r.C = 0u; s.A = 69069u; s.C = 1327217885u;
  for (int i = 0; i < 32; ++i) { r.R[i] = 1u << i; uint32_t b = 1u << i; b ^= b << 13; b ^= b >> 17; b ^= b << 5; s.R[i] = b; }
  for (uint32_t e = n; e; e >>= 1) {
    if (e & 1u) kiss_compose(r, s);
    const KissOp t = s;
    kiss_compose(s, t);
  }
  return r;
}

// The sub-columns' jump distances are n_g = changeSeed + g * per: one power for changeSeed, one for per, then one composition
// per sub-column (a component that redraws its seed every call rebuilds this table every call: 0.1 ms instead of the 1.5 ms
// that 140 independent exponentiations took -- the host side of a device-resident McICA step was bound by it).
inline void kiss_build_jumps(int nsub, int nlay, int icld, int changeSeed, std::vector<uint32_t> &out) {
  out.assign((size_t)nsub * kKissJumpWords, 0u);
  const uint32_t per = (icld == 3) ? 1u : (uint32_t)nlay;
  const uint32_t n0 = (uint32_t)(changeSeed < 0 ? 0 : changeSeed);
  KissOp cur = kiss_power(n0);
  const KissOp step = kiss_power(per);
  for (int g = 0; g < nsub; ++g) {
    const uint32_t n = n0 + (uint32_t)g * per;
    uint32_t *J = out.data() + (size_t)g * kKissJumpWords;
    J[0] = n;
    J[1] = cur.A; J[2] = cur.C;
    J[3] = n > 2u ? kiss_powmod(18000u, n - 2u, kKissM3) : 1u;
    J[4] = n > 2u ? kiss_powmod(30903u, n - 2u, kKissM4) : 1u;
    for (int i = 0; i < 32; ++i) J[8 + i] = cur.R[i];
    kiss_compose(cur, step);
  }
}

}  // namespace rrtmg
