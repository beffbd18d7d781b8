// mt_host_stream.cpp -- TEST INFRASTRUCTURE: the reference's Mersenne-twister sub-column mask as ONE sequential stream on the
// host, the way the reference generates it (checked against the reference Fortran's masks).  The library builds the same bits
// on the device by jump-ahead (climt_amd/csrc/rrtmg_mt_device.hip, rrtmg_mt_jump.cpp) and contains no host generator; the host
// emulation of the tests uses this one, and the device path is tested bit for bit against it.
#include <cstdint>
#include <vector>

namespace rrtmg {

// ---- MT19937 exactly as mcica_random_numbers.f90:77-302 (initialize_scalar, nextState, temper,
// getRandomReal); stream order of generate_stochastic_clouds: (sub-column, column, layer) for
// overlap 1/2, (sub-column, column) for overlap 3 (mcica_subcol_gen_sw.f90:360-367,:386-393,:420-428)
namespace {
struct MT {
  uint32_t st[624];
  int cur;
  explicit MT(int32_t seed) {
    st[0] = (uint32_t)seed;
    for (int i = 1; i < 624; ++i) st[i] = 1812433253u * (st[i - 1] ^ (st[i - 1] >> 30)) + (uint32_t)i;
    cur = 624;
  }
  static uint32_t twist(uint32_t u, uint32_t v) {
    const uint32_t mix = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (mix >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
  }
  void next_state() {
    for (int k = 0; k < 624 - 397; ++k) st[k] = st[k + 397] ^ twist(st[k], st[k + 1]);
    for (int k = 624 - 397; k < 623; ++k) st[k] = st[k + 397 - 624] ^ twist(st[k], st[k + 1]);
    st[623] = st[396] ^ twist(st[623], st[0]);
    cur = 0;
  }
  // advance the stream by n draws without tempering them (state regenerations only)
  void skip(long n) {
    while (n > 0) {
      if (cur >= 624) next_state();
      const long k = n < 624 - cur ? n : 624 - cur;
      cur += (int)k;
      n -= k;
    }
  }
  double real() {
    if (cur >= 624) next_state();
    uint32_t y = st[cur++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    const int32_t li = (int32_t)y;
    // getRandomReal: (localInt + 2.0**32_rb) is default-real (single precision) arithmetic for
    // negative localInt (mcica_random_numbers.f90:288-292)
    if (li < 0) return (double)((float)li + 4294967296.0f) / 4294967295.0;
    return (double)li / 4294967295.0;
  }
};
}  // namespace

// col0 / ncol_total: the ncol columns are columns col0 .. col0+ncol-1 of a grid of ncol_total columns (a shard of a
// multi-GPU run; ncol_total <= 0: not sharded).  The draws of the other shards' columns are skipped, so the shard gets
// exactly the bits the unsharded call gives these columns.
void mt_mask_host(int ncol, int nlay, int nsub, int icld, int seed, const double *cldfr, std::vector<uint64_t> &mask, int nw,
                  int col0, int ncol_total) {
  mask.assign((size_t)nsub * nw * ncol, 0ull);
  if (icld == 0) return;
  MT mt(seed);
  const double cldmin = 1.0e-20;
  const long per_col = icld == 3 ? 1 : nlay;                                    // draws per (sub-column, column)
  const long before = ncol_total > 0 ? (long)col0 * per_col : 0;
  const long after = ncol_total > 0 ? (long)(ncol_total - col0 - ncol) * per_col : 0;
  for (int g = 0; g < nsub; ++g) {
    mt.skip(before);
    for (int c = 0; c < ncol; ++c) {
      double cdf_prev = 0.0, cmax = 0.0;
      if (icld == 3) cmax = mt.real();
      for (int l = 0; l < nlay; ++l) {
        double cf = cldfr[(size_t)l * ncol + c];
        if (cf < cldmin) cf = 0.0;
        double cdf;
        if (icld == 3) {
          cdf = cmax;
        } else {
          cdf = mt.real();
          if (icld == 2 && l > 0) {
            double cfm = cldfr[(size_t)(l - 1) * ncol + c];
            if (cfm < cldmin) cfm = 0.0;
            if (cdf_prev > 1.0 - cfm) cdf = cdf_prev; else cdf = cdf * (1.0 - cfm);
          }
        }
        cdf_prev = cdf;
        if (cdf >= 1.0 - cf) mask[((size_t)g * nw + (l >> 6)) * ncol + c] |= 1ull << (l & 63);
      }
    }
    mt.skip(after);
  }
}

}  // namespace rrtmg
