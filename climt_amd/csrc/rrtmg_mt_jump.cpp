// rrtmg_mt_jump.cpp -- jump-ahead for the reference's Mersenne-twister stream (host part).
//
// The reference draws the McICA sub-column numbers of its default generator from ONE sequential MT19937 stream over
// (sub-column, column, layer) (mcica_subcol_gen_sw.f90:360-367, mcica_random_numbers.f90:77-302): 55 / 69 million draws per
// shortwave / longwave call at 8192 columns x 60 layers, 14.5 G for the 1440 x 720 x 100 grid -- which every rank of a sharded
// run would have to walk through to reach its own columns.  MT19937 is a linear recurrence over GF(2): with
//     x[m + 624] = x[m + 397] ^ twist(x[m], x[m + 1]),      draw n = temper(x[624 + n]),
// every bit sequence of x (from x[1] on) is annihilated by the same primitive polynomial phi of degree 19937, so the window
// 624 words ahead by J positions is a FIXED GF(2)-combination of the windows at the start:
//     x[1 + J + j] = XOR over the set bits i of (t^J mod phi) of x[1 + i + j],      j = 0 .. 623
// (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer 2008).  The polynomials depend only on the positions the segments of
// a call start at -- one segment per sub-column: (g * ncol_total + col0) * draws_per_column -- not on the seed: they are built
// once per grid shape here (phi by Berlekamp-Massey on the recurrence's own output, t^J by square-and-multiply, the
// segments' polynomials by one multiplication each) and applied on the device to the 20 561 words that follow each call's
// seed (rrtmg_mt_device.hip).
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace rrtmg {

namespace {
constexpr int kDeg = 19937;
constexpr int kW = 312;   // 64-bit words of a polynomial of degree <= 19967
typedef std::vector<uint64_t> Poly;

inline bool bit(const uint64_t *p, int i) { return (p[i >> 6] >> (i & 63)) & 1u; }

// untempered MT19937 words x[0 .. n) from a seed (mcica_random_numbers.f90:139-150 initialize_scalar, :97-121 nextState)
void mt_words(uint32_t seed, std::vector<uint32_t> &x, size_t n) {
  x.resize(n < 625 ? 625 : n);
  x[0] = seed;
  for (int i = 1; i < 624; ++i) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
  for (size_t m = 0; m + 624 < x.size(); ++m) {
    const uint32_t mix = (x[m] & 0x80000000u) | (x[m + 1] & 0x7fffffffu);
    x[m + 624] = x[m + 397] ^ (mix >> 1) ^ ((x[m + 1] & 1u) ? 0x9908b0dfu : 0u);
  }
}

// Berlekamp-Massey over GF(2): the shortest f with sum_i f_i s[n + i] = 0 for all n, deg f = L (f_L = 1).
Poly minimal_polynomial(const std::vector<uint8_t> &s) {
  const int N = (int)s.size(), words = N / 64 + 2;
  std::vector<uint64_t> C(words, 0), B(words, 0), T(words), R(words, 0);   // R bit i = s[n - i]
  C[0] = 1; B[0] = 1;
  int L = 0, m = -1;
  for (int n = 0; n < N; ++n) {
    for (int w = words - 1; w > 0; --w) R[w] = (R[w] << 1) | (R[w - 1] >> 63);
    R[0] = (R[0] << 1) | (uint64_t)(s[n] & 1);
    uint64_t acc = 0;
    for (int w = 0; w <= L / 64; ++w) acc ^= C[w] & R[w];
    if (!(__builtin_popcountll(acc) & 1)) continue;
    T = C;
    const int sh = n - m, ws = sh >> 6, bs = sh & 63;
    for (int w = words - 1; w >= ws; --w) {
      uint64_t v = B[w - ws] << bs;
      if (bs && w - ws - 1 >= 0) v |= B[w - ws - 1] >> (64 - bs);
      C[w] ^= v;
    }
    if (2 * L <= n) { L = n + 1 - L; B = T; m = n; }
  }
  // connection polynomial C (s[n] = sum_{i=1..L} C_i s[n - i]) -> f_{L - i} = C_i
  Poly f(kW + 1, 0);
  if (L != kDeg) return Poly();
  for (int i = 0; i <= L; ++i)
    if (bit(C.data(), i)) f[(L - i) >> 6] |= 1ull << ((L - i) & 63);
  return f;
}

const Poly &phi() {
  static Poly p;
  static std::once_flag once;
  std::call_once(once, [] {
    std::vector<uint32_t> x;
    const int N = 2 * kDeg + 64;
    mt_words(5489u, x, (size_t)N + 8);
    std::vector<uint8_t> s(N);
    for (int n = 0; n < N; ++n) s[n] = (uint8_t)(x[n + 1] & 1u);
    p = minimal_polynomial(s);
  });
  return p;
}

// acc = acc * t^8 mod phi, acc of degree < 19937: shift by one byte, fold the byte that left the top back in (red[o] = o(t) *
// t^19937 mod phi, o = 0..255)
struct Phi8 {
  uint64_t red[256][kW];
  Phi8() {
    const uint64_t *f = phi().data();
    std::memset(red, 0, sizeof red);
    // t^19937 = phi - t^19937 (the lower terms of phi); then red[2o] = red[o] * t, red[2o + 1] = red[2o] ^ red[1]
    for (int w = 0; w < kW; ++w) red[1][w] = f[w];
    red[1][kDeg >> 6] &= ~(1ull << (kDeg & 63));
    for (int o = 2; o < 256; ++o) {
      const uint64_t *h = red[o >> 1];
      uint64_t carry = 0;
      for (int w = 0; w < kW; ++w) { const uint64_t v = h[w]; red[o][w] = (v << 1) | carry; carry = v >> 63; }
      if (bit(red[o], kDeg)) { red[o][kDeg >> 6] &= ~(1ull << (kDeg & 63)); for (int w = 0; w < kW; ++w) red[o][w] ^= red[1][w]; }
      if (o & 1) for (int w = 0; w < kW; ++w) red[o][w] ^= red[1][w];
    }
  }
};
const Phi8 &phi8() {
  static const Phi8 *p = new Phi8();
  return *p;
}

// r = a * b mod phi (all of degree < 19937): Horner over the BYTES of a, with the 256 multiples v(t) * b of b in a table
// (0.9 ms instead of the 5.5 ms of a bit-by-bit Horner)
void mulmod(const uint64_t *a, const uint64_t *b, uint64_t *r) {
  const Phi8 &R = phi8();
  std::vector<uint64_t> tab((size_t)256 * kW, 0);
  auto T = [&](int v) { return &tab[(size_t)v * kW]; };
  std::memcpy(T(1), b, kW * sizeof(uint64_t));
  for (int v = 2; v < 256; ++v) {
    const uint64_t *h = T(v >> 1);
    uint64_t *d = T(v), carry = 0;
    for (int w = 0; w < kW; ++w) { const uint64_t x = h[w]; d[w] = (x << 1) | carry; carry = x >> 63; }
    if (bit(d, kDeg)) { d[kDeg >> 6] &= ~(1ull << (kDeg & 63)); for (int w = 0; w < kW; ++w) d[w] ^= R.red[1][w]; }
    if (v & 1) for (int w = 0; w < kW; ++w) d[w] ^= b[w];
  }
  uint64_t acc[kW + 1];
  std::memset(acc, 0, sizeof acc);
  const int nbytes = (kDeg + 7) / 8;   // 2493: bits 0 .. 19943 of a (the top ones are zero)
  for (int i = nbytes - 1; i >= 0; --i) {
    // acc *= t^8: the byte that crosses bit 19937 comes back through red[]
    const unsigned over = (unsigned)((acc[kDeg >> 6] >> ((kDeg & 63) - 8)) & 0xffu);   // bits 19929 .. 19936
    uint64_t carry = 0;
    for (int w = 0; w < kW; ++w) { const uint64_t v = acc[w]; acc[w] = (v << 8) | carry; carry = v >> 56; }
    acc[kDeg >> 6] &= (1ull << (kDeg & 63)) - 1;
    const unsigned av = (unsigned)((a[i >> 3] >> (8 * (i & 7))) & 0xffu);
    const uint64_t *ro = R.red[over], *tv = T((int)av);
    for (int w = 0; w < kW; ++w) acc[w] ^= ro[w] ^ tv[w];
  }
  std::memcpy(r, acc, kW * sizeof(uint64_t));
}

// r = t^e mod phi
void pow_t(uint64_t e, uint64_t *r) {
  const uint64_t *f = phi().data();
  uint64_t acc[kW], tmp[kW];
  std::memset(acc, 0, sizeof acc);
  acc[0] = 1;
  int top = 63;
  while (top > 0 && !((e >> top) & 1u)) --top;
  for (int b = top; b >= 0; --b) {
    if (b != top) { mulmod(acc, acc, tmp); std::memcpy(acc, tmp, sizeof acc); }
    if ((e >> b) & 1u) {
      uint64_t carry = 0;
      for (int w = 0; w < kW; ++w) { const uint64_t v = acc[w]; acc[w] = (v << 1) | carry; carry = v >> 63; }
      if (bit(acc, kDeg)) for (int w = 0; w < kW; ++w) acc[w] ^= f[w];
    }
  }
  std::memcpy(r, acc, sizeof acc);
}
}  // namespace

int mt_jump_words() { return kW; }

constexpr int kMtListMax = 19968 + 16;     // exponents of one polynomial, padded to a multiple of 16
constexpr int kMtListPad = 19937 + 624;   // (its words are beyond what the device keeps of the seed's stream: zeros)

// The jump polynomials of a call as lists of their set bits.  The call's draws are nsub runs (one per sub-column) that start
// at first + g * stride; each run is cut into npiece pieces of `piece` draws, so that segment (g, s) starts at draw
//     first + g * stride + s * piece            and has the polynomial            t^(that - 1) mod phi
// (a segment that starts at draw 0 needs no jump -- its window is the seed's own: count -1).  lists[(g * npiece + s) *
// kMtListMax ..] holds the exponents i of the polynomial's terms, padded to a multiple of 16 with kMtListPad, whose words read
// as zeros on the device; counts[] the padded lengths.  Built on a few threads: t^stride and the t^(s * piece) once, then
// one multiplication per segment (2.5 ms each; 140 sub-columns x 4 pieces: 0.2 s once per grid shape); cached, and COPIED out.
// Returns false if phi could not be established (never observed; the caller reports it).
bool mt_jump_lists(uint64_t first, uint64_t stride, int nsub, uint64_t piece, int npiece, std::vector<uint32_t> &lists, std::vector<int32_t> &counts) {
  static std::mutex mu;
  static std::map<std::vector<uint64_t>, std::pair<std::vector<uint32_t>, std::vector<int32_t>>> cache;
  std::lock_guard<std::mutex> lock(mu);
  if (phi().empty()) return false;
  const std::vector<uint64_t> key = {first, stride, (uint64_t)nsub, piece, (uint64_t)npiece};
  auto it = cache.find(key);
  if (it == cache.end()) {
    const int nseg = nsub * npiece;
    std::vector<uint64_t> P((size_t)nseg * kW, 0), step(kW), off((size_t)npiece * kW, 0);
    pow_t(stride, step.data());
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 8 ? 8 : nt);
    auto on_threads = [&](int n, const std::function<void(int, int)> &f) {
      const unsigned use = n < 16 ? 1 : nt;
      std::vector<std::thread> th;
      for (unsigned t = 1; t < use; ++t) th.emplace_back(f, (int)((long)n * t / use), (int)((long)n * (t + 1) / use));
      f(0, (int)((long)n / use));
      for (auto &x : th) x.join();
    };
    // the runs' first segments: a chain p[g] = p[g - 1] * t^stride, cut into stretches that start from their own t^(...)
    const int g0 = first == 0 ? 1 : 0;
    on_threads(nsub - g0, [&](int a, int b) {
      a += g0; b += g0;
      if (a >= b) return;
      pow_t(first + (uint64_t)a * stride - 1, &P[(size_t)a * npiece * kW]);
      for (int g = a + 1; g < b; ++g) mulmod(&P[(size_t)(g - 1) * npiece * kW], step.data(), &P[(size_t)g * npiece * kW]);
    });
    // the other pieces: p[g][s] = p[g][0] * t^(s * piece); run 0 of an unsharded call has no p[0][0]: t^(s * piece - 1) itself
    for (int sidx = 1; sidx < npiece; ++sidx) pow_t((uint64_t)sidx * piece, &off[(size_t)sidx * kW]);
    on_threads(nsub * (npiece - 1), [&](int a, int b) {
      for (int q = a; q < b; ++q) {
        const int g = q / (npiece - 1), sidx = 1 + q % (npiece - 1);
        uint64_t *dst = &P[((size_t)g * npiece + sidx) * kW];
        if (g < g0) pow_t((uint64_t)sidx * piece - 1, dst);
        else mulmod(&P[(size_t)g * npiece * kW], &off[(size_t)sidx * kW], dst);
      }
    });
    std::pair<std::vector<uint32_t>, std::vector<int32_t>> e;
    e.first.assign((size_t)nseg * kMtListMax, (uint32_t)kMtListPad);
    e.second.assign(nseg, 0);
    for (int k = 0; k < nseg; ++k) {
      if (first == 0 && k == 0) { e.second[k] = -1; continue; }
      uint32_t *L = &e.first[(size_t)k * kMtListMax];
      int n = 0;
      for (int i = 0; i < kDeg; ++i)
        if (bit(&P[(size_t)k * kW], i)) L[n++] = (uint32_t)i;
      e.second[k] = (n + 15) & ~15;
    }
    if (cache.size() > 8) cache.clear();
    it = cache.emplace(key, std::move(e)).first;
  }
  lists = it->second.first;
  counts = it->second.second;
  return true;
}

// (tests) the polynomial of one segment as a bit vector
void mt_jump_polynomial_host(uint64_t start, uint64_t *p) { pow_t(start - 1, p); }

}  // namespace rrtmg

#ifdef RRTMG_MT_JUMP_SELFTEST
#include <chrono>
#include <cstdio>
int main() {
  using namespace rrtmg;
  auto t0 = std::chrono::steady_clock::now();
  const bool ok = !phi().empty();
  auto t1 = std::chrono::steady_clock::now();
  printf("phi: %s (%.0f ms)\n", ok ? "degree 19937" : "FAILED", std::chrono::duration<double, std::milli>(t1 - t0).count());
  if (!ok) return 1;
  std::vector<uint32_t> x;
  for (uint64_t first : {0ull, 8192ull * 60 * 3 + 17}) {
    const uint64_t stride = 8192ull * 60, piece = 100000;
    const int nsub = 3, npiece = 4;
    mt_words(12345u, x, (size_t)(first + nsub * stride + 2000));
    std::vector<uint32_t> lists; std::vector<int32_t> counts;
    t0 = std::chrono::steady_clock::now();
    mt_jump_lists(first, stride, nsub, piece, npiece, lists, counts);
    t1 = std::chrono::steady_clock::now();
    printf("%d segment lists: %.0f ms\n", nsub * npiece, std::chrono::duration<double, std::milli>(t1 - t0).count());
    for (int k = 0; k < nsub * npiece; ++k) {
      const uint64_t n = first + (k / npiece) * stride + (k % npiece) * piece;
      if (counts[k] < 0) { printf("segment %d at draw %llu: seed window\n", k, (unsigned long long)n); continue; }
      uint32_t w[624] = {0};
      for (int q = 0; q < counts[k]; ++q) {
        const uint32_t i = lists[(size_t)k * kMtListMax + q];
        if (i == (uint32_t)kMtListPad) continue;
        for (int j = 0; j < 624; ++j) w[j] ^= x[1 + i + j];
      }
      int bad = 0;
      for (int j = 0; j < 624; ++j) bad += w[j] != x[n + j];
      printf("segment %d at draw %llu (%d terms): %s\n", k, (unsigned long long)n, counts[k], bad ? "MISMATCH" : "ok");
      if (bad) return 1;
    }
  }
  return 0;
}
#endif
