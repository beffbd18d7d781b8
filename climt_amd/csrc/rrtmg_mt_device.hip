// rrtmg_mt_device.hip -- the reference's Mersenne-twister sub-column masks on the GPU (gfx950).
//
// The reference's DEFAULT McICA generator is one sequential MT19937 stream over (sub-column, column, layer)
// (mcica_random_numbers.f90:77-302, mcica_subcol_gen_sw.f90:360-428).  Generated on the host it cost 0.4-0.5 s per spectrum at
// 8192 columns x 60 layers -- a hundred times the radiation itself -- and every rank of a sharded run had to walk through the
// draws of all the others.  Here the stream is cut into segments -- the draws of this call's columns for one sub-column are
// contiguous: one run per sub-column, a large run in pieces of ~256 K draws -- every segment's 624-word window is formed by
// polynomial jump-ahead (rrtmg_mt_jump.cpp) from the 20 561 words that follow the seed, and one workgroup per segment runs the
// recurrence from there:
//   mt_seed_kernel    <<<1, 256>>>       the seed's initial window and the words behind it, x[0 .. kMtBase)
//   mt_jump_kernel    <<<segments, 640>>> window of segment k = XOR over the terms t^i of its polynomial of x[1 + i ..]
//   mt_stream_kernel  <<<segments, 256>>> the segment's tempered 32-bit draws, in stream order, to HBM (per group of
//                     sub-columns whose draws fit 1 GB: the buffer is bounded whatever the grid)
//   mt_mask_kernel    <<<(sub-columns, tiles), 64>>>  draws -> cloud-mask bits, one thread per (column, sub-column), with the
//                     reference's conversion to a real number and its overlap rules
// Same bits as the sequential stream (GPU test against the host generator the reference masks were checked with).
#include "rrtmg_ctx.h"

namespace rrtmg {

bool mt_jump_lists(uint64_t first, uint64_t stride, int nsub, uint64_t piece, int npiece, std::vector<uint32_t> &lists, std::vector<int32_t> &counts);

constexpr int kMtN = 624, kMtM = 397, kMtDeg = 19937;
constexpr int kMtBase = kMtDeg + kMtN + 1;   // x[0 .. kMtBase): everything a jump reads (base window at x[1])
constexpr int kMtListMax = 19968 + 16;     // as in rrtmg_mt_jump.cpp: exponents of one polynomial, padded to a multiple of 16 ...
constexpr int kMtListPad = kMtDeg + kMtN;   // ... with this exponent, whose words lie behind x[0 .. kMtBase): zeros in the LDS copy
constexpr int kMtXs = (1 + kMtListPad + kMtN + 3) & ~3;   // LDS copy of x in mt_jump_kernel (a multiple of 16 bytes)
constexpr int kMtJumpLds = kMtXs * 4;

__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v) {
  const uint32_t mix = (u & 0x80000000u) | (v & 0x7fffffffu);
  return (mix >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// A workgroup of kMtGenThreads = 256 threads replaces the window st[0 .. 624) (LDS) by the next 624 words, in place, as
// nextState does (mcica_random_numbers.f90:97-121).  Word k needs the OLD k and k + 1 and, for k < 227, the old k + 397, for
// k >= 227 the NEW k - 227 (k = 623: the new word 0 as its neighbour) -- so the words fall into three runs, [0, 227),
// [227, 454), [454, 624), inside which nothing depends on anything written in the same run: one word per thread, every thread
// reads, barrier, every thread writes, barrier.  Three such rounds per 624 draws (one wavefront doing four words per lane took
// 3000 cycles per window: the rounds' LDS round trips in a row, nothing else resident to fill them).
// OUT: the tempered words go to out[0 .. limit) as they are formed.
constexpr int kMtGenThreads = 256;
template <bool OUT>
__device__ __forceinline__ void mt_next_block(uint32_t *st, int tid, uint32_t *out, long limit) {
  constexpr int kRun = kMtN - kMtM;   // 227
#pragma unroll
  for (int run = 0; run < 3; ++run) {
    const int lo = run * kRun, hi = run == 2 ? kMtN : lo + kRun;
    const int k = lo + tid;
    uint32_t v = 0;
    if (k < hi) {
      const uint32_t c = run == 0 ? st[k + kMtM] : st[k - kRun];
      v = c ^ mt_twist(st[k], st[k + 1 < kMtN ? k + 1 : 0]);
    }
    __syncthreads();
    if (k < hi) {
      st[k] = v;
      if (OUT && k < limit) out[k] = mt_temper(v);
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(kMtGenThreads) mt_seed_kernel(uint32_t seed, uint32_t *x) {
  __shared__ uint32_t st[kMtN];
  const int tid = threadIdx.x;
  if (tid == 0) {
    uint32_t v = seed;
    st[0] = v;
    for (int i = 1; i < kMtN; ++i) { v = 1812433253u * (v ^ (v >> 30)) + (uint32_t)i; st[i] = v; }   // initialize_scalar, :139-150
  }
  __syncthreads();
  for (int i = tid; i < kMtN; i += kMtGenThreads) x[i] = st[i];
  for (int base = kMtN; base < kMtBase; base += kMtN) {
    mt_next_block<false>(st, tid, nullptr, 0);
    for (int i = tid; i < kMtN && base + i < kMtBase; i += kMtGenThreads) x[base + i] = st[i];
    __syncthreads();
  }
}

// window[k][j] = x[n_k + j]: for a segment that starts at draw 0 (count -1) the seed's own window, else the jump.  The
// polynomial arrives as the list of its exponents: the same for every lane and written by nothing in this kernel, so they come
// through the scalar cache, 16 at a time, and the words of 16 terms are in flight together.  Ten wavefronts, one window word
// per lane: the LDS copy of x allows one workgroup per CU, and a wavefront has at most 15 LDS reads outstanding -- it takes
// that many wavefronts to keep the LDS busy (three wavefronts folding four words per lane: 0.38 ms per workgroup instead of 0.24).
constexpr int kMtJumpThreads = 640;
__global__ void __launch_bounds__(kMtJumpThreads) mt_jump_kernel(const uint32_t *__restrict__ x, const uint32_t *__restrict__ lists,
                                                               const int32_t *__restrict__ counts, uint32_t *__restrict__ windows) {
  extern __shared__ uint32_t xs[];                // kMtXs words of the seed's stream (zeros behind x[kMtBase))
  const int k = blockIdx.x;
  const int n = counts[k];
  for (int i = threadIdx.x; i < kMtXs; i += kMtJumpThreads) xs[i] = i < kMtBase ? x[i] : 0u;
  __syncthreads();
  uint32_t *w = windows + (long)k * kMtN;
  const int j = threadIdx.x < kMtN ? threadIdx.x : 0;
  if (n < 0) {
    if (threadIdx.x < kMtN) w[j] = xs[j];
    return;
  }
  const uint32_t *__restrict__ L = lists + (long)k * kMtListMax;
  const uint32_t *xj = xs + 1 + j;
  uint32_t a = 0;
  for (int q = 0; q < n; q += 16) {               // (the lists are padded to multiples of 16 with an exponent whose words are zeros)
    uint32_t v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = xj[L[q + u]];
#pragma unroll
    for (int u = 0; u < 16; ++u) a ^= v[u];
  }
  if (threadIdx.x < kMtN) w[j] = a;
}

// the tempered draws of segment k = (run g, piece sidx): out[g * count + sidx * piece ..], `piece` of them (the run's last piece:
// what is left of its `count`)
// (launched per GROUP of sub-columns, see mt_mask_device: seg0 = first segment of the group, whose draws start at out[0])
__global__ void __launch_bounds__(kMtGenThreads) mt_stream_kernel(const uint32_t *windows, long count, long piece, int npiece, int seg0, uint32_t *out) {
  __shared__ uint32_t st[kMtN];
  const int tid = threadIdx.x, k = seg0 + blockIdx.x, g = blockIdx.x / npiece, sidx = blockIdx.x % npiece;   // (seg0 is a multiple of npiece)
  for (int i = tid; i < kMtN; i += kMtGenThreads) st[i] = windows[(long)k * kMtN + i];
  __syncthreads();
  const long begin = (long)sidx * piece;
  const long len = count - begin < piece ? count - begin : piece;
  uint32_t *o = out + (long)g * count + begin;
  for (long base = 0; base < len; base += kMtN) mt_next_block<true>(st, tid, o + base, len - base);
}

// getRandomReal (mcica_random_numbers.f90:282-296): a negative localInt goes through DEFAULT-real (single precision) arithmetic
__device__ __forceinline__ double mt_real(uint32_t y) {
  const int32_t li = (int32_t)y;
  if (li < 0) return (double)((float)li + 4294967296.0f) / 4294967295.0;
  return (double)li / 4294967295.0;
}

// draws -> mask bits, one thread per (column, sub-column); the overlap rules of generate_stochastic_clouds
// (mcica_subcol_gen_sw.f90:360-367 random, :386-393 maximum-random, :420-428 maximum) as in the sequential host restatement (tests/emu/mt_host_stream.cpp).
// The stream keeps a column's draws together (layer fastest), the threads of a wavefront are 64 columns: the tile's
// 64 x per_col draws -- one contiguous run of the stream -- are read coalesced into LDS and each lane takes its column from
// there (row stride per_col | 1: odd, so the lanes' words sit in different banks).
// (launched per group of sub-columns: g0 = the group's first sub-column, whose draws start at draws[0])
__global__ void __launch_bounds__(64) mt_mask_kernel(int ncol, int nlay, int icld, const double *cldfr, const uint32_t *draws, uint64_t *mask, int nw, int g0) {
  extern __shared__ uint32_t sh[];
  const int lane = threadIdx.x, col0 = blockIdx.y * 64, col = col0 + lane, g = g0 + blockIdx.x;
  const int per_col = icld == 3 ? 1 : nlay, ld = per_col | 1;
  const int ncols_here = ncol - col0 < 64 ? ncol - col0 : 64;
  const uint32_t *src = draws + ((long)blockIdx.x * ncol + col0) * per_col;
  for (int i = lane; i < ncols_here * per_col; i += 64) sh[(i / per_col) * ld + i % per_col] = src[i];
  __syncthreads();
  if (col >= ncol) return;
  const double cldmin = 1.0e-20;
  const uint32_t *r = sh + lane * ld;
  double cdf_prev = 0.0, cmax = 0.0, cfm = 0.0;
  if (icld == 3) cmax = mt_real(r[0]);
  uint64_t bits = 0;
  for (int l = 0; l < nlay; ++l) {
    double cf = cldfr[(long)l * ncol + col];
    if (cf < cldmin) cf = 0.0;
    double cdf;
    if (icld == 3) {
      cdf = cmax;
    } else {
      cdf = mt_real(r[l]);
      if (icld == 2 && l > 0) {
        if (cdf_prev > 1.0 - cfm) cdf = cdf_prev; else cdf = cdf * (1.0 - cfm);
      }
    }
    cdf_prev = cdf;
    cfm = cf;
    if (cdf >= 1.0 - cf) bits |= 1ull << (l & 63);
    if ((l & 63) == 63 || l == nlay - 1) { mask[((long)g * nw + (l >> 6)) * ncol + col] = bits; bits = 0; }
  }
}

// The sub-column masks of columns col0 .. col0 + ncol - 1 of a grid of ncol_total columns (ncol_total <= 0: not sharded), all on
// stream s; cldfr and mask are device pointers.  which: 0 shortwave, 1 longwave (work buffers and polynomial caches apart: the two
// may be in flight on different streams).
int mt_mask_device(rrtmg_ctx *ctx, int which, int ncol, int nlay, int nsub, int icld, int seed, const double *cldfr, uint64_t *mask, int nw,
                   int col0, int ncol_total, hipStream_t s) {
  const size_t mask_bytes = (size_t)nsub * nw * ncol * sizeof(uint64_t);
  if (icld == 0) { RRTMG_HIP_CHECK(ctx, hipMemsetAsync(mask, 0, mask_bytes, s)); return RRTMG_OK; }
  const uint64_t per_col = icld == 3 ? 1 : (uint64_t)nlay;
  const uint64_t ncolT = ncol_total > 0 ? (uint64_t)ncol_total : (uint64_t)ncol;
  const uint64_t first = (ncol_total > 0 ? (uint64_t)col0 : 0) * per_col, stride = ncolT * per_col;
  const long count = (long)ncol * (long)per_col;
  // pieces of ~256 K draws (400 regenerations of the window: 0.2 ms for the workgroup that runs them), at most ~1000 segments
  int npiece = (int)((count + 262143) / 262144);
  if (npiece > 1024 / nsub) npiece = 1024 / nsub;
  if (npiece < 1) npiece = 1;
  const long piece = (count + npiece - 1) / npiece;
  const int nseg = nsub * npiece;
  const char *tag = which == 0 ? "sw.w.mt" : "lw.w.mt";
  uint32_t *x = (uint32_t *)ctx->buf(std::string(tag) + "x", (size_t)kMtBase * 4);
  uint32_t *win = (uint32_t *)ctx->buf(std::string(tag) + "win", (size_t)nseg * kMtN * 4);
  uint32_t *lists = (uint32_t *)ctx->buf(std::string(tag) + "lists", (size_t)nseg * kMtListMax * 4);
  int32_t *counts = (int32_t *)ctx->buf(std::string(tag) + "counts", (size_t)nseg * 4);
  // The draws leave the stream kernel through HBM and are read once by the mask kernel: nsub x count words -- 275 MB for the
  // longwave at 8192 x 60, 4.4 GB at 131 072 x 60 -- if all sub-columns were generated before any mask is formed.  The
  // sub-columns are therefore worked off in groups whose draws fit 1 GB (RRTMG_HIP_MT_DRAWS_MB; one group up to ~30 000 columns x 60
  // layers: the launches of a small grid are not cut into pieces that no longer fill the GPU); the buffer is reused by every group.
  size_t budget = (size_t)1 << 30;
  if (const char *env = getenv("RRTMG_HIP_MT_DRAWS_MB")) { const long mb = atol(env); if (mb > 0) budget = (size_t)mb << 20; }   // (tests: many groups on a small grid)
  int gsub = (int)(budget / ((size_t)count * 4));
  gsub = gsub < 1 ? 1 : (gsub > nsub ? nsub : gsub);
  uint32_t *draws = (uint32_t *)ctx->buf(std::string(tag) + "draws", (size_t)gsub * count * 4);
  if (!x || !win || !lists || !counts || !draws) return ctx->status;
  // the jump polynomials depend on where the segments start, not on the seed: uploaded when the grid shape changes
  uint64_t *key = ctx->mt_key[which];
  if (key[0] != first || key[1] != stride || key[2] != (uint64_t)nsub || key[3] != (uint64_t)piece || key[4] != (uint64_t)npiece || ctx->mt_dev[which] != lists) {
    std::vector<uint32_t> hl;
    std::vector<int32_t> hc;
    if (!mt_jump_lists(first, stride, nsub, (uint64_t)piece, npiece, hl, hc)) return ctx->fail(RRTMG_ERR_TABLES, "Mersenne-twister jump polynomials could not be built");
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(lists, hl.data(), hl.size() * 4, hipMemcpyHostToDevice, s));
    RRTMG_HIP_CHECK(ctx, hipMemcpyAsync(counts, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, s));
    RRTMG_HIP_CHECK(ctx, hipStreamSynchronize(s));   // (the host vectors go out of scope)
    key[0] = first; key[1] = stride; key[2] = (uint64_t)nsub; key[3] = (uint64_t)piece; key[4] = (uint64_t)npiece; ctx->mt_dev[which] = lists;
  }
  const bool big_lds = ctx->allow_dynamic_lds(1, (const void *)mt_jump_kernel, kMtJumpLds);
  if (!big_lds) return ctx->fail(RRTMG_ERR_HIP, "mt_jump_kernel: %d bytes of dynamic LDS refused", kMtJumpLds);
  const size_t mask_lds = (size_t)64 * ((icld == 3 ? 1 : nlay) | 1) * 4;   // (256 layers: 64.25 KB, just over what a kernel may have unasked)
  const bool mask_lds_ok = ctx->allow_dynamic_lds(2, (const void *)mt_mask_kernel, 64 * 257 * 4);
  if (mask_lds > 64 * 1024 && !mask_lds_ok) return ctx->fail(RRTMG_ERR_HIP, "mt_mask_kernel: %zu bytes of dynamic LDS refused", mask_lds);
  hipLaunchKernelGGL(mt_seed_kernel, dim3(1), dim3(kMtGenThreads), 0, s, (uint32_t)seed, x);
  hipLaunchKernelGGL(mt_jump_kernel, dim3(nseg), dim3(kMtJumpThreads), (size_t)kMtJumpLds, s, x, lists, counts, win);
  for (int g0 = 0; g0 < nsub; g0 += gsub) {
    const int ng = nsub - g0 < gsub ? nsub - g0 : gsub;
    hipLaunchKernelGGL(mt_stream_kernel, dim3(ng * npiece), dim3(kMtGenThreads), 0, s, win, count, piece, npiece, g0 * npiece, draws);
    hipLaunchKernelGGL(mt_mask_kernel, dim3(ng, (ncol + 63) / 64), dim3(64), mask_lds, s, ncol, nlay, icld, cldfr, draws, mask, nw, g0);
  }
  RRTMG_HIP_CHECK(ctx, hipGetLastError());
  return RRTMG_OK;
}

}  // namespace rrtmg
