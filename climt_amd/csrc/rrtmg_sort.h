// rrtmg_sort.h -- OPT-IN internal column order (rrtmg_hip_set_column_sort): cloud-free columns first, cloudy ones behind.
//
// A solve kernel variant is chosen per 64-column TILE (cloud-free / cloudy).  Where cloud-free columns are interleaved with cloudy
// ones more finely than a tile, every tile is cloudy and the cloud-free columns pay for both sky streams.  With the sort a
// device-resident call (memspace 1) runs on an internal copy of its inputs in which the cloud-free columns come first, padded to
// a tile boundary with replicas of the last of them, and the cloudy ones follow (the tail of the last tile: replicas again), so
// that no tile holds both kinds; the outputs are scattered back through the map.  Columns are independent in every routine
// (rrtmg_sw_rad.f90:616, rrtmg_lw_rad.nomcica.f90:453); the kissvec sub-column generator seeds per column from the column's own
// pressures, so the masks are the same wherever a column sits (the Mersenne twister's ONE stream is positional: such calls are
// not sorted).  Why it is not the default: a cloud-free column then runs in the clear-sky variant, whose shortwave differs from
// the cloudy variant's clear-sky stream by ~1e-12 W m^-2 (docs/EXPERIMENTS.md C) -- the default keeps a column's variant a function
// of its tile, so that tile-aligned shards reproduce the whole grid bit for bit.
//
// Everything is on the device and on the call's stream: classification (cldfr > 0 in any layer: what the preparation kernels
// use), a scan of the tile counts, the map, gathers of the inputs [rows][N][elem] -> [rows][N'][elem], the scatter of the
// outputs.  N' = 64 x (tiles + 1): what the host can size without knowing the counts.
#pragma once
#include <string>

#include "rrtmg_ctx.h"

namespace rrtmg {

struct SortHead { int32_t nclear, ncloudy, ncpad, pad; };

// one wavefront per tile: flag[col] = column has a cloud; cnt[tile] = how many of the tile's columns do
static __global__ void __launch_bounds__(64) sort_class_kernel(const double *cldfr, int ncol, int nlay, int32_t *flag, int32_t *cnt) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  bool c = false;
  if (col < ncol)
    for (int l = 0; l < nlay; ++l) c = c || cldfr[(long)l * ncol + col] > 0.0;
  if (col < ncol) flag[col] = c ? 1 : 0;
  const unsigned long long m = __ballot(c);
  if (threadIdx.x == 0) cnt[blockIdx.x] = __popcll(m);
}

// one workgroup: exclusive prefix of the tiles' cloudy counts (base[tile]) and the totals
static __global__ void __launch_bounds__(1024) sort_scan_kernel(const int32_t *cnt, int ntile, int ncol, int32_t *base, SortHead *head) {
  __shared__ int part[1024];
  const int t = threadIdx.x, per = (ntile + 1023) / 1024;
  int s = 0;
  for (int i = t * per; i < ntile && i < (t + 1) * per; ++i) s += cnt[i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = t == 0 ? 0 : part[t - 1];
  for (int i = t * per; i < ntile && i < (t + 1) * per; ++i) { base[i] = run; run += cnt[i]; }
  if (t == 1023) {
    const int ncloudy = part[1023], nclear = ncol - ncloudy;
    head->nclear = nclear; head->ncloudy = ncloudy; head->ncpad = (nclear + 63) / 64 * 64; head->pad = 0;
  }
}

// one wavefront per tile of SOURCE columns: src[slot] = column (stable within each kind), dst[slot] = column
static __global__ void __launch_bounds__(64) sort_map_kernel(const int32_t *flag, const int32_t *base, const SortHead *head, int ncol, int32_t *src, int32_t *dst) {
  const int lane = threadIdx.x, col = blockIdx.x * 64 + lane;
  const bool in = col < ncol, c = in && flag[col] != 0;
  const unsigned long long mc = __ballot(c), mk = __ballot(in && !c), lower = (1ull << lane) - 1ull;
  if (!in) return;
  const int cld_before = base[blockIdx.x], clr_before = blockIdx.x * 64 - cld_before;
  const int slot = c ? head->ncpad + cld_before + __popcll(mc & lower) : clr_before + __popcll(mk & lower);
  src[slot] = col; dst[slot] = col;
}
// the padding: replicas of the last cloud-free column behind the cloud-free block, of the last column of all behind the cloudy block
static __global__ void __launch_bounds__(64) sort_pad_kernel(const SortHead *head, int npad, int32_t *src, int32_t *dst) {
  const int slot = blockIdx.x * 64 + threadIdx.x;
  if (slot >= npad) return;
  const int nclear = head->nclear, ncpad = head->ncpad, end = ncpad + head->ncloudy;
  if (slot >= nclear && slot < ncpad) { src[slot] = src[nclear - 1]; dst[slot] = -1; }   // (nclear > 0 here: ncpad > nclear)
  else if (slot >= end) { src[slot] = head->ncloudy > 0 ? src[end - 1] : src[nclear - 1]; dst[slot] = -1; }
}
// in [rows][ncol][elem] -> out [rows][npad][elem].  elem = 1 (every array but the band-fastest cloud optics and masks): a thread
// owns one slot, reads its source column once and copies kSortRows rows of it (8 loads in flight, no index traffic per row)
constexpr int kSortRows = 8;
static __global__ void __launch_bounds__(256) sort_gather1_kernel(const double *in, double *out, const int32_t *src, int ncol, int npad, int rows) {
  const int slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= npad) return;
  const int c = src[slot], r0 = blockIdx.y * kSortRows;
  double v[kSortRows];
#pragma unroll
  for (int k = 0; k < kSortRows; ++k) if (r0 + k < rows) v[k] = __builtin_nontemporal_load(in + (long)(r0 + k) * ncol + c);
#pragma unroll
  for (int k = 0; k < kSortRows; ++k) if (r0 + k < rows) __builtin_nontemporal_store(v[k], out + (long)(r0 + k) * npad + slot);
}
static __global__ void __launch_bounds__(256) sort_gather_kernel(const double *in, double *out, const int32_t *src, int ncol, int npad, int elem) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)npad * elem) return;
  const int slot = (int)(i / elem), e = (int)(i - (long)slot * elem);
  const long r = blockIdx.y;
  out[(r * npad + slot) * elem + e] = in[(r * ncol + src[slot]) * elem + e];
}
// internal [rows][npad] -> user [rows][ncol]
static __global__ void __launch_bounds__(256) sort_scatter_kernel(const double *in, double *out, const int32_t *dst, int ncol, int npad, int rows) {
  const int slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= npad) return;
  const int col = dst[slot], r0 = blockIdx.y * kSortRows;
  if (col < 0) return;
  double v[kSortRows];
#pragma unroll
  for (int k = 0; k < kSortRows; ++k) if (r0 + k < rows) v[k] = __builtin_nontemporal_load(in + (long)(r0 + k) * npad + slot);
#pragma unroll
  for (int k = 0; k < kSortRows; ++k) if (r0 + k < rows) out[(long)(r0 + k) * ncol + col] = v[k];
}

struct ColumnSort {
  rrtmg_ctx *ctx;
  hipStream_t s;
  int N, L, Np;
  std::string prefix;
  int32_t *src = nullptr, *dst = nullptr;
  bool ok = true;
  ColumnSort(rrtmg_ctx *c, hipStream_t st, int ncol, int nlay, const char *pre) : ctx(c), s(st), N(ncol), L(nlay), Np(((ncol + 63) / 64 + 1) * 64), prefix(pre) {}
  template <class T> T *buf(const char *name, size_t n) {
    T *p = (T *)ctx->buf(prefix + name, n * sizeof(T));
    if (!p) ok = false;
    return p;
  }
  bool prepare(const double *cldfr) {
    const int ntile = (N + 63) / 64;
    int32_t *flag = buf<int32_t>("flag", N), *cnt = buf<int32_t>("cnt", ntile), *base = buf<int32_t>("base", ntile);
    SortHead *head = buf<SortHead>("head", 1);
    src = buf<int32_t>("src", Np); dst = buf<int32_t>("dst", Np);
    if (!ok) return false;
    hipLaunchKernelGGL(sort_class_kernel, dim3(ntile), dim3(64), 0, s, cldfr, N, L, flag, cnt);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, s, cnt, ntile, N, base, head);
    hipLaunchKernelGGL(sort_map_kernel, dim3(ntile), dim3(64), 0, s, flag, base, head, N, src, dst);
    hipLaunchKernelGGL(sort_pad_kernel, dim3(Np / 64), dim3(64), 0, s, head, Np, src, dst);
    return true;
  }
  // nullptr stays nullptr (an absent optional array)
  const double *gather(const char *name, const double *in, size_t rows, int elem = 1) {
    if (!in) return nullptr;
    double *out = buf<double>(name, rows * (size_t)Np * elem);
    if (!out) return nullptr;
    if (elem == 1) hipLaunchKernelGGL(sort_gather1_kernel, dim3((Np + 255) / 256, (unsigned)((rows + kSortRows - 1) / kSortRows)), dim3(256), 0, s, in, out, src, N, Np, (int)rows);
    else hipLaunchKernelGGL(sort_gather_kernel, dim3((unsigned)(((long)Np * elem + 255) / 256), (unsigned)rows), dim3(256), 0, s, in, out, src, N, Np, elem);
    return out;
  }
  double *out(const char *name, size_t rows) { return buf<double>(name, rows * (size_t)Np); }
  void scatter(const double *internal, double *user, size_t rows) {
    if (!internal || !user) return;
    hipLaunchKernelGGL(sort_scatter_kernel, dim3((Np + 255) / 256, (unsigned)((rows + kSortRows - 1) / kSortRows)), dim3(256), 0, s, internal, user, dst, N, Np, (int)rows);
  }
};

}  // namespace rrtmg
