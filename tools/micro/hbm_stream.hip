// micro-benchmark (development tool): what does HBM sustain on this part for the access patterns of the radiation step?
//   read   : every thread sums 16-byte loads (the flux kernels' pattern)         -> GB/s read
//   copy   : 16-byte loads + 16-byte stores, plain and non-temporal              -> GB/s read + written
//   slab   : every wavefront streams through its OWN region in 1 KB runs, writing it and reading it back later -- the solve
//            kernels' scratch slab (thousands of interleaved streams, writes and reads mixed)
// Buffers of 4 GB: far beyond the 256 MB Infinity Cache.   build: hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_stream.hip -o tools/micro/hbm_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void k_read(const d2 *a, size_t n, double *out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const d2 v = __builtin_nontemporal_load(a + i); s += v.x + v.y; }
  if (s == 1.2345) *out = s;
}
template <bool NT> __global__ void k_copy(const d2 *a, d2 *b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i); else b[i] = a[i];
  }
}
// one wavefront per region of `rows` 1 KB rows: write all rows (NT), then read them back in reverse (NT): the slab pattern
template <bool NT> __global__ void k_slab(d2 *a, int rows, double *out) {
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  d2 *p = a + wave * (size_t)rows * 64 + (threadIdx.x & 63);
  d2 v = {1.0 + threadIdx.x, 2.0};
  for (int r = 0; r < rows; ++r) { if (NT) __builtin_nontemporal_store(v, p + (size_t)r * 64); else p[(size_t)r * 64] = v; v.x += 1.0; }
  double s = 0.0;
  for (int r = rows - 1; r >= 0; --r) { const d2 w = NT ? __builtin_nontemporal_load(p + (size_t)r * 64) : p[(size_t)r * 64]; s += w.x + w.y; }
  if (s == 1.2345) *out = s;
}
__global__ void k_slab_hybrid(d2 *a, int rows, int keep, double *out) {
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  d2 *p = a + wave * (size_t)rows * 64 + (threadIdx.x & 63);
  d2 v = {1.0 + threadIdx.x, 2.0};
  for (int r = 0; r < rows - keep; ++r) { __builtin_nontemporal_store(v, p + (size_t)r * 64); v.x += 1.0; }
  for (int r = rows - keep; r < rows; ++r) { p[(size_t)r * 64] = v; v.x += 1.0; }
  double s = 0.0;
  for (int r = rows - 1; r >= rows - keep; --r) { const d2 w = p[(size_t)r * 64]; s += w.x + w.y; }
  for (int r = rows - keep - 1; r >= 0; --r) { const d2 w = __builtin_nontemporal_load(p + (size_t)r * 64); s += w.x + w.y; }
  if (s == 1.2345) *out = s;
}
// which half of the pair decides whether the Infinity Cache holds a row: the store or the load?
template <bool NTS, bool NTL> __global__ void k_slab2(d2 *a, int rows, double *out) {
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  d2 *p = a + wave * (size_t)rows * 64 + (threadIdx.x & 63);
  d2 v = {1.0 + threadIdx.x, 2.0};
  for (int r = 0; r < rows; ++r) { if (NTS) __builtin_nontemporal_store(v, p + (size_t)r * 64); else p[(size_t)r * 64] = v; v.x += 1.0; }
  double s = 0.0;
  for (int r = rows - 1; r >= 0; --r) { const d2 w = NTL ? __builtin_nontemporal_load(p + (size_t)r * 64) : p[(size_t)r * 64]; s += w.x + w.y; }
  if (s == 1.2345) *out = s;
}
template <class F> static double timeit(F f, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  double best = 1e9;
  for (int k = 0; k < reps; ++k) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
  return best;
}
int main() {
  const size_t bytes = (size_t)4 << 30, n = bytes / 16;
  d2 *a, *b; double *out;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 8);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  for (int blocks : {256 * 4, 256 * 8, 256 * 16}) {
    const double tr = timeit([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, out); });
    const double tc = timeit([&] { hipLaunchKernelGGL(k_copy<false>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    const double tn = timeit([&] { hipLaunchKernelGGL(k_copy<true>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    printf("blocks %5d: read %.0f GB/s | copy %.0f GB/s (read + written) | copy non-temporal %.0f GB/s\n", blocks, bytes / tr / 1e6, 2.0 * bytes / tc / 1e6, 2.0 * bytes / tn / 1e6);
  }
  // slab: 2048 / 4096 / 8192 concurrent wavefront streams, each writing then reading back `rows` KB
  for (int waves : {2048, 4096, 8192, 16384}) {
    const int rows = (int)(bytes / 1024 / waves);
    const double t = timeit([&] { hipLaunchKernelGGL(k_slab<true>, dim3(waves / 4), dim3(256), 0, 0, a, rows, out); });
    printf("slab %5d wavefront streams x %d KB each: %.0f GB/s (written + read back)\n", waves, rows, 2.0 * waves * (double)rows * 1024 / t / 1e6);
  }
  // the same pattern with a LIVE footprint around the 256 MB Infinity Cache: does the most-recently-written-first read-back of the
  // slab hit it?  (2048 wavefronts = the longwave solve's residency, 240 KB = its clear-sky rows per wavefront at 60 layers)
  for (int kb : {60, 120, 240, 480, 960}) {
    const int waves = 2048, rows = kb;
    const double t = timeit([&] { hipLaunchKernelGGL(k_slab<true>, dim3(waves / 4), dim3(256), 0, 0, a, rows, out); }, 9);
    const double u = timeit([&] { hipLaunchKernelGGL(k_slab<false>, dim3(waves / 4), dim3(256), 0, 0, a, rows, out); }, 9);
    printf("slab %5d wavefront streams x %4d KB each = %4.0f MB live: non-temporal %.0f GB/s, plain %.0f GB/s (written + read back)\n", waves, rows, waves * (double)rows / 1024,
           2.0 * waves * (double)rows * 1024 / t / 1e6, 2.0 * waves * (double)rows * 1024 / u / 1e6);
  }
  // hybrid: 2048 streams x 240 KB (480 MB live), the last-written `keep` KB of every stream cacheable, the rest non-temporal
  for (int keep : {0, 30, 60, 100, 120}) {
    const double t = timeit([&] { hipLaunchKernelGGL(k_slab_hybrid, dim3(2048 / 4), dim3(256), 0, 0, a, 240, keep, out); }, 9);
    printf("hybrid 2048 streams x 240 KB, last %3d KB plain (%3.0f MB cacheable): %.0f GB/s\n", keep, 2048.0 * keep / 1024, 2.0 * 2048 * 240.0 * 1024 / t / 1e6);
  }
  {
    const double t1 = timeit([&] { hipLaunchKernelGGL((k_slab2<false, true>), dim3(2048 / 4), dim3(256), 0, 0, a, 120, out); }, 9);
    const double t2 = timeit([&] { hipLaunchKernelGGL((k_slab2<true, false>), dim3(2048 / 4), dim3(256), 0, 0, a, 120, out); }, 9);
    printf("2048 streams x 120 KB (240 MB live): plain stores + non-temporal loads %.0f GB/s | non-temporal stores + plain loads %.0f GB/s\n",
           2.0 * 2048 * 120.0 * 1024 / t1 / 1e6, 2.0 * 2048 * 120.0 * 1024 / t2 / 1e6);
  }
  // memory the L2 does not cache (hipDeviceMallocUncached): does the Infinity Cache still hold it?
  d2 *u = nullptr;
  if (hipExtMallocWithFlags((void **)&u, (size_t)1 << 30, hipDeviceMallocUncached) == hipSuccess) {
    for (int kb : {60, 120, 240, 480}) {
      const double t = timeit([&] { hipLaunchKernelGGL(k_slab<false>, dim3(2048 / 4), dim3(256), 0, 0, u, kb, out); }, 9);
      printf("uncached memory, plain accesses, 2048 streams x %3d KB = %3.0f MB live: %.0f GB/s\n", kb, 2048.0 * kb / 1024, 2.0 * 2048 * (double)kb * 1024 / t / 1e6);
    }
  } else printf("hipExtMallocWithFlags(hipDeviceMallocUncached) failed\n");
  return 0;
}
