// micro-benchmark (development tool): do two kernels enqueued on ONE stream overlap when the second is launched with
// hipExtAnyOrderLaunch (no barrier bit in its AQL packet)?  Each kernel: 128 workgroups of 256 threads spinning ~1 ms --
// half of an MI355X's 256 CUs.  Serialized: ~2 ms; overlapped: ~1 ms.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/anyorder.hip -o /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long ticks, int *out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}
static double run(hipStream_t s, hipStream_t s2, int mode, long long ticks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEvent_t f, j; hipEventCreateWithFlags(&f, hipEventDisableTiming); hipEventCreateWithFlags(&j, hipEventDisableTiming);
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  if (mode == 0) {          // plain: two launches on one stream
    hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, ticks, (int *)nullptr);
    hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, ticks, (int *)nullptr);
  } else if (mode == 1) {   // second launch any-order
    hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, ticks, (int *)nullptr);
    hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, (int *)nullptr);
  } else {                  // fork / join over a second stream
    hipEventRecord(f, s); hipStreamWaitEvent(s2, f, 0);
    hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, ticks, (int *)nullptr);
    hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s2, ticks, (int *)nullptr);
    hipEventRecord(j, s2); hipStreamWaitEvent(s, j, 0);
  }
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const long long ticks = 100000;   // wall_clock64 runs at 100 MHz: 1 ms
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      double best = 1e9;
      for (int k = 0; k < 5; ++k) { const double t = run(s, s2, mode, ticks); if (t < best) best = t; }
      printf("rep %d mode %d (%s): %.3f ms\n", rep, mode, mode == 0 ? "in order" : mode == 1 ? "any-order" : "fork/join", best);
    }
  // cost of an empty fork / join pair against two empty launches
  for (int mode = 0; mode < 3; ++mode) {
    double best = 1e9;
    for (int k = 0; k < 20; ++k) { const double t = run(s, s2, mode, 0); if (t < best) best = t; }
    printf("empty kernels, mode %d: %.4f ms\n", mode, best);
  }
  return 0;
}
