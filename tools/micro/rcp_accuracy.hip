#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  double b = x[i];
  double r = __builtin_amdgcn_rcp(b);
  r0[i] = r;
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r); r1[i] = r;
  r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r); r2[i] = r;
}
int main() {
  const int n = 1 << 20; std::vector<double> x(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
  double *dx, *d0, *d1, *d2; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) { long double t = 1.0L / (long double)x[i]; e0 = fmax(e0, fabs((double)((a[i] - t) / t))); e1 = fmax(e1, fabs((double)((b[i] - t) / t))); e2 = fmax(e2, fabs((double)((c[i] - t) / t))); }
  printf("max rel err: raw rcp %.3e (%.1f ulp), 1 NR %.3e (%.2f ulp), 2 NR %.3e (%.2f ulp)\n", e0, e0 / 1.11e-16, e1, e1 / 1.11e-16, e2, e2 / 1.11e-16);
  return 0;
}
