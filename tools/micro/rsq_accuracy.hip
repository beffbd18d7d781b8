// Accuracy of the quick square root used in the shortwave two-stream operators (rrtmg_common.h qsqrt):
// v_rsq_f64 + one coupled Goldschmidt iteration + one residual correction, against sqrtl.  Development tool:
//   hipcc --offload-arch=gfx950 -O2 tools/micro/rsq_accuracy.hip -o /tmp/rsq && /tmp/rsq
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n) {
  int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
  const double a = x[i];
  const double y = __builtin_amdgcn_rsq(a);
  r0[i] = a * y;
  double g = a * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  r1[i] = g;
  g = __builtin_fma(__builtin_fma(-g, g, a), h, g);
  r2[i] = g;
}
int main() {
  const int n = 1 << 20; std::vector<double> x(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 61) - 50); }
  double *dx, *d0, *d1, *d2; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0; long exact = 0;
  for (int i = 0; i < n; ++i) {
    long double t = sqrtl((long double)x[i]);
    e0 = fmax(e0, fabs((double)((a[i] - t) / t))); e1 = fmax(e1, fabs((double)((b[i] - t) / t))); e2 = fmax(e2, fabs((double)((c[i] - t) / t)));
    exact += c[i] == sqrt(x[i]);
  }
  printf("max rel err over 2^-50..2^10: x*rsq %.3e, + coupled iteration %.3e, + residual step %.3e (%.2f ulp); equal to sqrt() in %.4f %% of cases\n",
         e0, e1, e2, e2 / 1.11e-16, 100.0 * exact / n);
  return 0;
}
