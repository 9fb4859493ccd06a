// Accuracy of v_rsq_f64 and of one / two Newton steps on it (max relative error vs long double on the host).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double y = __builtin_amdgcn_rsq(v);
  r0[i] = y;
  double h = 0.5 * v * y, e = fma(-h, y, 0.5);
  y = fma(y, e, y);
  r1[i] = y;
  h = 0.5 * v * y; e = fma(-h, y, 0.5);
  y = fma(y, e, y);
  r2[i] = y;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  long double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) {
    long double t = 1.0L / sqrtl((long double)x[i]);
    m0 = fmaxl(m0, fabsl(a[i] - t) / t); m1 = fmaxl(m1, fabsl(b[i] - t) / t); m2 = fmaxl(m2, fabsl(c[i] - t) / t);
  }
  printf("max rel err: v_rsq_f64 %.3Le  +1 Newton %.3Le  +2 Newton %.3Le  (eps = 1.11e-16)\n", m0, m1, m2);
  return 0;
}
