// fp64 division / Markstein / sqrt cost in a dependent chain and as independent work (one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int KIND>
__global__ void k(double* out, long long* cyc, double seed) {
  double x = seed + threadIdx.x * 1e-9, y = 1.0000001 + threadIdx.x * 1e-12, z = 0.5, a0 = 1.1, a1 = 1.2, a2 = 1.3, a3 = 1.4;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < REP; ++i) {
    if (KIND == 0) x = y / x;                                   // dependent IEEE divisions
    if (KIND == 1) { a0 = y / a0; a1 = y / a1; a2 = y / a2; a3 = y / a3; }  // 4 independent per iteration
    if (KIND == 2) { double q = x * z; double r = fma(-y, q, x); x = fma(r, z, q) + 1.0; }  // Markstein step chain
    if (KIND == 3) x = sqrt(x) + 1.0;
    if (KIND == 4) { a0 = fma(a0, y, z); a1 = fma(a1, y, z); a2 = fma(a2, y, z); a3 = fma(a3, y, z); }  // 4 indep fma
    if (KIND == 5) { a0 = a0 * y; a1 = a1 * y; a2 = a2 * y; a3 = a3 * y; }  // 4 indep mul
    if (KIND == 6) { a0 = a0 + y; a1 = a1 + y; a2 = a2 + y; a3 = a3 + y; }  // 4 indep add
    if (KIND == 7) { a0 = fmax(a0, y); a1 = fmin(a1, y); a2 = fabs(a2) + 0.0; a3 = -a3; }
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x + a0 + a1 + a2 + a3;
  if (threadIdx.x == 0) cyc[KIND] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMallocManaged(&cyc, 32 * 8);
  k<0><<<1, 64>>>(out, cyc, 1.5); k<1><<<1, 64>>>(out, cyc, 1.5); k<2><<<1, 64>>>(out, cyc, 1.5); k<3><<<1, 64>>>(out, cyc, 1.5);
  k<4><<<1, 64>>>(out, cyc, 1.5); k<5><<<1, 64>>>(out, cyc, 1.5); k<6><<<1, 64>>>(out, cyc, 1.5); k<7><<<1, 64>>>(out, cyc, 1.5);
  hipDeviceSynchronize();
  printf("dependent IEEE fp64 division: %.1f cycles\n", (double)cyc[0] / REP);
  printf("independent IEEE fp64 divisions: %.1f cycles each\n", (double)cyc[1] / REP / 4);
  printf("Markstein quotient step (3 ops + add), dependent: %.1f cycles\n", (double)cyc[2] / REP);
  printf("dependent sqrt(fp64)+add: %.1f cycles\n", (double)cyc[3] / REP);
  printf("independent v_fma_f64: %.1f  v_mul_f64: %.1f  v_add_f64: %.1f  max/min/abs/neg mix: %.1f cycles each\n", (double)cyc[4] / REP / 4,
         (double)cyc[5] / REP / 4, (double)cyc[6] / REP / 4, (double)cyc[7] / REP / 4);
  return 0;
}
