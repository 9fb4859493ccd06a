// Verifies the lane layout of v_mfma_f64_16x16x4_f64 assumed by the 16x16 backward pass
//   A[i][k]: lane = i + 16 k          B[k][j]: lane = j + 16 k
//   D[i][j]: lane = j + 16 (i % 4), register i / 4      (cdna_hip_programming.md, "f64 MFMA")
// with an asymmetric product, and measures the dependent-issue cost of the instruction (through C and
// through the B operand) for one wavefront alone on its SIMD.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) d[v * 64 + l] = c[v];
}
__global__ void timing(double* out, long long* cyc, int iters) {
  const int l = threadIdx.x;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  d4 c = {0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);  // chain through C
  long long t1 = __builtin_readcyclecounter();
  double x = b;
  for (int i = 0; i < iters; ++i) {  // chain through B: the result feeds the next product's B operand
    d4 z = {0, 0, 0, 0};
    z = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x, z, 0, 0, 0);
    x = z[0] * 1e-3;
  }
  long long t2 = __builtin_readcyclecounter();
  d4 c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {  // three independent accumulators
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c3, 0, 0, 0);
  }
  long long t3 = __builtin_readcyclecounter();
  out[l] = c[0] + c[3] + x + c2[1] + c3[2];
  if (l == 0) {
    cyc[0] = t1 - t0;
    cyc[1] = t2 - t1;
    cyc[2] = t3 - t2;
  }
}
int main() {
  double ha[64], hb[64], hd[256];
  double A[16][4], B[4][16];
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 4; ++k) A[i][k] = 1.0 + i * 0.5 + k * 7.0;
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 16; ++j) B[k][j] = 2.0 + k * 3.0 - j * 0.25 + (k * j) * 0.01;
  for (int l = 0; l < 64; ++l) {
    ha[l] = A[l & 15][l >> 4];
    hb[l] = B[l >> 4][l & 15];
  }
  double *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 2048);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(da, db, dd);
  hipMemcpy(hd, dd, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double ref = 0;
      for (int k = 0; k < 4; ++k) ref += A[i][k] * B[k][j];
      const double got = hd[(i / 4) * 64 + j + 16 * (i % 4)];
      // (the instruction accumulates k = 0..3 in its own order: compare to a few ulp, the layout errors are O(1))
      if (fabs(got - ref) > 1e-12 * fabs(ref) && ++bad < 8) printf("mismatch D[%d][%d]: got %.17g want %.17g\n", i, j, got, ref);
    }
  printf("mfma_f64_16x16x4 layout (A lane i+16k, B lane j+16k, D lane j+16(i%%4) reg i/4): %s\n", bad ? "WRONG" : "confirmed");
  long long* dc; double* dout; long long hc[3];
  hipMalloc(&dc, 24); hipMalloc(&dout, 512);
  const int iters = 4096;
  timing<<<1, 64>>>(dout, dc, iters);
  timing<<<1, 64>>>(dout, dc, iters);
  hipMemcpy(hc, dc, 24, hipMemcpyDeviceToHost);
  printf("mfma f64 16x16x4 dep on C       %.1f cycles/op\n", (double)hc[0] / iters);
  printf("mfma f64 16x16x4 dep on B (+mul) %.1f cycles/op\n", (double)hc[1] / iters);
  printf("3 independent mfma f64 16x16x4  %.1f cycles per instruction\n", (double)hc[2] / iters / 3);
  return 0;
}
