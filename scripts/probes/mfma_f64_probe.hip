// Discovers the lane layout of v_mfma_f64_4x4x4f64 (4 blocks of 4x4x4) empirically.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const double* a, const double* b, double* d) {
  int l = threadIdx.x;
  double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
  d[l] = r;
}
int main() {
  double ha[64], hb[64], hd[64];
  double *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  // test 1: a[l] = one-hot at lane la, b[l] = one-hot at lane lb -> which d lane gets 1?
  printf("nonzero outputs for (la, lb) within block 0 and a cross-block pair:\n");
  for (int la = 0; la < 64; ++la) {
    for (int lb = 0; lb < 64; ++lb) {
      if (!((la < 16 && lb < 16) || (la == 17 && lb >= 16 && lb < 32) || (la == 5 && lb == 21))) continue;
      for (int i = 0; i < 64; ++i) { ha[i] = (i == la); hb[i] = (i == lb); }
      hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
      probe<<<1, 64>>>(da, db, dd);
      hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
      for (int i = 0; i < 64; ++i) if (hd[i] != 0.0) printf("a@%d b@%d -> d@%d = %g\n", la, lb, i, hd[i]);
    }
  }
  return 0;
}
