// Where do the waves of one workgroup land?  Launches workgroups of 3 (and 4) wavefronts with a large LDS footprint
// (one workgroup per CU, like k_sweep_fused) and prints the SIMD id (HW_REG_HW_ID bits 5:4) each wavefront ran on.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
  extern __shared__ unsigned char smem[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
  if (threadIdx.x == 0) smem[0] = 1;
}
int main() {
  unsigned* d;
  hipMalloc(&d, 512 * 8 * 4);
  for (int waves : {3, 4}) {
    hipMemset(d, 0xff, 512 * 8 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024);
    probe<<<256, waves * 64, 130 * 1024>>>(d);
    unsigned h[512 * 8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < 256; ++b) {
      unsigned mask = 0;
      for (int w = 0; w < waves; ++w) mask |= 1u << ((h[b * 8 + w] >> 4) & 3);
      hist[__builtin_popcount(mask)]++;
    }
    printf("%d waves per workgroup: workgroups whose waves ran on 1 / 2 / 3 / 4 distinct SIMDs: %d / %d / %d / %d\n", waves, hist[1], hist[2],
           hist[3], hist[4]);
    printf("  first workgroups (simd ids): ");
    for (int b = 0; b < 8; ++b) {
      for (int w = 0; w < waves; ++w) printf("%u", (h[b * 8 + w] >> 4) & 3);
      printf(" ");
    }
    printf("\n");
  }
  return 0;
}
