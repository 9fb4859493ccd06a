// Dependent-chain latencies of the instructions on the AL-iLQR critical path (one wave per SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 -o latency_probe latency_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int KIND>
__global__ void k(double* out, long long* cyc, double seed) {
  double x = seed + threadIdx.x * 1e-9, y = 1.0000001, z = 0.5;
  int xi = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
    if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
    if (KIND == 3) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
    if (KIND == 4) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
    if (KIND == 5) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %0, %1, %2\n s_nop 4" : "+v"(x) : "v"(y), "v"(z));
    if (KIND == 6) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(xi));
    if (KIND == 7) asm volatile("v_permlane16_swap_b32 %0, %0" : "+v"(xi));
    if (KIND == 8) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(xi) : "v"((int)(threadIdx.x * 4) ^ 4));
    if (KIND == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(xi) : "v"(xi));
    if (KIND == 10) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(y) : "v"(z), "v"(z)); }
    if (KIND == 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xi) : "v"(xi), "v"(xi));
    if (KIND == 12) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
    if (KIND == 13) asm volatile("v_mov_b64 %0, %0" : "+v"(x));
    if (KIND == 14) { asm volatile("v_cmp_gt_f64 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_f64 %0, %0, %1\n1:\n s_or_b64 exec, exec, s[20:21]" : "+v"(x) : "v"(y) : "vcc", "s20", "s21"); }
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x + y + xi;
  if (threadIdx.x == 0) cyc[KIND] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMallocManaged(&cyc, 32 * 8);
#define RUN(K) k<K><<<1, 64>>>(out, cyc, 1.5); 
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14)
  hipDeviceSynchronize();
  const char* names[] = {"v_fma_f64 dep", "v_mul_f64 dep", "v_add_f64 dep", "v_rsq_f64 dep", "mfma f64 4x4x4 dep on C", "mfma f64 dep on A (+s_nop 4)",
                         "v_mov_b32_dpp dep", "v_permlane16_swap dep", "ds_bpermute+wait dep", "v_cndmask_b32 dep", "2 indep v_fma_f64 chains (per pair)",
                         "v_fma_f32 dep", "v_rcp_f64 dep", "v_mov_b64 dep", "cmp+saveexec+branch+add"};
  for (int i = 0; i < 15; ++i) printf("%-40s %.1f cycles/op\n", names[i], (double)cyc[i] / REP);
  return 0;
}
