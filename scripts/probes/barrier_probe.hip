// Cost of the per-knot hand-off of the forward pass: LDS write + s_waitcnt lgkmcnt(0) + s_barrier with
// 1..4 waves per workgroup, and the LDS read latency behind a barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 512
template <int MODE>
__global__ void k(double* out, long long* cyc, int slot) {
  __shared__ double buf[2][5][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double x = 1.0 + lane * 1e-9, acc = 0.0;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < REP; ++i) {
    if (MODE == 0) {  // barrier only
      asm volatile("s_barrier" ::: "memory");
    } else if (MODE == 1) {  // wave 0 writes 5 doubles, everybody: waitcnt + barrier
      if (wave == 0) {
#pragma unroll
        for (int e = 0; e < 5; ++e) buf[i & 1][e][lane] = x + e;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (MODE == 2) {  // + consumers read the 5 doubles after the barrier and use them
      if (wave == 0) {
#pragma unroll
        for (int e = 0; e < 5; ++e) buf[i & 1][e][lane] = x + e;
        x = x * 1.0000001;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (wave != 0) {
#pragma unroll
        for (int e = 0; e < 5; ++e) acc += buf[i & 1][e][lane];
      }
    } else if (MODE == 3) {  // dependent LDS read chain (latency)
      int idx = (int)x & 63;
      x = buf[0][0][idx] + 1.0;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x + acc;
  if (threadIdx.x == 0) cyc[slot] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 256 * 8); hipMallocManaged(&cyc, 64 * 8);
  int s = 0;
  for (int waves = 1; waves <= 4; ++waves) {
    k<0><<<1, 64 * waves>>>(out, cyc, s++);
    k<1><<<1, 64 * waves>>>(out, cyc, s++);
    k<2><<<1, 64 * waves>>>(out, cyc, s++);
  }
  k<3><<<1, 64>>>(out, cyc, s++);
  hipDeviceSynchronize();
  s = 0;
  for (int waves = 1; waves <= 4; ++waves) {
    printf("%d wave(s): barrier %.1f | write+waitcnt+barrier %.1f | +consumer reads %.1f cycles/iteration\n", waves,
           (double)cyc[s] / REP, (double)cyc[s + 1] / REP, (double)cyc[s + 2] / REP);
    s += 3;
  }
  printf("dependent ds_read_b64 chain: %.1f cycles/read\n", (double)cyc[s] / REP);
  return 0;
}
