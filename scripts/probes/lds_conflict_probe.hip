// LDS bank-conflict counter calibration (round 3): what does SQ_LDS_BANK_CONFLICT count for the
// access patterns of k_forward2?  Each kernel issues kReps LDS reads per wave of ONE pattern; run under
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -- ./lds_conflict_probe
// and divide the counters by the launches' instruction counts.
//   build: hipcc --offload-arch=gfx950 -O3 -o lds_conflict_probe lds_conflict_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kReps = 4096;

// stride in BYTES between consecutive lanes, W = access width in bytes; GROUP = lanes that share one address (1: none)
template <int W, int STRIDE, int GROUP>
__global__ __launch_bounds__(64) void k_read(double* out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 160 * 1024 / 8; i += 64) reinterpret_cast<double*>(smem)[i] = (double)i;
  __syncthreads();
  const unsigned base = (unsigned)((lane / GROUP) * STRIDE);
  double acc = 0.0;
  for (int r = 0; r < kReps; ++r) {
    const unsigned a = (base + (unsigned)(r & 7) * 16u * 64u) & (160u * 1024u - 64u);
    if (W == 4) {
      float v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
      acc += (double)v;
    } else if (W == 8) {
      double v;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
      acc += v;
    } else {
      typedef double v2d __attribute__((ext_vector_type(2)));
      v2d v;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
      acc += v.x + v.y;
    }
  }
  out[blockIdx.x * 64 + lane] = acc;
}

template <int W, int STRIDE, int GROUP>
void run(const char* what, double* d_out) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_read<W, STRIDE, GROUP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_read<W, STRIDE, GROUP>), dim3(256), dim3(64), 160 * 1024, nullptr, d_out);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_read<W, STRIDE, GROUP>), dim3(256), dim3(64), 160 * 1024, nullptr, d_out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::printf("%-64s W=%2d stride=%4d group=%2d : %.1f us per launch (%d reads per wave)\n", what, W, STRIDE, GROUP, 1e3 * ms, kReps);
}

int main() {
  double* d_out = nullptr;
  hipMalloc((void**)&d_out, 256 * 64 * sizeof(double));
  run<8, 8, 1>("b64, consecutive lanes (the hand-off slots)", d_out);
  run<8, 16, 1>("b64, every other element (2-way conflict expected)", d_out);
  run<8, 256, 1>("b64, all lanes on one bank pair (32-way conflict expected)", d_out);
  run<8, 17696, 20>("b64, 20 lanes per address, instances 17696 B apart (staged rows)", d_out);
  run<16, 16, 1>("b128, consecutive lanes (staging writes' pattern)", d_out);
  run<16, 17696, 20>("b128, 20 lanes per address, instances 17696 B apart (nominal knot)", d_out);
  run<4, 4, 1>("b32, consecutive lanes", d_out);
  // round 6: can a different distance between the instances' staged blocks remove the counted conflicts of
  // "20 lanes per address, three addresses per wave"?  Distances that put the three addresses on other banks / bank groups:
  run<8, 17696 + 8, 20>("b64, 20 lanes per address, instances 17704 B apart", d_out);
  run<8, 17696 + 32, 20>("b64, 20 lanes per address, instances 17728 B apart", d_out);
  run<8, 17696 + 64, 20>("b64, 20 lanes per address, instances 17760 B apart", d_out);
  run<8, 17696 + 96, 20>("b64, 20 lanes per address, instances 17792 B apart", d_out);
  run<8, 17696 + 128, 20>("b64, 20 lanes per address, instances 17824 B apart", d_out);
  run<8, 17696 + 224, 20>("b64, 20 lanes per address, instances 17920 B apart (multiple of 256)", d_out);
  run<8, 17696, 32>("b64, 32 lanes per address (two addresses per wave, one per half)", d_out);
  run<8, 17696, 64>("b64, all 64 lanes on one address", d_out);
  run<8, 17696, 16>("b64, 16 lanes per address (four addresses per wave)", d_out);
  run<16, 17696 + 64, 20>("b128, 20 lanes per address, instances 17760 B apart", d_out);
  run<16, 17696 + 224, 20>("b128, 20 lanes per address, instances 17920 B apart", d_out);
  run<16, 17696, 32>("b128, 32 lanes per address", d_out);
  run<16, 17696, 64>("b128, all 64 lanes on one address", d_out);
  hipDeviceSynchronize();
  hipFree(d_out);
  return 0;
}
