// Microbenchmark: sustained wave64 f32 VALU issue rate on gfx950 (independent v_mul/v_add chains, no memory).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PK>
__global__ __launch_bounds__(64) void k(float *out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 1.0000001f, c = 1e-7f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // 8 x (8 mul + 8 add) = 128 VALU per iteration, 8 independent chains
      a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
      a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  float *d; hipMalloc(&d, sizeof(float) * 64 * 65536);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int iters = 2000;
  for (int wps : {1, 2, 4, 8}) {          // waves per SIMD
    int blocks = p.multiProcessorCount * 4 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<0><<<blocks, 64>>>(d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<0><<<blocks, 64>>>(d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * iters * 128.0;
    double per_simd_per_s = insts / (ms * 1e-3) / (p.multiProcessorCount * 4);
    printf("waves/SIMD %d: %.3f ms, %.1f G wave-instr/s chip-wide, %.3f G/s per SIMD (clock %d MHz -> %.2f cycles per VALU instr per SIMD)\n",
           wps, ms, insts / (ms * 1e-3) / 1e9, per_simd_per_s / 1e9, p.clockRate / 1000, (p.clockRate * 1e3) / per_simd_per_s);
  }
  return 0;
}
