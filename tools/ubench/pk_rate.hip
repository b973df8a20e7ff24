// Microbenchmark: v_pk_mul_f32 / v_pk_add_f32 issue rate vs v_mul_f32 / v_add_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void k_pk(float *out, int iters, float seed) {
  f2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const f2 m = {1.0000001f, 1.0000002f}, c = {1e-7f, 2e-7f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
      a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
    }
  }
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
}
int main() {
  float *d; (void)hipMalloc(&d, sizeof(float) * 64 * 65536);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int iters = 2000;
  for (int wps : {1, 2, 4, 8}) {
    int blocks = p.multiProcessorCount * 4 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_pk<<<blocks, 64>>>(d, 10, 1.0f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k_pk<<<blocks, 64>>>(d, iters, 1.0f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * iters * 128.0;
    double per_simd = insts / (ms * 1e-3) / (p.multiProcessorCount * 4);
    printf("packed f32, waves/SIMD %d: %.3f ms, %.3f G packed-instr/s per SIMD -> %.2f cycles per v_pk instr per SIMD\n",
           wps, ms, per_simd / 1e9, (p.clockRate * 1e3) / per_simd);
  }
  return 0;
}
