// Does hipGetLastError() still report a failed launch after a later launch succeeded?  (The launchers of the library check
// it once after a group of launches.)  And what does a 1-D grid of 2^32 work-items or more do?
// build + run (GPU box): hipcc --offload-arch=gfx950 -O2 tools/ubench/last_error.hip -o /tmp/last_error && /tmp/last_error
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_touch(unsigned long long *count) { if (threadIdx.x == 0) atomicAdd(count, 1ULL); }
int main() {
  unsigned long long *d = nullptr, h = 0;
  hipMalloc(&d, 8); hipMemset(d, 0, 8);
  hipLaunchKernelGGL(k_touch, dim3(1), dim3(2048), 0, nullptr, d);          // invalid: 2048 threads per workgroup
  hipLaunchKernelGGL(k_touch, dim3(4), dim3(64), 0, nullptr, d);            // valid
  hipError_t e = hipGetLastError();
  printf("after a failed and then a good launch: hipGetLastError = %d (%s)\n", (int)e, hipGetErrorString(e));
  e = hipGetLastError();
  printf("read again: %d\n", (int)e);
  hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); printf("workgroups run: %llu (expected 4)\n", h);
  hipMemset(d, 0, 8);
  hipLaunchKernelGGL(k_touch, dim3((1u << 26) + 5u), dim3(64), 0, nullptr, d);   // 2^32 + 320 work-items
  e = hipGetLastError();
  printf("grid of 2^26 + 5 workgroups of 64: launch error %d (%s)\n", (int)e, hipGetErrorString(e));
  e = hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("sync %d, workgroups run: %llu (asked for %u)\n", (int)e, h, (1u << 26) + 5u);
  hipMemset(d, 0, 8);
  hipLaunchKernelGGL(k_touch, dim3((1u << 26) - 8u), dim3(64), 0, nullptr, d);
  e = hipGetLastError(); hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("grid of 2^26 - 8 workgroups of 64: launch error %d, workgroups run: %llu (asked for %u)\n", (int)e, h, (1u << 26) - 8u);
  hipMemset(d, 0, 8);
  hipLaunchKernelGGL(k_touch, dim3(1u << 10, 1u << 10, 1u << 7), dim3(64), 0, nullptr, d);    // 2^27 workgroups, 2^33 work-items, 3-D
  e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize(); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("3-D grid of 2^27 workgroups of 64: launch error %d, sync %d, workgroups run: %llu (asked for %llu)\n", (int)e, (int)e2, h, 1ULL << 27);
  return 0;
}
