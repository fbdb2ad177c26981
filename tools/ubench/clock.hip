// s_memtime ticks per s_memrealtime tick (100 MHz constant clock) while a VALU-heavy loop runs:
// the shader clock the step kernel actually sees.  hipcc --offload-arch=gfx950 -O3 -o clock clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, double* sink, int iters) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) a = __builtin_fma(a, b, 1e-9);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  const int blocks = 4096;
  unsigned long long* out; double* sink;
  hipMalloc(&out, blocks * 16); hipMalloc(&sink, blocks * 64 * 8);
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, sink, 20000);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("rep %d: s_memtime %llu ticks, s_memrealtime %llu ticks (100 MHz) -> %.1f MHz\n", rep, h[0], h[1],
           100.0 * (double)h[0] / (double)h[1]);
  }
  return 0;
}
