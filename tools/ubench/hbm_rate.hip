// What the HBM delivers for the step kernel's traffic mix, measured with trivial streaming kernels:
//   read-only, write-only (zeros, 16 B per lane), copy (1:1) and a 62 MB read : 141 MB write mix
// (the fused step's FETCH / WRITE per launch at 64 x 4096), each as one launch of 4096 one-wave
// workgroups (the step's geometry) and as a 1024-thread-block grid-stride kernel.
// hipcc --offload-arch=gfx950 -O3 -o hbm_rate hbm_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ void k_write(v4* dst, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) dst[i] = (v4){0.f, 0.f, 0.f, 0.f};
}
__global__ void k_write_nt(v4* dst, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) __builtin_nontemporal_store((v4){0.f, 0.f, 0.f, 0.f}, &dst[i]);
}
// the step's shape with non-temporal stores
__global__ void k_mix_nt(const v4* src, v4* dst, int nr, int nw, float* sink) {
  const size_t wv = blockIdx.x;
  const v4* s = src + wv * (size_t)nr * 64;
  v4* d = dst + wv * (size_t)nw * 64;
  v4 a = {0, 0, 0, 0};
  for (int i = 0; i < nr; ++i) a += s[(size_t)i * 64 + threadIdx.x];
  a.x = a.x * 0.f;
  for (int i = 0; i < nw; ++i) __builtin_nontemporal_store((v4){a.x, 0.f, 0.f, 0.f}, &d[(size_t)i * 64 + threadIdx.x]);
  if (a.y == 123.456f) sink[0] = a.y;
}
__global__ void k_read(const v4* src, size_t n, float* sink) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  v4 a = {0, 0, 0, 0};
  for (; i < n; i += st) a += src[i];
  if (a.x + a.y + a.z + a.w == 123.456f) sink[0] = a.x;
}
__global__ void k_copy(const v4* src, v4* dst, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) dst[i] = src[i];
}
// each wave: contiguous slab; reads nr chunks then writes nw chunks (the step's shape: load, compute, store)
__global__ void k_mix(const v4* src, v4* dst, int nr, int nw, float* sink) {
  const size_t wv = blockIdx.x;
  const v4* s = src + wv * (size_t)nr * 64;
  v4* d = dst + wv * (size_t)nw * 64;
  v4 a = {0, 0, 0, 0};
  for (int i = 0; i < nr; ++i) a += s[(size_t)i * 64 + threadIdx.x];
  a.x = a.x * 0.f;
  for (int i = 0; i < nw; ++i) d[(size_t)i * 64 + threadIdx.x] = (v4){a.x, 0.f, 0.f, 0.f};
  if (a.y == 123.456f) sink[0] = a.y;
}
template <class F> float timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main(int argc, char** argv) {
  // `hbm_rate warm`: every launch touches the SAME 256 MB slot, so the lines are resident in the 256 MiB Infinity
  // Cache (the regime of back-to-back env steps); default: rotate through 8 slots (nothing is resident)
  const int nslots = (argc > 1 && argv[1][0] == 'w') ? 1 : 8;
  printf("%s\n", nslots == 1 ? "WARM: one slot, Infinity-Cache resident" : "COLD: rotating through 2 GB");
  const size_t MB = 1000000, bytesW = 141 * MB, bytesR = 62 * MB;
  // rotate through 8 slots of 256 MB per buffer (2 GB each): every launch touches memory that
  // left the Infinity Cache (256 MiB) long ago.  Largest access per launch: 141 MB < slot.
  const size_t slot_bytes = 256 * MB, total = 8 * slot_bytes;
  v4 *A, *B; float* sink;
  if (hipMalloc(&A, total) != hipSuccess || hipMalloc(&B, total) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) {
    printf("hipMalloc failed\n"); return 1;
  }
  hipMemset(A, 0, total); hipMemset(B, 0, total);
  int slot = 0;
  auto next = [&](size_t) { slot = (slot + 1) % nslots; return (size_t)slot * (slot_bytes / 16); };
  for (int geom = 0; geom < 2; ++geom) {
    dim3 g = geom ? dim3(4096) : dim3(2048), t = geom ? dim3(64) : dim3(1024);
    float w = timeit([&] { size_t o = next(0); hipLaunchKernelGGL(k_write, g, t, 0, 0, B + o, bytesW / 16); }, 50);
    float r = timeit([&] { size_t o = next(0); hipLaunchKernelGGL(k_read, g, t, 0, 0, A + o, bytesR / 16, sink); }, 50);
    float c = timeit([&] { size_t o = next(0); hipLaunchKernelGGL(k_copy, g, t, 0, 0, A + o, B + o, (size_t)100 * MB / 16); }, 50);
    float wn = timeit([&] { size_t o = next(0); hipLaunchKernelGGL(k_write_nt, g, t, 0, 0, B + o, bytesW / 16); }, 50);
    printf("grid %d x %d: write 141 MB non-temporal %.1f us (%.2f TB/s)\n", g.x, t.x, wn * 1e3, bytesW / wn / 1e9);
    printf("grid %d x %d: write 141 MB %.1f us (%.2f TB/s) | read 62 MB %.1f us (%.2f TB/s) | copy 100+100 MB %.1f us (%.2f TB/s)\n",
           g.x, t.x, w * 1e3, bytesW / w / 1e9, r * 1e3, bytesR / r / 1e9, c * 1e3, 200.0 * MB / c / 1e9);
  }
  // the step's shape: 4096 waves, each reads 236 B/lane (15 chunks) then writes 538 B/lane (34 chunks)
  {
    const int nr = 15, nw = 34;
    float m = timeit([&] { size_t o = next(0); hipLaunchKernelGGL(k_mix, dim3(4096), dim3(64), 0, 0, A + o, B + o, nr, nw, sink); }, 50);
    const double by = 4096.0 * 64 * 16 * (nr + nw);
    printf("mix 4096 waves, read %d then write %d chunks per lane: %.1f MB in %.1f us (%.2f TB/s)\n", nr, nw, by / 1e6, m * 1e3, by / m / 1e9);
    float mn = timeit([&] { size_t o = next(0); hipLaunchKernelGGL(k_mix_nt, dim3(4096), dim3(64), 0, 0, A + o, B + o, nr, nw, sink); }, 50);
    printf("  same with non-temporal stores: %.1f us (%.2f TB/s)\n", mn * 1e3, by / mn / 1e9);
  }
  return 0;
}
