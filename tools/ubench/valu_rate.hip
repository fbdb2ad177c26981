// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU
// instructions the step kernel is made of, on gfx950.  8 independent chains per lane,
// 1 / 2 / 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, int iters, double a, double b) {
  double d[8];
  float f[8];
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p[8];
  for (int i = 0; i < 8; ++i) { d[i] = a + i + threadIdx.x; f[i] = (float)d[i]; p[i] = (v2f){f[i], f[i] + 1.f}; }
  float fa = (float)a, fb = (float)b;
  v2f pa = {fa, fa}, pb = {fb, fb};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a), "v"(b));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 1) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(a));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 2) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(a));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 3) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(fa), "v"(fb));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 4) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pa), "v"(pb));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 5) {
#define X(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 6) {
#define X(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[i]));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 7) {
#define X(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 8) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 9) {
#define X(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(d[i]), "v"(a), "v"(f[i]), "v"(fa) : "vcc");
      REP8(X)
#undef X
    } else if (OP == 10) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fa));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 11) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[i]) : "v"(fa));
      REP8(X) REP8(X)
#undef X
    } else if (OP == 12) {  // dependent chain of fma_f64: latency
      asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                   "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                   "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                   "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                   : "+v"(d[0]) : "v"(a), "v"(b));
    } else if (OP == 13) {  // the compiler's a / b (div_scale, rcp, fma chain, div_fmas, div_fixup)
      for (int i = 0; i < 8; ++i) d[i] = a / d[i];
    } else if (OP == 14) {
      for (int i = 0; i < 8; ++i) d[i] = __builtin_sqrt(d[i]);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += d[i] + f[i] + p[i].x + p[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int per_iter) {
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * 4 * wps;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 64);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, cyc, 10, 1.000001, 0.5);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters, 1.000001, 0.5);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double n = (double)iters * per_iter;
    printf("%-22s waves/SIMD %d: %6.2f cycles per instr per wave, %6.2f per SIMD  (kernel %.3f ms => %.2f GHz-cycles/instr/SIMD at 2.4)\n",
           name, wps, mean / n, mean / n / wps, ms, ms * 1e-3 * 2.4e9 / (n * wps));
    hipFree(out); hipFree(cyc);
  }
}

int main() {
  run<0>("v_fma_f64", 16); run<1>("v_mul_f64", 16); run<2>("v_add_f64", 16);
  run<3>("v_fma_f32", 16); run<4>("v_pk_fma_f32", 16); run<10>("v_mul_f32", 16); run<11>("v_add_u32", 16);
  run<5>("v_rcp_f64", 16); run<6>("v_sqrt_f64", 16); run<7>("v_rsq_f64", 16);
  run<8>("v_cvt_f32_f64", 16); run<9>("v_cmp_f64+cndmask", 16);
  run<12>("dep chain v_fma_f64", 16); run<13>("a / b (f64, 8 chains)", 8); run<14>("sqrt(f64) (8 chains)", 8);
  return 0;
}
