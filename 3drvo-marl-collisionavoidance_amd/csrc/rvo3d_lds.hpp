// rvo3d_lds.hpp -- LDS views of a workgroup: the fp64 image (exact stage), the fp32 image (filters), request
// masks and the zero-fill bookkeeping; sizes shared by host and device.
// Part of the gfx950 device code (see rvo3d_device.hpp for the overview).
#pragma once

#include "rvo3d_math.hpp"

namespace rvo3d {

// ---- LDS views ---------------------------------------------------------------
struct Lds {
  double *x, *y, *z, *vx, *vy, *vz, *r, *prio;  // [T] fp64 image (exact stage)
  int* kept;                                     // [T] rows kept by the final sweep
  int* any_reset;                                // [epb]
  int* far;                                      // [epb] a drone is outside the fp32 filter's bound
  // fp32 image, each env's N slots stored twice ([el][2N]) so that neighbour
  // d + k (mod N) is slot d + k; and the exact-stage request masks
  float* w[12];                                  // x y z r [FL] (stored twice); vx vy vz kd ax ay az prio [FS]
  unsigned long long* mask2;                     // [NW][T] (word-major, see mi()) bit j: run pair_eval(me, j)
  int T;
};

// Floats per "doubled" fp32 array (x y z r).  A drone d reaches neighbour d + k, k <= N/2, at
// slot d + k: one env per workgroup (NW > 1) stores its N slots followed by a second copy of
// the first N/2 + 1 only (1.5 N instead of 2 N: at N = 256 that is what lets four workgroups
// share a CU's 160 KiB).  One-wave workgroups (NW == 1: T = 64, epb * N <= 64) keep two full
// copies per env at a fixed array length, so every LDS array sits at a compile-time offset
// from one base (address = base + constant + 4 * index).
__host__ __device__ inline int f32_len_nw(int T, int N, int epb, int NW) {
  (void)T; (void)epb;
  return NW == 1 ? 128 : ((N + (N >> 1) + 1 + 3) & ~3);
}
__host__ __device__ inline int f32_single_nw(int N, int epb, int NW) {
  return NW == 1 ? 64 : ((epb * N + 3) & ~3);
}

__device__ __forceinline__ Lds carve_lds(unsigned char* base, int T, int nm, int epb, int N,
                                         int NW) {
  Lds L;
  double* d = reinterpret_cast<double*>(base);
  L.x = d; L.y = d + T; L.z = d + 2 * T; L.vx = d + 3 * T; L.vy = d + 4 * T; L.vz = d + 5 * T;
  L.r = d + 6 * T; L.prio = d + 7 * T;
  L.mask2 = reinterpret_cast<unsigned long long*>(d + 8 * T);
  L.kept = reinterpret_cast<int*>(L.mask2 + (size_t)T * NW);
  float* wf = reinterpret_cast<float*>(L.kept + T);
  const int FL = f32_len_nw(T, N, epb, NW), FS = f32_single_nw(N, epb, NW);
  // order: WX WY WZ WR doubled, then the single-copy arrays
  L.w[0] = wf; L.w[1] = wf + FL; L.w[2] = wf + 2 * FL; L.w[6] = wf + 3 * FL;
  float* ws = wf + 4 * (size_t)FL;
  L.w[3] = ws; L.w[4] = ws + FS; L.w[5] = ws + 2 * FS; L.w[7] = ws + 3 * FS; L.w[8] = ws + 4 * FS;
  L.w[9] = ws + 5 * FS; L.w[10] = ws + 6 * FS; L.w[11] = ws + 7 * FS;
  L.any_reset = reinterpret_cast<int*>(ws + 8 * (size_t)FS);
  L.far = L.any_reset + epb;
  L.T = T;
  return L;
}
__host__ __device__ inline size_t lds_bytes(int T, int nm, int epb, int N, int NW) {
  (void)nm;
  return (size_t)T * 8 * 8 + (size_t)T * NW * 8 + (size_t)T * 4 +
         (size_t)f32_len_nw(T, N, epb, NW) * 16 + (size_t)f32_single_nw(N, epb, NW) * 32 +
         (size_t)epb * 8 + 16;
}

__device__ __forceinline__ Drone lds_drone(const Lds& L, int k) {
  Drone d;
  d.x = L.x[k]; d.y = L.y[k]; d.z = L.z[k]; d.vx = L.vx[k]; d.vy = L.vy[k]; d.vz = L.vz[k];
  d.r = L.r[k]; d.prio = L.prio[k];
  return d;
}

}  // namespace rvo3d
