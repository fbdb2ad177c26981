// rvo3d_device.hpp -- gfx950 device code of the batched 3D-RVO drone step.
//
// One thread = one drone; a workgroup holds EPB whole environments so the
// whole step (RVO reward sweep -> integrate -> observation sweep -> optional
// auto-reset + re-observe -> outputs) is ONE launch with workgroup barriers
// between the phases.  Neighbour position / velocity / radius / priority are
// staged in LDS (64 B per drone).
//
// Cost structure (fp64 VALU is the scarce resource):
//   * stage G (packed fp32, full lane utilisation): a CONSERVATIVE candidate
//     filter over all neighbours - in range (|dp|^2 <= 100 + band), approaching
//     (v.rel > -eps) or nearly touching - builds a 64-bit candidate mask per
//     lane.  The bands bound the fp32 error (host-computed from the map size),
//     so a pair the reference would act on is never dropped;
//   * stage X (fp64, candidates only) repeats every test exactly: squared norms
//     against T(tau) = max{x : sqrt(x) <= tau} (equivalent to the reference's
//     `norm(..) <= tau` for a correctly rounded sqrt), collision, v.rel, then a
//     conservative cone pre-filter (beta >= alpha + 1e-4 rad, no asin/acos/div)
//     and only for the rest asin / acos / the TTC quadratic;
//   * outputs are quantised without fp64 divisions where that is provably exact
//     after the float32 cast, and observations are written once, coalesced.
//
// Decision arithmetic is fp64 and follows the reference's evaluation order
// (compile with -ffp-contract=off; the explicit __builtin_fma calls model the
// OpenBLAS ddot the reference goes through, see DESIGN.md "arithmetic model").
// Reference citations are relative to the reference root.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvo3d {

constexpr double kExpRadius = 0.2;       // rvo_inter.py:11
constexpr double kCtimeThreshold = 2.0;  // rvo_inter.py:11
constexpr double kDeg2Rad = 0.017453292519943295;
constexpr double kPi = 3.141592653589793;
constexpr int kMaxThreads = 512;
// cone pre-filter slack: inside the cone implies beta_raw <= alpha_raw (both
// roundings to 2 decimals considered), so beta_raw >= alpha_raw + kDelta is
// surely outside.  cos/sin of kDelta = 1e-4 rad:
constexpr double kCosD = 0.999999995;              // cos(1e-4) rounded down
constexpr double kSinD = 1.0000000000e-4;          // >= sin(1e-4)

// Parameters that are used in one place each, outside the pair loops.  They live in device
// memory and are read through the constant address space (scalar loads at the point of use)
// instead of riding along in the kernel-argument registers for the whole kernel.
struct Cold {
  double map[3];
  double T5, T04;     // max{x : sqrt(x) <= 5 | 0.4}  (rvo_inter.py:104; drone.py:15)
  double cen[3];      // fp32 candidate filter (stage G): centre,
  double act_scale;   // 10^action_decimals or 0 (no re-quantisation)
  unsigned long long zf_m40;  // ceil(2^40 / zf_q)
  float cmax;         // |centred coordinate| bound the bands were computed for
  float kdot;         // fp32 error bound of v.rel per unit |v|_1
  uint32_t zf_div;    // zero-fill: units (8 B or 4 B) per row of the VO region
  uint32_t zf_magic;  // ceil(2^32 / zf_div)
  uint32_t zf_q;      // 16-B zero-fill (W even): row bytes / 8
  int nb;
  const double* bld;       // [nb][4]
  const double* pow95;     // [P]   0.95 ** k, host libm (ir_gym.py:283)
  // building grid over the map's xy plane (rvo3d_load_world): cell (ix, iy) lists every
  // building whose 5 m gate circle reaches the cell, kBgridK + 1 u16 per cell = count, indices;
  // count 0xffff = more than kBgridK: test all.  bgx == 0: no grid.
  const uint16_t* bgrid;
  int bgx, bgy;
  double bg_inv;           // 1 / cell size
};
constexpr int kBgridK = 15;
typedef const __attribute__((address_space(4))) Cold ColdC;

struct Params {
  int E, N, P, nm, env_train, epb, W;
  int action_f64;     // 1: actions are double
  int dv_cached;      // 1: dvk_a/dvk_b hold des_vel of the current state (skip the pre-move dronestate)
  int g_cached;       // 1: gcache holds the in-range words (stage G) of the current state
  int action_mode;    // 0: absolute action; 1: policy increment (trainer glue, multi_ppo.py:196-205)
  float acceler;      // ir_gym.acceler as numpy sees it next to a float32 array (float32)
  int ablate;         // diagnostics only (env RVO3D_ABLATE): bit k skips phase k, results invalid
  int zf16;           // per call: obs is 16-B aligned and W is even -> 16-B zero-fill
  double T10;         // max{x : sqrt(x) <= 10}  (rvo_inter.py:96)
  // fp32 candidate filter (stage G): error bands
  float t10f;      // T10 + band, rounded up
  float band;      // fp32 error bound of a squared distance at <= 10.5 m
  // fp32 cone pre-filter (stage X1)
  int nw;          // ceil(N / 64) rounded up to a power of two: words per request mask
  float x1_gap;    // below this d2 - R^2 the cone filter is skipped (pair passes)
  float x1_k2;     // slack factor on K^2
  float x1_cs2;    // (cos-space error bound)^2: dp < 0 and dp^2 > cs2*d2*w2 is surely outside
  // All per-drone arrays live in one arena, struct-of-arrays with a common element stride
  // S = EN rounded up to 64 (EN = E*N): array k of a block starts at element k*S.  Three base
  // pointers instead of thirty keep the kernel's scalar registers free of spills.
  //   f64: px py pz vx vy vz yaw pitch real_len max_dev extra_len | cur[3] prev[3] (the
  //        waypoints wp[i], wp[i-1] of the drone's waypoint index i) |
  //        route_len radius prio dev0 | wp [P][3] | row_iet [nm]
  //   i32: wp_idx n_points | dvk_a dvk_b (des_vel of the current state) | dv0_a dv0_b (of the
  //        reset state) | gcache [nw] (stage-G words of the current state) | row_pk [nm]
  //   u8:  arrive dest
  double* f64;
  int32_t* i32;
  uint8_t* u8;
  uint32_t S;
  enum { F_PX, F_PY, F_PZ, F_VX, F_VY, F_VZ, F_YAW, F_PITCH, F_REAL_LEN, F_MAX_DEV, F_EXTRA_LEN,
         F_CUR, F_PREV = F_CUR + 3,
         F_ROUTE_LEN = F_PREV + 3, F_RADIUS, F_PRIO, F_DEV0, F_WP };
  __host__ __device__ double* f(int k) const { return f64 + (size_t)k * S; }
  // mutable state
  __host__ __device__ double* px() const { return f(F_PX); }
  __host__ __device__ double* py() const { return f(F_PY); }
  __host__ __device__ double* pz() const { return f(F_PZ); }
  __host__ __device__ double* vx() const { return f(F_VX); }
  __host__ __device__ double* vy() const { return f(F_VY); }
  __host__ __device__ double* vz() const { return f(F_VZ); }
  __host__ __device__ double* yaw() const { return f(F_YAW); }
  __host__ __device__ double* pitch() const { return f(F_PITCH); }
  __host__ __device__ double* real_len() const { return f(F_REAL_LEN); }
  __host__ __device__ double* max_dev() const { return f(F_MAX_DEV); }
  __host__ __device__ double* extra_len() const { return f(F_EXTRA_LEN); }
  __host__ __device__ int32_t* wp_idx() const { return i32; }
  __host__ __device__ uint8_t* arrive() const { return u8; }
  __host__ __device__ uint8_t* dest() const { return u8 + S; }
  // static world
  __host__ __device__ double* route_len() const { return f(F_ROUTE_LEN); }
  __host__ __device__ double* radius() const { return f(F_RADIUS); }
  __host__ __device__ double* prio() const { return f(F_PRIO); }
  __host__ __device__ double* dev0() const { return f(F_DEV0); }  // deviation in the reset state
  __host__ __device__ double* wp(int k, int c) const { return f(F_WP + 3 * k + c); }  // [P][3]
  // drone.current_des / previous_des (drone.py:24-30, 172-192), kept next to the state so
  // that no load has to wait for the waypoint index
  __host__ __device__ double* cur(int c) const { return f(F_CUR + c); }
  __host__ __device__ double* prev(int c) const { return f(F_PREV + c); }
  __host__ __device__ int32_t* n_points() const { return i32 + S; }
  // des_vel = k / 1000 (drone.py:199-210), packed (dv_encode): of the current state, written by
  // every step / observe (valid unless the state was set from outside: dv_cached), and of the
  // reset state (static)
  __host__ __device__ uint32_t* dvk_a() const { return reinterpret_cast<uint32_t*>(i32) + (size_t)2 * S; }
  __host__ __device__ uint32_t* dvk_b() const { return reinterpret_cast<uint32_t*>(i32) + (size_t)3 * S; }
  __host__ __device__ uint32_t* dv0_a() const { return reinterpret_cast<uint32_t*>(i32) + (size_t)4 * S; }
  __host__ __device__ uint32_t* dv0_b() const { return reinterpret_cast<uint32_t*>(i32) + (size_t)5 * S; }
  // stage-G result of the sweep that ended the last step / observe: word w of drone g has bit b
  // set when neighbour d + 32w + b + 1 is possibly within 10 m.  The next step's sweep A runs
  // on the same state and starts from it (g_cached).
  __host__ __device__ uint32_t* gcache(int w) const { return reinterpret_cast<uint32_t*>(i32) + (size_t)(6 + w) * S; }
  // kept VO rows of the sweep in flight, [nm] arrays (touched only when a pair is flagged)
  __host__ __device__ double* row_iet(int s) const { return f(F_WP + 3 * P + s); }  // 1/(t+0.2)
  __host__ __device__ uint32_t* row_pk(int s) const {                    // (alpha*100) << 16 | j
    return reinterpret_cast<uint32_t*>(i32) + (size_t)(6 + nw + s) * S;
  }
  __host__ __device__ static size_t f64_arrays(int P_, int nm_) { return F_WP + 3 * (size_t)P_ + (nm_ > 0 ? nm_ : 1); }
  __host__ __device__ static size_t i32_arrays(int nm_, int nw_) { return 6 + (size_t)nw_ + (size_t)(nm_ > 0 ? nm_ : 1); }
  uint32_t* err;
  const Cold* cold_;   // device copy of the rarely used parameters
  __device__ __forceinline__ ColdC& cold() const { return *(ColdC*)cold_; }
  unsigned long long* dbg;  // diagnostics: per-workgroup s_memtime stamps [blocks][16], or null
  // per-call I/O
  const void* actions;
  float* obs;
  int32_t* vo_count;
  float* reward;
  uint8_t *done, *info, *finish, *reset_mask;
};

// diagnostic aid: phase stamps of lane 0, only when a stamp buffer is attached
// (rvo3d_debug_stamps; tools/stamps.py)
#define RVO3D_STAMP(i)                                                                  \
  do {                                                                                  \
    if (P.dbg && threadIdx.x == 0) P.dbg[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

// ---- arithmetic primitives -------------------------------------------------
__device__ __forceinline__ double sq(double x) { return x * x; }  // reference: pow(x, 2)
__device__ __forceinline__ double dot3b(double ax, double ay, double az, double bx,
                                        double by, double bz) {
  return __builtin_fma(az, bz, __builtin_fma(ay, by, ax * bx));  // OpenBLAS ddot, n = 3
}
__device__ __forceinline__ double norm3b(double x, double y, double z) {
  return __builtin_sqrt(dot3b(x, y, z, x, y, z));
}
__device__ __forceinline__ double norm2sq(double x, double y) { return __builtin_fma(y, y, x * x); }
__device__ __forceinline__ bool finite_d(double q) { return __builtin_fabs(q) < __builtin_inf(); }

// np.round(x, 2) as the float32 the caller stores: float(rint(x*100)/100).
// float(k * 0.01) == float(k / 100.0) for every integer |k| < 2^24 (k/100 is never
// within 3e-10 relative of a float32 rounding tie, k*0.01 is within 2e-16 of it);
// larger magnitudes take the division.
__device__ __forceinline__ float round2_f32(double x) {
  const double k = __builtin_rint(x * 100.0);
  return __builtin_fabs(k) < 16777216.0 ? (float)(k * 0.01) : (float)(k / 100.0);
}
// Correctly rounded k / 1000 without a division: q = k * RN(1/1000) followed by one
// fma residual correction (Markstein: exact for every finite k; 1000's significand
// is not all ones).  tests/test_numeric_shortcuts.py checks it against k / 1000.0.
__device__ __forceinline__ double k_over_1000(double k) {
  const double q = k * 0.001;
  const double r = __builtin_fma(-q, 1000.0, k);
  const double c = q + r * 0.001;
  return finite_d(k) ? c : k;
}

// des_vel is k / 1000 with integer |k| <= 1000 (np.round(., 3) of a unit vector, drone.py:210)
// or 0: three 11-bit fields (k + 1024) and the signs of zeros.  a == ~0u marks "not of that
// form" (NaN input): the reader recomputes.
constexpr uint32_t kDvInvalid = 0xffffffffu;
__device__ __forceinline__ void dv_encode(const double dv[3], uint32_t& a, uint32_t& b) {
  const double k0 = __builtin_rint(dv[0] * 1000.0), k1 = __builtin_rint(dv[1] * 1000.0),
               k2 = __builtin_rint(dv[2] * 1000.0);
  const bool ok = __builtin_fabs(k0) <= 1023.0 && __builtin_fabs(k1) <= 1023.0 &&
                  __builtin_fabs(k2) <= 1023.0 && k_over_1000(k0) == dv[0] &&
                  k_over_1000(k1) == dv[1] && k_over_1000(k2) == dv[2];
  const uint32_t nz = (uint32_t)(k0 == 0.0 && __builtin_signbit(dv[0])) |
                      ((uint32_t)(k1 == 0.0 && __builtin_signbit(dv[1])) << 1) |
                      ((uint32_t)(k2 == 0.0 && __builtin_signbit(dv[2])) << 2);
  a = ok ? ((uint32_t)((int)k0 + 1024) | ((uint32_t)((int)k1 + 1024) << 16)) : kDvInvalid;
  b = (uint32_t)((int)(ok ? k2 : 0.0) + 1024) | (nz << 16);
}
__device__ __forceinline__ bool dv_decode(uint32_t a, uint32_t b, double dv[3]) {
  if (a == kDvInvalid) return false;
  const int k0 = (int)(a & 0xffffu) - 1024, k1 = (int)(a >> 16) - 1024, k2 = (int)(b & 0xffffu) - 1024;
  dv[0] = (b & (1u << 16)) ? -0.0 : k_over_1000((double)k0);
  dv[1] = (b & (1u << 17)) ? -0.0 : k_over_1000((double)k1);
  dv[2] = (b & (1u << 18)) ? -0.0 : k_over_1000((double)k2);
  return true;
}

// Python round(x, 2): correctly rounded decimal, ties to even (vel_obs3D.py:15).
// Returns the integer c with round(x, 2) == c / 100.0.
__device__ __forceinline__ double py_round2_c(double x) {
  double p = x * 100.0;
  double e = __builtin_fma(x, 100.0, -p);
  double c = __builtin_floor(p);
  double d = (p - (c + 0.5)) + e;
  if (d > 0.0) c += 1.0;
  else if (d == 0.0 && (((long long)c) & 1)) c += 1.0;
  return c;
}
__device__ __forceinline__ double clampd(double x, double lo, double hi) {
  return x < lo ? lo : (x > hi ? hi : x);
}
__device__ __forceinline__ double np_mod(double a, double b) {  // npy_divmod remainder
  double m = fmod(a, b);
  if (m != 0.0) {
    if ((b < 0.0) != (m < 0.0)) m += b;
  } else {
    m = __builtin_copysign(0.0, b);
  }
  return m;
}

// ---- per-drone pieces --------------------------------------------------------
// drone.cal_des_vel (drone.py:199-210, 340-352, 319-328): np.round(dir, 3) with
// dir = [cos az cos el, sin az cos el, sin el].  dir equals dif/|dif| to a few
// ulp, so when dif/|dif|*1000 is further than 1e-7 from a rounding tie the
// rounded integers are the same and no trigonometry is needed; otherwise the
// reference's exact sequence runs.
__device__ __forceinline__ void des_vel(const Params& P, const double p[3], const double cur[3],
                                        double out[3]) {
  const double dx = cur[0] - p[0], dy = cur[1] - p[1], dz = cur[2] - p[2];
  const double d2 = dot3b(dx, dy, dz, dx, dy, dz);
  if (d2 > P.cold().T04) {  // norm > goal_threshold
    const double inv = 1000.0 / __builtin_sqrt(d2);
    const double ux = dx * inv, uy = dy * inv, uz = dz * inv;
    double kx = __builtin_rint(ux), ky = __builtin_rint(uy), kz = __builtin_rint(uz);
    const double m = __builtin_fmin(__builtin_fmin(0.5 - __builtin_fabs(ux - kx),
                                                   0.5 - __builtin_fabs(uy - ky)),
                                    0.5 - __builtin_fabs(uz - kz));
    if (!(m > 1e-7)) {  // near a tie (or NaN): the reference's trig sequence
      const double az = atan2(dy, dx);
      const double el = atan2(dz, __builtin_sqrt(norm2sq(dx, dy)));
      double sa, ca, se, ce;
      sincos(az, &sa, &ca);
      sincos(el, &se, &ce);
      kx = __builtin_rint((1.0 * (ca * ce)) * 1000.0);
      ky = __builtin_rint((1.0 * (sa * ce)) * 1000.0);
      kz = __builtin_rint((1.0 * se) * 1000.0);
    }
    out[0] = k_over_1000(kx); out[1] = k_over_1000(ky); out[2] = k_over_1000(kz);
  } else {
    out[0] = out[1] = out[2] = 0.0;
  }
}

// drone.calculate_deviation (drone.py:366-406)
__device__ __forceinline__ double deviation(const double a[3], const double b[3],
                                            const double p[3]) {
  double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
  double mag = __builtin_sqrt(sq(dx) + sq(dy) + sq(dz));
  if (mag == 0.0) return 0.0;
  double hx = dx / mag, hy = dy / mag, hz = dz / mag;
  double qx0 = p[0] - a[0], qy0 = p[1] - a[1], qz0 = p[2] - a[2];
  double t = qx0 * hx + qy0 * hy + qz0 * hz;
  double qx = a[0] + t * hx, qy = a[1] + t * hy, qz = a[2] + t * hz;
  return __builtin_sqrt(sq(p[0] - qx) + sq(p[1] - qy) + sq(p[2] - qz));
}

__device__ __forceinline__ bool arrived(const Params& P, const double p[3], const double d[3]) {
  const double x = p[0] - d[0], y = p[1] - d[1], z = p[2] - d[2];
  return dot3b(x, y, z, x, y, z) <= P.cold().T04;  // norm <= 0.4, drone.py:172
}

// vel_obs3D.cal_vo_exp_tim (vel_obs3D.py:145-182)
__device__ __forceinline__ double vo_exp_time(double rx, double ry, double rz, double rvx,
                                              double rvy, double rvz, double ra, double rb) {
  double r = ra + rb;
  double ux = -rvx, uy = -rvy, uz = -rvz;
  double a = sq(ux) + sq(uy) + sq(uz);
  double b = 2 * rx * ux + 2 * ry * uy + 2 * rz * uz;
  double c = sq(rx) + sq(ry) + sq(rz) - sq(r);
  if (c <= 0) return 0.0;
  double temp = sq(b) - 4 * a * c;
  if (temp <= 0) return __builtin_inf();
  double s = __builtin_sqrt(temp);
  double t1 = (-b + s) / (2 * a);
  double t2 = (-b - s) / (2 * a);
  if (t1 < 0 && t2 < 0) return -1.0;
  double t3 = t1 >= 0 ? t1 : __builtin_inf();
  double t4 = t2 >= 0 ? t2 : __builtin_inf();
  return t4 < t3 ? t4 : t3;  // python min(t3, t4)
}

struct Drone {  // the 8 values a neighbour contributes (drone.dronestate[0:8])
  double x, y, z, vx, vy, vz, r, prio;
};

struct PairOut {
  bool collision, flag;
  double t, iet, md;
  int alpha_c;  // alpha == alpha_c / 100.0
};

// ---- LDS views ---------------------------------------------------------------
struct Lds {
  double *x, *y, *z, *vx, *vy, *vz, *r, *prio;  // [T] fp64 image (exact stage)
  int* kept;                                     // [T] rows kept by the final sweep
  uint32_t* zc;                                  // [2T] per row: first / end 16-B chunk of its zero run
  int* any_reset;                                // [epb]
  int* far;                                      // [epb] a drone is outside the fp32 filter's bound
  // fp32 image, each env's N slots stored twice ([el][2N]) so that neighbour
  // d + k (mod N) is slot d + k; and the exact-stage request masks
  float* w[12];                                  // x y z r [FL] (stored twice); vx vy vz kd ax ay az prio [FS]
  unsigned long long* mask2;                     // [T][NW] bit j: run pair_eval(me, j)
  int T;
};

// floats per fp32 array: two copies of every env of the workgroup
__host__ __device__ inline int f32_len(int T, int N, int epb) { return (2 * epb * N + 3) & ~3; }
// One-wave workgroups (NW == 1: T = 64, epb * N <= 64) use fixed array lengths, so every LDS
// array sits at a compile-time offset from one base (address = base + constant + 4 * index).
__host__ __device__ inline int f32_len_nw(int T, int N, int epb, int NW) {
  return NW == 1 ? 128 : f32_len(T, N, epb);
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
  L.zc = reinterpret_cast<uint32_t*>(L.kept + T);
  float* wf = reinterpret_cast<float*>(L.zc + 2 * T);
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
  return (size_t)T * 8 * 8 + (size_t)T * NW * 8 + (size_t)T * 12 +
         (size_t)f32_len_nw(T, N, epb, NW) * 16 + (size_t)f32_single_nw(N, epb, NW) * 32 +
         (size_t)epb * 8 + 16;
}

__device__ __forceinline__ Drone lds_drone(const Lds& L, int k) {
  Drone d;
  d.x = L.x[k]; d.y = L.y[k]; d.z = L.z[k]; d.vx = L.vx[k]; d.vy = L.vy[k]; d.vz = L.vz[k];
  d.r = L.r[k]; d.prio = L.prio[k];
  return d;
}

// min_dis of a kept row, recomputed from LDS exactly as pair_eval computed it.
__device__ __forceinline__ double pair_md(const Drone& S, const Drone& O) {
  double rx = O.x - S.x, ry = O.y - S.y, rz = O.z - S.z;
  return __builtin_sqrt(sq(ry) + sq(rx) + sq(rz)) - O.r;
}

// Stage X: the exact fp64 evaluation of one candidate pair = the neighbour gate of
// rvo_inter.preprocess (rvo_inter.py:90-97) followed by rvo_inter.config_vo_circle2
// (rvo_inter.py:116-196) with get_alpha / get_PAA / vo_out_jud_vector / get_beta
// (vel_obs3D.py:8-66, rvo_inter.py:212-228).  `a` is the action after the
// "< 1e-5 -> 0" rule (rvo_inter.py:118).
__device__ __forceinline__ PairOut pair_eval(const Params& P, const Drone& S, const Lds& L, int k,
                                             const double a[3]) {
  PairOut o;
  o.flag = false; o.collision = false; o.t = 0.0; o.iet = 0.0; o.md = 0.0; o.alpha_c = 0;
  const double rx = L.x[k] - S.x, ry = L.y[k] - S.y, rz = L.z[k] - S.z;
  const double d2 = dot3b(rx, ry, rz, rx, ry, rz);  // np.linalg.norm(dif) ** 2 (sign-symmetric)
  // gate: norm <= 10 (rvo_inter.py:96) and not the very same position (rvo_inter.py:92)
  if (!(d2 <= P.T10)) return o;
  if (d2 == 0.0 && rx == 0.0 && ry == 0.0 && rz == 0.0) return o;
  const double Or = L.r[k];
  const double ssum = sq(ry) + sq(rx) + sq(rz);  // dis ** 2 as rvo_inter.py:135 sums it
  const double R = S.r + Or;
  // dis <= thr without the sqrt unless ssum is within 1e-15 (relative) of thr^2
  const double thr = P.env_train ? R : (S.r - kExpRadius + Or);
  const double thr2 = thr * thr;
  bool coll;
  if (ssum < thr2 * (1.0 - 1e-15)) coll = thr >= 0;
  else if (ssum > thr2 * (1.0 + 1e-15)) coll = false;
  else coll = __builtin_sqrt(ssum) <= thr;
  if (coll) { o.collision = true; return o; }
  const double dotp = S.vx * rx + S.vy * ry + S.vz * rz;
  if (dotp <= 0) return o;
  const double Ovx = L.vx[k], Ovy = L.vy[k], Ovz = L.vz[k], Oprio = L.prio[k];
  // get_PAA (vel_obs3D.py:19-32); x / (x + x) == 0.5 exactly
  const double pr = (S.prio == Oprio) ? 0.5 : S.prio / (S.prio + Oprio);
  const double paax = pr * (2 * S.x + (S.vx + Ovx));
  const double paay = pr * (2 * S.y + (S.vy + Ovy));
  const double paaz = pr * (2 * S.z + (S.vz + Ovz));
  const double wx = (S.x + 2 * a[0]) - paax, wy = (S.y + 2 * a[1]) - paay,
               wz = (S.z + 2 * a[2]) - paaz;
  const double dp = dot3b(rx, ry, rz, wx, wy, wz);
  // dp <= 0: cos <= 0 (or AB == 0 -> cos := 0), beta >= pi/2, beta_c >= 157 >= alpha_c: outside
  if (dp <= 0) return o;
  const double w2 = dot3b(wx, wy, wz, wx, wy, wz);
  // Conservative pre-filter.  Inside needs alpha_c >= beta_c + 1, which implies
  // beta_raw <= alpha_raw; so cos(beta) < cos(alpha + 1e-4) is surely outside.
  // |ab| cos(alpha + d) = cos d sqrt(d2 - R^2) - sin d R =: K, cos(beta) = dp / (|ab| |w|).
  // The square root is taken in fp32 (1e-7 relative); the 1e-5 slack on K^2 covers it.
  const double K = kCosD * (double)__builtin_sqrtf((float)(d2 - R * R)) - kSinD * R;
  if (K > 0 && dp * dp < (w2 * (K * K)) * (1.0 - 1e-5)) return o;
  const double nab = __builtin_sqrt(d2);
  const double alpha_c = py_round2_c(asin(R / nab));
  const double AB = nab * __builtin_sqrt(w2);
  const double cosang = (AB != 0) ? dp / AB : 0.0;
  const double beta_c = __builtin_rint(acos(cosang) * 100.0);  // NaN when |cos| > 1 (np.arccos)
  if (!(alpha_c > beta_c)) return o;  // alpha > beta on the rounded values (rvo_inter.py:226)
  const double rvx = 2 * a[0] - Ovx - S.vx, rvy = 2 * a[1] - Ovy - S.vy,
               rvz = 2 * a[2] - Ovz - S.vz;
  const double t = vo_exp_time(rx, ry, rz, rvx, rvy, rvz, S.r, Or);
  if (t < kCtimeThreshold) {
    o.flag = true;
    o.t = t;
    o.iet = 1 / (t + 0.2);
    o.md = __builtin_sqrt(ssum) - Or;
    o.alpha_c = (int)alpha_c;
  }
  return o;
}

// Insert one flagged pair into the kept VO rows of lane `tid` (LDS), keeping the nm
// most urgent in the order of list.sort(reverse=True, key=(-iet, min_dis)) (stable):
// ascending iet, then descending min_dis, then ascending j; slot 0 = least urgent
// kept.  The order is total, so the result does not depend on insertion order.
__device__ __forceinline__ int insert_row(const Params& P, const Lds& L, int g, int lbase,
                                          const Drone& S, const PairOut& po, int j, int kept) {
  const size_t T = P.S;  // slot stride of the row scratch
  double* const iet = P.row_iet(0) + g;
  uint32_t* const pk = P.row_pk(0) + g;
  // position among kept rows: first slot whose row is more urgent than the new one
  int pos = kept;
  for (int s = 0; s < kept; ++s) {
    const double ie = iet[s * T];
    bool new_first;  // new row sorts before slot s
    if (po.iet != ie) new_first = po.iet < ie;
    else {
      const int js = (int)(pk[s * T] & 0xffffu);
      const double mds = pair_md(S, lds_drone(L, lbase + js));
      new_first = (po.md != mds) ? (po.md > mds) : (j < js);
    }
    if (new_first) { pos = s; break; }
  }
  const uint32_t packed = ((uint32_t)po.alpha_c << 16) | (uint32_t)j;
  if (kept < P.nm) {  // grow: shift [pos, kept) up by one
    for (int s = kept; s > pos; --s) {
      iet[s * T] = iet[(s - 1) * T];
      pk[s * T] = pk[(s - 1) * T];
    }
    iet[pos * T] = po.iet;
    pk[pos * T] = packed;
    ++kept;
  } else if (pos > 0) {  // full: drop slot 0 (least urgent), insert at pos-1
    for (int s = 0; s < pos - 1; ++s) {
      iet[s * T] = iet[(s + 1) * T];
      pk[s * T] = pk[(s + 1) * T];
    }
    iet[(pos - 1) * T] = po.iet;
    pk[(pos - 1) * T] = packed;
  }
  return kept;
}

// ===== the pair pipeline: whole envs per workgroup, fp32 filters, exact stage on request =====
enum { WX = 0, WY, WZ, WVX, WVY, WVZ, WR, WKD, WAX, WAY, WAZ, WPRIO };

// fp32 image of one drone, written to both copies of its env segment.
__device__ __forceinline__ void stage_f32(const Params& P, const Lds& L, int el, int d,
                                          bool active, const double p[3], const double v[3],
                                          const double az[3], double r, double prio) {
  if (!active) return;
  const double cx = p[0] - P.cold().cen[0], cy = p[1] - P.cold().cen[1], cz = p[2] - P.cold().cen[2];
  const float fvx = (float)v[0], fvy = (float)v[1], fvz = (float)v[2];
  const float val[12] = {(float)cx, (float)cy, (float)cz, fvx, fvy, fvz, (float)r,
                         P.cold().kdot * (__builtin_fabsf(fvx) + __builtin_fabsf(fvy) + __builtin_fabsf(fvz)) + 1e-30f,
                         (float)az[0], (float)az[1], (float)az[2], (float)prio};
  const int o = el * 2 * P.N + d, os = el * P.N + d;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    if (k == WX || k == WY || k == WZ || k == WR) { L.w[k][o] = val[k]; L.w[k][o + P.N] = val[k]; }
    else L.w[k][os] = val[k];
  }
  const double cm = (double)P.cold().cmax;
  if (!(__builtin_fabs(cx) <= cm && __builtin_fabs(cy) <= cm && __builtin_fabs(cz) <= cm))
    L.far[el] = 1;
}

// Offsets 1..N/2 a drone is responsible for, as NW words of 32 bits (bit b of word w =
// offset 32w + b + 1).  With N even the offset N/2 belongs to the drones d < N/2 only.
template <int NW>
__device__ __forceinline__ void valid_offsets(int N, int d, uint32_t valid[NW]) {
  const int H = N >> 1;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const int n = H - 32 * w;
    valid[w] = n >= 32 ? 0xffffffffu : (n > 0 ? ((1u << n) - 1u) : 0u);
  }
  if (!(N & 1) && d >= H && H > 0) {
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (((H - 1) >> 5) == w) valid[w] &= ~(1u << ((H - 1) & 31));
  }
}

// Stage G for the offsets of word w (packed fp32, two offsets per instruction): squared
// distance to neighbour d + k against the threshold(s).  TOUCHONLY: possibly touching
// (and in range); else: possibly in range.
template <bool TOUCHONLY>
__device__ __forceinline__ uint32_t gate_word(const Params& P, const Lds& L, int o0, int w, int H,
                                              float mex, float mey, float mez, float mer,
                                              uint32_t* range_out = nullptr) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f sx = {mex, mex}, sy = {mey, mey}, sz = {mez, mez}, sr = {mer, mer};
  const int kend = (H - 32 * w) < 32 ? (H - 32 * w) : 32;  // offsets in this word
  uint32_t m = 0u, mr = 0u;
#pragma unroll 8
  for (int b = 0; b < kend; b += 2) {
    const int o = o0 + 32 * w + b + 1;
    const v2f dx = (v2f){L.w[WX][o], L.w[WX][o + 1]} - sx;
    const v2f dy = (v2f){L.w[WY][o], L.w[WY][o + 1]} - sy;
    const v2f dz = (v2f){L.w[WZ][o], L.w[WZ][o + 1]} - sz;
    v2f d2 = dx * dx;
    d2 = __builtin_elementwise_fma(dy, dy, d2);
    d2 = __builtin_elementwise_fma(dz, dz, d2);
    uint32_t b0 = d2.x <= P.t10f, b1 = d2.y <= P.t10f;
    if (TOUCHONLY) {
      if (range_out) mr |= (b0 | (b1 << 1)) << b;  // the in-range word on the side
      const v2f rs = (v2f){L.w[WR][o], L.w[WR][o + 1]} + sr;
      const v2f rc = __builtin_elementwise_fma(rs * rs, (v2f){1.00001f, 1.00001f},
                                               (v2f){P.band, P.band});
      b0 &= (uint32_t)(d2.x <= rc.x);
      b1 &= (uint32_t)(d2.y <= rc.y);
    }
    m |= (b0 | (b1 << 1)) << b;
  }
  if (TOUCHONLY && range_out) *range_out = mr;
  return m;
}

// Symmetric sweep: every unordered pair {i, j} of an env is examined once, by the
// drone whose index d satisfies j = d + k (mod N), 1 <= k <= N/2.
//   stage G  (packed fp32, all offsets): possibly in range;
//   stage X1 (fp32, candidates): possibly approaching / touching and a conservative
//            cone pre-filter, for both directions; survivors request the exact
//            evaluation from the owner (bit masks, LDS atomics);
//   stage X2 (fp64, requested pairs only): pair_eval.
// G and X1 only ever drop pairs that pair_eval would return "nothing" for.
// NW = ceil(N / 64): words per request mask (64 drones) and per offset mask (32 offsets).
template <int NW, bool ROWS, bool TOUCH>
__device__ __forceinline__ int sweep_env(const Params& P, const Lds& L, int lane, int el, int d,
                                         int g, bool active, const Drone& S, const double a[3],
                                         bool zero_act, bool& flag, double& tmin,
                                         bool& collision, uint32_t gw[NW], bool have_gw) {
  flag = false;
  tmin = __builtin_inf();
  int kept = 0;
  const int N = P.N, H = N >> 1;
#pragma unroll
  for (int w = 0; w < NW; ++w) L.mask2[lane * NW + w] = 0ull;
  __syncthreads();
  unsigned long long m2r = 0ull;  // NW == 1: my own requests stay in a register
  if (active) {
    const int o0 = el * 2 * N + d, os0 = el * N + d;
    const float mex = L.w[WX][o0], mey = L.w[WY][o0], mez = L.w[WZ][o0];
    const float mvx = L.w[WVX][os0], mvy = L.w[WVY][os0], mvz = L.w[WVZ][os0];
    const float mer = L.w[WR][o0], mkd = L.w[WKD][os0], mprio = L.w[WPRIO][os0];
    const float max_ = zero_act ? 0.f : L.w[WAX][os0], may = zero_act ? 0.f : L.w[WAY][os0],
                maz = zero_act ? 0.f : L.w[WAZ][os0];
    const bool far = L.far[el] != 0;
    uint32_t valid[NW];
    valid_offsets<NW>(N, d, valid);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      if (ROWS && !TOUCH) RVO3D_STAMP(10);
      // stage G, unless the words of this very state are on file (gw in, have_gw)
      uint32_t cand;
      if (have_gw) cand = gw[w];
      else {
        cand = far ? valid[w] : (gate_word<false>(P, L, o0, w, H, mex, mey, mez, mer) & valid[w]);
        gw[w] = cand;
      }
      if (P.ablate & 64) cand = 0;
      if (ROWS && !TOUCH) RVO3D_STAMP(11);
      // stage X1, two candidate pairs per trip in packed fp32 (a lane with an odd
      // count repeats its last candidate: the requests are idempotent ORs)
      typedef float v2f __attribute__((ext_vector_type(2)));
      const v2f mex2 = {mex, mex}, mey2 = {mey, mey}, mez2 = {mez, mez}, mer2 = {mer, mer};
      const v2f mvx2 = {mvx, mvx}, mvy2 = {mvy, mvy}, mvz2 = {mvz, mvz};
      const v2f tax = {2.f * max_, 2.f * max_}, tay = {2.f * may, 2.f * may},
                taz = {2.f * maz, 2.f * maz};
      const int fr = far ? 1 : 0;
      while (cand) {
        const int kb0 = __builtin_ctz(cand);
        cand &= cand - 1;
        const bool two = cand != 0u;
        const int kb1 = two ? __builtin_ctz(cand) : kb0;
        cand &= cand - 1;
        const int off0 = 32 * w + kb0 + 1, off1 = 32 * w + kb1 + 1;
        const int oa = o0 + off0, ob = o0 + off1;
        int jd0 = d + off0, jd1 = d + off1;
        if (jd0 >= N) jd0 -= N;
        if (jd1 >= N) jd1 -= N;
        const int ja = el * N + jd0, jb = el * N + jd1;  // slots in the single-copy arrays
#define RVO3D_LD2(K, i0, i1) ((v2f){L.w[K][i0], L.w[K][i1]})
        // straight-line fp32; booleans are combined bitwise on purpose (no branches)
        const v2f dx = RVO3D_LD2(WX, oa, ob) - mex2, dy = RVO3D_LD2(WY, oa, ob) - mey2,
                  dz = RVO3D_LD2(WZ, oa, ob) - mez2;
        const v2f jvx = RVO3D_LD2(WVX, ja, jb), jvy = RVO3D_LD2(WVY, ja, jb),
                  jvz = RVO3D_LD2(WVZ, ja, jb);
        const v2f jr = RVO3D_LD2(WR, oa, ob), jkd = RVO3D_LD2(WKD, ja, jb),
                  jprio = RVO3D_LD2(WPRIO, ja, jb);
        const v2f z2 = {0.f, 0.f};
        const v2f ajx = zero_act ? z2 : RVO3D_LD2(WAX, ja, jb),
                  ajy = zero_act ? z2 : RVO3D_LD2(WAY, ja, jb),
                  ajz = zero_act ? z2 : RVO3D_LD2(WAZ, ja, jb);
#undef RVO3D_LD2
        const v2f d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        const v2f rs = jr + mer2;
        const v2f rs2 = rs * rs;
        const v2f tch = __builtin_elementwise_fma(rs2, (v2f){1.00001f, 1.00001f},
                                                  (v2f){P.band, P.band});
        // possibly approaching, each direction (v.rel > -eps)
        const v2f vi = __builtin_elementwise_fma(
            mvz2, dz, __builtin_elementwise_fma(mvy2, dy, mvx2 * dx));
        const v2f vj = __builtin_elementwise_fma(
            jvz, dz, __builtin_elementwise_fma(jvy, dy, jvx * dx));
        // cone pre-filter: |ab| cos(alpha + 2e-3), slack x1_k2 on its square
        const v2f gap = d2 - rs2;  // d^2 - R^2
        // raw v_sqrt_f32 (1 ulp): the filter's slack covers it
        const v2f sq_ = {__builtin_amdgcn_sqrtf(__builtin_fmaxf(gap.x, 0.f)),
                         __builtin_amdgcn_sqrtf(__builtin_fmaxf(gap.y, 0.f))};
        // K <= 0: the second clause below is empty (its bound becomes 0)
        const v2f Kr = (v2f){0.999998f, 0.999998f} * sq_ - (v2f){2.0e-3f, 2.0e-3f} * rs;
        const v2f K = {__builtin_fmaxf(Kr.x, 0.f), __builtin_fmaxf(Kr.y, 0.f)};
        const v2f K2 = K * K * (v2f){P.x1_k2, P.x1_k2};
        const v2f hf = {0.5f, 0.5f};
        const v2f hx = hf * (mvx2 + jvx), hy = hf * (mvy2 + jvy), hz = hf * (mvz2 + jvz);
        // w_i = 2 a_i - (v_i + v_j) / 2  (get_PAA with equal priorities)
        const v2f wix = tax - hx, wiy = tay - hy, wiz = taz - hz;
        const v2f dpi = __builtin_elementwise_fma(
            dz, wiz, __builtin_elementwise_fma(dy, wiy, dx * wix));
        const v2f wi2 = __builtin_elementwise_fma(
            wiz, wiz, __builtin_elementwise_fma(wiy, wiy, wix * wix));
        // seen from j: rel -> -rel, w_j = 2 a_j - (v_i + v_j) / 2
        const v2f two2 = {2.f, 2.f};
        const v2f wjx = two2 * ajx - hx, wjy = two2 * ajy - hy, wjz = two2 * ajz - hz;
        const v2f dpj = -__builtin_elementwise_fma(
            dz, wjz, __builtin_elementwise_fma(dy, wjy, dx * wjx));
        const v2f wj2 = __builtin_elementwise_fma(
            wjz, wjz, __builtin_elementwise_fma(wjy, wjy, wjx * wjx));
        const v2f cs = (v2f){P.x1_cs2, P.x1_cs2} * d2;
        // signed squares: s = dp |dp|.  Surely outside the cone: cos < -cs (s < -cs w2), or
        // 0 <= cos < cos(alpha + delta) with slack (0 <= s < K^2 w2); one bound per sign of s
        // and ONE comparison s < bound (a NaN compares false: the pair is kept).
        const v2f si = dpi * __builtin_elementwise_abs(dpi), sj = dpj * __builtin_elementwise_abs(dpj);
        const v2f ci = -(cs * wi2), cj = -(cs * wj2), ki = wi2 * K2, kj = wj2 * K2;
#define RVO3D_X1_HALF(c, jd, pi, pj)                                                          \
        {                                                                                     \
          const int touch = TOUCH & (int)(d2.c <= tch.c);                                     \
          const int ai = vi.c > -mkd, aj = vj.c < jkd.c;                                      \
          const int filt = (int)(gap.c >= P.x1_gap) & (int)(jprio.c == mprio);                \
          const int oi = si.c < (si.c < 0.f ? ci.c : ki.c);                                   \
          const int oj = sj.c < (sj.c < 0.f ? cj.c : kj.c);                                   \
          pi = (fr | touch | (ai & ~(filt & oi))) & 1;                                        \
          pj = (fr | touch | (aj & ~(filt & oj))) & 1;                                        \
        }
        bool pi0, pj0, pi1, pj1;
        RVO3D_X1_HALF(x, jd0, pi0, pj0)
        RVO3D_X1_HALF(y, jd1, pi1, pj1)
#undef RVO3D_X1_HALF
        pi1 &= two; pj1 &= two;
        if (NW == 1) {
          m2r |= ((unsigned long long)pi0 << jd0) | ((unsigned long long)pi1 << jd1);
        } else {
          if (pi0) atomicOr(&L.mask2[lane * NW + (jd0 >> 6)], 1ull << (jd0 & 63));
          if (pi1) atomicOr(&L.mask2[lane * NW + (jd1 >> 6)], 1ull << (jd1 & 63));
        }
        if (pj0) atomicOr(&L.mask2[(el * N + jd0) * NW + (d >> 6)], 1ull << (d & 63));
        if (pj1) atomicOr(&L.mask2[(el * N + jd1) * NW + (d >> 6)], 1ull << (d & 63));
      }
    }
  }
  if (ROWS && !TOUCH) RVO3D_STAMP(12);
  if (!ROWS) RVO3D_STAMP(14);
  __syncthreads();
  if (active && !(P.ablate & 32)) {
    const int lbase = el * N;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      unsigned long long m2 = L.mask2[lane * NW + w] | (w == 0 ? m2r : 0ull);
      while (m2) {  // stage X2: exact, requested pairs only
        const int j = 64 * w + __builtin_ctzll(m2);
        m2 &= m2 - 1;
        const PairOut po = pair_eval(P, S, L, lbase + j, a);
        if (TOUCH && po.collision) collision = true;
        if (po.flag) {
          flag = true;
          if (po.t < tmin) tmin = po.t;
          if (ROWS && P.nm > 0) kept = insert_row(P, L, g, lbase, S, po, j, kept);
        }
      }
    }
  }
  if (ROWS && !TOUCH) RVO3D_STAMP(13);
  if (!ROWS) RVO3D_STAMP(15);
  return kept;
}

// Collision-only sweep: exactly the collision_flag part of rvo_inter.config_vo_inf
// (rvo_inter.py:40-48) - a neighbour inside the 10 m gate, not at the very same
// position, with dis <= r + mr (env_train) - for every drone of the env.  Each
// unordered pair is tested once; the fp32 stage only selects pairs that are possibly
// touching, the decision itself is fp64.
template <int NW>
__device__ __forceinline__ bool collide_env(const Params& P, const Lds& L, int lane, int el,
                                            int d, bool active, const Drone& S,
                                            uint32_t gw[NW]) {
  const int N = P.N, H = N >> 1;
  L.mask2[lane * NW] = 0ull;
  __syncthreads();
  bool coll = false;
  if (active) {
    const int o0 = el * 2 * N + d;
    const float mex = L.w[WX][o0], mey = L.w[WY][o0], mez = L.w[WZ][o0], mer = L.w[WR][o0];
    const bool far = L.far[el] != 0;
    uint32_t valid[NW];
    valid_offsets<NW>(N, d, valid);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      // one pass over the offsets: possibly touching (cand) and possibly in range (gw, the
      // stage-G words of this post-move state for the rows sweep that follows)
      uint32_t cand = valid[w];
      gw[w] = valid[w];
      if (!far) {
        uint32_t rng;
        cand = gate_word<true>(P, L, o0, w, H, mex, mey, mez, mer, &rng) & valid[w];
        gw[w] = rng & valid[w];
      }
      while (cand) {  // exact decision, both drones of the pair
        const int kb = __builtin_ctz(cand);
        cand &= cand - 1;
        int jd = d + 32 * w + kb + 1;
        if (jd >= N) jd -= N;
        const int k = el * N + jd;
        const double rx = L.x[k] - S.x, ry = L.y[k] - S.y, rz = L.z[k] - S.z;
        const double d2 = dot3b(rx, ry, rz, rx, ry, rz);
        if (!(d2 <= P.T10)) continue;
        if (d2 == 0.0 && rx == 0.0 && ry == 0.0 && rz == 0.0) continue;
        const double Or = L.r[k];
        const double dis = __builtin_sqrt(sq(ry) + sq(rx) + sq(rz));
        bool ci, cj;
        if (P.env_train) {
          ci = cj = dis <= S.r + Or;
        } else {  // rvo_inter.py:145-147: r - exp_radius + mr, evaluated from each side
          ci = dis <= S.r - kExpRadius + Or;
          cj = dis <= Or - kExpRadius + S.r;
        }
        if (ci) coll = true;
        if (cj) atomicOr(&L.mask2[k * NW], 1ull);
      }
    }
  }
  __syncthreads();
  if (active && (L.mask2[lane * NW] & 1ull)) coll = true;
  return coll;
}

// One-wave workgroups: bring the stage-G word of every drone up to date after the drones in
// `reset_lanes` (ballot) moved to their start positions (fp32 image already restaged).  Only
// pairs with a reset drone change: each reset drone r is tested against all lanes of its env
// at once (same arithmetic as gate_word, so the bits equal a full recomputation); the pair's
// owner - the end whose offset to the other is <= N/2 - takes the bit, and r itself rebuilds
// its word from the ballot rotated to its own offset order.
__device__ __forceinline__ uint32_t regate_resets(const Params& P, const Lds& L, int tid, int el,
                                                  int d, bool active,
                                                  unsigned long long reset_lanes, uint32_t gw) {
  const int N = P.N, H = N >> 1;
  uint32_t valid[1];
  valid_offsets<1>(N, d, valid);
  const int o0 = el * 2 * N + d;
  const float mex = L.w[WX][o0], mey = L.w[WY][o0], mez = L.w[WZ][o0];
  while (reset_lanes) {
    const int rl = __builtin_ctzll(reset_lanes);
    reset_lanes &= reset_lanes - 1;
    const int rel = __builtin_amdgcn_readlane(el, rl), rd = __builtin_amdgcn_readlane(d, rl);
    const int orr = rel * 2 * N + rd;
    const float dx = L.w[WX][orr] - mex, dy = L.w[WY][orr] - mey, dz = L.w[WZ][orr] - mez;
    float d2 = dx * dx;
    d2 = __builtin_fmaf(dy, dy, d2);
    d2 = __builtin_fmaf(dz, dz, d2);
    const bool same = active && el == rel && tid != rl;
    const bool inr = same && d2 <= P.t10f;
    int k = rd - d;  // offset from me to r
    if (k < 0) k += N;
    if (same && k >= 1 && k <= H) {  // I own the pair (if the offset is mine at all)
      const uint32_t bit = 1u << (k - 1);
      gw = (gw & ~bit) | ((inr ? bit : 0u) & valid[0]);
    }
    const unsigned long long bal = __ballot(inr);
    if (tid == rl) {
      unsigned long long seg = bal >> (rel * N), rot;
      if (N == 64) {
        const int sh = (rd + 1) & 63;
        rot = sh ? ((seg >> sh) | (seg << (64 - sh))) : seg;
      } else {
        seg &= (1ull << N) - 1ull;
        const int sh = rd + 1;  // 1..N
        rot = ((seg >> sh) | (seg << (N - sh))) & ((1ull << N) - 1ull);
      }
      gw = (uint32_t)rot & valid[0];
    }
  }
  return gw;
}

// building gate + check_col_with_budilding (rvo_inter.py:99-105, 198-209)
__device__ __forceinline__ bool building_test(const double* const bld, int b, const Drone& S,
                                              double T5) {
  const double bx = bld[4 * b], by = bld[4 * b + 1], bh = bld[4 * b + 2], br = bld[4 * b + 3];
  const double ex = S.x - bx, ey = S.y - by;
  // h > z - 2 and norm <= 5 (gate), z <= h, then dis <= r + br on the few that pass
  if ((bh > S.z - 2) & (norm2sq(ex, ey) <= T5) & (S.z <= bh))
    return __builtin_sqrt(sq(ex) + sq(ey)) <= S.r + br;
  return false;
}
__device__ __forceinline__ bool building_hit(const Params& P, const Drone& S) {
  const int nb = P.cold().nb;
  if (nb == 0) return false;
  bool hit = false;
  const double* const bld = P.cold().bld;
  const double T5 = P.cold().T5;
  const int gx = P.cold().bgx;
  if (gx > 0) {
    // only the buildings listed for the drone's cell can pass the 5 m gate (the lists are
    // conservative; a drone outside the map collides anyway and NaN passes no test)
    const int gy = P.cold().bgy;
    const double inv = P.cold().bg_inv;
    int ix = (int)__builtin_floor(S.x * inv), iy = (int)__builtin_floor(S.y * inv);
    ix = ix < 0 ? 0 : (ix > gx - 1 ? gx - 1 : ix);
    iy = iy < 0 ? 0 : (iy > gy - 1 ? gy - 1 : iy);
    const uint16_t* const cell = P.cold().bgrid + (size_t)(ix * gy + iy) * (kBgridK + 1);
    const int cnt = cell[0];
    if (cnt != 0xffff) {
      for (int k = 0; k < cnt; ++k) hit |= building_test(bld, cell[1 + k], S, T5);
      return hit;
    }
  }
#pragma unroll 4
  for (int b = 0; b < nb; ++b) hit |= building_test(bld, b, S, T5);
  return hit;
}

// Proprioceptive part of one observation row: np.round of [state, vel, radius,
// priority, des_vel, deviation] (ir_gym.py:208-229 / :353-355), 12 floats.  The last four
// arrive already rounded (`tail`, made by proprio_tail before the final sweep so that the
// fp64 values are dead across it).
struct ProprioTail { float dv0, dv1, dv2, dev; bool bad; };
__device__ __forceinline__ ProprioTail proprio_tail(const double dv[3], double dev) {
  ProprioTail t;
  t.dv0 = round2_f32(dv[0]); t.dv1 = round2_f32(dv[1]); t.dv2 = round2_f32(dv[2]);
  t.dev = round2_f32(dev);
  t.bad = !(finite_d(dv[0]) && finite_d(dv[1]) && finite_d(dv[2]) && finite_d(dev));
  return t;
}
__device__ __forceinline__ void write_proprio(const Params& P, int g, const Drone& S,
                                              const ProprioTail& t) {
  float* o = P.obs + (size_t)g * P.W;
  const double v[8] = {S.x, S.y, S.z, S.vx, S.vy, S.vz, S.r, S.prio};
  float f[12];
  bool bad = t.bad;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    f[k] = round2_f32(v[k]);
    bad |= !finite_d(v[k]);
  }
  f[8] = t.dv0; f[9] = t.dv1; f[10] = t.dv2; f[11] = t.dev;
  if ((P.W & 1) == 0) {  // rows are 8-B aligned
    float2* o2 = reinterpret_cast<float2*>(o);
#pragma unroll
    for (int k = 0; k < 6; ++k) o2[k] = make_float2(f[2 * k], f[2 * k + 1]);
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k) o[k] = f[k];
  }
  if (bad) atomicOr(P.err, 1u);
}

// The kept VO rows of one observation row (np.round(., 2) of [PAA, rel, alpha, min_dis,
// iet] per row, ascending urgency), its vo_count, and the bookkeeping of the zero run
// behind them (written by zero_fill()).
__device__ __forceinline__ void write_vo_rows(const Params& P, const Lds& L, int tid, int lbase,
                                              int g, const Drone& S, int kept) {
  float* o = P.obs + (size_t)g * P.W;
  bool bad = false;
  for (int s = 0; s < kept; ++s) {
    const uint32_t pk = P.row_pk(s)[g];
    const int j = (int)(pk & 0xffffu);
    const Drone O = lds_drone(L, lbase + j);
    const double pr = (S.prio == O.prio) ? 0.5 : S.prio / (S.prio + O.prio);
    double row[9];
    row[0] = pr * (2 * S.x + (S.vx + O.vx));  // get_PAA, vel_obs3D.py:19-32
    row[1] = pr * (2 * S.y + (S.vy + O.vy));
    row[2] = pr * (2 * S.z + (S.vz + O.vz));
    row[3] = O.x - S.x; row[4] = O.y - S.y; row[5] = O.z - S.z;
    row[6] = (double)(pk >> 16) / 100.0;
    row[7] = pair_md(S, O);
    row[8] = P.row_iet(s)[g];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      bad |= !finite_d(row[k]);
      o[12 + 9 * s + k] = round2_f32(row[k]);
    }
  }
  // with 8-B zero-fill units an odd 9*kept leaves one float for this lane
  if ((P.W & 1) == 0 && ((9 * kept) & 1) && kept < P.nm) o[12 + 9 * kept] = 0.0f;
  if (P.zf16) {
    // 16-B zero-fill: this lane writes the 8-B pieces that do not fill a 16-B chunk at
    // either end of its zero run and publishes the run as a chunk range [c0, c1)
    const unsigned long long rb = 4ull * (unsigned)P.W;
    const unsigned long long row_b = rb * (unsigned long long)g;
    unsigned long long zs = row_b + 4ull * (unsigned)((12 + 9 * kept + 1) & ~1);
    unsigned long long ze = row_b + rb;
    char* ob = reinterpret_cast<char*>(P.obs);
    if (zs < ze && (zs & 8)) { *reinterpret_cast<float2*>(ob + zs) = make_float2(0.f, 0.f); zs += 8; }
    if (zs < ze && (ze & 8)) { ze -= 8; *reinterpret_cast<float2*>(ob + ze) = make_float2(0.f, 0.f); }
    if (zs > ze) zs = ze;
    L.zc[2 * tid] = (uint32_t)(zs >> 4);
    L.zc[2 * tid + 1] = (uint32_t)(ze >> 4);
  }
  P.vo_count[g] = kept;
  if (bad) atomicOr(P.err, 1u);
}

// Cooperative, coalesced zero padding of the VO region of every row of this
// workgroup: rows [row0, row0 + nrows) are contiguous in memory; L.kept holds
// the kept count per row.  Unit = float2 when W is even (rows 8-B aligned),
// float otherwise.
__device__ __forceinline__ void zero_fill(const Params& P, const Lds& L, int tid, int row0,
                                          int nrows) {
  if (P.zf16) {
    // rows [row0, row0 + nrows) occupy bytes [rb*row0, rb*(row0+nrows)); every 16-B chunk
    // that starts inside a row's published zero run is stored, fully coalesced
    const unsigned long long rb = 4ull * (unsigned)P.W;
    const uint32_t cbeg = (uint32_t)((rb * (unsigned)row0 + 15) >> 4);
    const uint32_t cend = (uint32_t)((rb * (unsigned)(row0 + nrows)) >> 4);
    float4* ob = reinterpret_cast<float4*>(P.obs);
    const uint2* zc2 = reinterpret_cast<const uint2*>(L.zc);
    const unsigned long long m40 = P.cold().zf_m40;
    // four chunks per trip: the run lookups (one 8-B LDS read each) are issued together
    // and nothing in the body branches, so a trip costs one LDS round trip, not eight
    for (uint32_t c = cbeg + tid; c < cend; c += 4 * L.T) {
      uint2 z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t cu = c + u * L.T;
        const uint32_t cc = cu < cend ? cu : c;  // clamp: the lookup stays inside this block
        const uint32_t grow = (uint32_t)(((unsigned long long)(2u * cc) * m40) >> 40);
        z[u] = zc2[grow - (uint32_t)row0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t cu = c + u * L.T;
        if ((cu < cend) & (cu >= z[u].x) & (cu < z[u].y)) ob[cu] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    return;
  }
  const uint32_t per_row = P.cold().zf_div;
  if (per_row == 0) return;
  const uint32_t total = (uint32_t)nrows * per_row;
  float* base = P.obs + (size_t)row0 * P.W + 12;
  if ((P.W & 1) == 0) {
    for (uint32_t q = tid; q < total; q += L.T) {
      const uint32_t row = (uint32_t)(((uint64_t)q * P.cold().zf_magic) >> 32);
      const uint32_t c = q - row * per_row;       // float2 index inside the VO region
      const uint32_t first = (9u * (uint32_t)L.kept[row] + 1u) >> 1;  // first all-zero unit
      if (c >= first)
        *reinterpret_cast<float2*>(base + (size_t)row * P.W + 2 * c) = make_float2(0.f, 0.f);
    }
  } else {
    for (uint32_t q = tid; q < total; q += L.T) {
      const uint32_t row = (uint32_t)(((uint64_t)q * P.cold().zf_magic) >> 32);
      const uint32_t c = q - row * per_row;
      if (c >= 9u * (uint32_t)L.kept[row]) base[(size_t)row * P.W + c] = 0.0f;
    }
  }
}

__device__ __forceinline__ void load_wp(const Params& P, int g, int k, double out[3]) {
  out[0] = P.wp(k, 0)[g];
  out[1] = P.wp(k, 1)[g];
  out[2] = P.wp(k, 2)[g];
}

__device__ __forceinline__ void load3(double* const a0, double* const a1, double* const a2, int g,
                                      double out[3]) {
  out[0] = a0[g]; out[1] = a1[g]; out[2] = a2[g];
}
__device__ __forceinline__ void store3(double* const a0, double* const a1, double* const a2, int g,
                                       const double v[3]) {
  a0[g] = v[0]; a1[g] = v[1]; a2[g] = v[2];
}
#define RVO3D_LOAD_CUR(P, g, out) load3((P).cur(0), (P).cur(1), (P).cur(2), g, out)
#define RVO3D_LOAD_PREV(P, g, out) load3((P).prev(0), (P).prev(1), (P).prev(2), g, out)
#define RVO3D_STORE_CUR(P, g, v) store3((P).cur(0), (P).cur(1), (P).cur(2), g, v)
#define RVO3D_STORE_PREV(P, g, v) store3((P).prev(0), (P).prev(1), (P).prev(2), g, v)

// ir_gym.rvo_reward_cal (ir_gym.py:64-133), the part that does not depend on the sweep:
// angle_punish + vel_penalty.  The sweep's safety term is added afterwards in the
// reference's order, (punish + vel_penalty) + safety: rvo_reward_k().
__device__ __forceinline__ double rvo_reward_pre(const double dv[3], const double a[3]) {
  // des_vel is already a 3-decimal value: np.round(., 3) again is the identity
  const double d0 = dv[0], d1 = dv[1], d2 = dv[2];
  const double vel_penalty = 0.2 * norm3b(a[0], a[1], a[2]) / norm3b(d0, d1, d2);
  const double eps = 1e-8;
  const double magA = __builtin_sqrt(sq(d0) + sq(d1) + sq(d2) + eps);
  const double magB = __builtin_sqrt(sq(a[0]) + sq(a[1]) + sq(a[2]) + eps);
  const double dotp = d0 * a[0] + d1 * a[1] + d2 * a[2];
  double c = dotp / (magA * magB);  // magA, magB >= 1e-4: the `< 1e-6` branch is dead
  c = c < -1.0 + eps ? -1.0 + eps : (c > 1.0 - eps ? 1.0 - eps : c);
  // angle bins (ir_gym.py:91-100) on ang = acos(c): compare c with the cosines of
  // the bin edges; acos itself only when c is within 1e-12 of an edge.
  const double C18 = 0.984807753012208, C6 = 0.8660254037844387, C3 = 0.5000000000000001,
               C2 = 6.123233995736766e-17;
  double punish;
  if (c == 0.0) punish = -4;  // acos(0) == pi/2 exactly: not < pi/2
  else if (__builtin_fabs(c - C18) > 1e-12 && __builtin_fabs(c - C6) > 1e-12 &&
           __builtin_fabs(c - C3) > 1e-12 && __builtin_fabs(c - C2) > 1e-12) {
    punish = c > C18 ? 3 : (c > C6 ? 1 : (c > C3 ? 0.5 : (c > C2 ? 0 : -4)));
    if (c != c) punish = -4;
  } else {
    const double ang = acos(c);
    if (ang < kPi / 18) punish = 3;
    else if (ang < kPi / 6) punish = 1;
    else if (ang < kPi / 3) punish = 0.5;
    else if (ang < kPi / 2) punish = 0;
    else punish = -4;
  }
  return punish + vel_penalty;
}
// Returns the integer k with np.round(total, 3) == k / 1000 (or inf / nan, survey Q9).
__device__ __forceinline__ double rvo_reward_k(double pre, bool flag, double tmin) {
  double safety = 0;
  if (flag) {
    double urgency = 0;
    if (tmin < 2) urgency = -8.0 * exp(-tmin / 0.5);
    safety = -2.5 + urgency;
  }
  return __builtin_rint((pre + safety) * 1000.0);
}

// ir_gym.mov_reward (ir_gym.py:256-311); returns k with round(., 3) == k / 1000
__device__ __forceinline__ double mov_reward_k(const Params& P, bool collision, bool arrive_r,
                                               int waypoint_num, int n_points_m1, bool dest_r,
                                               double dev, bool len_flag, double exlen) {
  if (collision) return -50000.0;  // -50
  double reward = 0;
  if (arrive_r) reward += 3.0 * P.cold().pow95[n_points_m1 - waypoint_num];
  if (dest_r) reward += 20.0;
  const double d = dev * 10;
  const double dev_pen = -1.5 * (2 / (1 + exp(-(d - 5) / 0.3)));
  double ex_pen = 0;
  if (len_flag) {
    ex_pen = -0.3 * log(exlen + 1 + 1e-6);
    if (ex_pen < -6 || ex_pen != ex_pen) ex_pen = -6;
  }
  return __builtin_rint((reward + dev_pen + ex_pen) * 1000.0);
}

// mdin.py:28 adds two np.round(., 3) values in fp64; k / 1000 is formed exactly
// (k_over_1000) so the sum, cancellation included, is the reference's double.
__device__ __forceinline__ float reward_f32(double k1, double k2) {
  return (float)(k_over_1000(k1) + k_over_1000(k2));
}

enum Mode { kObserve = 0, kStep = 1, kStepAutoReset = 2 };

// One-wave workgroups (N <= 64) are register-limited: 128 VGPRs = 4 waves per SIMD, i.e. the
// 4096 waves of 64 x 4096 are all resident at once.  Larger N is LDS-limited (3 per SIMD).
#ifndef RVO3D_WAVES_ATTR
#define RVO3D_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(NW == 1 ? 4 : 3)))
#endif

// The whole environment step, one launch.
template <int MODE, int NW>
__global__ void __launch_bounds__(64 * NW) RVO3D_WAVES_ATTR env_kernel(const Params P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, T = NW == 1 ? 64 : (int)blockDim.x, N = P.N;
  const Lds L = carve_lds(smem, T, P.nm, P.epb, N, NW);
  const int el = tid / N;
  const int d = tid - el * N;
  const int e0 = blockIdx.x * P.epb;
  const int e = e0 + el;
  const bool active = (el < P.epb) && (e < P.E);
  const int g = active ? e * N + d : 0;
  const int lbase = el * N;
  const int nrows = ((P.E - e0) < P.epb ? (P.E - e0) : P.epb) * N;  // rows of this workgroup
  constexpr bool LITE = (MODE == kStepAutoReset);

  // Register discipline: values are loaded right before the phase that needs them and
  // stored as soon as they are final, so that across the sweeps little more than the
  // drone's own 8-value record and its action stay live (registers = waves per SIMD).
  RVO3D_STAMP(0);
  Drone S;
  S.x = S.y = S.z = S.vx = S.vy = S.vz = 0.0; S.r = 0.2; S.prio = 5;
  double a[3] = {0, 0, 0}, cur[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
  double dev = 0, max_dev = 0;
  int wpi = 1;

  // ---- phase 0: the drone's own record and its action; everything else about the pre-move
  //      state (waypoints, des_vel, deviation) is fetched after sweep A, which needs none of it
  if (active) {
    S.x = P.px()[g]; S.y = P.py()[g]; S.z = P.pz()[g];
    S.vx = P.vx()[g]; S.vy = P.vy()[g]; S.vz = P.vz()[g];
    S.r = P.radius()[g]; S.prio = P.prio()[g];
    if (MODE != kObserve) {
      if (P.action_mode == 1) {
        // The trainer's glue (multi_ppo.py:196-205), in numpy's own types:
        //   a_inc = np.round(sample, 2)                  float32: rint(a * 100f) / 100f
        //   abs   = np.round(acceler * a_inc + vel, 2)   float32 product, widened, + float64
        const float* A = static_cast<const float*>(P.actions) + (size_t)g * 3;
        const double vv[3] = {S.vx, S.vy, S.vz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float r = __builtin_rintf(A[k] * 100.0f) / 100.0f;
          const double x = (double)(P.acceler * r) + vv[k];
          a[k] = __builtin_rint(x * 100.0) / 100.0;
        }
      } else {
        if (P.action_f64) {
          const double* A = static_cast<const double*>(P.actions) + (size_t)g * 3;
          a[0] = A[0]; a[1] = A[1]; a[2] = A[2];
        } else {
          const float* A = static_cast<const float*>(P.actions) + (size_t)g * 3;
          a[0] = (double)A[0]; a[1] = (double)A[1]; a[2] = (double)A[2];
        }
        if (P.cold().act_scale > 0) {
          a[0] = __builtin_rint(a[0] * P.cold().act_scale) / P.cold().act_scale;
          a[1] = __builtin_rint(a[1] * P.cold().act_scale) / P.cold().act_scale;
          a[2] = __builtin_rint(a[2] * P.cold().act_scale) / P.cold().act_scale;
        }
      }
    }
  }
  RVO3D_STAMP(1);
  double az[3] = {a[0], a[1], a[2]};  // action as the RVO code sees it (rvo_inter.py:118)
  if (norm3b(a[0], a[1], a[2]) < 1e-5) az[0] = az[1] = az[2] = 0.0;
  const double zero3[3] = {0, 0, 0};

  if (tid < P.epb) { L.any_reset[tid] = 0; L.far[tid] = 0; }
  L.kept[tid] = 0;
  __syncthreads();  // flags zeroed before anyone raises them
  L.x[tid] = S.x; L.y[tid] = S.y; L.z[tid] = S.z;
  L.vx[tid] = S.vx; L.vy[tid] = S.vy; L.vz[tid] = S.vz;
  L.r[tid] = S.r; L.prio[tid] = S.prio;
  {
    const double p[3] = {S.x, S.y, S.z}, v[3] = {S.vx, S.vy, S.vz};
    stage_f32(P, L, el, d, active, p, v, az, S.r, S.prio);
  }
  __syncthreads();

  bool flag, collision = false;
  double tmin;

  if (MODE == kObserve) {
    if (active) {  // drone.dronestate (drone.py:254-263)
      double prev[3];
      max_dev = P.max_dev()[g];
      RVO3D_LOAD_CUR(P, g, cur);
      RVO3D_LOAD_PREV(P, g, prev);
      const double p[3] = {S.x, S.y, S.z};
      des_vel(P, p, cur, dv);
      dev = deviation(prev, cur, p);
      if (dev > max_dev) max_dev = dev;
    }
    uint32_t gw[NW];
    const int kept = sweep_env<NW, true, true>(P, L, tid, el, d, g, active, S, zero3, true, flag,
                                               tmin, collision, gw, false);
    if (active) {
      write_proprio(P, g, S, proprio_tail(dv, dev));
      write_vo_rows(P, L, tid, lbase, g, S, kept);
      L.kept[tid] = kept;
      P.max_dev()[g] = max_dev;
      uint32_t dvk_a, dvk_b;
      dv_encode(dv, dvk_a, dvk_b);
      P.dvk_a()[g] = dvk_a; P.dvk_b()[g] = dvk_b;
#pragma unroll
      for (int w = 0; w < NW; ++w) P.gcache(w)[g] = gw[w];
    }
    __syncthreads();
    zero_fill(P, L, tid, e0 * N, nrows);
    return;
  }

  RVO3D_STAMP(2);
  // ---- sweep A: ir_gym.rvo_reward_list_cal on the pre-move state (ir_gym.py:50-62)
  uint32_t gw[NW];
  const bool have_gw = P.g_cached != 0;  // the previous step ended in this very state
  if (have_gw && active) {
#pragma unroll
    for (int w = 0; w < NW; ++w) gw[w] = P.gcache(w)[g];
  }
  sweep_env<NW, false, false>(P, L, tid, el, d, g, active && !(P.ablate & 1), S, az, false, flag,
                              tmin, collision, gw, have_gw);
  // ---- everything else about this drone arrives in ONE batch of loads now (none of the
  //      addresses depends on a loaded value), then: drone.dronestate on the pre-move state
  //      (drone.py:254-263) and the RVO reward - the state is the one the previous step (or
  //      observe / reset) ended in, so its des_vel is on file and its deviation is already in
  //      max_deviation; only a state set from outside is recomputed - and
  //      drone.move_forward + kinematicstep (drone.py:96-129, 435-490), the post-move
  //      dronestate and the arrival flags of ir_gym.observation_reward (:168-193)
  double rew_k = 0;
  double mov_nc = 0;  // mov_reward (k form) if the step turns out collision-free
  bool f_dest = false;
  RVO3D_STAMP(3);
  if (active) {
    max_dev = P.max_dev()[g];
    RVO3D_LOAD_CUR(P, g, cur);
    bool have = false;
    if (P.dv_cached) have = dv_decode(P.dvk_a()[g], P.dvk_b()[g], dv);
    if (!have) {
      double prev[3];
      RVO3D_LOAD_PREV(P, g, prev);
      const double p[3] = {S.x, S.y, S.z};
      des_vel(P, p, cur, dv);
      dev = deviation(prev, cur, p);
      if (dev > max_dev) max_dev = dev;
    }
    rew_k = rvo_reward_k(rvo_reward_pre(dv, a), flag, tmin);
  }
  __syncthreads();  // everyone is done with the pre-move LDS image
  if (active) {
    double prev[3];
    wpi = P.wp_idx()[g];
    RVO3D_LOAD_PREV(P, g, prev);
    double yaw = P.yaw()[g], pitch = P.pitch()[g], real_len = P.real_len()[g];
    double extra_len = P.extra_len()[g];
    const double route_len = P.route_len()[g];
    const int npts = P.n_points()[g];
    bool f_arrive = P.arrive()[g] != 0;
    f_dest = P.dest()[g] != 0;

    double speed = norm3b(S.vx, S.vy, S.vz);
    const double acc = clampd(a[0] * 1.0, -1.0, 1.0);
    const double dyaw = clampd(a[1] * 90.0, -90.0, 90.0);
    const double dpit = clampd(a[2] * 90.0, -90.0, 90.0);
    const double nv = speed + acc;
    speed = (0.0 > nv) ? 0.0 : nv;
    yaw = np_mod(yaw + dyaw, 360.0);
    pitch = clampd(pitch + dpit, -90.0, 90.0);
    double nvx = 0.0, nvy = 0.0, nvz = 0.0;
    if (!f_dest) {  // `stop` := map_size (env_base.py:142, drone.py:107): parked once finished
      double sy, cy, sp, cp;
      sincos(yaw * kDeg2Rad, &sy, &cy);
      sincos(pitch * kDeg2Rad, &sp, &cp);
      nvx = speed * cp * cy; nvy = speed * cp * sy; nvz = speed * sp;
    }
    const double q0 = S.x, q1 = S.y, q2 = S.z;
    S.x = S.x + nvx; S.y = S.y + nvy; S.z = S.z + nvz;
    S.vx = nvx; S.vy = nvy; S.vz = nvz;
    real_len = real_len + norm3b(S.x - q0, S.y - q1, S.z - q2);
    const double p[3] = {S.x, S.y, S.z};
    // the destination matters only next to a waypoint: fetched on demand (rare)
    double dst[3] = {0, 0, 0};
    if (f_arrive || arrived(P, p, cur)) load_wp(P, g, npts - 1, dst);
    if (arrived(P, p, cur)) {  // drone.py:116-129
      const bool at_dst = arrived(P, p, dst);
      if (at_dst) extra_len = real_len - route_len;  // destination_arrive side effect
      if (!at_dst && wpi < npts - 1) {
        wpi += 1;
        prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
        load_wp(P, g, wpi, cur);
        RVO3D_STORE_CUR(P, g, cur);
        RVO3D_STORE_PREV(P, g, prev);
        f_arrive = false;
      }
    }
    // dronestate on the post-move state
    des_vel(P, p, cur, dv);
    dev = deviation(prev, cur, p);
    if (dev > max_dev) max_dev = dev;
    // arrival flags (ir_gym.py:168-181)
    bool arrive_r = false, dest_r = false;
    const int waypoint_num = wpi;
    if (!f_arrive && arrived(P, p, cur)) { f_arrive = true; arrive_r = true; }
    if (f_arrive) {
      if (arrived(P, p, dst)) {
        extra_len = real_len - route_len;
        if (!f_dest) { f_dest = true; dest_r = true; }
      }
    }
    const double exlen = real_len - route_len + 4;
    mov_nc = mov_reward_k(P, false, arrive_r, waypoint_num, npts - 1, dest_r, dev, exlen > 0,
                          exlen);
    collision = building_hit(P, S);
    if (S.x < 0 || S.x > P.cold().map[0] || S.y < 0 || S.y > P.cold().map[1] || S.z < 0 || S.z > P.cold().map[2])
      collision = true;  // drone.drone_out_map, drone.py:213-225
    // final for this step unless the drone is reset below
    P.yaw()[g] = yaw; P.pitch()[g] = pitch; P.real_len()[g] = real_len; P.extra_len()[g] = extra_len;
    P.wp_idx()[g] = wpi;
    P.arrive()[g] = f_arrive ? 1 : 0; P.dest()[g] = f_dest ? 1 : 0;
    P.info[g] = f_arrive ? 1 : 0;
    P.finish[g] = f_dest ? 1 : 0;
  }
  L.x[tid] = S.x; L.y[tid] = S.y; L.z[tid] = S.z;
  L.vx[tid] = S.vx; L.vy[tid] = S.vy; L.vz[tid] = S.vz;
  {
    const double p[3] = {S.x, S.y, S.z}, v[3] = {S.vx, S.vy, S.vz};
    stage_f32(P, L, el, d, active, p, v, az, S.r, S.prio);
  }
  __syncthreads();

  RVO3D_STAMP(4);
  // ---- sweep B: the pair part of ir_gym.observation_reward (ir_gym.py:197).
  // In the fused auto-reset step an env that resets discards the step's VO rows (its
  // observation is recomputed after the reset), so a collision-only sweep runs first,
  // the resets are settled, and then ONE sweep produces the rows - on the post-move
  // state with the action, or on the post-reset state with action 0.
  int kept = 0;
  if (LITE) {
    if (collide_env<NW>(P, L, tid, el, d, active && !(P.ablate & 2), S, gw)) collision = true;
  } else {
    kept = sweep_env<NW, true, true>(P, L, tid, el, d, g, active && !(P.ablate & 2), S, az, false,
                                     flag, tmin, collision, gw, false);
  }
  bool do_reset = false;
  if (active) {
    P.reward[g] = reward_f32(rew_k, collision ? -50000.0 : mov_nc);  // mdin.py:28
    P.done[g] = collision ? 1 : 0;
    do_reset = LITE && (collision || f_dest);
  }

  RVO3D_STAMP(5);
  if (LITE) {
    if (active && P.reset_mask) P.reset_mask[g] = do_reset ? 1 : 0;
    if (do_reset) L.any_reset[el] = 1;
    __syncthreads();  // sweep reads done; any_reset visible
    if (do_reset) {  // drone.reset (drone.py:270-291); extra_len survives
      double p[3];
      load_wp(P, g, 0, p);
      S.x = p[0]; S.y = p[1]; S.z = p[2]; S.vx = S.vy = S.vz = 0.0;
      // dronestate of the start state: static, tabulated by rvo3d_load_world (dv0_kernel)
      dev = P.dev0()[g];
      load_wp(P, g, 1, cur);
      if (!dv_decode(P.dv0_a()[g], P.dv0_b()[g], dv)) {
        des_vel(P, p, cur, dv);
        dev = deviation(p, cur, p);  // previous_des = waypoints[0] = the start position
      }
      RVO3D_STORE_CUR(P, g, cur);
      RVO3D_STORE_PREV(P, g, p);
      max_dev = dev > 0.0 ? dev : 0.0;
      P.wp_idx()[g] = 1; P.arrive()[g] = 0; P.dest()[g] = 0;
      P.real_len()[g] = 0.0; P.yaw()[g] = 0.0; P.pitch()[g] = 0.0;
      L.x[tid] = S.x; L.y[tid] = S.y; L.z[tid] = S.z;
      L.vx[tid] = 0.0; L.vy[tid] = 0.0; L.vz[tid] = 0.0;
      const double v0[3] = {0, 0, 0};
      stage_f32(P, L, el, d, true, p, v0, az, S.r, S.prio);
    }
  }
  // everything about this drone except its VO rows is final now.  The stores wait until
  // after the last sweep (vector memory returns in order: a load behind a store waits for
  // it, and measured: stores issued here cost 1.5 %); the drone's record and the rounded
  // floats of des_vel / deviation stay live across the sweep.
  if (active) {
    uint32_t dvk_a, dvk_b;
    dv_encode(dv, dvk_a, dvk_b);  // des_vel, on file for the next step
    P.dvk_a()[g] = dvk_a; P.dvk_b()[g] = dvk_b;
  }
  const ProprioTail ptail = proprio_tail(dv, dev);
  if (LITE) {
    __syncthreads();
    RVO3D_STAMP(6);
    // rows for every env: ir_gym.observation_reward's VO part (the env kept its state) or
    // ir_gym.env_observation with action 0 (the env reset a drone, ir_gym.py:372-383)
    const bool env_reset = active && (L.any_reset[el] != 0);
    bool c2 = false;
    const double* aa = env_reset ? zero3 : az;
    // stage G: the collision sweep delivered the words of the post-move state; only pairs
    // with a reset drone changed since (larger envs: recompute when the env reset anyone)
    bool have_gw2 = !env_reset;
    if (NW == 1) {
      const unsigned long long rlanes = __ballot(do_reset);
      if (L.far[el] != 0) {
        uint32_t valid[1];
        valid_offsets<1>(N, d, valid);
        gw[0] = valid[0];
      } else {
        gw[0] = regate_resets(P, L, tid, el, d, active, rlanes, gw[0]);
      }
      have_gw2 = true;
    }
    if (P.ablate & 2) have_gw2 = false;  // diagnostics: the collision sweep was skipped
    kept = sweep_env<NW, true, false>(P, L, tid, el, d, g, active && !(P.ablate & 4), S, aa,
                                      env_reset, flag, tmin, c2, gw, have_gw2);
  }
  RVO3D_STAMP(7);
  if (active) {
    if (!(P.ablate & 8)) {
      write_vo_rows(P, L, tid, lbase, g, S, kept);
      write_proprio(P, g, S, ptail);
    }
    L.kept[tid] = kept;
    P.max_dev()[g] = max_dev;
#pragma unroll
    for (int w = 0; w < NW; ++w) P.gcache(w)[g] = gw[w];
    P.px()[g] = S.x; P.py()[g] = S.y; P.pz()[g] = S.z;
    P.vx()[g] = S.vx; P.vy()[g] = S.vy; P.vz()[g] = S.vz;
  }
  __syncthreads();  // L.kept complete
  RVO3D_STAMP(8);
  if (!(P.ablate & 16)) zero_fill(P, L, tid, e0 * N, nrows);
  RVO3D_STAMP(9);
}

// ---- small state kernels -------------------------------------------------------
// drone.reset (drone.py:270-291) for masked envs / drones.
__global__ void reset_kernel(const Params P, const uint8_t* env_mask, const uint8_t* drone_mask) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.E * P.N) return;
  if (env_mask && !env_mask[g / P.N]) return;
  if (drone_mask && !drone_mask[g]) return;
  double s[3];
  load_wp(P, g, 0, s);
  P.px()[g] = s[0]; P.py()[g] = s[1]; P.pz()[g] = s[2];
  P.vx()[g] = 0.0; P.vy()[g] = 0.0; P.vz()[g] = 0.0;
  P.wp_idx()[g] = 1; P.arrive()[g] = 0; P.dest()[g] = 0;
  P.real_len()[g] = 0.0; P.max_dev()[g] = 0.0; P.yaw()[g] = 0.0; P.pitch()[g] = 0.0;
  double c1[3];
  load_wp(P, g, 1, c1);
  RVO3D_STORE_CUR(P, g, c1);
  RVO3D_STORE_PREV(P, g, s);
  // des_vel of the start state is on file; a start state with a non-zero deviation (only
  // with non-finite waypoints) is left to the step's own dronestate
  const bool plain = P.dev0()[g] == 0.0;
  P.dvk_a()[g] = plain ? P.dv0_a()[g] : kDvInvalid;
  P.dvk_b()[g] = P.dv0_b()[g];
}

// cur / prev from the waypoint index (rvo3d_load_world; rvo3d_set_state with wp_idx)
__global__ void wpcache_kernel(const Params P) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.E * P.N) return;
  const int np = P.n_points()[g];
  int i = P.wp_idx()[g];
  i = i < 1 ? 1 : (i > np - 1 ? np - 1 : i);  // the clamp only guards the table lookup
  double v[3];
  load_wp(P, g, i, v);
  RVO3D_STORE_CUR(P, g, v);
  load_wp(P, g, i - 1, v);
  RVO3D_STORE_PREV(P, g, v);
}

// rvo3d_load_world: dronestate of every drone's reset state (drone.py:254-263 after
// drone.reset, :270-291): des_vel towards waypoint 1 and the deviation from the first leg.
__global__ void dv0_kernel(const Params P) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.E * P.N) return;
  double p[3], cur[3], dv[3];
  load_wp(P, g, 0, p);
  load_wp(P, g, 1, cur);
  des_vel(P, p, cur, dv);
  uint32_t a, b;
  dv_encode(dv, a, b);
  P.dv0_a()[g] = a; P.dv0_b()[g] = b;
  P.dev0()[g] = deviation(p, cur, p);
}

__global__ void des_vel_kernel(const Params P, double* out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.E * P.N) return;
  double p[3] = {P.px()[g], P.py()[g], P.pz()[g]}, cur[3], dv[3];
  RVO3D_LOAD_CUR(P, g, cur);
  des_vel(P, p, cur, dv);
  out[3 * (size_t)g] = dv[0]; out[3 * (size_t)g + 1] = dv[1]; out[3 * (size_t)g + 2] = dv[2];
}

// ---- classical RVO velocity selection -----------------------------------
// uaisa_env/vel_obs/reciprocal_vel_obs.py:19-166 as intended (the class cannot run: list
// attribute assignment :109, slices :63-69/:105, missing return :119-124), built from the
// helpers it calls: get_alpha / get_PAA / get_rvo_array / get_beta / cal_exp_tim
// (vel_obs3D.py:8-66, 104-143).  PARITY UNPINNED for the driver loop; the helpers' arithmetic
// is pinned through the CPU restatement by tests/golden/rvo_vel.npz (tests/test_rvo_vel.py).
// One workgroup per env, one thread per drone; the env's records
// are staged in LDS; candidates (<= 64: acceler <= 1) are tested against every neighbour's
// velocity obstacle with one bit per candidate.
struct RvoVelArgs { double vmax[3]; double acceler; };

__device__ __forceinline__ int arange_len(double lo, double hi) {  // len(np.arange(lo, hi, 0.5))
  const double n = __builtin_ceil((hi - lo) / 0.5);
  return n > 0 ? (int)n : 0;
}
__device__ __forceinline__ double arange_at(double lo, int k) {  // numpy fills start + k * delta
  const double next = lo + 0.5;
  return k == 0 ? lo : (k == 1 ? next : lo + k * (next - lo));
}

__global__ void rvo_vel_kernel(const Params P, const RvoVelArgs A, double* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = P.N, d = threadIdx.x, e = blockIdx.x;
  const int T = blockDim.x;
  double* const lx = reinterpret_cast<double*>(smem);  // x y z vx vy vz r prio, [T] each
  const bool active = d < N;
  const int g = active ? e * N + d : e * N;
  Drone S;
  S.x = P.px()[g]; S.y = P.py()[g]; S.z = P.pz()[g];
  S.vx = P.vx()[g]; S.vy = P.vy()[g]; S.vz = P.vz()[g];
  S.r = P.radius()[g]; S.prio = P.prio()[g];
  lx[d] = S.x; lx[T + d] = S.y; lx[2 * T + d] = S.z;
  lx[3 * T + d] = S.vx; lx[4 * T + d] = S.vy; lx[5 * T + d] = S.vz;
  lx[6 * T + d] = S.r; lx[7 * T + d] = S.prio;
  __syncthreads();
  if (!active) return;
  double cur[3], des[3];
  RVO3D_LOAD_CUR(P, g, cur);
  const double p[3] = {S.x, S.y, S.z}, v0[3] = {S.vx, S.vy, S.vz};
  des_vel(P, p, cur, des);
  double lo[3];
  int cnt[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lo[k] = clampd(v0[k] - A.acceler, -A.vmax[k], A.vmax[k]);  // np.clip
    const double hi = clampd(v0[k] + A.acceler, -A.vmax[k], A.vmax[k]);
    cnt[k] = arange_len(lo[k], hi);
    if (cnt[k] > 4) cnt[k] = 4;  // host checks acceler <= 1
  }
  const int C = cnt[0] * cnt[1] * cnt[2];  // candidate c = (ix * cnt1 + iy) * cnt2 + iz
  unsigned long long live = 0ull, inside = 0ull;
  for (int c = 0; c < C; ++c) {
    const int iz = c % cnt[2], iy = (c / cnt[2]) % cnt[1], ix = c / (cnt[2] * cnt[1]);
    const double vx = arange_at(lo[0], ix), vy = arange_at(lo[1], iy), vz = arange_at(lo[2], iz);
    if (!(__builtin_sqrt(sq(vx) + sq(vy) + sq(vz)) < 0.3)) live |= 1ull << c;
  }
  double tc_min = __builtin_inf();
  for (int j = 0; j < N; ++j) {
    if (j == d) continue;
    const double bx = lx[j], by = lx[T + j], bz = lx[2 * T + j];
    const double fx = S.x - bx, fy = S.y - by, fz = S.z - bz;  // agent - drone (:40-43)
    if (!(dot3b(fx, fy, fz, fx, fy, fz) <= P.T10)) continue;   // norm <= 10
    const double bvx = lx[3 * T + j], bvy = lx[4 * T + j], bvz = lx[5 * T + j];
    const double br = lx[6 * T + j], bprio = lx[7 * T + j];
    // cal_exp_tim (vel_obs3D.py:104-143)
    {
      const double wx = S.vx - bvx, wy = S.vy - bvy, wz = S.vz - bvz, r = S.r + br;
      const double qa = sq(wx) + sq(wy) + sq(wz);
      const double qb = 2 * fx * wx + 2 * fy * wy + 2 * fz * wz;
      const double qc = sq(fx) + sq(fy) + sq(fz) - sq(r);
      double tc;
      if (qc <= 0) tc = 0.0;
      else {
        const double temp = sq(qb) - 4 * qa * qc;
        if (temp <= 0) tc = __builtin_inf();
        else {
          const double sr = __builtin_sqrt(temp);
          const double t1 = (-qb + sr) / (2 * qa), t2 = (-qb - sr) / (2 * qa);
          const double t3 = t1 >= 0 ? t1 : __builtin_inf(), t4 = t2 >= 0 ? t2 : __builtin_inf();
          tc = t3 < t4 ? t3 : t4;
        }
      }
      if (tc < tc_min) tc_min = tc;
    }
    const double ax = bx - S.x, ay = by - S.y, az = bz - S.z;  // get_rvo_array
    const double nab = norm3b(ax, ay, az);
    const double q = (S.r + br) / nab;
    const double alpha = (q <= 1.0) ? py_round2_c(asin(q)) / 100.0 : 1.57;  // get_alpha
    const double pr = S.prio / (S.prio + bprio);                            // get_PAA
    const double pax = pr * (2 * S.x + (S.vx + bvx) * 1), pay = pr * (2 * S.y + (S.vy + bvy) * 1),
                 paz = pr * (2 * S.z + (S.vz + bvz) * 1);
    for (int c = 0; c < C; ++c) {  // vo_out2 (:103-117)
      if (!((live >> c) & 1ull)) continue;
      const int iz = c % cnt[2], iy = (c / cnt[2]) % cnt[1], ix = c / (cnt[2] * cnt[1]);
      const double wx = (S.x + arange_at(lo[0], ix) * 1) - pax,
                   wy = (S.y + arange_at(lo[1], iy) * 1) - pay,
                   wz = (S.z + arange_at(lo[2], iz) * 1) - paz;
      const double AB = nab * norm3b(wx, wy, wz);
      const double cs = (AB != 0) ? dot3b(ax, ay, az, wx, wy, wz) / AB : 0.0;  // get_beta
      const double beta = __builtin_rint(acos(cs) * 100.0) / 100.0;
      if (alpha > beta) inside |= 1ull << c;
    }
  }
  const double tc_inv = (tc_min == 0) ? __builtin_inf() : 1.0 / tc_min;
  bool have_out = false, have_in = false;
  double best_out = 0, best_in = 0, so[3] = {0, 0, 0}, si[3] = {0, 0, 0};
  for (int c = 0; c < C; ++c) {  // vel_select (:119-124): Python min keeps the first minimum
    if (!((live >> c) & 1ull)) continue;
    const int iz = c % cnt[2], iy = (c / cnt[2]) % cnt[1], ix = c / (cnt[2] * cnt[1]);
    const double vx = arange_at(lo[0], ix), vy = arange_at(lo[1], iy), vz = arange_at(lo[2], iz);
    const double dd = __builtin_sqrt(sq(des[0] - vx) + sq(des[1] - vy) + sq(des[2] - vz));
    if (!((inside >> c) & 1ull)) {
      if (!have_out || dd < best_out) { best_out = dd; so[0] = vx; so[1] = vy; so[2] = vz; have_out = true; }
    } else {
      const double pen = 1 * tc_inv + dd;
      if (!have_in || pen < best_in) { best_in = pen; si[0] = vx; si[1] = vy; si[2] = vz; have_in = true; }
    }
  }
  double* o = out + 3 * (size_t)g;
  if (have_out) { o[0] = so[0]; o[1] = so[1]; o[2] = so[2]; }
  else if (have_in) { o[0] = si[0]; o[1] = si[1]; o[2] = si[2]; }
  else { o[0] = 0.0; o[1] = 0.0; o[2] = 0.0; }
}

// AoS <-> SoA copies for get_state / set_state
__global__ void aos3_to_soa(const double* src, double* x, double* y, double* z, int n) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  x[g] = src[3 * (size_t)g]; y[g] = src[3 * (size_t)g + 1]; z[g] = src[3 * (size_t)g + 2];
}
__global__ void soa_to_aos3(const double* x, const double* y, const double* z, double* dst, int n) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  dst[3 * (size_t)g] = x[g]; dst[3 * (size_t)g + 1] = y[g]; dst[3 * (size_t)g + 2] = z[g];
}

}  // namespace rvo3d
