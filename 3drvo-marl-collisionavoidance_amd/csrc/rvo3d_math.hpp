// rvo3d_math.hpp -- Arithmetic primitives in the reference's evaluation order, exact rounding shortcuts, and the
// per-drone pieces: des_vel, deviation, arrival, the TTC quadratic; Drone / PairOut records.
// Part of the gfx950 device code (see rvo3d_device.hpp for the overview).
#pragma once

#include "rvo3d_params.hpp"

namespace rvo3d {

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
  // fmod without the library call where it is plain: |a| < b is a itself, b <= a < 2b is
  // a - b, exact (Sterbenz) - the yaw of a step, [0, 360) plus at most 90 either way, never
  // leaves that range (tests/test_numeric_shortcuts.py)
  double m;
  if (b > 0.0 && a > -b && a < 2.0 * b) m = a >= b ? a - b : a;
  else m = fmod(a, b);
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

}  // namespace rvo3d
