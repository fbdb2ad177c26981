// rvo3d_params.hpp -- Parameters of the step kernels: constants, the rarely used ones read through the
// constant address space (Cold), the kernel-argument block with its arena accessors (Params).
// Part of the gfx950 device code (see rvo3d_device.hpp for the overview).
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
  float cmax;         // |centred coordinate| bound the bands were computed for
  float kdot;         // fp32 error bound of v.rel per unit |v|_1
  uint32_t zf_div;    // zero-fill: units (8 B or 4 B) per row of the VO region
  uint32_t zf_magic;  // ceil(2^32 / zf_div)
  uint32_t zf_q;      // 16-B row writer (W even): row bytes / 8, else 0
  // two-phase row writer, part 1: for thread quad tq = tid / 4, bit i = "the 64-B block this quad
  // meets in trip i holds no proprio byte" (a full workgroup's rows; zf_iters trips, 0: no table)
  int zf_iters;
  uint32_t zmask[128];
  int nb;
  const double* bld;       // [nb][4]
  const double* pow95;     // [P]   0.95 ** k, host libm (ir_gym.py:283)
  // building grid over the map's xy plane (rvo3d_load_world): cell (ix, iy) lists every
  // building a drone inside the cell could hit (within min(5 m gate, largest drone radius +
  // building radius) of the cell), kBgridK + 1 u16 per cell = count, indices;
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
#ifdef RVO3D_DIAG
  int ablate;         // diagnostics build only (tools/): bit k skips phase k, results invalid
#endif
  int zf16;           // per call: obs is 16-B aligned and W is even -> 16-B zero-fill
  int uniform_rp;     // 1: every drone has radius r0 and priority prio0 (the reference's constants,
  double r0, prio0;   //    drone.py:14-15): the step does not read the two arrays (16 B per drone-step)
  double T10;         // max{x : sqrt(x) <= 10}  (rvo_inter.py:96)
  // fp32 candidate filter (stage G): error bands
  float t10n;      // -nextafter(T10 + band): stage G's fma chain starts here, in range iff it ends < 0
  float bandn;     // band + t10n: the same start for the "possibly touching" threshold of stage G
  float band;      // fp32 error bound of a squared distance at <= 10.5 m
  // fp32 cone pre-filter (stage X1)
  int nw;          // ceil(N / 64) (1..4; 8 beyond 256 drones): words per request mask, 32-offset words per drone
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
#ifdef RVO3D_DIAG
  unsigned long long* dbg;  // diagnostics build only: per-workgroup s_memtime stamps [blocks][32], or null
#endif
  // per-call I/O
  const void* actions;
  float* obs;
  int32_t* vo_count;
  float* reward;
  double* reward64;   // optional: the same reward as the reference returns it, float64 (mdin.py:28)
  uint8_t *done, *info, *finish, *reset_mask;
};

// Diagnostics exist only in the -DRVO3D_DIAG build (librvo3d_hip_diag.so, made and loaded by
// tools/ alone): phase stamps of lane 0 (rvo3d_debug_stamps; tools/stamps.py) and phase
// ablation (RVO3D_ABLATE; tools/pmc_ablate.sh).  The product library contains neither: no
// environment variable and no call can make its step skip work.
#ifdef RVO3D_DIAG
#define RVO3D_STAMP(i)                                                                  \
  do {                                                                                  \
    if (P.dbg && threadIdx.x == 0) P.dbg[(size_t)blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define RVO3D_ABLATED(bits) ((P.ablate & (bits)) != 0)
#else
#define RVO3D_STAMP(i) do { } while (0)
#define RVO3D_ABLATED(bits) false
#endif

}  // namespace rvo3d
