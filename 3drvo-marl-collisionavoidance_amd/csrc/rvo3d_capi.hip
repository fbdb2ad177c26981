// rvo3d_capi.hip -- the C-ABI of include/rvo3d.h over the gfx950 kernels.
// Host side: handle + device buffers + launches.  No torch types, no
// exceptions across the boundary, no allocation in step/observe.
#include "../../include/rvo3d.h"
#include "rvo3d_device.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

using rvo3d::Params;

namespace {
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                            \
  do {                                                                           \
    hipError_t _e = (expr);                                                      \
    if (_e != hipSuccess)                                                        \
      return fail(RVO3D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// "Nothing throws" (include/rvo3d.h): every entry point that returns a status runs between these two.
// A host allocation that fails (std::vector staging in rvo3d_load_world, the error string itself) or
// any other C++ exception becomes RVO3D_ERR_INVALID instead of crossing the extern "C" boundary.
int api_caught(const char* what) noexcept {
  try {
    g_err = std::string("C++ exception: ") + what;
  } catch (...) {
    try { g_err.clear(); } catch (...) { }
  }
  return RVO3D_ERR_INVALID;
}
#define RVO3D_API_BEGIN try {
#define RVO3D_API_END                                              \
  }                                                                \
  catch (const std::exception& e) { return api_caught(e.what()); } \
  catch (...) { return api_caught("unknown"); }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// x ** 2 as the reference computes it: glibc pow (the volatile exponent keeps
// the compiler from folding the call into x * x).
volatile double kTwo = 2.0;
inline double py_sq(double x) { return std::pow(x, kTwo); }

// max{x : sqrt(x) <= tau} for the host's correctly rounded sqrt: `norm <= tau`
// in the reference is exactly `norm^2 <= sq_threshold(tau)` on the device.
double sq_threshold(double tau) {
  volatile double x = tau * tau;
  while (std::sqrt(std::nextafter((double)x, INFINITY)) <= tau) x = std::nextafter((double)x, INFINITY);
  while (std::sqrt((double)x) > tau) x = std::nextafter((double)x, -INFINITY);
  return x;
}
}  // namespace

struct rvo3d_env {
  rvo3d_config cfg;
  Params P;             // pointers into `arena`
  rvo3d::Cold cold;     // host copy of the rarely used parameters (device copy: P.cold_)
  void* arena = nullptr;
  size_t arena_bytes = 0;
  bool world_loaded = false;
  bool dv_valid = false;  // dvk_a/dvk_b describe the current state (see Params::dv_cached)
  bool g_valid = false;   // gcache describes the current state (see Params::g_cached)
  int threads = 0, blocks = 0, lds = 0;
};

namespace {

// One device allocation, carved into 256-B aligned struct-of-arrays fields.
int carve(rvo3d_env* h) {
  const rvo3d_config& c = h->cfg;
  const size_t EN = (size_t)c.num_envs * c.num_drones;
  const size_t S = align_up(EN, 64);  // common element stride of every per-drone array
  Params& P = h->P;
  rvo3d::Cold& C = h->cold;
  P.S = (uint32_t)S;
  const size_t nf = Params::f64_arrays(c.max_points, c.neighbors_num);
  const size_t ni = Params::i32_arrays(c.neighbors_num, P.nw);
  struct Field { void** slot; size_t bytes; };
  std::vector<Field> f = {
      {(void**)&P.f64, nf * S * 8},
      {(void**)&P.i32, ni * S * 4},
      {(void**)&P.u8, 2 * S},
      {(void**)&C.bld, (size_t)(c.num_buildings > 0 ? c.num_buildings : 1) * 4 * 8},
      {(void**)&C.pow95, (size_t)c.max_points * 8},
      {(void**)&C.bgrid, (size_t)(C.bgx > 0 ? C.bgx * C.bgy : 1) * (rvo3d::kBgridK + 1) * 2},
      {(void**)&P.err, 256},
      {(void**)&P.cold_, sizeof(rvo3d::Cold)},
  };
  size_t total = 0;
  for (auto& x : f) total += align_up(x.bytes, 256);
  HIP_TRY(hipMalloc(&h->arena, total));
  HIP_TRY(hipMemset(h->arena, 0, total));
  h->arena_bytes = total;
  size_t off = 0;
  for (auto& x : f) {
    *x.slot = static_cast<char*>(h->arena) + off;
    off += align_up(x.bytes, 256);
  }
  HIP_TRY(hipMemcpy((void*)P.cold_, &C, sizeof C, hipMemcpyHostToDevice));  // complete by now
  return RVO3D_OK;
}

// Makes the handle's device current for the duration of one API call and puts the caller's
// device back afterwards (a single-process multi-GPU program keeps its own current device).
struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  int enter(int dev) {
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess && prev != dev) {
      e = hipSetDevice(dev);
      changed = e == hipSuccess;
    }
    if (e != hipSuccess) return fail(RVO3D_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return RVO3D_OK;
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

int check(rvo3d_env* h, bool need_world, DeviceGuard& g) {
  if (!h) return fail(RVO3D_ERR_INVALID, "null handle");
  if (need_world && !h->world_loaded)
    return fail(RVO3D_ERR_STATE, "rvo3d_load_world has not been called");
  return g.enter(h->cfg.device);
}

// The compile-time-N instantiation (NFIX) a handle's shape selects, 0 = the generic kernel of its NW.
// One place decides it: launch_nw() launches it and rvo3d_kernel_name() reports it.
// Which instantiation a handle's shape runs on: the compile-time ring size NFIX (0 = the generic kernel of
// its NW) and whether N is smaller than it (padded: ghost lanes).  One place decides; launch_nw() launches
// it and rvo3d_kernel_name() reports it.
struct Pick { int nfix; bool pad; };
Pick pick_kernel(const Params& P) {
  if (P.nw == 1) {
    // a one-wave workgroup of epb envs: segments of 64 / epb lanes
    const int seg = (P.epb == 1 || P.epb == 2 || P.epb == 4) ? 64 / P.epb : 0;
    if (seg && P.N == seg) return {seg, false};
    if (seg >= 32 && P.N < seg) return {seg, true};  // 33..63 drones on the 64 kernel, 22..31 on the 32 one
    return {0, false};
  }
  // multi-wave workgroups: the compile-time kernels take any N up to their size
  if (P.nw == 2) return {128, true};
  if (P.nw == 3) return {192, true};
  if (P.nw == 4) return {256, true};
  return {0, false};
}

template <int MODE, int NW, int NFIX, bool TRAIN, bool PAD>
void launch_inst(rvo3d_env* h, const Params& P, hipStream_t s) {
  hipLaunchKernelGGL((rvo3d::env_kernel<MODE, NW, NFIX, TRAIN, PAD>), dim3(h->blocks), dim3(h->threads), h->lds, s, P);
}
template <int MODE, int NW, int NFIX, bool PAD>
void launch_train(rvo3d_env* h, const Params& P, hipStream_t s) {
  // both env_train modes have their instantiations (the evaluator of train/policy_test.py:46 runs env_train = False)
  if (P.env_train) launch_inst<MODE, NW, NFIX, true, PAD>(h, P, s);
  else launch_inst<MODE, NW, NFIX, false, PAD>(h, P, s);
}

template <int MODE, int NW>
int launch_nw(rvo3d_env* h, const Params& P, hipStream_t s) {
  const Pick k = pick_kernel(P);
  if constexpr (NW == 1) {
    if (k.nfix == 64) { if (k.pad) launch_train<MODE, 1, 64, true>(h, P, s); else launch_train<MODE, 1, 64, false>(h, P, s); }
    else if (k.nfix == 32) { if (k.pad) launch_train<MODE, 1, 32, true>(h, P, s); else launch_train<MODE, 1, 32, false>(h, P, s); }
    else if (k.nfix == 16) launch_train<MODE, 1, 16, false>(h, P, s);
    else launch_train<MODE, 1, 0, false>(h, P, s);
  } else if constexpr (NW == 2 || NW == 3 || NW == 4) {
    launch_train<MODE, NW, 64 * NW, true>(h, P, s);
  } else {
    launch_train<MODE, NW, 0, false>(h, P, s);
  }
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
}

template <int MODE>
int launch(rvo3d_env* h, const Params& P, hipStream_t s) {
  switch (P.nw) {
    case 1: return launch_nw<MODE, 1>(h, P, s);
    case 2: return launch_nw<MODE, 2>(h, P, s);
    case 3: return launch_nw<MODE, 3>(h, P, s);
    case 4: return launch_nw<MODE, 4>(h, P, s);
    default: return launch_nw<MODE, 8>(h, P, s);
  }
}

// (only the generic kernels can need more than 64 KiB: the compile-time ones stop at 256 drones = 40 KiB)
template <int MODE, int NW>
hipError_t allow_lds(int bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)rvo3d::env_kernel<MODE, NW, 0, true, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)rvo3d::env_kernel<MODE, NW, 0, false, false>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
template <int NW>
bool allow_lds_all(int bytes) {
  return allow_lds<rvo3d::kObserve, NW>(bytes) == hipSuccess &&
         allow_lds<rvo3d::kStep, NW>(bytes) == hipSuccess &&
         allow_lds<rvo3d::kStepAutoReset, NW>(bytes) == hipSuccess;
}

}  // namespace

#ifndef RVO3D_MLP_WAVES
#define RVO3D_MLP_WAVES 8
#endif
namespace {
constexpr int kMlpWaves = RVO3D_MLP_WAVES;  // waves per workgroup of policy_mlp_kernel
template <int KS1>
int launch_policy_mlp(const rvo3d::PolicyMlpArgs& A, unsigned grid, hipStream_t s) {
  constexpr int lds = rvo3d::mlp_lds_bytes(KS1);
  // more than 64 KB of dynamic LDS needs the attribute, once per device (the function object is per device) and
  // instantiation; a lost race between two threads sets it twice, which is harmless
  static uint64_t attr_set = 0;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !((attr_set >> dev) & 1)) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(rvo3d::policy_mlp_kernel<KS1, kMlpWaves>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (dev >= 0 && dev < 64) attr_set |= (uint64_t)1 << dev;
  }
  hipLaunchKernelGGL((rvo3d::policy_mlp_kernel<KS1, kMlpWaves>), dim3(grid), dim3(64 * kMlpWaves), lds, s, A);
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
}
}  // namespace

// optional noise counter in device memory (rvo3d_rollout_set_step_counter): added to the `step` of every sampling launch,
// advanced by rvo3d_rollout_account - lets a caller replay a captured launch sequence (a HIP graph) with fresh noise
#include <atomic>
namespace { std::atomic<uint64_t*> g_step_dev{nullptr}; }

extern "C" {

int rvo3d_version(void) { return RVO3D_VERSION; }
const char* rvo3d_last_error(void) { return g_err.c_str(); }

int rvo3d_create(const rvo3d_config* cfg, rvo3d_env** out) {
  RVO3D_API_BEGIN
  if (!cfg || !out) return fail(RVO3D_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->num_envs < 1 || cfg->num_drones < 1 || cfg->num_drones > rvo3d::kMaxThreads)
    return fail(RVO3D_ERR_INVALID, "need num_envs >= 1 and 1 <= num_drones <= 512");
  if (cfg->max_points < 2 || cfg->num_buildings < 0 || cfg->neighbors_num < 0)
    return fail(RVO3D_ERR_INVALID, "need max_points >= 2, num_buildings >= 0, neighbors_num >= 0");
  if ((long long)cfg->num_envs * cfg->num_drones > (1ll << 30))
    return fail(RVO3D_ERR_INVALID, "num_envs * num_drones too large");
  if (cfg->action_decimals > 9) return fail(RVO3D_ERR_INVALID, "action_decimals must be <= 9");
  DeviceGuard dg;
  if (int rc0 = dg.enter(cfg->device)) return rc0;

  rvo3d_env* h = new (std::nothrow) rvo3d_env();
  if (!h) return fail(RVO3D_ERR_INVALID, "out of host memory");
  // owns h until the very end: every early return - and an exception - frees the handle and its arena
  struct Owner {
    rvo3d_env* p;
    ~Owner() {
      if (!p) return;
      if (p->arena) (void)hipFree(p->arena);
      delete p;
    }
  } owner{h};
  h->cfg = *cfg;
  Params& P = h->P;
  rvo3d::Cold& C = h->cold;
  std::memset(&P, 0, sizeof P);
  std::memset(&C, 0, sizeof C);
  P.E = cfg->num_envs; P.N = cfg->num_drones; P.P = cfg->max_points;
  C.nb = cfg->num_buildings; P.nm = cfg->neighbors_num; P.env_train = cfg->env_train ? 1 : 0;
  P.W = 12 + 9 * P.nm;
#ifdef RVO3D_DIAG
  if (const char* ab = std::getenv("RVO3D_ABLATE")) P.ablate = std::atoi(ab);  // diagnostics build only
#endif
  C.act_scale = cfg->action_decimals >= 0 ? std::pow(10.0, cfg->action_decimals) : 0.0;
  for (int k = 0; k < 3; ++k) C.map[k] = cfg->map_size[k];
  P.T10 = sq_threshold(10.0);  // rvo_inter.py:96
  C.T5 = sq_threshold(5.0);    // rvo_inter.py:104
  C.bgx = C.bgy = 0; C.bg_inv = 0.0;
  if (cfg->num_buildings > 0 && cfg->map_size[0] > 0 && cfg->map_size[1] > 0 &&
      std::isfinite(cfg->map_size[0]) && std::isfinite(cfg->map_size[1])) {
    // xy grid for the building gate: ~8 m cells, at most 64 x 64
    const double cs = std::fmax(8.0, std::fmax(cfg->map_size[0], cfg->map_size[1]) / 64.0);
    C.bgx = (int)std::ceil(cfg->map_size[0] / cs); C.bgy = (int)std::ceil(cfg->map_size[1] / cs);
    if (C.bgx < 1) C.bgx = 1;
    if (C.bgy < 1) C.bgy = 1;
    C.bg_inv = 1.0 / cs;
  }
  C.T04 = sq_threshold(0.4);   // drone.py:15 goal_threshold
  {
    // fp32 candidate filter (stage G).  Coordinates are centred on the map and
    // assumed within cmax of it (envs with a drone further out bypass the filter).
    // u = 2^-24.  A centred coordinate carries <= u*cmax of rounding, a difference
    // of two <= eD = u*(2*cmax + 2*10.5); a squared distance at |d| <= 10.5 is off by
    // <= 2*sqrt(3)*10.5*eD + 3*eD^2 + 8u*10.5^2; v.rel by <= |v|_1*(eD + 4u*10.5).
    // Every band below is twice its bound.
    double mx = std::fmax(C.map[0], std::fmax(C.map[1], C.map[2]));
    if (!(mx > 0)) mx = 1.0;
    for (int k = 0; k < 3; ++k) C.cen[k] = 0.5 * C.map[k];
    const double cmax = 0.75 * mx + 16.0;
    const double u = std::ldexp(1.0, -24);
    const double eD = u * (2.0 * cmax + 21.0);
    const double band = 2.0 * (2.0 * 1.7320508 * 10.5 * eD + 3.0 * eD * eD + 8.0 * u * 110.25);
    C.cmax = (float)cmax;
    P.band = std::nextafter((float)band, INFINITY);
    // stage G tests the sign of dx^2 + dy^2 + dz^2 - t' (one fma chain): the chain's own rounding
    // (<= 3 ulp of ~110) is inside the band, which is twice the bound as it is
    const float t10f = std::nextafter((float)(P.T10 + band), INFINITY);
    P.t10n = -std::nextafter(t10f, INFINITY);
    P.bandn = P.band + P.t10n;
    C.kdot = std::nextafter((float)(2.0 * (eD + 4.0 * u * 10.5) * 1.001), INFINITY);
    // stage X1 (wave mode).  With gap = d^2 - R^2 >= x1_gap = 512*band the relative
    // error of gap is <= 1/1024 and |rel| >= sqrt(gap); a direction cosine then
    // carries <= cs = 4*(sqrt(3)*eD/sqrt(gap) + 8u) of error.  K^2 is compared with
    // slack 1 - (4e-3 + 4*cs): 2e-3 for gap's error, the rest for dp, w2 and K.
    const double gap = 512.0 * band;
    const double cs = 4.0 * (1.7320508 * eD / std::sqrt(gap) + 8.0 * u);
    P.x1_gap = (float)gap;
    P.x1_k2 = (float)(1.0 - (4e-3 + 4.0 * cs));
    const double cs_out = cs > 1e-3 ? cs : 1e-3;
    P.x1_cs2 = (float)(cs_out * cs_out);
  }

  // Launch geometry: whole envs per workgroup.  N <= 64: a workgroup is ONE wave holding
  // floor(64 / N) envs (its barriers are free, every wave is scheduled independently);
  // larger envs get one workgroup of ceil(N / 64) waves each.
  const int N = P.N;
  int nw = (N + 63) / 64;
  P.nw = nw <= 4 ? nw : 8;  // 1..4 waves: the compile-time kernels for 64 / 128 / 192 / 256 drones; beyond: generic
  int epb = P.nw == 1 ? 64 / N : 1;
  if (epb > P.E) epb = P.E;
  // nw = 2 / 3 / 4: the compile-time kernels for 128 / 192 / 256 drones, any N up to that (ghost lanes)
  const int ring = (P.nw >= 2 && P.nw <= 4) ? 64 * P.nw : N;
  const int threads = P.nw == 1 ? 64 : (int)align_up((size_t)ring, 64);
  size_t lds = rvo3d::lds_bytes(threads, P.nm, epb, ring, P.nw);
#ifdef RVO3D_DIAG
  if (const char* pad = std::getenv("RVO3D_LDS_PAD")) lds += (size_t)std::atoi(pad);  // diagnostics build only: cap occupancy
#endif
  if (lds > 160 * 1024) {
    return fail(RVO3D_ERR_INVALID, "neighbors_num * num_drones needs more than 160 KiB of LDS");
  }
  P.epb = epb;
  // zero-fill geometry: units per row of the VO region (float2 if rows are 8-B aligned)
  C.zf_div = (uint32_t)((P.W & 1) == 0 ? (P.W - 12) / 2 : (P.W - 12));
  C.zf_magic = 0;
  if (C.zf_div > 0) {
    const uint32_t m = (uint32_t)(((1ull << 32) + C.zf_div - 1) / C.zf_div);
    bool ok = true;
    const uint64_t qmax = (uint64_t)threads * C.zf_div;
    for (uint64_t q = 0; q < qmax && ok; ++q) ok = ((q * m) >> 32) == q / C.zf_div;
    if (!ok) {
      return fail(RVO3D_ERR_INVALID, "neighbors_num too large for the zero-fill index trick");
    }
    C.zf_magic = m;
  }
  C.zf_q = (P.W & 1) == 0 ? (uint32_t)(P.W / 2) : 0u;  // row bytes / 8: the 16-B row writer applies
  // early_zero_blocks: which of its trips a thread quad stores in depends on W and the quad only
  C.zf_iters = 0;
  static_assert(rvo3d::kMaxThreads / 4 <= sizeof(C.zmask) / sizeof(C.zmask[0]),
                "Cold::zmask has one word per thread quad of the largest workgroup");
  std::memset(C.zmask, 0, sizeof C.zmask);
  {
    const uint32_t rb = 4u * (uint32_t)P.W;
    const uint32_t rows_full = (uint32_t)epb * (uint32_t)N, nwv = (uint32_t)threads / 64u;
    const uint32_t nblk = rows_full * rb >> 6;
    const uint32_t iters = (nblk + 16u * nwv - 1u) / (16u * nwv);
    if (C.zf_q != 0 && P.W >= 48 && (rows_full & 7u) == 0 && iters <= 32u) {
      for (uint32_t tq = 0; tq < (uint32_t)threads / 4u; ++tq) {
        uint32_t m = 0;
        for (uint32_t i = 0; i < iters; ++i) {
          const uint32_t blk = tq + 16u * nwv * i;  // tq = wave * 16 + (lane / 4)
          if (blk >= nblk) break;
          const uint32_t o = (blk * 64u) % rb;
          if (o >= 48u && o + 64u <= rb) m |= 1u << i;
        }
        C.zmask[tq] = m;
      }
      C.zf_iters = (int)iters;
    }
  }
  h->threads = threads;
  h->blocks = (P.E + epb - 1) / epb;
  h->lds = (int)lds;
  if (lds > 64 * 1024) {
    const bool ok = P.nw == 1 ? allow_lds_all<1>((int)lds) : (P.nw == 8 && allow_lds_all<8>((int)lds));
    if (!ok) {
      return fail(RVO3D_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    }
  }
  int rc = carve(h);
  if (rc != RVO3D_OK) return rc;
  owner.p = nullptr;
  *out = h;
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_destroy(rvo3d_env* h) {
  RVO3D_API_BEGIN
  if (!h) return RVO3D_OK;
  DeviceGuard dg;
  (void)dg.enter(h->cfg.device);
  (void)hipDeviceSynchronize();
  if (h->arena) (void)hipFree(h->arena);
  delete h;
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_load_world(rvo3d_env* h, const double* waypoints, const int32_t* n_points,
                     const double* buildings, const double* radius, const double* priority,
                     void* stream) {
  RVO3D_API_BEGIN
  DeviceGuard dg;
  int rc = check(h, false, dg);
  if (rc) return rc;
  if (!waypoints || !n_points) return fail(RVO3D_ERR_INVALID, "waypoints / n_points are required");
  const Params& P = h->P;
  const rvo3d::Cold& C = h->cold;
  if (C.nb > 0 && !buildings) return fail(RVO3D_ERR_INVALID, "buildings required when num_buildings > 0");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t EN = (size_t)P.E * P.N;
  std::vector<double> wp((size_t)P.P * 3 * EN), rl(EN), rad(EN), pri(EN), p95(P.P);
  for (size_t g = 0; g < EN; ++g) {
    const int np = n_points[g];
    if (np < 2 || np > P.P) return fail(RVO3D_ERR_INVALID, "n_points entries must be in [2, max_points]");
    const double* src = waypoints + g * P.P * 3;
    double total = 0.0;  // drone.calculate_total_length (drone.py:409-429)
    for (int k = 0; k < P.P; ++k) {
      const int kk = k < np ? k : np - 1;  // pad with the destination
      for (int c = 0; c < 3; ++c) wp[((size_t)k * 3 + c) * EN + g] = src[kk * 3 + c];
      if (k + 1 < np) {
        const double dx = src[(k + 1) * 3] - src[k * 3], dy = src[(k + 1) * 3 + 1] - src[k * 3 + 1],
                     dz = src[(k + 1) * 3 + 2] - src[k * 3 + 2];
        total += std::sqrt(py_sq(dx) + py_sq(dy) + py_sq(dz));
      }
    }
    rl[g] = total;
    rad[g] = radius ? radius[g] : 0.2;
    pri[g] = priority ? priority[g] : 5.0;
  }
  {
    // one radius and one priority for every drone (bit-identical doubles): the step takes them from
    // its argument block instead of reading 16 B per drone-step
    bool uni = true;
    for (size_t g = 1; g < EN && uni; ++g)
      uni = std::memcmp(&rad[g], &rad[0], 8) == 0 && std::memcmp(&pri[g], &pri[0], 8) == 0;
    h->P.uniform_rp = uni ? 1 : 0;
    h->P.r0 = rad[0];
    h->P.prio0 = pri[0];
  }
  for (int k = 0; k < P.P; ++k) p95[k] = std::pow(0.95, (double)k);  // ir_gym.py:283
  // [P][3] rows of EN doubles into arrays of stride S
  HIP_TRY(hipMemcpy2DAsync((void*)P.wp(0, 0), (size_t)P.S * 8, wp.data(), EN * 8, EN * 8,
                           (size_t)P.P * 3, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync((void*)P.n_points(), n_points, EN * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync((void*)P.route_len(), rl.data(), EN * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync((void*)P.radius(), rad.data(), EN * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync((void*)P.prio(), pri.data(), EN * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync((void*)C.pow95, p95.data(), (size_t)P.P * 8, hipMemcpyHostToDevice, s));
  std::vector<uint16_t> grid;
  if (C.nb > 0) {
    HIP_TRY(hipMemcpyAsync((void*)C.bld, buildings, (size_t)C.nb * 32, hipMemcpyHostToDevice, s));
    if (C.bgx > 0) {
      // cell (ix, iy) = [ix*cs, (ix+1)*cs] x [iy*cs, (iy+1)*cs], widened by 1e-3 m (the device
      // finds the cell with floor(x / cs) in floating point) and unbounded at the map's edge
      // (clamped lookups); a building is listed where a drone inside the cell could hit it:
      // within the 5 m gate AND within (largest drone radius + building radius) of its axis
      // (rvo_inter.py:104, :207) - with 0.2 m drones that is 2 cells per building instead of 5
      const int K = rvo3d::kBgridK;
      const double cs = 1.0 / C.bg_inv;
      double rmax = 0.0;
      for (size_t g = 0; g < EN; ++g) {
        if (rad[g] != rad[g]) rmax = INFINITY;  // a NaN radius: no pruning beyond the gate
        else if (rad[g] > rmax) rmax = rad[g];
      }
      grid.assign((size_t)C.bgx * C.bgy * (K + 1), 0);
      for (int ix = 0; ix < C.bgx; ++ix)
        for (int iy = 0; iy < C.bgy; ++iy) {
          uint16_t* cell = &grid[((size_t)ix * C.bgy + iy) * (K + 1)];
          const double x0 = ix == 0 ? -INFINITY : ix * cs, x1 = ix == C.bgx - 1 ? INFINITY : (ix + 1) * cs;
          const double y0 = iy == 0 ? -INFINITY : iy * cs, y1 = iy == C.bgy - 1 ? INFINITY : (iy + 1) * cs;
          int n = 0;
          bool overflow = false;
          for (int b = 0; b < C.nb && !overflow; ++b) {
            const double bx = buildings[4 * b], by = buildings[4 * b + 1];
            double reach = rmax + buildings[4 * b + 3];
            if (!(reach < 5.0)) reach = 5.0;  // the gate (also a NaN radius)
            reach += 1e-3;
            const double dx = bx < x0 ? x0 - bx : (bx > x1 ? bx - x1 : 0.0);
            const double dy = by < y0 ? y0 - by : (by > y1 ? by - y1 : 0.0);
            if (!(dx * dx + dy * dy > reach * reach)) {  // also keeps NaN centres
              if (n == K || b > 0xfffe) overflow = true;
              else cell[1 + n++] = (uint16_t)b;
            }
          }
          cell[0] = overflow ? 0xffff : (uint16_t)n;
        }
      HIP_TRY(hipMemcpyAsync((void*)C.bgrid, grid.data(), grid.size() * 2, hipMemcpyHostToDevice, s));
    }
  }
  HIP_TRY(hipMemsetAsync(P.extra_len(), 0, EN * 8, s));
  const int tb = 256;
  hipLaunchKernelGGL(rvo3d::dv0_kernel, dim3((unsigned)((EN + tb - 1) / tb)), dim3(tb), 0, s, P);
  hipLaunchKernelGGL(rvo3d::reset_kernel, dim3((unsigned)((EN + tb - 1) / tb)), dim3(tb), 0, s, P,
                     (const uint8_t*)nullptr, (const uint8_t*)nullptr);
  hipLaunchKernelGGL(rvo3d::wpcache_kernel, dim3((unsigned)((EN + tb - 1) / tb)), dim3(tb), 0, s, P);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));  // the host staging vectors die here
  h->world_loaded = true;
  h->dv_valid = true;  // reset_kernel filed the des_vel of every start state
  h->g_valid = false;
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_reset(rvo3d_env* h, const uint8_t* env_mask, void* stream) {
  RVO3D_API_BEGIN
  DeviceGuard dg;
  int rc = check(h, true, dg);
  if (rc) return rc;
  const size_t EN = (size_t)h->P.E * h->P.N;
  h->g_valid = false;  // positions change: the stage-G words on file are stale
  hipLaunchKernelGGL(rvo3d::reset_kernel, dim3((unsigned)((EN + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), h->P, env_mask, (const uint8_t*)nullptr);
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_reset_drones(rvo3d_env* h, const uint8_t* drone_mask, void* stream) {
  RVO3D_API_BEGIN
  DeviceGuard dg;
  int rc = check(h, true, dg);
  if (rc) return rc;
  if (!drone_mask) return fail(RVO3D_ERR_INVALID, "drone_mask is required");
  const size_t EN = (size_t)h->P.E * h->P.N;
  h->g_valid = false;
  hipLaunchKernelGGL(rvo3d::reset_kernel, dim3((unsigned)((EN + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), h->P, (const uint8_t*)nullptr, drone_mask);
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_observe(rvo3d_env* h, float* obs, int32_t* vo_count, void* stream) {
  RVO3D_API_BEGIN
  DeviceGuard dg;
  int rc = check(h, true, dg);
  if (rc) return rc;
  if (!obs || !vo_count) return fail(RVO3D_ERR_INVALID, "obs / vo_count are required");
  Params P = h->P;
  P.obs = obs; P.vo_count = vo_count;
  P.zf16 = (h->cold.zf_q != 0 && (reinterpret_cast<uintptr_t>(obs) & 15) == 0 && P.nm > 0) ? 1 : 0;
  rc = launch<rvo3d::kObserve>(h, P, static_cast<hipStream_t>(stream));
  if (rc == RVO3D_OK) h->dv_valid = h->g_valid = true;  // observe files des_vel and the stage-G words
  return rc;
  RVO3D_API_END
}

static int step_common(rvo3d_env* h, const void* actions, int32_t action_dtype, float* obs,
                       int32_t* vo_count, float* reward, uint8_t* done, uint8_t* info,
                       uint8_t* finish, uint8_t* reset_mask, bool autoreset, void* stream) {
  RVO3D_API_BEGIN
  DeviceGuard dg;
  int rc = check(h, true, dg);
  if (rc) return rc;
  if (!actions || !obs || !vo_count || !reward || !done || !info || !finish)
    return fail(RVO3D_ERR_INVALID, "null I/O pointer");
  if (action_dtype != RVO3D_F32 && action_dtype != RVO3D_F64)
    return fail(RVO3D_ERR_INVALID, "action_dtype must be RVO3D_F32 or RVO3D_F64");
  Params P = h->P;
  P.actions = actions; P.action_f64 = action_dtype == RVO3D_F64;
  P.obs = obs; P.vo_count = vo_count; P.reward = reward;
  P.zf16 = (h->cold.zf_q != 0 && (reinterpret_cast<uintptr_t>(obs) & 15) == 0 && P.nm > 0) ? 1 : 0;
  P.done = done; P.info = info; P.finish = finish; P.reset_mask = reset_mask;
  P.dv_cached = h->dv_valid ? 1 : 0;
  P.g_cached = h->g_valid ? 1 : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = autoreset ? launch<rvo3d::kStepAutoReset>(h, P, s) : launch<rvo3d::kStep>(h, P, s);
  if (rc == RVO3D_OK) h->dv_valid = h->g_valid = true;  // every step files both for the state it ends in
  return rc;
  RVO3D_API_END
}

int rvo3d_step(rvo3d_env* h, const void* actions, int32_t action_dtype, float* obs,
               int32_t* vo_count, float* reward, uint8_t* done, uint8_t* info, uint8_t* finish,
               void* stream) {
  RVO3D_API_BEGIN
  return step_common(h, actions, action_dtype, obs, vo_count, reward, done, info, finish, nullptr,
                     false, stream);
  RVO3D_API_END
}

static int step_policy_common(rvo3d_env* h, const float* a_inc, float acceler, float* obs,
                              int32_t* vo_count, float* reward, uint8_t* done, uint8_t* info,
                              uint8_t* finish, uint8_t* reset_mask, bool autoreset, void* stream) {
  RVO3D_API_BEGIN
  if (!h) return fail(RVO3D_ERR_INVALID, "null handle");
  h->P.action_mode = 1;
  h->P.acceler = acceler;
  const int rc = step_common(h, a_inc, RVO3D_F32, obs, vo_count, reward, done, info, finish,
                             reset_mask, autoreset, stream);
  h->P.action_mode = 0;
  return rc;
  RVO3D_API_END
}

int rvo3d_step_policy(rvo3d_env* h, const float* a_inc, float acceler, float* obs,
                      int32_t* vo_count, float* reward, uint8_t* done, uint8_t* info,
                      uint8_t* finish, uint8_t* reset_mask, int32_t autoreset, void* stream) {
  RVO3D_API_BEGIN
  return step_policy_common(h, a_inc, acceler, obs, vo_count, reward, done, info, finish,
                            reset_mask, autoreset != 0, stream);
  RVO3D_API_END
}

int rvo3d_step_autoreset(rvo3d_env* h, const void* actions, int32_t action_dtype, float* obs,
                         int32_t* vo_count, float* reward, uint8_t* done, uint8_t* info,
                         uint8_t* finish, uint8_t* reset_mask, void* stream) {
  RVO3D_API_BEGIN
  return step_common(h, actions, action_dtype, obs, vo_count, reward, done, info, finish,
                     reset_mask, true, stream);
  RVO3D_API_END
}

int rvo3d_policy_sample(const rvo3d_policy_heads* hd, int64_t rows, float std_factor, uint64_t seed,
                        uint64_t step, float* act, float* logp, float* val, float* dbg_mu, float* dbg_raw,
                        void* stream) {
  RVO3D_API_BEGIN
  if (!hd || !hd->h_pi || !hd->h_v || !hd->log_std || !act || !logp || !val)
    return fail(RVO3D_ERR_INVALID, "null pointer");
  if (rows < 0) return fail(RVO3D_ERR_INVALID, "rows < 0");
  if (rows == 0) return RVO3D_OK;
  rvo3d::PolicySampleArgs A;
  A.h_pi = hd->h_pi; A.h_v = hd->h_v; A.ld_pi = hd->ld_pi; A.ld_v = hd->ld_v;
  A.hidden = hd->hidden; A.tanh_out = hd->hidden == 0 ? 0 : hd->tanh_out;  // (mu given: already activated)
  A.w_pi = hd->w_pi; A.b_pi = hd->b_pi; A.w_v = hd->w_v; A.b_v = hd->b_v; A.log_std = hd->log_std;
  A.std_factor = std_factor; A.seed = seed; A.step = step; A.rows = rows;
  A.step_dev = g_step_dev.load();
  A.act = act; A.logp = logp; A.val = val; A.dbg_mu = dbg_mu; A.dbg_raw = dbg_raw;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hd->hidden == 0) {
    if (hd->ld_pi < 3 || hd->ld_v < 1) return fail(RVO3D_ERR_INVALID, "hidden == 0 needs ld_pi >= 3 and ld_v >= 1");
    hipLaunchKernelGGL(rvo3d::policy_sample_direct_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, A);
  } else {
    if (!hd->w_pi || !hd->b_pi || !hd->w_v || !hd->b_v) return fail(RVO3D_ERR_INVALID, "head weights are required");
    const bool bf = hd->dtype == RVO3D_BF16;
    if (!bf && hd->dtype != RVO3D_F32) return fail(RVO3D_ERR_INVALID, "dtype must be RVO3D_F32 or RVO3D_BF16");
    const int per_chunk = bf ? 256 : 128;  // 32 lanes x 16 bytes
    const int nch = hd->hidden / per_chunk;
    if (hd->hidden % per_chunk != 0 || (nch != 1 && nch != 2 && nch != 4 && nch != 8) || hd->hidden > 1024)
      return fail(RVO3D_ERR_INVALID, "hidden must be 128 / 256 / 512 / 1024 (float32) or 256 / 512 / 1024 (bfloat16)");
    const size_t es = bf ? 2 : 4;
    if (hd->ld_pi < hd->hidden || hd->ld_v < hd->hidden || (hd->ld_pi * es) % 16 || (hd->ld_v * es) % 16 ||
        (reinterpret_cast<uintptr_t>(hd->h_pi) & 15) || (reinterpret_cast<uintptr_t>(hd->h_v) & 15))
      return fail(RVO3D_ERR_INVALID, "hidden activations must be 16-byte aligned rows of at least `hidden` elements");
    const dim3 grid((unsigned)((rows + 127) / 128)), blk(256);  // 4 waves x 32 rows
#define RVO3D_PS(T, N) hipLaunchKernelGGL((rvo3d::policy_sample_kernel<T, N>), grid, blk, 0, s, A)
    if (bf) {
      if (nch == 1) RVO3D_PS(rvo3d::bf16_t, 1); else if (nch == 2) RVO3D_PS(rvo3d::bf16_t, 2); else RVO3D_PS(rvo3d::bf16_t, 4);
    } else {
      if (nch == 1) RVO3D_PS(float, 1); else if (nch == 2) RVO3D_PS(float, 2); else if (nch == 4) RVO3D_PS(float, 4); else RVO3D_PS(float, 8);
    }
#undef RVO3D_PS
  }
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
  RVO3D_API_END
}

int64_t rvo3d_policy_mlp_blob_bytes(int32_t obs_width) {
  if (obs_width < 1 || obs_width > 126) return -1;
  return 2 * rvo3d::mlp_net_bytes(rvo3d::mlp_ks1(obs_width));
}

int rvo3d_policy_mlp_pack(const rvo3d_mlp_weights* pi, const rvo3d_mlp_weights* v, int32_t obs_width, void* blob,
                          void* stream) {
  RVO3D_API_BEGIN
  if (!pi || !v || !blob) return fail(RVO3D_ERR_INVALID, "null pointer");
  if (obs_width < 1 || obs_width > 126) return fail(RVO3D_ERR_INVALID, "obs_width must be 1..126");
  if (reinterpret_cast<uintptr_t>(blob) & 15) return fail(RVO3D_ERR_INVALID, "blob must be 16-byte aligned");
  const rvo3d_mlp_weights* n[2] = {pi, v};
  rvo3d::MlpPackArgs A;
  A.k_in = obs_width; A.ks1 = rvo3d::mlp_ks1(obs_width);
  for (int i = 0; i < 2; ++i) {
    if (!n[i]->w1 || !n[i]->b1 || !n[i]->w2 || !n[i]->b2 || !n[i]->w3 || !n[i]->b3)
      return fail(RVO3D_ERR_INVALID, "null weight pointer");
    A.w1[i] = n[i]->w1; A.b1[i] = n[i]->b1; A.w2[i] = n[i]->w2; A.b2[i] = n[i]->b2; A.w3[i] = n[i]->w3; A.b3[i] = n[i]->b3;
  }
  A.blob = static_cast<unsigned char*>(blob);
  hipLaunchKernelGGL(rvo3d::mlp_pack_kernel, dim3(64, 2), dim3(256), 0, static_cast<hipStream_t>(stream), A);
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_policy_mlp_sample(const void* blob, int32_t obs_width, const float* obs, int64_t obs_ld, int64_t rows,
                            const int32_t* vo_count, int32_t state_dim, int32_t row_dim, int32_t tanh_out, const float* log_std, float std_factor, uint64_t seed, uint64_t step,
                            float* act, float* logp, float* val, float* dbg_mu, float* dbg_raw, void* stream) {
  RVO3D_API_BEGIN
  if (!blob || !obs || !log_std || !act || !logp || !val) return fail(RVO3D_ERR_INVALID, "null pointer");
  if (obs_width < 1 || obs_width > 126) return fail(RVO3D_ERR_INVALID, "obs_width must be 1..126");
  if (rows < 0 || obs_ld < obs_width) return fail(RVO3D_ERR_INVALID, "rows < 0 or obs_ld < obs_width");
  if (rows > 0 && ((rows - 1) * obs_ld + obs_width) * 4 > (int64_t)0x7fffffff)
    return fail(RVO3D_ERR_INVALID, "the observation array must stay below 2 GiB per call (32-bit buffer offsets): split the rows");
  if (reinterpret_cast<uintptr_t>(obs) & 3) return fail(RVO3D_ERR_INVALID, "obs must be 4-byte aligned");
  if (reinterpret_cast<uintptr_t>(blob) & 15) return fail(RVO3D_ERR_INVALID, "blob must be 16-byte aligned");
  if (rows == 0) return RVO3D_OK;
  int dev = 0, cus = 0;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int ks1 = rvo3d::mlp_ks1(obs_width);
  rvo3d::PolicyMlpArgs A;
  A.blob = static_cast<const unsigned char*>(blob); A.net_bytes = rvo3d::mlp_net_bytes(ks1);
  A.obs = obs; A.ld_obs = obs_ld; A.k_in = obs_width;
  A.cnt = vo_count; A.state_dim = state_dim; A.row_dim = row_dim;
  if (vo_count && (state_dim < 0 || row_dim < 1 || state_dim > obs_width))
    return fail(RVO3D_ERR_INVALID, "vo_count needs 0 <= state_dim <= obs_width and row_dim >= 1");
  A.S = rvo3d::PolicySampleArgs{};
  A.S.tanh_out = tanh_out; A.S.log_std = log_std; A.S.std_factor = std_factor; A.S.seed = seed; A.S.step = step;
  A.S.step_dev = g_step_dev.load();
  A.S.rows = rows; A.S.act = act; A.S.logp = logp; A.S.val = val; A.S.dbg_mu = dbg_mu; A.S.dbg_raw = dbg_raw;
  // one workgroup per CU, half of them per network; every wave takes 64 rows per trip
  const int64_t nchunks = (rows + 63) / 64;
  int64_t G = (nchunks + kMlpWaves - 1) / kMlpWaves;
  const int64_t Gmax = cus >= 2 ? cus / 2 : 1;
  if (G > Gmax) G = Gmax;
  const unsigned grid = (unsigned)(2 * G);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (ks1) {
    case 1: return launch_policy_mlp<1>(A, grid, s);
    case 2: return launch_policy_mlp<2>(A, grid, s);
    case 3: return launch_policy_mlp<3>(A, grid, s);
    case 4: return launch_policy_mlp<4>(A, grid, s);
    case 5: return launch_policy_mlp<5>(A, grid, s);
    case 6: return launch_policy_mlp<6>(A, grid, s);
    case 7: return launch_policy_mlp<7>(A, grid, s);
    default: return launch_policy_mlp<8>(A, grid, s);
  }
  RVO3D_API_END
}

int rvo3d_reader_zero_features(const float* obs, int64_t obs_ld, int64_t rows, int32_t state_dim, int32_t feat_dim,
                               const float* ln_w, const float* ln_b, float sum_h0, float sumsq_h0, float ln_eps,
                               float* out, int64_t out_ld, const int32_t* vo_count, int32_t* list, int32_t* count,
                               void* stream) {
  RVO3D_API_BEGIN
  if (!obs || !ln_w || !ln_b || !out) return fail(RVO3D_ERR_INVALID, "null pointer");
  if ((vo_count != nullptr) != (list != nullptr) || (vo_count != nullptr) != (count != nullptr))
    return fail(RVO3D_ERR_INVALID, "vo_count, list and count go together");
  if (state_dim < 1 || state_dim > rvo3d::kReaderMaxSd || feat_dim <= state_dim)
    return fail(RVO3D_ERR_INVALID, "need 1 <= state_dim <= 32 and feat_dim > state_dim");
  // (the kernel reads the state in 16-byte pieces: the last one may take up to three floats beyond state_dim, still inside the row)
  if (rows < 0 || obs_ld < (state_dim + 3) / 4 * 4 || out_ld < state_dim + 8)
    return fail(RVO3D_ERR_INVALID, "rows / row strides too small");
  if (rows == 0) return RVO3D_OK;
  rvo3d::ZeroFeatArgs A{obs, obs_ld, rows, state_dim, feat_dim, ln_w, ln_b, sum_h0, sumsq_h0, ln_eps, out, out_ld,
                        vo_count, list, count};
  hipLaunchKernelGGL(rvo3d::reader_zero_features_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), A);
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_policy_rows(const rvo3d_rnn_policy* net, const float* obs, int64_t obs_ld, const int32_t* vo_count,
                      const int32_t* list, int32_t* count, int32_t* done_blocks, int32_t tanh_out, const float* log_std,
                      float std_factor, uint64_t seed, uint64_t step, float* act, float* logp, float* val, void* stream) {
  RVO3D_API_BEGIN
  if (!net || !obs || !vo_count || !list || !count || !done_blocks || !log_std || !act || !logp || !val)
    return fail(RVO3D_ERR_INVALID, "null pointer");
  if (!net->w_ih_f || !net->w_hh_f || !net->b_ih_f || !net->b_hh_f || !net->ln_w || !net->ln_b)
    return fail(RVO3D_ERR_INVALID, "null reader weight");
  const bool bi = net->w_ih_r != nullptr;
  if (bi != (net->w_hh_r != nullptr) || bi != (net->b_ih_r != nullptr) || bi != (net->b_hh_r != nullptr))
    return fail(RVO3D_ERR_INVALID, "the reverse direction needs all four of w_ih_r / w_hh_r / b_ih_r / b_hh_r");
  if (net->hidden < 1 || net->hidden > 256 || net->in_dim < 1 || net->in_dim > rvo3d::kReaderMaxIn || net->state_dim < 1 ||
      net->state_dim > rvo3d::kReaderMaxSd || net->slots < 1 || net->slots > 16)
    return fail(RVO3D_ERR_INVALID, "hidden <= 256, in_dim <= 16, state_dim <= 32, slots <= 16");
  if (obs_ld < net->state_dim + net->slots * net->in_dim) return fail(RVO3D_ERR_INVALID, "obs_ld too small");
  const rvo3d_mlp_weights* m[2] = {&net->pi, &net->v};
  rvo3d::PolicyRowsArgs A;
  A.obs = obs; A.obs_ld = obs_ld; A.cnt = vo_count; A.list = list; A.count = count; A.done_blocks = done_blocks;
  A.state_dim = net->state_dim; A.in_dim = net->in_dim; A.H = net->hidden; A.slots = net->slots;
  A.w_ih[0] = net->w_ih_f; A.w_hh[0] = net->w_hh_f; A.b_ih[0] = net->b_ih_f; A.b_hh[0] = net->b_hh_f;
  A.w_ih[1] = net->w_ih_r; A.w_hh[1] = net->w_hh_r; A.b_ih[1] = net->b_ih_r; A.b_hh[1] = net->b_hh_r;
  A.ln_w = net->ln_w; A.ln_b = net->ln_b; A.eps = net->ln_eps;
  for (int i = 0; i < 2; ++i) {
    if (!m[i]->w1 || !m[i]->b1 || !m[i]->w2 || !m[i]->b2 || !m[i]->w3 || !m[i]->b3)
      return fail(RVO3D_ERR_INVALID, "null head weight");
    A.w1[i] = m[i]->w1; A.b1[i] = m[i]->b1; A.w2[i] = m[i]->w2; A.b2[i] = m[i]->b2; A.w3[i] = m[i]->w3; A.b3[i] = m[i]->b3;
  }
  A.S = rvo3d::PolicySampleArgs{};
  A.S.tanh_out = tanh_out; A.S.log_std = log_std; A.S.std_factor = std_factor; A.S.seed = seed; A.S.step = step;
  A.S.step_dev = g_step_dev.load();
  A.S.act = act; A.S.logp = logp; A.S.val = val;
  hipLaunchKernelGGL(rvo3d::policy_rows_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), A);
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_reader_first_step(const rvo3d_gru_reader* rd, const float* obs, int64_t obs_ld, int64_t rows, void* feat,
                            int32_t feat_dtype, int64_t feat_ld, void* stream) {
  RVO3D_API_BEGIN
  if (!rd || !obs || !feat || !rd->w_ih_f || !rd->b_ih_f || !rd->b_hh_f || !rd->ln_w || !rd->ln_b)
    return fail(RVO3D_ERR_INVALID, "null pointer");
  if ((rd->w_ih_r != nullptr) != (rd->b_ih_r != nullptr) || (rd->w_ih_r != nullptr) != (rd->b_hh_r != nullptr))
    return fail(RVO3D_ERR_INVALID, "the reverse direction needs all three of w_ih_r / b_ih_r / b_hh_r");
  if (rd->hidden < 64 || rd->hidden > 256 || rd->hidden % 64 != 0 || rd->in_dim != 9 || rd->state_dim < 0 ||
      rd->state_dim > rvo3d::kReaderMaxSd)
    return fail(RVO3D_ERR_INVALID, "hidden must be 64 / 128 / 192 / 256, in_dim 9, state_dim <= 32");
  if (feat_dtype != RVO3D_F32 && feat_dtype != RVO3D_BF16) return fail(RVO3D_ERR_INVALID, "feat_dtype must be RVO3D_F32 or RVO3D_BF16");
  if (rows < 0 || obs_ld < rd->state_dim + rd->in_dim || feat_ld < rd->state_dim + rd->hidden)
    return fail(RVO3D_ERR_INVALID, "rows / row strides too small");
  if (feat_dtype == RVO3D_BF16 && ((reinterpret_cast<uintptr_t>(feat) & 7) || (feat_ld & 3)))
    return fail(RVO3D_ERR_INVALID, "bf16 features: feat 8-byte aligned, feat_ld a multiple of 4");
  if (rows == 0) return RVO3D_OK;
  rvo3d::ReaderArgs A;
  A.w_ih_f = rd->w_ih_f; A.b_ih_f = rd->b_ih_f; A.b_hh_f = rd->b_hh_f;
  A.w_ih_r = rd->w_ih_r; A.b_ih_r = rd->b_ih_r; A.b_hh_r = rd->b_hh_r;
  A.ln_w = rd->ln_w; A.ln_b = rd->ln_b; A.H = rd->hidden; A.IN = rd->in_dim; A.SD = rd->state_dim; A.eps = rd->ln_eps;
  A.obs = obs; A.obs_ld = obs_ld; A.rows = rows; A.feat = feat; A.feat_bf16 = feat_dtype == RVO3D_BF16; A.feat_ld = feat_ld;
  const int64_t groups = (rows + rvo3d::kReaderRows - 1) / rvo3d::kReaderRows;
  // a few workgroups per CU, each looping over row groups: the unit's weights are loaded once per workgroup
  const unsigned grid = (unsigned)(groups < 256 * 8 ? groups : 256 * 8);
  hipLaunchKernelGGL((rvo3d::reader_first_step_kernel<9>), dim3(grid), dim3((unsigned)rd->hidden), 0,
                     static_cast<hipStream_t>(stream), A);
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_rollout_set_step_counter(uint64_t* device_counter) {
  RVO3D_API_BEGIN
  g_step_dev.store(device_counter);
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_rollout_account(int32_t E, int32_t N, const float* reward, const uint8_t* done, const uint8_t* finish,
                          int32_t sanitize, int32_t max_ep_len, int32_t epoch_end, float* rew_slot, float* ep_ret,
                          int32_t* ep_len, uint8_t* cut_slot, uint8_t* extra_mask, double* sums, int32_t* any_extra,
                          void* stream) {
  RVO3D_API_BEGIN
  if (E < 1 || N < 1 || N > rvo3d::kMaxThreads) return fail(RVO3D_ERR_INVALID, "need num_envs >= 1 and 1 <= num_drones <= 512");
  if (!reward || !done || !finish || !rew_slot || !ep_ret || !ep_len || !cut_slot || !extra_mask || !sums || !any_extra)
    return fail(RVO3D_ERR_INVALID, "null pointer");
  rvo3d::AccountArgs A{E, N, reward, done, finish, sanitize, max_ep_len, epoch_end, rew_slot, ep_ret, ep_len,
                       cut_slot, extra_mask, sums, any_extra, g_step_dev.load()};
  hipLaunchKernelGGL(rvo3d::rollout_account_kernel, dim3((unsigned)E), dim3((unsigned)align_up((size_t)N, 64)), 0,
                     static_cast<hipStream_t>(stream), A);
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_set_reward_f64(rvo3d_env* h, double* reward64) {
  RVO3D_API_BEGIN
  if (!h) return fail(RVO3D_ERR_INVALID, "null handle");
  h->P.reward64 = reward64;
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_des_vel(rvo3d_env* h, double* des_vel, void* stream) {
  RVO3D_API_BEGIN
  DeviceGuard dg;
  int rc = check(h, true, dg);
  if (rc) return rc;
  if (!des_vel) return fail(RVO3D_ERR_INVALID, "des_vel is required");
  const size_t EN = (size_t)h->P.E * h->P.N;
  hipLaunchKernelGGL(rvo3d::des_vel_kernel, dim3((unsigned)((EN + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), h->P, des_vel);
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_rvo_vel(rvo3d_env* h, const double* vmax, double acceler, double* out_vel, void* stream) {
  RVO3D_API_BEGIN
  DeviceGuard dg;
  int rc = check(h, true, dg);
  if (rc) return rc;
  if (!vmax || !out_vel) return fail(RVO3D_ERR_INVALID, "vmax / out_vel are required");
  if (!(acceler >= 0.0 && acceler <= 1.0))
    return fail(RVO3D_ERR_INVALID, "acceler must be in [0, 1] (at most 4 candidates per axis)");
  rvo3d::RvoVelArgs A;
  for (int k = 0; k < 3; ++k) A.vmax[k] = vmax[k];
  A.acceler = acceler;
  const int T = (int)align_up((size_t)h->P.N, 64);
  hipLaunchKernelGGL(rvo3d::rvo_vel_kernel, dim3((unsigned)h->P.E), dim3((unsigned)T),
                     (size_t)T * 8 * sizeof(double), static_cast<hipStream_t>(stream), h->P, A, out_vel);
  HIP_TRY(hipGetLastError());
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_state_ptrs(rvo3d_env* h, rvo3d_state_view* out) {
  RVO3D_API_BEGIN
  if (!h || !out) return fail(RVO3D_ERR_INVALID, "null argument");
  const Params& P = h->P;
  out->px = P.px(); out->py = P.py(); out->pz = P.pz(); out->vx = P.vx(); out->vy = P.vy(); out->vz = P.vz();
  out->yaw = P.yaw(); out->pitch = P.pitch(); out->real_len = P.real_len(); out->max_dev = P.max_dev();
  out->extra_len = P.extra_len(); out->wp_idx = P.wp_idx(); out->arrive = P.arrive(); out->dest = P.dest();
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_get_state(rvo3d_env* h, double* pos, double* vel, double* yaw, double* pitch,
                    double* real_len, double* max_dev, double* extra_len, int32_t* wp_idx,
                    uint8_t* arrive, uint8_t* dest, void* stream) {
  RVO3D_API_BEGIN
  DeviceGuard dg;
  int rc = check(h, true, dg);
  if (rc) return rc;
  const Params& P = h->P;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int EN = P.E * P.N;
  const dim3 grid((EN + 255) / 256), blk(256);
  if (pos) hipLaunchKernelGGL(rvo3d::soa_to_aos3, grid, blk, 0, s, P.px(), P.py(), P.pz(), pos, EN);
  if (vel) hipLaunchKernelGGL(rvo3d::soa_to_aos3, grid, blk, 0, s, P.vx(), P.vy(), P.vz(), vel, EN);
  HIP_TRY(hipGetLastError());
  const hipMemcpyKind k = hipMemcpyDeviceToDevice;
  if (yaw) HIP_TRY(hipMemcpyAsync(yaw, P.yaw(), (size_t)EN * 8, k, s));
  if (pitch) HIP_TRY(hipMemcpyAsync(pitch, P.pitch(), (size_t)EN * 8, k, s));
  if (real_len) HIP_TRY(hipMemcpyAsync(real_len, P.real_len(), (size_t)EN * 8, k, s));
  if (max_dev) HIP_TRY(hipMemcpyAsync(max_dev, P.max_dev(), (size_t)EN * 8, k, s));
  if (extra_len) HIP_TRY(hipMemcpyAsync(extra_len, P.extra_len(), (size_t)EN * 8, k, s));
  if (wp_idx) HIP_TRY(hipMemcpyAsync(wp_idx, P.wp_idx(), (size_t)EN * 4, k, s));
  if (arrive) HIP_TRY(hipMemcpyAsync(arrive, P.arrive(), (size_t)EN, k, s));
  if (dest) HIP_TRY(hipMemcpyAsync(dest, P.dest(), (size_t)EN, k, s));
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_set_state(rvo3d_env* h, const double* pos, const double* vel, const double* yaw,
                    const double* pitch, const double* real_len, const double* max_dev,
                    const double* extra_len, const int32_t* wp_idx, const uint8_t* arrive,
                    const uint8_t* dest, void* stream) {
  RVO3D_API_BEGIN
  DeviceGuard dg;
  int rc = check(h, true, dg);
  if (rc) return rc;
  h->dv_valid = h->g_valid = false;  // the next step recomputes the pre-move dronestate and stage G
  const Params& P = h->P;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int EN = P.E * P.N;
  const dim3 grid((EN + 255) / 256), blk(256);
  if (pos) hipLaunchKernelGGL(rvo3d::aos3_to_soa, grid, blk, 0, s, pos, P.px(), P.py(), P.pz(), EN);
  if (vel) hipLaunchKernelGGL(rvo3d::aos3_to_soa, grid, blk, 0, s, vel, P.vx(), P.vy(), P.vz(), EN);
  HIP_TRY(hipGetLastError());
  const hipMemcpyKind k = hipMemcpyDeviceToDevice;
  if (yaw) HIP_TRY(hipMemcpyAsync(P.yaw(), yaw, (size_t)EN * 8, k, s));
  if (pitch) HIP_TRY(hipMemcpyAsync(P.pitch(), pitch, (size_t)EN * 8, k, s));
  if (real_len) HIP_TRY(hipMemcpyAsync(P.real_len(), real_len, (size_t)EN * 8, k, s));
  if (max_dev) HIP_TRY(hipMemcpyAsync(P.max_dev(), max_dev, (size_t)EN * 8, k, s));
  if (extra_len) HIP_TRY(hipMemcpyAsync(P.extra_len(), extra_len, (size_t)EN * 8, k, s));
  if (wp_idx) {  // current / previous waypoint follow the index
    HIP_TRY(hipMemcpyAsync(P.wp_idx(), wp_idx, (size_t)EN * 4, k, s));
    hipLaunchKernelGGL(rvo3d::wpcache_kernel, grid, blk, 0, s, P);
  }
  if (arrive) HIP_TRY(hipMemcpyAsync(P.arrive(), arrive, (size_t)EN, k, s));
  if (dest) HIP_TRY(hipMemcpyAsync(P.dest(), dest, (size_t)EN, k, s));
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_error_flags(rvo3d_env* h, uint32_t* flags, void* stream) {
  RVO3D_API_BEGIN
  DeviceGuard dg;
  int rc = check(h, false, dg);
  if (rc) return rc;
  if (!flags) return fail(RVO3D_ERR_INVALID, "flags is required");
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(flags, h->P.err, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemsetAsync(h->P.err, 0, 4, s));
  HIP_TRY(hipStreamSynchronize(s));
  return RVO3D_OK;
  RVO3D_API_END
}

#ifdef RVO3D_DIAG
// diagnostics build only (librvo3d_hip_diag.so, tools/): attach a device buffer
// [blocks][16] of s_memtime stamps, or NULL to detach
int rvo3d_debug_stamps(rvo3d_env* h, unsigned long long* stamps) {
  RVO3D_API_BEGIN
  if (!h) return fail(RVO3D_ERR_INVALID, "null handle");
  h->P.dbg = stamps;
  return RVO3D_OK;
  RVO3D_API_END
}
#endif

int rvo3d_kernel_name(rvo3d_env* h, int32_t mode, char* buf, int32_t cap) {
  RVO3D_API_BEGIN
  if (!h || !buf || cap < 1) return fail(RVO3D_ERR_INVALID, "null handle / buffer");
  if (mode < 0 || mode > 2) return fail(RVO3D_ERR_INVALID, "mode: 0 observe, 1 step, 2 step + auto-reset");
  const Pick k = pick_kernel(h->P);
  std::snprintf(buf, (size_t)cap, "rvo3d::env_kernel<%d, %d, %d, %s, %s>", (int)mode, h->P.nw, k.nfix,
                h->P.env_train ? "true" : "false", k.pad ? "true" : "false");
  return RVO3D_OK;
  RVO3D_API_END
}

int rvo3d_launch_info(rvo3d_env* h, int32_t* threads, int32_t* envs_per_block, int32_t* blocks,
                      int32_t* lds_bytes) {
  RVO3D_API_BEGIN
  if (!h) return fail(RVO3D_ERR_INVALID, "null handle");
  if (threads) *threads = h->threads;
  if (envs_per_block) *envs_per_block = h->P.epb;
  if (blocks) *blocks = h->blocks;
  if (lds_bytes) *lds_bytes = h->lds;
  return RVO3D_OK;
  RVO3D_API_END
}

}  // extern "C"
