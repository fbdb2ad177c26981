// rvo3d_pairs.hpp -- The pair pipeline: exact pair evaluation (stage X2) and kept-row insertion, the fp32 stages G
// and X1, the symmetric sweep, the collision-only sweep and the incremental re-gating after resets.
// Part of the gfx950 device code (see rvo3d_device.hpp for the overview).
#pragma once

#include <type_traits>

#include "rvo3d_lds.hpp"

namespace rvo3d {

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
// TRAIN: rvo_inter.env_train as a compile-time constant (the env_train = False code - another
// collision threshold and the "math domain error" report - stays out of the training kernels).
template <bool TRAIN>
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
  const double thr = TRAIN ? R : (S.r - kExpRadius + Or);
  const double thr2 = thr * thr;
  bool coll;
  if (ssum < thr2 * (1.0 - 1e-15)) coll = thr >= 0;
  else if (ssum > thr2 * (1.0 + 1e-15)) coll = false;
  else coll = __builtin_sqrt(ssum) <= thr;
  if (coll) { o.collision = true; return o; }
  const double dotp = S.vx * rx + S.vy * ry + S.vz * rz;
  if (dotp <= 0) return o;
  if (!TRAIN && d2 < R * R * (1.0 + 1e-12)) {
    // env_train = False, r - 0.2 + mr < dis < r + mr, approaching: get_alpha's
    // asin((r + mr) / norm) raises ValueError("math domain error") in the reference
    // (vel_obs3D.py:13) and aborts the step; here the pair is "no VO" and the event is
    // reported through the error word (RVO3D_FLAG_DOMAIN_ERROR)
    // (r + mr) / norm > 1  <=>  norm < r + mr (both correctly rounded): no division, and the
    // square root only inside the 1e-15 band around (r + mr)^2
    const double R2 = R * R;
    if (d2 < R2 * (1.0 - 1e-15) || (d2 < R2 * (1.0 + 1e-15) && __builtin_sqrt(d2) < R)) {
      atomicOr(P.err, 2u);
      return o;
    }
  }
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
  // (g made opaque: otherwise the two row-scratch addresses are hoisted out of the exact-stage
  // loop and carried - spilled, at 128 / 256 drones - across it for this rare path)
  asm volatile("" : "+v"(g));
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

// Request-mask word w of the drone in slot `slot` (slot = its thread index).  Word-major, [NW][T]: the lanes of a
// wave touch consecutive 8-byte words - conflict-free - where a drone-major [T][NW] layout put them 8 NW bytes
// apart (NW = 4: four-way bank conflicts on every clear / hand-over / read of the masks).
template <int NW>
__device__ __forceinline__ int mi(const Lds& L, int slot, int w) {
  return NW == 1 ? slot : w * L.T + slot;
}

// ===== the pair pipeline: whole envs per workgroup, fp32 filters, exact stage on request =====
enum { WX = 0, WY, WZ, WVX, WVY, WVZ, WR, WKD, WAX, WAY, WAZ, WPRIO };

// fp32 image of one drone, written to both copies of its env segment (NW > 1: the second copy
// holds the first N/2 + 1 drones only, see f32_len_nw).
template <int NW>
__device__ __forceinline__ void stage_f32(const Params& P, const Lds& L, int el, int d,
                                          bool active, const double p[3], const double v[3],
                                          const double az[3], double r, double prio) {
  if (!active) return;
  const double cx = p[0] - P.cold().cen[0], cy = p[1] - P.cold().cen[1], cz = p[2] - P.cold().cen[2];
  const float fvx = (float)v[0], fvy = (float)v[1], fvz = (float)v[2];
  // kd: fp32 error bound of v.rel ("possibly approaching" is v.rel > -kd, seen from the other
  // side v.rel < kd).  A drone whose velocity is EXACTLY zero (reset, or speed clamped at 0,
  // drone.py:452 - common) has v.rel == 0 exactly, which the reference treats as "not
  // approaching" (rvo_inter.py:159: dot <= 0 returns): kd = -1 makes both tests fail, instead
  // of sending every in-range pair of two resting drones (w = 0: the cone filter cannot decide)
  // to the exact stage - 0.59 of the 0.60 requests per lane of the rows sweep were these.
  const bool vzero = v[0] == 0.0 && v[1] == 0.0 && v[2] == 0.0;
  const float val[12] = {(float)cx, (float)cy, (float)cz, fvx, fvy, fvz, (float)r,
                         vzero ? -1.0f
                               : P.cold().kdot * (__builtin_fabsf(fvx) + __builtin_fabsf(fvy) + __builtin_fabsf(fvz)) + 1e-30f,
                         (float)az[0], (float)az[1], (float)az[2], (float)prio};
  const int o = el * 2 * P.N + d, os = el * P.N + d;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    if (k == WX || k == WY || k == WZ || k == WR) {
      L.w[k][o] = val[k];
      if (NW == 1 || d <= (P.N >> 1)) L.w[k][o + P.N] = val[k];
    } else L.w[k][os] = val[k];
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

// Stage G arithmetic, shared by every place that decides "possibly in range" (gate_word,
// regate_resets): acc = dx^2 + dy^2 + dz^2 - t', one fma chain starting from -t' with
// t' = nextafter(T10 + band): the pair is possibly in range iff acc < 0 (d2 < t' <=> d2 <= T10 + band up to
// the rounding of the chain, which the doubled band covers), i.e. iff the SIGN BIT of acc is set -
// one v_alignbit per pair shifts it into the word, no compare / select / or.
__device__ __forceinline__ float gate_acc(float dx, float dy, float dz, float t10n) {
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, __builtin_fmaf(dx, dx, t10n)));
}
__device__ __forceinline__ uint32_t shift_in_sign(uint32_t m, float acc) {
  return __builtin_amdgcn_alignbit(m, __builtin_bit_cast(uint32_t, acc), 31);  // (m << 1) | sign(acc)
}

#ifndef RVO3D_G_UNROLL
#define RVO3D_G_UNROLL 8
#endif
// Stage G for the offsets of word w (packed fp32, two offsets per instruction), from the
// highest offset of the word down, so that bit b of the result is offset 32w + b + 1.
// TOUCHONLY: possibly touching (and in range), with the in-range word on the side; else:
// possibly in range.  Bits above the word's last offset may be set: callers mask with valid[].
template <bool TOUCHONLY, bool UNIFORM_R = false>
__device__ __forceinline__ uint32_t gate_word(const Params& P, const Lds& L, int o0, int w, int H,
                                              float mex, float mey, float mez, float mer,
                                              uint32_t* range_out = nullptr) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f sx = {mex, mex}, sy = {mey, mey}, sz = {mez, mez}, sr = {mer, mer};
  const v2f tn = {P.t10n, P.t10n};
  const int kend = (H - 32 * w) < 32 ? (H - 32 * w) : 32;  // offsets in this word
  uint32_t mr = 0u, mt = 0u;
  // one radius for every drone (the usual world): the touch bound is the same for all pairs,
  // the very value the per-pair arithmetic below would produce
  const v2f ru = sr + sr;
  const v2f rcu = __builtin_elementwise_fma(ru * ru, (v2f){1.00001f, 1.00001f},
                                            (v2f){P.bandn, P.bandn});
#pragma unroll RVO3D_G_UNROLL
  for (int b = (kend - 1) & ~1; b >= 0; b -= 2) {
    const int o = o0 + 32 * w + b + 1;
    const v2f dx = (v2f){L.w[WX][o], L.w[WX][o + 1]} - sx;
    const v2f dy = (v2f){L.w[WY][o], L.w[WY][o + 1]} - sy;
    const v2f dz = (v2f){L.w[WZ][o], L.w[WZ][o + 1]} - sz;
    v2f acc = __builtin_elementwise_fma(dx, dx, tn);
    acc = __builtin_elementwise_fma(dy, dy, acc);
    acc = __builtin_elementwise_fma(dz, dz, acc);  // == gate_acc per half
    mr = shift_in_sign(shift_in_sign(mr, acc.y), acc.x);
    if (TOUCHONLY) {
      // possibly touching: d2 <= (r + mr)^2 * 1.00001 + band, as the sign of
      // (d2 - t') - ((r + mr)^2 * 1.00001 + band - t')
      v2f rc = rcu;
      if (!UNIFORM_R) {
        const v2f rs = (v2f){L.w[WR][o], L.w[WR][o + 1]} + sr;
        rc = __builtin_elementwise_fma(rs * rs, (v2f){1.00001f, 1.00001f},
                                       (v2f){P.bandn, P.bandn});
      }
      const v2f at = acc - rc;
      mt = shift_in_sign(shift_in_sign(mt, at.y), at.x);
    }
  }
  if (TOUCHONLY) {
    if (range_out) *range_out = mr;
    return mt & mr;
  }
  return mr;
}

// Stage G for workgroups of exactly 64 NW drones (the compile-time 128 / 256 kernels): a drone's
// 32 NW offsets are NW words, and word s of drone e is computed by lane e + 32 s, which tests ITS
// 32 nearest followers (slots d + 1 .. d + 32) against NW "own" drones d, d - 32, d - 64, ... at
// once.  Every neighbour record is read from LDS once per lane instead of once per (lane, word):
// stage G at 256 drones was bound by LDS bandwidth (3 x 8 B per lane and two offsets, 16 waves per
// CU: 12 of every 32 cycles per SIMD, four SIMDs on one LDS), not by its arithmetic - which is
// the same as gate_word's, bit for bit.  The words travel to their owners through the request-mask
// slots (u64: possibly in range | possibly touching << 32); the caller synchronises.
#ifndef RVO3D_GS_UNROLL
#define RVO3D_GS_UNROLL 4
#endif
template <int NW, bool TOUCH, bool UNIFORM_R>
__device__ __forceinline__ void gate_words_shared(const Params& P, const Lds& L, int d) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  constexpr int N = 64 * NW;
  const v2f tn = {P.t10n, P.t10n};
  v2f ox[NW], oy[NW], oz[NW], orr[NW], rcu[NW];
  uint32_t mr[NW], mt[NW];
  int own[NW];
#pragma unroll
  for (int s = 0; s < NW; ++s) {
    int e = d - 32 * s;
    if (e < 0) e += N;
    own[s] = e;
    const float x = L.w[WX][e], y = L.w[WY][e], z = L.w[WZ][e], r = L.w[WR][e];
    ox[s] = (v2f){x, x}; oy[s] = (v2f){y, y}; oz[s] = (v2f){z, z}; orr[s] = (v2f){r, r};
    const v2f ru = orr[s] + orr[s];
    rcu[s] = __builtin_elementwise_fma(ru * ru, (v2f){1.00001f, 1.00001f}, (v2f){P.bandn, P.bandn});
    mr[s] = 0u; mt[s] = 0u;
  }
#pragma unroll RVO3D_GS_UNROLL
  for (int b = 30; b >= 0; b -= 2) {
    const int o = d + b + 1;
    const v2f X = {L.w[WX][o], L.w[WX][o + 1]}, Y = {L.w[WY][o], L.w[WY][o + 1]},
              Z = {L.w[WZ][o], L.w[WZ][o + 1]};
    v2f R = {0.f, 0.f};
    if (TOUCH && !UNIFORM_R) R = (v2f){L.w[WR][o], L.w[WR][o + 1]};
#pragma unroll
    for (int s = 0; s < NW; ++s) {
      const v2f dx = X - ox[s], dy = Y - oy[s], dz = Z - oz[s];
      v2f acc = __builtin_elementwise_fma(dx, dx, tn);
      acc = __builtin_elementwise_fma(dy, dy, acc);
      acc = __builtin_elementwise_fma(dz, dz, acc);
      mr[s] = shift_in_sign(shift_in_sign(mr[s], acc.y), acc.x);
      if (TOUCH) {
        v2f rc = rcu[s];
        if (!UNIFORM_R) {
          const v2f rs = R + orr[s];
          rc = __builtin_elementwise_fma(rs * rs, (v2f){1.00001f, 1.00001f}, (v2f){P.bandn, P.bandn});
        }
        const v2f at = acc - rc;
        mt[s] = shift_in_sign(shift_in_sign(mt[s], at.y), at.x);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < NW; ++s)
    L.mask2[mi<NW>(L, own[s], s)] =
        (unsigned long long)mr[s] | ((unsigned long long)(TOUCH ? (mt[s] & mr[s]) : 0u) << 32);
}

// The offsets one X1 loop walks, as a bit set of CW words (32 offsets each): pop-lowest, test, set.
template <int CW> struct OffsetSet;
template <> struct OffsetSet<1> {
  uint32_t a;
  __device__ __forceinline__ static OffsetSet load(const uint32_t* g) { return {g[0]}; }
  __device__ __forceinline__ static OffsetSet none() { return {0u}; }
  __device__ __forceinline__ bool any() const { return a != 0u; }
  __device__ __forceinline__ int lowest() const { return __builtin_ctz(a); }
  __device__ __forceinline__ void drop() { a &= a - 1u; }
  __device__ __forceinline__ void add(int b, int on) { a |= (uint32_t)(on & 1) << b; }
  __device__ __forceinline__ void store(uint32_t* g) const { g[0] = a; }
  __device__ __forceinline__ int count() const { return __builtin_popcount(a); }
};
template <> struct OffsetSet<2> {
  unsigned long long a;
  __device__ __forceinline__ static OffsetSet load(const uint32_t* g) {
    return {(unsigned long long)g[0] | ((unsigned long long)g[1] << 32)};
  }
  __device__ __forceinline__ static OffsetSet none() { return {0ull}; }
  __device__ __forceinline__ bool any() const { return a != 0ull; }
  __device__ __forceinline__ int lowest() const { return __builtin_ctzll(a); }
  __device__ __forceinline__ void drop() { a &= a - 1ull; }
  __device__ __forceinline__ void add(int b, int on) { a |= (unsigned long long)(on & 1) << b; }
  __device__ __forceinline__ void store(uint32_t* g) const { g[0] = (uint32_t)a; g[1] = (uint32_t)(a >> 32); }
  __device__ __forceinline__ int count() const { return __builtin_popcountll(a); }
};
template <> struct OffsetSet<3> {  // 96 offsets (192-drone ring): 64 + 32
  unsigned long long lo;
  uint32_t hi;
  __device__ __forceinline__ static OffsetSet load(const uint32_t* g) {
    return {(unsigned long long)g[0] | ((unsigned long long)g[1] << 32), g[2]};
  }
  __device__ __forceinline__ static OffsetSet none() { return {0ull, 0u}; }
  __device__ __forceinline__ bool any() const { return lo != 0ull || hi != 0u; }
  __device__ __forceinline__ int lowest() const { return lo != 0ull ? __builtin_ctzll(lo) : 64 + __builtin_ctz(hi); }
  __device__ __forceinline__ void drop() {
    if (lo != 0ull) lo &= lo - 1ull; else hi &= hi - 1u;
  }
  __device__ __forceinline__ void add(int b, int on) {
    if (b < 64) lo |= (unsigned long long)(on & 1) << b; else hi |= (uint32_t)(on & 1) << (b - 64);
  }
  __device__ __forceinline__ void store(uint32_t* g) const { g[0] = (uint32_t)lo; g[1] = (uint32_t)(lo >> 32); g[2] = hi; }
  __device__ __forceinline__ int count() const { return __builtin_popcountll(lo) + __builtin_popcount(hi); }
};
template <> struct OffsetSet<4> {
  unsigned long long lo, hi;
  __device__ __forceinline__ static OffsetSet load(const uint32_t* g) {
    return {(unsigned long long)g[0] | ((unsigned long long)g[1] << 32),
            (unsigned long long)g[2] | ((unsigned long long)g[3] << 32)};
  }
  __device__ __forceinline__ static OffsetSet none() { return {0ull, 0ull}; }
  __device__ __forceinline__ bool any() const { return (lo | hi) != 0ull; }
  __device__ __forceinline__ int lowest() const {
    return lo != 0ull ? __builtin_ctzll(lo) : 64 + __builtin_ctzll(hi);
  }
  __device__ __forceinline__ void drop() {
    if (lo != 0ull) lo &= lo - 1ull; else hi &= hi - 1ull;
  }
  __device__ __forceinline__ void add(int b, int on) {
    const unsigned long long v = (unsigned long long)(on & 1) << (b & 63);
    if (b < 64) lo |= v; else hi |= v;
  }
  __device__ __forceinline__ void store(uint32_t* g) const {
    g[0] = (uint32_t)lo; g[1] = (uint32_t)(lo >> 32); g[2] = (uint32_t)hi; g[3] = (uint32_t)(hi >> 32);
  }
  __device__ __forceinline__ int count() const { return __builtin_popcountll(lo) + __builtin_popcountll(hi); }
};

// Symmetric sweep: every unordered pair {i, j} of an env is examined once, by the
// drone whose index d satisfies j = d + k (mod N), 1 <= k <= N/2.
//   stage G  (packed fp32, all offsets): possibly in range;
//   stage X1 (fp32, candidates): possibly approaching / touching and a conservative
//            cone pre-filter, for both directions; survivors request the exact
//            evaluation from the owner (bit masks, LDS atomics);
//   stage X2 (fp64, requested pairs only): pair_eval.
// G and X1 only ever drop pairs that pair_eval would return "nothing" for.
// NW = ceil(N / 64): words per request mask (64 drones) and per offset mask (32 offsets).
template <int NW, bool ROWS, bool TOUCH, bool TRAIN, bool GSHARE = false>
__device__ __forceinline__ int sweep_env(const Params& P, const Lds& L, int lane, int el, int d,
                                         int g, bool active, const Drone& S, const double a[3],
                                         bool zero_act, bool& flag, double& tmin,
                                         bool& collision, uint32_t gw[NW], bool have_gw) {
  flag = false;
  tmin = __builtin_inf();
  int kept = 0;
  const int N = P.N, H = N >> 1;
  if (GSHARE && !have_gw) {
    // (have_gw is uniform over the workgroup: one env, and P.g_cached / "the env reset somebody")
    // stage G first, shared between the lanes (gate_words_shared); the words arrive in the
    // owners' request-mask slots, which the owners then clear for stage X1
    // (the last readers of the request masks - the exact stage of the previous sweep, each lane its
    // own slots - are at least one barrier behind: every caller stages or integrates in between)
    const bool far = L.far[el] != 0;
    // (every lane of the workgroup, ghosts of a padded env included: lane e + 32 s computes word s of drone e)
    if (!far) gate_words_shared<NW, false, false>(P, L, d);
    __syncthreads();
    if (active) {
      uint32_t valid[NW];
      valid_offsets<NW>(P.N, d, valid);
#pragma unroll
      for (int w = 0; w < NW; ++w)
        gw[w] = far ? valid[w] : ((uint32_t)L.mask2[mi<NW>(L, lane, w)] & valid[w]);
    }
    have_gw = true;
  }
#pragma unroll
  for (int w = 0; w < NW; ++w) L.mask2[mi<NW>(L, lane, w)] = 0ull;
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
    // stage G for every word, unless the words of this very state are on file (gw in, have_gw)
    if (ROWS && !TOUCH) RVO3D_STAMP(10);
    if (!have_gw) {
#pragma unroll
      for (int w = 0; w < NW; ++w)
        gw[w] = far ? valid[w] : (gate_word<false>(P, L, o0, w, H, mex, mey, mez, mer) & valid[w]);
    }
    if (ROWS && !TOUCH) RVO3D_STAMP(11);
    // stage X1, two candidate pairs per trip in packed fp32 (a lane with an odd count repeats
    // its last candidate: the requests are idempotent ORs).  One loop walks a lane's whole
    // candidate set up to 128 offsets (OffsetSet): fewer, fuller trips than one loop per word.
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f mex2 = {mex, mex}, mey2 = {mey, mey}, mez2 = {mez, mez}, mer2 = {mer, mer};
    const v2f mvx2 = {mvx, mvx}, mvy2 = {mvy, mvy}, mvz2 = {mvz, mvz};
    // (the cone filter below works with 2 w = 4 a - (v_i + v_j): every term of its comparisons is
    // homogeneous of degree 2 in w and a factor 2 is exact in binary, so the decisions are those of
    // w = 2 a - (v_i + v_j) / 2 bit for bit, without the three halvings per trip)
    const v2f tax = {4.f * max_, 4.f * max_}, tay = {4.f * may, 4.f * may},
              taz = {4.f * maz, 4.f * maz};
    const int fr = far ? 1 : 0;
#ifndef RVO3D_X1_CW
#define RVO3D_X1_CW 4
#endif
    // words per candidate loop: a lane's whole set in one loop up to 128 offsets (256 drones)
    constexpr int CW = NW < RVO3D_X1_CW ? NW : RVO3D_X1_CW;
#pragma unroll
    for (int w = 0; w < NW / CW; ++w) {
      OffsetSet<CW> cand = OffsetSet<CW>::load(gw + CW * w);
      if (RVO3D_ABLATED(64)) cand = OffsetSet<CW>::none();
      OffsetSet<CW> keep = OffsetSet<CW>::none();  // candidates with somebody possibly approaching (ROWS: filed for the next sweep A)
#ifdef RVO3D_DIAG
      if (P.dbg && ROWS && RVO3D_ABLATED(128)) {  // diagnostics build, RVO3D_ABLATE bit 128 = count: X1 candidates of this workgroup in the rows sweep (sum, max per lane)
        const int c = cand.count();
        atomicAdd(&P.dbg[(size_t)blockIdx.x * 32 + 24], (unsigned long long)c);
        atomicMax(&P.dbg[(size_t)blockIdx.x * 32 + 25], (unsigned long long)c);
      }
#endif
      while (cand.any()) {
        const int kb0 = cand.lowest();
        cand.drop();
        const bool two = cand.any();
        const int kb1 = two ? cand.lowest() : kb0;
        cand.drop();  // (of an empty set: still empty)
        const int off0 = 32 * CW * w + kb0 + 1, off1 = 32 * CW * w + kb1 + 1;
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
        const v2f hx = mvx2 + jvx, hy = mvy2 + jvy, hz = mvz2 + jvz;
        // 2 w_i, w_i = 2 a_i - (v_i + v_j) / 2  (get_PAA with equal priorities)
        const v2f wix = tax - hx, wiy = tay - hy, wiz = taz - hz;
        const v2f dpi = __builtin_elementwise_fma(
            dz, wiz, __builtin_elementwise_fma(dy, wiy, dx * wix));
        const v2f wi2 = __builtin_elementwise_fma(
            wiz, wiz, __builtin_elementwise_fma(wiy, wiy, wix * wix));
        // seen from j: rel -> -rel, 2 w_j = 4 a_j - (v_i + v_j)
        const v2f four2 = {4.f, 4.f};
        const v2f wjx = four2 * ajx - hx, wjy = four2 * ajy - hy, wjz = four2 * ajz - hz;
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
          /* lane predicates, combined with & | ! (no short circuit: no branches; the        \
             compiler keeps them as wave masks and combines them on the scalar unit) */       \
          const bool touch = TOUCH & (d2.c <= tch.c);                                         \
          const bool ai = vi.c > -mkd, aj = vj.c < jkd.c;                                     \
          const bool filt = (gap.c >= P.x1_gap) & (jprio.c == mprio);                         \
          const bool oi = si.c < (si.c < 0.f ? ci.c : ki.c);                                  \
          const bool oj = sj.c < (sj.c < 0.f ? cj.c : kj.c);                                  \
          pi = far | touch | (ai & !(filt & oi));                                             \
          pj = far | touch | (aj & !(filt & oj));                                             \
          if (ROWS) keep.add(kbit, (int)(far | ai | aj));                                     \
        }
        bool pi0, pj0, pi1, pj1;
        {
          const int kbit = kb0;
          RVO3D_X1_HALF(x, jd0, pi0, pj0)
        }
        {
          const int kbit = kb1;
          RVO3D_X1_HALF(y, jd1, pi1, pj1)
        }
#undef RVO3D_X1_HALF
        pi1 &= two; pj1 &= two;
        if (NW == 1) {
          m2r |= ((unsigned long long)pi0 << jd0) | ((unsigned long long)pi1 << jd1);
        } else {
          if (pi0) atomicOr(&L.mask2[mi<NW>(L, lane, jd0 >> 6)], 1ull << (jd0 & 63));
          if (pi1) atomicOr(&L.mask2[mi<NW>(L, lane, jd1 >> 6)], 1ull << (jd1 & 63));
        }
        if (pj0) atomicOr(&L.mask2[mi<NW>(L, el * N + jd0, d >> 6)], 1ull << (d & 63));
        if (pj1) atomicOr(&L.mask2[mi<NW>(L, el * N + jd1, d >> 6)], 1ull << (d & 63));
      }
      // The words filed for the next step's sweep A (same state, another action) keep only the
      // pairs in which somebody is possibly approaching: "approaching" does not depend on the
      // action, and sweep A (no touch test) asks nothing of the others - about half of the in-range
      // pairs drop out of its X1.
      if (ROWS) keep.store(gw + CW * w);
    }
  }
  if (ROWS && !TOUCH) RVO3D_STAMP(12);
  if (!ROWS) RVO3D_STAMP(14);
  __syncthreads();
#ifdef RVO3D_DIAG
  if (P.dbg && active && RVO3D_ABLATED(128)) {  // diagnostics build, bit 128 = count: X2 requests of this workgroup (sum, max per lane)
    int c = 0;
    for (int w = 0; w < NW; ++w) c += __builtin_popcountll(L.mask2[mi<NW>(L, lane, w)] | (w == 0 ? m2r : 0ull));
    atomicAdd(&P.dbg[(size_t)blockIdx.x * 32 + (ROWS ? 22 : 20)], (unsigned long long)c);
    atomicMax(&P.dbg[(size_t)blockIdx.x * 32 + (ROWS ? 23 : 21)], (unsigned long long)c);
  }
#endif
  if (active && !RVO3D_ABLATED(32)) {
    const int lbase = el * N;
#pragma unroll  // (a rolled loop over the words is smaller but 4 % slower at 128 and 256 drones)
    for (int w = 0; w < NW; ++w) {
      unsigned long long m2 = L.mask2[mi<NW>(L, lane, w)] | (w == 0 ? m2r : 0ull);
      while (m2) {  // stage X2: exact, requested pairs only
        const int j = 64 * w + __builtin_ctzll(m2);
        m2 &= m2 - 1;
        const PairOut po = pair_eval<TRAIN>(P, S, L, lbase + j, a);
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
template <int NW, bool TRAIN, bool GSHARE = false>
__device__ __forceinline__ bool collide_env(const Params& P, const Lds& L, int lane, int el,
                                            int d, bool active, const Drone& S,
                                            uint32_t gw[NW]) {
  const int N = P.N, H = N >> 1;
  uint32_t touchw[NW];  // GSHARE: the possibly-touching words, from gate_words_shared
  if (GSHARE) {
    // (the request masks were last read by sweep A's exact stage, before the integrate barriers)
    const bool far = L.far[el] != 0;  // uniform over the workgroup (one env)
    if (!far) {  // (every lane, ghosts included: see sweep_env)
      if (P.uniform_rp) gate_words_shared<NW, true, true>(P, L, d);
      else gate_words_shared<NW, true, false>(P, L, d);
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const unsigned long long v = L.mask2[mi<NW>(L, lane, w)];
      gw[w] = (uint32_t)v;
      touchw[w] = (uint32_t)(v >> 32);
      if (w) L.mask2[mi<NW>(L, lane, w)] = 0ull;
    }
  }
  L.mask2[mi<NW>(L, lane, 0)] = 0ull;
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
      if (GSHARE) {
        if (!far) cand = touchw[w] & valid[w];
        gw[w] = far ? valid[w] : (gw[w] & valid[w]);
      } else {
        gw[w] = valid[w];
        if (!far) {
          uint32_t rng;
          cand = (P.uniform_rp ? gate_word<true, true>(P, L, o0, w, H, mex, mey, mez, mer, &rng)
                               : gate_word<true, false>(P, L, o0, w, H, mex, mey, mez, mer, &rng)) & valid[w];
          gw[w] = rng & valid[w];
        }
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
        if (TRAIN) {
          ci = cj = dis <= S.r + Or;
        } else {  // rvo_inter.py:145-147: r - exp_radius + mr, evaluated from each side
          ci = dis <= S.r - kExpRadius + Or;
          cj = dis <= Or - kExpRadius + S.r;
          // the shell r - 0.2 + mr < dis < r + mr: the reference raises "math domain error"
          // for the side(s) that approach (see pair_eval); the rows sweep that follows only
          // sees the envs that did not reset, so the event is reported here as well
          const double R = S.r + Or;
          if (d2 < R * R * (1.0 + 1e-12) && __builtin_sqrt(d2) < R) {
            const bool ai = !ci && S.vx * rx + S.vy * ry + S.vz * rz > 0;
            const bool aj = !cj && L.vx[k] * -rx + L.vy[k] * -ry + L.vz[k] * -rz > 0;
            if (ai || aj) atomicOr(P.err, 2u);
          }
        }
        if (ci) coll = true;
        if (cj) atomicOr(&L.mask2[mi<NW>(L, k, 0)], 1ull);
      }
    }
  }
  __syncthreads();
  if (active && (L.mask2[mi<NW>(L, lane, 0)] & 1ull)) coll = true;
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
    const bool same = active && el == rel && tid != rl;
    const bool inr = same && __builtin_signbit(gate_acc(dx, dy, dz, P.t10n));
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

}  // namespace rvo3d
