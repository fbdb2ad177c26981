// rvo3d_step.hpp -- The environment step kernel and its per-drone parts: building gate, observation writers,
// zero fill, rewards, env_kernel<MODE, NW>.
// Part of the gfx950 device code (see rvo3d_device.hpp for the overview).
#pragma once

#include "rvo3d_pairs.hpp"

namespace rvo3d {

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
    // the count and the first seven entries in one 16-B load (lists are short: most cells have
    // none or one), the records of two listed buildings per round trip (entry 1 pads the batch:
    // testing a building twice changes nothing); longer lists are walked from memory
    const uint16_t* const cell = P.cold().bgrid + (size_t)(ix * gy + iy) * (kBgridK + 1);
    const uint4 c0 = *reinterpret_cast<const uint4*>(cell);
    const int cnt = (int)(c0.x & 0xffffu);
    if (cnt != 0xffff) {
      if (cnt >= 1) {
        const int b0 = (int)(c0.x >> 16), b1 = cnt >= 2 ? (int)(c0.y & 0xffffu) : b0;
        hit |= building_test(bld, b0, S, T5);
        hit |= building_test(bld, b1, S, T5);
      }
      if (cnt >= 3) {
        const int b0 = (int)(c0.y >> 16), b1 = cnt >= 4 ? (int)(c0.z & 0xffffu) : b0;
        hit |= building_test(bld, b0, S, T5);
        hit |= building_test(bld, b1, S, T5);
      }
      for (int k = 5; k <= cnt; ++k) hit |= building_test(bld, (int)cell[k], S, T5);
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
  // pin the four floats here: left alone the compiler sinks the final multiply and the conversion
  // below the sweep and keeps (or spills) the four doubles across it instead
  asm volatile("" : "+v"(t.dv0), "+v"(t.dv1), "+v"(t.dv2), "+v"(t.dev));
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

// 16-B row writer, part 1: the lane's proprioceptive floats go to LDS (float2 [row][6], laid
// over the fp32 image, which is dead by now) together with the start of the row's zero run in
// 8-B units; row_fill_pairs() then writes proprio and zeros of all rows with coalesced 16-B stores.
__device__ __forceinline__ void stage_row(const Params& P, const Lds& L, int tid, int g,
                                          const Drone& S, const ProprioTail& t, int kept) {
  const double v[8] = {S.x, S.y, S.z, S.vx, S.vy, S.vz, S.r, S.prio};
  float f[12];
  bool bad = t.bad;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    f[k] = round2_f32(v[k]);
    bad |= !finite_d(v[k]);
  }
  f[8] = t.dv0; f[9] = t.dv1; f[10] = t.dv2; f[11] = t.dev;
  float4* pro = reinterpret_cast<float4*>(L.w[0]) + 3 * tid;
  pro[0] = make_float4(f[0], f[1], f[2], f[3]);
  pro[1] = make_float4(f[4], f[5], f[6], f[7]);
  pro[2] = make_float4(f[8], f[9], f[10], f[11]);
  L.kept[tid] = ((12 + 9 * kept + 1) & ~1) >> 1;
  // with 8-B units an odd 9*kept leaves one float between the kept rows and the zero run
  if (((9 * kept) & 1) && kept < P.nm) P.obs[(size_t)g * P.W + 12 + 9 * kept] = 0.0f;
  if (bad) atomicOr(P.err, 1u);
}

// 16-B row writer, part 2 (W even, obs 16-B aligned).  Two consecutive rows (an even row and
// its successor, counted over the whole obs tensor) are 2 * W * 4 = 16 * q bytes starting on a
// 16-B boundary: q = W / 2 chunks of 16 B.  One wave-instruction stores one row pair: lane c
// owns chunk c of every pair, i.e. two fixed 8-B units (row of the pair, unit inside the row)
// computed once; per pair it looks up the zero runs of its one or two rows and assembles the
// chunk - proprio bytes from LDS, zeros inside a row's zero run, nothing inside the kept VO
// rows (their lane wrote them).  A half outside the workgroup's rows or inside kept rows turns
// the store into an 8-B one.  The waves of the workgroup take the pairs round-robin.
template <int NW>
__device__ __forceinline__ void row_fill_pairs(const Params& P, const Lds& L, int tid, int row0,
                                               int nrows) {
  const int q = P.W >> 1;  // 8-B units per row = 16-B chunks per row pair
  const int ln = tid & 63, wv = tid >> 6;
  const int nwv = NW == 1 ? 1 : (L.T >> 6);  // waves of this workgroup (<= NW; T = N rounded up to 64)
  const int pair0 = row0 >> 1;
  const int npairs = ((row0 + nrows + 1) >> 1) - pair0;
  const int rbase = 2 * pair0 - row0;  // local row of the first pair's first row: 0 or -1
  const float2* pro2 = reinterpret_cast<const float2*>(L.w[0]);
  const uint32_t pair_bytes = 8u * (uint32_t)P.W;
  char* const obsb = reinterpret_cast<char*>(P.obs) + (size_t)pair0 * pair_bytes;
  // Row indices -1 and nrows occur at the two ends of the range; the LDS words read for them are
  // valid memory next to the arrays and never used (in0 / in1 are false there).
  for (int cb = 0; cb < q; cb += 64) {  // one trip unless a row has more than 64 chunks
    const int c = cb + ln;
    const bool lane_on = c < q;
    const int h0 = 2 * c, h1 = 2 * c + 1;        // the chunk's halves, as units of the pair
    const int s0 = h0 >= q ? 1 : 0, s1 = h1 >= q ? 1 : 0;
    const int u0 = h0 - (s0 ? q : 0), u1 = h1 - (s1 ? q : 0);
    const bool p0 = u0 < 6, p1 = u1 < 6;         // proprio units
    const int pu0 = p0 ? u0 : 5, pu1 = p1 ? u1 : 5;
    uint32_t off = (uint32_t)wv * pair_bytes + 16u * (uint32_t)c;
    const uint32_t ostep = (uint32_t)nwv * pair_bytes;
    // four pairs per trip: their LDS lookups are issued together (one round trip, not four)
    // and the stores are two flat predicated regions (whole chunk: nearly always; one half:
    // only next to kept rows and at the ends of the range)
    for (int i0 = wv; i0 < npairs; i0 += 4 * nwv) {
      int z0[4], z1[4];
      float2 d0[4], d1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * nwv;  // may run past npairs: reads stay inside the LDS allocation
        const int lr0 = rbase + 2 * i + s0, lr1 = rbase + 2 * i + s1;
        z0[u] = L.kept[lr0]; z1[u] = L.kept[lr1];
        d0[u] = pro2[lr0 * 6 + pu0]; d1[u] = pro2[lr1 * 6 + pu1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * nwv;
        const int lr0 = rbase + 2 * i + s0, lr1 = rbase + 2 * i + s1;
        const bool live = lane_on & (i < npairs);
        const bool v0 = live & ((unsigned)lr0 < (unsigned)nrows) & (p0 | (u0 >= z0[u]));
        const bool v1 = live & ((unsigned)lr1 < (unsigned)nrows) & (p1 | (u1 >= z1[u]));
        const float2 a = p0 ? d0[u] : make_float2(0.f, 0.f);
        const float2 b = p1 ? d1[u] : make_float2(0.f, 0.f);
        char* const pc = obsb + (off + (uint32_t)u * ostep);
        if (v0 & v1) *reinterpret_cast<float4*>(pc) = make_float4(a.x, a.y, b.x, b.y);
        if (v0 != v1) *reinterpret_cast<float2*>(pc + (v1 ? 8 : 0)) = v1 ? b : a;
      }
      off += 4u * ostep;
    }
  }
}

// ---- two-phase row writer (fused auto-reset step; workgroup rows a multiple of 8, W even, >= 48) --
// The workgroup's rows are then a byte range that starts and ends on 64-B boundaries.  Seen in 64-B
// blocks, a block either holds proprio bytes of some row ("late": 2 of every ~6.4) or only
// VO-region bytes ("early").  The early blocks need no data at all: their zeros are stored BEFORE
// the rows sweep - 69 % of the observation bytes leave while the sweep computes instead of in the
// store burst at the end of the kernel - and a lane whose sweep then keeps VO rows writes them over
// the zeros afterwards (after s_waitcnt vmcnt(0): the zeros of its own wave have landed; workgroups
// of several waves drain before the sweep's second barrier).  The late blocks are written after the
// sweep as 128-B windows, one per row, from the proprio staged in LDS (part 2 below).
// Ordering: `s_waitcnt vmcnt(0)` (+ the workgroup barrier for several waves) makes the early zeros land before a
// kept row is stored over them.  That holds because the waves of a workgroup share one CU's memory pipeline - the
// library is built WITHOUT threadgroup-split mode (no -mtgsplit: hipcc's default), where waves of a workgroup
// could sit on different CUs and a store's completion would not order it against another CU's stores.
// Full workgroups only (the host tabulates the block pattern of a full workgroup, zf_iters > 0:
// W even, 48 <= W <= 128, rows per workgroup a multiple of 8); the last, partial workgroup of a
// launch and other widths take the row-pair writer.
__device__ __forceinline__ bool two_phase_rows(const Params& P, int row0, int nrows, int full_rows) {
  return P.zf16 && P.cold().zf_iters > 0 && nrows == full_rows && (row0 & 7) == 0;
}

// part 1: zeros of every 64-B block that holds no proprio byte.  One wave-instruction = 16 blocks
// (1 KB, whole lines).
template <int NW>
__device__ __forceinline__ void early_zero_blocks(const Params& P, const Lds& L, int tid, int row0,
                                                  int nrows) {
  const uint32_t rb = 4u * (uint32_t)P.W;  // row bytes (a multiple of 8)
  const int ln = tid & 63, wv = tid >> 6;
  const int nwv = NW == 1 ? 1 : (L.T >> 6);
  (void)nrows;  // a full workgroup (two_phase_rows)
  char* const base = reinterpret_cast<char*>(P.obs) + (size_t)row0 * rb;
  const uint32_t blk = (uint32_t)wv * 16u + ((uint32_t)ln >> 2);
  typedef float v4f __attribute__((ext_vector_type(4)));
  // which trips store is a property of the thread quad, tabulated by the host (rvo3d_create): no
  // per-trip offset arithmetic.  Streaming stores: whole 64-B blocks nobody reads back (-1 %; the
  // windows of part 2, whose lines are shared with the kept rows, are faster as ordinary stores)
  const int iters = P.cold().zf_iters;
  uint32_t m = P.cold().zmask[tid >> 2];
  char* p = base + (size_t)blk * 64u + 16u * ((uint32_t)ln & 3u);
  const size_t step = (size_t)1024 * (size_t)nwv;
  // four trips per loop pass (bits beyond the last trip are 0: nothing is stored for them)
  for (int i = 0; i < iters; i += 4, p += 4 * step, m >>= 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if ((m >> u) & 1u) {
#if RVO3D_ZERO_NT
        __builtin_nontemporal_store((v4f){0.f, 0.f, 0.f, 0.f}, reinterpret_cast<v4f*>(p + u * step));
#else
        *reinterpret_cast<v4f*>(p + u * step) = (v4f){0.f, 0.f, 0.f, 0.f};
#endif
      }
  }
}

// part 2: per row the 128-B window that starts at the 64-B block holding the row's first byte: the
// tail of the previous row's VO region (s bytes), the 48 proprio bytes, the head of this row's VO
// region.  Eight rows per wave-instruction, lane = (row, 16-B chunk); 8-B halves as in
// row_fill_pairs: proprio from LDS, zeros inside a zero run, nothing inside kept rows.
template <int NW>
__device__ __forceinline__ void late_row_windows(const Params& P, const Lds& L, int tid, int row0,
                                                 int nrows) {
  const uint32_t rb = 4u * (uint32_t)P.W, q = rb >> 3;  // row bytes, 8-B units per row
  const int ln = tid & 63, wv = tid >> 6;
  const int nwv = NW == 1 ? 1 : (L.T >> 6);
  const float2* pro2 = reinterpret_cast<const float2*>(L.w[0]);
  char* const base = reinterpret_cast<char*>(P.obs) + (size_t)row0 * rb;
  const int ch = ln & 7;
  const int r0 = wv * 8 + (ln >> 3);
  // A trip advances every lane by 8 * nwv rows = a multiple of 64 B (W is even): where the lane's
  // chunk sits relative to its row - which of its two 8-B units belong to the previous row's tail,
  // which are proprio, which unit of the row they are - is the same in every trip.  Only the row
  // itself (its kept count, its proprio floats) changes.
  const uint32_t rs0 = rb * (uint32_t)r0;      // row start of the first trip, relative to the workgroup's range
  const uint32_t s8 = (rs0 & 63u) >> 3;        // units of the window that belong to row r - 1
  const uint32_t ua = 2u * (uint32_t)ch, ub = ua + 1u;
  const bool pa = ua < s8, pb = ub < s8;       // in the previous row's tail (never in row 0: s8 = 0 there)
  const uint32_t uia = pa ? q - s8 + ua : ua - s8, uib = pb ? q - s8 + ub : ub - s8;
  const bool proa = uia < 6u, prob = uib < 6u;
  const int pia = (int)(proa ? uia : 5u), pib = (int)(prob ? uib : 5u);
  const int da_row = pa ? 1 : 0, db_row = pb ? 1 : 0;
  char* pc = base + (rs0 & ~63u) + 16u * (uint32_t)ch;
  const size_t step = (size_t)8 * (size_t)nwv * rb;
  for (int r = r0; r < nrows; r += 8 * nwv, pc += step) {
    const int ra = r - da_row, rbw = r - db_row;
    const int za = L.kept[ra], zb = L.kept[rbw];
    const float2 da = pro2[ra * 6 + pia], db = pro2[rbw * 6 + pib];
    const bool va = proa | (uia >= (uint32_t)za), vb = prob | (uib >= (uint32_t)zb);
    const float2 a = proa ? da : make_float2(0.f, 0.f), b = prob ? db : make_float2(0.f, 0.f);
    if (va & vb) *reinterpret_cast<float4*>(pc) = make_float4(a.x, a.y, b.x, b.y);
    if (va != vb) *reinterpret_cast<float2*>(pc + (vb ? 8 : 0)) = vb ? b : a;
  }
}

// The kept VO rows of one observation row (np.round(., 2) of [PAA, rel, alpha, min_dis,
// iet] per row, ascending urgency) and its vo_count.  The zeros behind them are the row
// writers' business (row_fill_pairs / the two-phase writer / zero_fill).
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
  P.vo_count[g] = kept;
  if (bad) atomicOr(P.err, 1u);
}

// Generic row writer (W odd, or obs not 16-B aligned; the 16-B path is stage_row + row_fill_pairs).
// The zero run behind the kept rows of one observation row: with 8-B zero-fill units an odd
// 9 * kept leaves one float for this lane.
__device__ __forceinline__ void publish_zero_run(const Params& P, int g, int kept) {
  if ((P.W & 1) == 0 && ((9 * kept) & 1) && kept < P.nm) P.obs[(size_t)g * P.W + 12 + 9 * kept] = 0.0f;
}

// Cooperative, coalesced zero padding of the VO region of every row of this
// workgroup: rows [row0, row0 + nrows) are contiguous in memory; L.kept holds
// the kept count per row.  Unit = float2 when W is even (rows 8-B aligned),
// float otherwise.  q < T * per_row (validated by rvo3d_create for the 32-bit magic).
__device__ __forceinline__ void zero_fill(const Params& P, const Lds& L, int tid, int row0,
                                          int nrows) {
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

// mdin.py:28 adds two np.round(., 3) values in fp64; k / 1000 is formed exactly (k_over_1000)
// so the sum, cancellation included, is the reference's double (stored as float32, and as it is
// into the optional float64 output).

enum Mode { kObserve = 0, kStep = 1, kStepAutoReset = 2 };

#ifndef RVO3D_ZERO_NT
#define RVO3D_ZERO_NT 1  // streaming (non-temporal) stores for the early zero blocks
#endif
#ifndef RVO3D_STAGGER_ZEROS
#define RVO3D_STAGGER_ZEROS 0
#endif
#ifndef RVO3D_DEPHASE
#define RVO3D_DEPHASE 0
#endif

// 128 VGPRs = 4 waves per SIMD.  One-wave workgroups (N <= 64): the 4096 waves of 64 x 4096 are
// all resident at once.  N <= 256 (one env per workgroup of 2 or 4 waves): the fourth wave per
// SIMD is what lets 4 workgroups of 256 drones (40 KB of LDS each) share a CU, and shortens the
// tail for the shapes in between (100 drones x 2048 envs: 67.8 -> 55.6 us) - at the price of
// some spills above 128 drones.  N > 256 is LDS-bound to one or two workgroups per CU: 3 per SIMD.
#ifndef RVO3D_WAVES_ATTR
#define RVO3D_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(NW <= 4 ? 4 : 3)))
#endif

// The whole environment step, one launch.
// NFIX > 0: the instantiation for exactly NFIX drones per env (16, 32, 64: 64 / NFIX envs per
// one-wave workgroup; 128, 256: one env per workgroup of NFIX threads): N, the envs per
// workgroup and the workgroup size are compile-time constants, so the LDS layout and the
// index arithmetic of the sweeps fold.  NFIX = 0: any N <= 64 * NW.
// TRAIN = rvo_inter.env_train (rvo_inter.py:14), a compile-time constant of the instantiation.
// PAD: the compile-time kernel takes any N <= NFIX (ghost lanes, see below).  Multi-wave NFIX kernels are always
// padded; the one-wave ones exist in both flavours (the exact one keeps N a compile-time constant everywhere).
template <int MODE, int NW, int NFIX = 0, bool TRAIN = true, bool PAD = (NFIX != 0 && NW > 1)>
__global__ void __launch_bounds__(64 * NW) RVO3D_WAVES_ATTR env_kernel(const Params Pin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Params P = Pin;
  // Inside the kernel P.N is the size of the RING the pair sweeps walk (neighbour d + k mod P.N) and of the
  // LDS layout; Nr is the number of drones an env really has.  They differ only in the padded compile-time
  // kernels (PAD): those take any N <= NFIX, the lanes d >= N of an env's segment are ghost drones parked at
  // infinity - never in range of anybody, never active - so envs of 33..63, 65..127, 129..255 (...) drones run
  // on the kernels whose index arithmetic folds (no generic-N spills; stage G shared between the lanes
  // at 128 / 256).
  static_assert(!PAD || NFIX != 0, "a padded kernel has a compile-time ring size");
  const int Nr = Pin.N;
  if (NFIX) { P.N = NFIX; P.epb = NFIX <= 64 ? 64 / NFIX : 1; }
  const int tid = threadIdx.x, T = (NW == 1 || NFIX) ? 64 * NW : (int)blockDim.x, N = P.N;
  const Lds L = carve_lds(smem, T, P.nm, P.epb, N, NW);
  const int el = tid / N;
  const int d = tid - el * N;
  const int e0 = blockIdx.x * P.epb;
  const int e = e0 + el;
  const bool active = (el < P.epb) && (e < P.E) && (!PAD || d < Nr);
  const int g = active ? e * (PAD ? Nr : N) + d : 0;
  const int lbase = el * N;
  const int full_rows = P.epb * (PAD ? Nr : N);                                            // rows of a full workgroup
  const int nrows = ((P.E - e0) < P.epb ? (P.E - e0) : P.epb) * (PAD ? Nr : N);            // rows of this workgroup
  const int row0 = e0 * (PAD ? Nr : N);
  const int lrow = PAD ? el * Nr + d : tid;  // this drone's row among the workgroup's rows (row writers)
  constexpr bool LITE = (MODE == kStepAutoReset);
  // stage G shared between the lanes (gate_words_shared): workgroups of exactly 64 NW drones
  constexpr bool GSH = NW > 1 && NFIX == 64 * NW;

  // Register discipline: values are loaded right before the phase that needs them and
  // stored as soon as they are final, so that across the sweeps little more than the
  // drone's own 8-value record and its action stay live (registers = waves per SIMD).
  RVO3D_STAMP(0);
#if RVO3D_STAGGER_ZEROS
  // (experiment, off: measured slower at 64 x 4096 both cache-warm, 40.1 -> 44.6 us, and cache-cold, 60.9 -> 63.3 us)
  // Half of the workgroups store their early zero blocks (two-phase row writer: 69 % of the observation
  // bytes, no data needed) at the very START, the other half where the reset decision is taken.  When a
  // launch finds nothing in the caches (a rollout: policy GEMMs ran since the last step) all waves of the
  // one resident round move through the phases together - everybody loads, everybody computes, everybody
  // stores - and the memory system idles while they compute; with the two halves out of step the stores
  // of one half fill the load / compute phases of the other.
#if RVO3D_STAGGER_ZEROS == 2
  // by the wave's slot on its SIMD (HW_ID bits 3:0), so that the waves sharing one SIMD are out of step
  unsigned hw_id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
  const bool odd_half = NW == 1 ? (hw_id & 1) : ((blockIdx.x >> 3) & 1);
#else
  const bool odd_half = (blockIdx.x >> 3) & 1;
#endif
  const bool zeros_first = LITE && odd_half && two_phase_rows(P, row0, nrows, full_rows);
  if (zeros_first) early_zero_blocks<NW>(P, L, tid, row0, nrows);
#else
  const bool zeros_first = false;
#endif
#if RVO3D_DEPHASE
  // (experiment, off) waves in odd slots of their SIMD start RVO3D_DEPHASE x ~0.94 us (2048 cycles) late
  if (NW == 1) {
    unsigned hw_slot;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_slot));
    if (hw_slot & 1)
      for (int i = 0; i < RVO3D_DEPHASE; ++i) __builtin_amdgcn_s_sleep(32);
  }
#endif
  Drone S;
  S.x = S.y = S.z = S.vx = S.vy = S.vz = 0.0; S.r = 0.2; S.prio = 5;
  if (PAD && d >= Nr) {
    // a ghost: its fp64 record (restaged with everybody's below) sits at +inf - no exact test passes - and its
    // fp32 record, which stage_f32 never touches again, far outside every gate (1e30^2 overflows to +inf:
    // the sign test of stage G says "not in range"; never NaN against a real drone)
    S.x = S.y = S.z = __builtin_inf();
    const int o = el * 2 * N + d, os = el * N + d;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const float v = (k == WX || k == WY || k == WZ) ? 1e30f : (k == WKD ? -1.0f : 0.0f);
      if (k == WX || k == WY || k == WZ || k == WR) {
        L.w[k][o] = v;
        if (d <= (N >> 1)) L.w[k][o + N] = v;
      } else L.w[k][os] = v;
    }
  }
  double a[3] = {0, 0, 0}, cur[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
  double dev = 0, max_dev = 0;
  int wpi = 1;

  // ---- phase 0: the drone's own record and its action; everything else about the pre-move
  //      state (waypoints, des_vel, deviation) is fetched after sweep A, which needs none of it
  uint32_t gw[NW];  // candidate words (stage G): on file from the previous step if it ended in this state
  const bool have_gw = MODE != kObserve && P.g_cached != 0;
  if (active) {
    S.x = P.px()[g]; S.y = P.py()[g]; S.z = P.pz()[g];
    S.vx = P.vx()[g]; S.vy = P.vy()[g]; S.vz = P.vz()[g];
    if (P.uniform_rp) { S.r = P.r0; S.prio = P.prio0; }
    else { S.r = P.radius()[g]; S.prio = P.prio()[g]; }
    if (have_gw) {  // requested with the record: sweep A's stage X1 starts from them (no load behind the staging barrier)
#pragma unroll
      for (int w = 0; w < NW; ++w) gw[w] = P.gcache(w)[g];
    }
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
    stage_f32<NW>(P, L, el, d, active, p, v, az, S.r, S.prio);
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
    const int kept = sweep_env<NW, true, true, TRAIN, GSH>(P, L, tid, el, d, g, active, S, zero3, true, flag,
                                               tmin, collision, gw, false);
    if (active) {
      write_vo_rows(P, L, tid, lbase, g, S, kept);
      if (P.zf16) stage_row(P, L, lrow, g, S, proprio_tail(dv, dev), kept);
      else {
        write_proprio(P, g, S, proprio_tail(dv, dev));
        publish_zero_run(P, g, kept);
        L.kept[lrow] = kept;
      }
      P.max_dev()[g] = max_dev;
      uint32_t dvk_a, dvk_b;
      dv_encode(dv, dvk_a, dvk_b);
      P.dvk_a()[g] = dvk_a; P.dvk_b()[g] = dvk_b;
#pragma unroll
      for (int w = 0; w < NW; ++w) P.gcache(w)[g] = gw[w];
    }
    __syncthreads();
    if (P.zf16) row_fill_pairs<NW>(P, L, tid, row0, nrows);
    else zero_fill(P, L, tid, row0, nrows);
    return;
  }

  RVO3D_STAMP(2);
  // ---- sweep A: ir_gym.rvo_reward_list_cal on the pre-move state (ir_gym.py:50-62)
  sweep_env<NW, false, false, TRAIN, GSH>(P, L, tid, el, d, g, active && !RVO3D_ABLATED(1), S, az, false, flag,
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
  // (the whole batch is issued here, ahead of the barrier below - a fence the compiler does not move
  // loads across: what the integration needs arrives while the RVO reward is being computed)
  // (up to 128 drones per env; at 256 the eighteen registers this keeps busy over the barrier cost more
  // than the latency they hide)
  constexpr bool kLoadAhead = NW <= 2;
  double prev[3] = {0, 0, 0}, yaw = 0, pitch = 0, real_len = 0, route_len = 0;
  int npts = 2;
  bool f_arrive_in = false, f_dest_in = false;
#define RVO3D_LOAD_INTEGRATION_STATE()                                               \
  {                                                                                  \
    wpi = P.wp_idx()[g];                                                             \
    RVO3D_LOAD_PREV(P, g, prev);                                                     \
    yaw = P.yaw()[g]; pitch = P.pitch()[g]; real_len = P.real_len()[g];              \
    route_len = P.route_len()[g];                                                    \
    npts = P.n_points()[g];                                                          \
    f_arrive_in = P.arrive()[g] != 0; f_dest_in = P.dest()[g] != 0;                  \
  }
  if (active) {
    max_dev = P.max_dev()[g];
    RVO3D_LOAD_CUR(P, g, cur);
    uint32_t dvk_a = 0, dvk_b = 0;
    if (P.dv_cached) { dvk_a = P.dvk_a()[g]; dvk_b = P.dvk_b()[g]; }
    if (kLoadAhead) RVO3D_LOAD_INTEGRATION_STATE()
    bool have = false;
    if (P.dv_cached) have = dv_decode(dvk_a, dvk_b, dv);
    if (!have) {
      if (!kLoadAhead) RVO3D_LOAD_PREV(P, g, prev);
      const double p[3] = {S.x, S.y, S.z};
      des_vel(P, p, cur, dv);
      dev = deviation(prev, cur, p);
      if (dev > max_dev) max_dev = dev;
    }
    rew_k = rvo_reward_k(rvo_reward_pre(dv, a), flag, tmin);
  }
  RVO3D_STAMP(18);
  __syncthreads();  // everyone is done with the pre-move LDS image
  if (active) {
    if (!kLoadAhead) RVO3D_LOAD_INTEGRATION_STATE()
#undef RVO3D_LOAD_INTEGRATION_STATE
    // extra_len is only ever WRITTEN by the step (drone.py:188, ir_gym.py:176): not loaded, and
    // stored only by the drones that set it; wp_idx / arrive / dest likewise only when they change
    bool ex_set = false;
    double extra_len = 0.0;
    const int wpi_in = wpi;
    bool f_arrive = f_arrive_in;
    f_dest = f_dest_in;

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
      if (at_dst) { extra_len = real_len - route_len; ex_set = true; }  // destination_arrive side effect
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
        ex_set = true;
        if (!f_dest) { f_dest = true; dest_r = true; }
      }
    }
    const double exlen = real_len - route_len + 4;
    mov_nc = mov_reward_k(P, false, arrive_r, waypoint_num, npts - 1, dest_r, dev, exlen > 0,
                          exlen);
    RVO3D_STAMP(19);
    collision = building_hit(P, S);
    if (S.x < 0 || S.x > P.cold().map[0] || S.y < 0 || S.y > P.cold().map[1] || S.z < 0 || S.z > P.cold().map[2])
      collision = true;  // drone.drone_out_map, drone.py:213-225
    // final for this step unless the drone is reset below
    P.yaw()[g] = yaw; P.pitch()[g] = pitch; P.real_len()[g] = real_len;
    if (ex_set) P.extra_len()[g] = extra_len;
    if (wpi != wpi_in) P.wp_idx()[g] = wpi;
    if (f_arrive != f_arrive_in) P.arrive()[g] = f_arrive ? 1 : 0;
    if (f_dest != f_dest_in) P.dest()[g] = f_dest ? 1 : 0;
    P.info[g] = f_arrive ? 1 : 0;
    P.finish[g] = f_dest ? 1 : 0;
  }
  L.x[tid] = S.x; L.y[tid] = S.y; L.z[tid] = S.z;
  L.vx[tid] = S.vx; L.vy[tid] = S.vy; L.vz[tid] = S.vz;
  {
    const double p[3] = {S.x, S.y, S.z}, v[3] = {S.vx, S.vy, S.vz};
    stage_f32<NW>(P, L, el, d, active, p, v, az, S.r, S.prio);
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
    if (collide_env<NW, TRAIN, GSH>(P, L, tid, el, d, active && !RVO3D_ABLATED(2), S, gw)) collision = true;
  } else {
    kept = sweep_env<NW, true, true, TRAIN, GSH>(P, L, tid, el, d, g, active && !RVO3D_ABLATED(2), S, az, false,
                                     flag, tmin, collision, gw, false);
  }
  bool do_reset = false;
  if (active) {
    const double rew64 = k_over_1000(rew_k) + k_over_1000(collision ? -50000.0 : mov_nc);  // mdin.py:28
    P.reward[g] = (float)rew64;
    if (P.reward64) P.reward64[g] = rew64;
    P.done[g] = collision ? 1 : 0;
    do_reset = LITE && (collision || f_dest);
  }

  RVO3D_STAMP(5);
  if (LITE) {
    if (active && P.reset_mask) P.reset_mask[g] = do_reset ? 1 : 0;
    if (do_reset) L.any_reset[el] = 1;
    __syncthreads();  // sweep reads done; any_reset visible
    // the reset drones' start state is requested first; then - the workgroup's early zero blocks
    // (two-phase row writer: every 64-B block without proprio bytes) are stored, which takes a
    // few thousand cycles of store issue and needs no data at all: the loads land meanwhile
    double p[3] = {0, 0, 0}, rcur[3] = {0, 0, 0}, rdev = 0.0;
    uint32_t rdv_a = 0, rdv_b = 0;
    if (do_reset) {  // (dronestate of the start state: static, tabulated by rvo3d_load_world, dv0_kernel)
      load_wp(P, g, 0, p);
      rdev = P.dev0()[g];
      load_wp(P, g, 1, rcur);
      rdv_a = P.dv0_a()[g]; rdv_b = P.dv0_b()[g];
    }
    RVO3D_STAMP(26);
    // (one-wave workgroups: 64 x 4096 -2 %; with several waves per workgroup the blocks go out right
    // before the rows sweep instead, which measured better there)
    if (NW == 1 && !zeros_first && two_phase_rows(P, row0, nrows, full_rows) && !RVO3D_ABLATED(16)) early_zero_blocks<NW>(P, L, tid, row0, nrows);
    RVO3D_STAMP(27);
    if (do_reset) {  // drone.reset (drone.py:270-291); extra_len survives
      S.x = p[0]; S.y = p[1]; S.z = p[2]; S.vx = S.vy = S.vz = 0.0;
      dev = rdev;
      cur[0] = rcur[0]; cur[1] = rcur[1]; cur[2] = rcur[2];
      if (!dv_decode(rdv_a, rdv_b, dv)) {
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
      stage_f32<NW>(P, L, el, d, true, p, v0, az, S.r, S.prio);
    }
    // an env that reset somebody is observed with action 0 (ir_gym.py:372-383): its drones zero the
    // fp32 copy of their action now (any_reset is visible since the barrier above; the rows sweep is
    // behind the next one), so stage X1 needs no per-trip "action or zero" selects
    if (active && L.any_reset[el] != 0) {
      const int os = el * N + d;
      L.w[WAX][os] = 0.f; L.w[WAY][os] = 0.f; L.w[WAZ][os] = 0.f;
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
  // max_deviation is final as well; stored here - unlike the rest of the record - because two
  // registers less across the sweep are worth more than the store costs (config 5: -1.5 %)
  if (active) P.max_dev()[g] = max_dev;
  if (LITE) {
    __syncthreads();
    RVO3D_STAMP(6);
    // rows for every env: ir_gym.observation_reward's VO part (the env kept its state) or
    // ir_gym.env_observation with action 0 (the env reset a drone, ir_gym.py:372-383)
    // (ghost lanes of a padded env follow their env: the shared stage G below needs every lane, and its
    // barrier every wave)
    const bool env_reset = (active || (PAD && el < P.epb)) && (L.any_reset[el] != 0);
    bool c2 = false;
    const double aa[3] = {env_reset ? 0.0 : az[0], env_reset ? 0.0 : az[1], env_reset ? 0.0 : az[2]};
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
    if (RVO3D_ABLATED(2)) have_gw2 = false;  // diagnostics: the collision sweep was skipped
    if (NW > 1 && !zeros_first && two_phase_rows(P, row0, nrows, full_rows) && !RVO3D_ABLATED(16)) early_zero_blocks<NW>(P, L, tid, row0, nrows);
    kept = sweep_env<NW, true, false, TRAIN, GSH>(P, L, tid, el, d, g, active && !RVO3D_ABLATED(4), S, aa,
                                      false, flag, tmin, c2, gw, have_gw2);
  }
  RVO3D_STAMP(7);
  // kept rows first (their loads from the row scratch would otherwise queue behind the fill's
  // stores); then every other byte of the rows - proprio and zeros - in coalesced 16-B stores
  // (a lane writing its own 48 proprio bytes costs as much as the whole zero fill: 64 rows =
  // 64 partial cache lines per store instruction); the state stores drain behind them.
  const bool two_phase = LITE && two_phase_rows(P, row0, nrows, full_rows);
  if (NW > 1 && two_phase) {  // every wave's early zeros have landed before any wave writes kept rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (active) {
    if (!RVO3D_ABLATED(8)) {
      // kept rows go over zeros this wave (or, before the sweep's barrier, this workgroup)
      // stored earlier: those stores have to have landed
      if (two_phase) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      write_vo_rows(P, L, tid, lbase, g, S, kept);
      RVO3D_STAMP(16);
      if (P.zf16) stage_row(P, L, lrow, g, S, ptail, kept);
      else { write_proprio(P, g, S, ptail); publish_zero_run(P, g, kept); }
    }
    if (!P.zf16) L.kept[lrow] = kept;
  }
  __syncthreads();  // the staged rows / L.kept complete
  RVO3D_STAMP(17);
  if (!RVO3D_ABLATED(16)) {
    if (two_phase) late_row_windows<NW>(P, L, tid, row0, nrows);
    else if (P.zf16) row_fill_pairs<NW>(P, L, tid, row0, nrows);
    else zero_fill(P, L, tid, row0, nrows);
  }
  RVO3D_STAMP(8);
  if (active) {
#pragma unroll
    for (int w = 0; w < NW; ++w) P.gcache(w)[g] = gw[w];
    P.px()[g] = S.x; P.py()[g] = S.y; P.pz()[g] = S.z;
    P.vx()[g] = S.vx; P.vy()[g] = S.vy; P.vz()[g] = S.vz;
  }
  RVO3D_STAMP(9);
}

}  // namespace rvo3d
