// rvo3d_rollout_kernels.hpp -- The trainer's per-step glue around the env step, on the device: the two output heads of
// the actor-critic + action sampling + log-probability + buffer stores in one pass over the hidden activations
// (policy_sample_kernel), and the episode bookkeeping behind the env step (rollout_account_kernel).
// Reference: train/policy/multi_ppo.py:193-281 (the rollout loop), policy_rnn_ac.py:57-69, 197-235 (ac.step, the Gaussian actor).
// Part of the gfx950 device code (see rvo3d_device.hpp for the overview).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvo3d {

// ---- Philox4x32-10 (Salmon et al., SC'11), counter = (row, step), key = seed -------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t lo0 = c0 * 0xD2511F53u, hi0 = __umulhi(c0, 0xD2511F53u);
    const uint32_t lo1 = c2 * 0xCD9E8D57u, hi1 = __umulhi(c2, 0xCD9E8D57u);
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// three standard normals from one Philox block (Box-Muller on (0, 1) uniforms)
__device__ __forceinline__ void normal3(uint64_t seed, uint64_t step, uint64_t row, float eps[3]) {
  uint32_t x[4];
  philox4x32_10((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), x);
  const float k = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)x[0] + 0.5f) * k, u1 = ((float)x[1] + 0.5f) * k;
  const float u2 = ((float)x[2] + 0.5f) * k, u3 = ((float)x[3] + 0.5f) * k;
  // (float)x rounds up to 2^32 for the top 128 values: u = 1 then, log(1) = 0 - still finite
  const float r0 = __builtin_sqrtf(-2.0f * __logf(u0 < 1.0f ? u0 : 0.99999994f));
  const float r1 = __builtin_sqrtf(-2.0f * __logf(u2 < 1.0f ? u2 : 0.99999994f));
  float s0, c0, s1, c1;
  __sincosf(6.283185307179586f * u1, &s0, &c0);
  __sincosf(6.283185307179586f * u3, &s1, &c1);
  eps[0] = r0 * c0; eps[1] = r0 * s0; eps[2] = r1 * c1;
  (void)s1;
}

struct PolicySampleArgs {
  const void* h_pi;      // [rows][ld_pi] hidden activations of the actor (T), or - hidden == 0 - mu [rows][3] float
  const void* h_v;       // [rows][ld_v]  hidden activations of the critic (T), or - hidden == 0 - v [rows] float
  int64_t ld_pi, ld_v;   // row strides in elements
  int32_t hidden;        // H (a multiple of 32 * 16 / sizeof(T), see NCH), or 0: no heads, mu / v given
  int32_t tanh_out;      // 1: mu = tanh(W h + b)   (output_activation = nn.Tanh, policy_rnn_ac.py:197)
  const float* w_pi;     // [3][H]  nn.Linear weight of the actor's last layer
  const float* b_pi;     // [3]
  const float* w_v;      // [H]
  const float* b_v;      // [1]
  const float* log_std;  // [3]     GaussianActor.log_std (policy_rnn_ac.py:198)
  float std_factor;
  uint64_t seed, step;
  const uint64_t* step_dev;  // optional: a counter in device memory added to `step` (rvo3d_rollout_set_step_counter)
  int64_t rows;
  float* act;            // [rows][3]  np.round(a, 2) in float32, what the buffer stores (multi_ppo.py:197) and the env steps from
  float* logp;           // [rows]     log-probability of the UNROUNDED sample (policy_rnn_ac.py:63-64)
  float* val;            // [rows]
  float* dbg_mu;         // optional [rows][3]
  float* dbg_raw;        // optional [rows][3]: the unrounded sample
};

// GaussianActor._distribution (policy_rnn_ac.py:217-222): std = clamp(std_factor * exp(log_std) + 1e-6, 1e-4, 10) - the
// same for every row of a launch: computed once per lane, not once per row
struct SampleConsts { float sd[3], inv2var[3], log_sd[3]; uint64_t step; };
__device__ __forceinline__ SampleConsts sample_consts(const PolicySampleArgs& A) {
  SampleConsts C;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float sd = A.std_factor * __expf(A.log_std[k]) + 1e-6f;
    sd = sd < 1e-4f ? 1e-4f : (sd > 10.0f ? 10.0f : sd);
    C.sd[k] = sd; C.inv2var[k] = 1.0f / (2.0f * sd * sd); C.log_sd[k] = __logf(sd);
  }
  C.step = A.step + (A.step_dev ? *A.step_dev : 0);
  return C;
}
// tanh from one exp and one reciprocal: 1 - 2 / (e^2x + 1); exact limits at both ends, absolute error < 3e-7
__device__ __forceinline__ float tanh_fast(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}
__device__ __forceinline__ void finish_row(const PolicySampleArgs& A, const SampleConsts& C, int64_t row, float z0,
                                           float z1, float z2) {
  // a ~ Normal(mu, std); logp = sum_k log N(a_k; mu_k, std_k)
  float mu[3] = {z0, z1, z2};
  if (A.tanh_out) { mu[0] = tanh_fast(z0); mu[1] = tanh_fast(z1); mu[2] = tanh_fast(z2); }
  float eps[3];
  normal3(A.seed, C.step, (uint64_t)row, eps);
  float lp = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float a = mu[k] + C.sd[k] * eps[k];
    const float d = a - mu[k];
    lp += -(d * d) * C.inv2var[k] - C.log_sd[k] - 0.9189385332046727f;
    A.act[row * 3 + k] = __builtin_rintf(a * 100.0f) / 100.0f;  // np.round(float32, 2): rint(a * 100) / 100
    if (A.dbg_mu) A.dbg_mu[row * 3 + k] = mu[k];
    if (A.dbg_raw) A.dbg_raw[row * 3 + k] = a;
  }
  A.logp[row] = lp;
}
__device__ __forceinline__ void finish_row(const PolicySampleArgs& A, int64_t row, float z0, float z1, float z2) {
  finish_row(A, sample_consts(A), row, z0, z1, z2);
}

// hidden == 0: mu and v come from the caller's own network (e.g. the biGRU actor-critic); one lane per row.
__global__ void __launch_bounds__(256) policy_sample_direct_kernel(const PolicySampleArgs A) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= A.rows) return;
  const float* mu = static_cast<const float*>(A.h_pi) + row * A.ld_pi;
  finish_row(A, row, mu[0], mu[1], mu[2]);
  A.val[row] = static_cast<const float*>(A.h_v)[row * A.ld_v];
}

// The two heads ([H] -> 3 and [H] -> 1) as one pass over the hidden activations: HBM-bound (2 H elements read
// per row, 20 bytes written), no matrix core involved.  A wave takes 32 rows at a time: lanes 0-31 read the
// actor's hidden row, lanes 32-63 the critic's, 16 B per lane and chunk (one row = NCH chunks of 32 lanes);
// the head weights sit in registers.  Four rows are reduced together (a butterfly that halves the live values
// per step: 18 cross-lane moves for four rows of three sums instead of 60), after eight such quads every lane
// of a half-wave owns one row's sums and finishes it: tanh, sample, log-probability, stores.
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPL = 4;
  __device__ static void unpack(const uint4& v, float x[4]) {
    x[0] = __builtin_bit_cast(float, v.x); x[1] = __builtin_bit_cast(float, v.y);
    x[2] = __builtin_bit_cast(float, v.z); x[3] = __builtin_bit_cast(float, v.w);
  }
};
struct bf16_t { uint16_t bits; };
template <> struct Elem<bf16_t> {
  static constexpr int EPL = 8;
  __device__ static void unpack(const uint4& v, float x[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[2 * i] = __builtin_bit_cast(float, w[i] << 16);
      x[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
  }
};

template <typename T, int NCH>
__global__ void __launch_bounds__(256) policy_sample_kernel(const PolicySampleArgs A) {
  constexpr int EPL = Elem<T>::EPL;
  const int lane = threadIdx.x & 63, l32 = lane & 31;
  const bool is_v = lane >= 32;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t row0 = wave * 32;
  if (row0 >= A.rows) return;  // whole waves only: no barrier in this kernel
  const T* const base = static_cast<const T*>(is_v ? A.h_v : A.h_pi);
  const int64_t ld = is_v ? A.ld_v : A.ld_pi;
  // head weights of this lane's elements: actor lanes three rows of w_pi, critic lanes w_v and two zero rows
  float w0[NCH][EPL], w1[NCH][EPL], w2[NCH][EPL];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const int e = (c * 32 + l32) * EPL + i;
      w0[c][i] = is_v ? A.w_v[e] : A.w_pi[e];
      w1[c][i] = is_v ? 0.f : A.w_pi[A.hidden + e];
      w2[c][i] = is_v ? 0.f : A.w_pi[2 * A.hidden + e];
    }
  float mine0 = 0.f, mine1 = 0.f, mine2 = 0.f;  // the sums of the row this lane ends up owning
  const int b4 = (l32 >> 4) & 1, b3 = (l32 >> 3) & 1, q_mine = l32 & 7;
  // raw 16-byte chunks of four rows; the next quad's are requested before the current quad is reduced
  typedef uint4 raw_t;
  raw_t cur[4][NCH], nxt[4][NCH];
  auto request = [&](int q, raw_t dst[4][NCH]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int64_t row = row0 + 4 * q + r;
      if (row >= A.rows) row = A.rows - 1;  // (a ragged tail re-reads the last row; its results are not stored)
      const T* src = base + row * ld + (int64_t)l32 * EPL;
#pragma unroll
      for (int c = 0; c < NCH; ++c) dst[r][c] = *reinterpret_cast<const raw_t*>(src + c * 32 * EPL);
    }
  };
  request(0, cur);
  // (rolled: fully unrolled the compiler hoists every quad's loads and the kernel needs 142 VGPRs = 3 waves per SIMD)
#pragma unroll 1
  for (int q = 0; q < 8; ++q) {
    if (q + 1 < 8) request(q + 1, nxt);
    float p[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float x[EPL];
        Elem<T>::unpack(cur[r][c], x);
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
          s0 = __builtin_fmaf(x[i], w0[c][i], s0);
          s1 = __builtin_fmaf(x[i], w1[c][i], s1);
          s2 = __builtin_fmaf(x[i], w2[c][i], s2);
        }
      }
      p[r][0] = s0; p[r][1] = s1; p[r][2] = s2;
    }
    // butterfly over the 32 lanes of the half-wave.  xor 16: lanes with bit 4 clear keep rows 0, 1 and hand
    // rows 2, 3 over (and the other way round); xor 8: bit 3 picks one of the two; xor 4, 2, 1: plain sums.
    float k2[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float keep = b4 ? p[2 + j][k] : p[j][k];
        const float give = b4 ? p[j][k] : p[2 + j][k];
        k2[j][k] = keep + __shfl_xor(give, 16, 64);
      }
    float k1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float keep = b3 ? k2[1][k] : k2[0][k];
      const float give = b3 ? k2[0][k] : k2[1][k];
      k1[k] = keep + __shfl_xor(give, 8, 64);
    }
#pragma unroll
    for (int m = 4; m >= 1; m >>= 1)
#pragma unroll
      for (int k = 0; k < 3; ++k) k1[k] += __shfl_xor(k1[k], m, 64);
    // every lane with bits (4, 3) = (b4, b3) now holds row 4 q + 2 b4 + b3 of the quad; lane q of them keeps it
    if (q == q_mine) { mine0 = k1[0]; mine1 = k1[1]; mine2 = k1[2]; }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < NCH; ++c) cur[r][c] = nxt[r][c];
  }
  const int64_t row = row0 + 4 * q_mine + 2 * b4 + b3;
  if (row >= A.rows) return;
  if (is_v) A.val[row] = mine0 + A.b_v[0];
  else finish_row(A, row, mine0 + A.b_pi[0], mine1 + A.b_pi[1], mine2 + A.b_pi[2]);
}

// ---- episode bookkeeping of one rollout step (multi_ppo.py:217-281), one thread per drone, one env per workgroup ----
struct AccountArgs {
  int32_t E, N;
  const float* reward;     // [E][N]  the env step's reward (may be inf / nan, survey Q9)
  const uint8_t* done;     // [E][N]
  const uint8_t* finish;   // [E][N]
  int32_t sanitize;        // 1: inf / nan rewards enter the buffer as 0
  int32_t max_ep_len;      // timeout: ep_len > max_ep_len (multi_ppo.py:266)
  int32_t epoch_end;       // 1: last step of the epoch - every path ends, every drone is reset (multi_ppo.py:244-264)
  float* rew_slot;         // [E][N]  buffer slot of this step
  float* ep_ret;           // [E][N]  running episode return
  int32_t* ep_len;         // [E][N]
  uint8_t* cut_slot;       // [E]     finish_path(0) for every drone of the env behind this step
  uint8_t* extra_mask;     // [E][N]  drones the trainer still has to reset (timeouts, epoch end) - the env step reset done | finish itself
  double* sums;            // [E][2]  per env: += sum of finished episodes' returns, += their number (one workgroup
                           //         owns an env's pair: no atomics - 8192 atomic adds per step on two addresses
                           //         took 100 us; the caller sums over the envs when it reads the statistic)
  int32_t* any_extra;      // [1]     |= 1 when extra_mask has a non-zero byte
  uint64_t* step_dev;      // optional: the noise counter of rvo3d_rollout_set_step_counter, advanced by one per call
};

__global__ void __launch_bounds__(512) rollout_account_kernel(const AccountArgs A) {
  __shared__ int s_term, s_extra;
  __shared__ double s_sum[8], s_cnt[8];
  const int e = blockIdx.x, d = threadIdx.x;
  if (d == 0) { s_term = 0; s_extra = 0; }
  if (e == 0 && d == 0 && A.step_dev) *A.step_dev += 1;  // one rollout step done: the next policy call draws new noise
  __syncthreads();
  double my_sum = 0.0, my_cnt = 0.0;
  if (d < A.N) {
    const int64_t g = (int64_t)e * A.N + d;
    const float r = A.reward[g];
    const float rf = (r == r && r != __builtin_inff() && r != -__builtin_inff()) ? r : 0.0f;  // nan_to_num(., 0, 0, 0)
    A.rew_slot[g] = A.sanitize ? rf : r;
    const float ret = A.ep_ret[g] + rf;
    const int len = A.ep_len[g] + 1;
    const bool fin = A.finish[g] != 0, by_step = fin || A.done[g] != 0;
    const bool timeout = len > A.max_ep_len;
    const bool ended = by_step || timeout || A.epoch_end;
    const bool extra = (timeout || A.epoch_end) && !by_step;
    if (fin || timeout) s_term = 1;   // any drone of the env (multi_ppo.py:229): benign race, same value
    if (extra) s_extra = 1;
    A.extra_mask[g] = extra ? 1 : 0;
    if (ended) { my_sum = (double)ret; my_cnt = 1.0; }
    A.ep_ret[g] = ended ? 0.0f : ret;
    A.ep_len[g] = ended ? 0 : len;
  }
  // workgroup sums: wave reduction, then one atomic pair per workgroup
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    my_sum += __shfl_xor(my_sum, m, 64);
    my_cnt += __shfl_xor(my_cnt, m, 64);
  }
  if ((d & 63) == 0) { s_sum[d >> 6] = my_sum; s_cnt[d >> 6] = my_cnt; }
  __syncthreads();
  if (d == 0) {
    double ts = 0.0, tc = 0.0;
    for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) { ts += s_sum[w]; tc += s_cnt[w]; }
    if (tc != 0.0) { A.sums[2 * (size_t)e] += ts; A.sums[2 * (size_t)e + 1] += tc; }
    if (s_term || A.epoch_end) A.cut_slot[e] = 1;
    if (s_extra) atomicOr(A.any_extra, 1);
  }
}

// ---- rnn_Reader for one-step sequences (policy_rnn_ac.py:75-127: biGRU over the VO rows, sum of the two final hidden
//      states, concat with the proprioceptive part, LayerNorm) ----
// In a rollout nearly every drone has zero or one VO row, and a GRU over a one-step sequence from h = 0 is one cell
// evaluation without the recurrent GEMM: r = s(W_ir x + b_ir + b_hr), z = s(W_iz x + b_iz + b_hz),
// n = tanh(W_in x + b_in + r b_hn), h = (1 - z) n.  With in_dim = 9 that is 54 multiply-adds per hidden unit and
// direction: far too little for a GEMM (K = 9) and, issued as PyTorch ops, a dozen elementwise passes over [rows, 3 H]
// float32 temporaries (2.9 ms per 262144 rows).  Here: one thread per hidden unit with its 2 x 3 x in_dim weights in
// registers, eight rows per trip (their 21 leading floats staged in LDS and read as broadcasts), the LayerNorm's two
// row reductions done for the eight rows at once; the row's features leave as the (padded) A operand of the first MLP
// layer.  Reads 84 B and writes (state_dim + H) elements per row: compute-bound on the transcendentals (~0.2 ms).
// Rows with more than one VO row are recomputed by the caller (PyTorch, a small gathered batch) afterwards.
struct ReaderArgs {
  const float *w_ih_f, *b_ih_f, *b_hh_f;  // [3H][IN], [3H], [3H]   (nn.GRU: gates r, z, n)
  const float *w_ih_r, *b_ih_r, *b_hh_r;  // reverse direction or null
  const float *ln_w, *ln_b;               // [SD + H]
  int32_t H, IN, SD;
  float eps;
  const float* obs; int64_t obs_ld, rows;
  void* feat; int32_t feat_bf16; int64_t feat_ld;
};

// (v_rcp_f32 / v_rsq_f32 directly, 1 ulp: __frcp_rn is the correctly rounded reciprocal, a ten-instruction sequence)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * fast_sigmoid(2.0f * x) - 1.0f; }
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

constexpr int kReaderRows = 8;   // rows per trip
constexpr int kReaderMaxIn = 16; // in_dim <= 16
constexpr int kReaderMaxSd = 32; // state_dim <= 32

template <int IN>
__global__ void __launch_bounds__(256) reader_first_step_kernel(const ReaderArgs A) {
  __shared__ float s_in[kReaderRows][kReaderMaxSd + kReaderMaxIn];
  __shared__ float s_red[kReaderRows][4], s_red2[kReaderRows][4];
  __shared__ float s_stat[2][kReaderRows];  // mean, 1 / sqrt(var + eps) of the eight rows
  __shared__ __attribute__((aligned(16))) uint16_t s_out[kReaderRows][kReaderMaxSd + 256];  // bf16 rows on their way out
  const int u = threadIdx.x, H = A.H, SD = A.SD, D = SD + H;
  const int lane = u & 63, wv = u >> 6, nwv = (H + 63) >> 6;
  const bool bi = A.w_ih_r != nullptr;
  // this unit's weights: [dir][gate][IN]; biases folded where the cell allows it
  float w[2][3][IN], c_r[2], c_z[2], b_in[2], b_hn[2];
#pragma unroll
  for (int dir = 0; dir < 2; ++dir) {
    const float* wih = dir ? A.w_ih_r : A.w_ih_f;
    const float* bih = dir ? A.b_ih_r : A.b_ih_f;
    const float* bhh = dir ? A.b_hh_r : A.b_hh_f;
    const bool on = dir == 0 || bi;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int k = 0; k < IN; ++k) w[dir][g][k] = on ? wih[(size_t)(g * H + u) * IN + k] : 0.f;
    c_r[dir] = on ? bih[u] + bhh[u] : 0.f;
    c_z[dir] = on ? bih[H + u] + bhh[H + u] : 0.f;
    b_in[dir] = on ? bih[2 * H + u] : 0.f;
    b_hn[dir] = on ? bhh[2 * H + u] : 0.f;
  }
  const float lw = A.ln_w[SD + u], lb = A.ln_b[SD + u];
  const float lw0 = u < SD ? A.ln_w[u] : 0.f, lb0 = u < SD ? A.ln_b[u] : 0.f;
  // one cell evaluation of this unit from h = 0 (both directions summed).  Packed fp32 (v_pk_fma_f32) across GATES,
  // not rows - the weights pair up as they sit in registers ((r, z) of a direction; n of the two directions), only the
  // row's nine inputs are splatted: 27 instead of 54 multiply-add instructions per row.  Per unit, row and direction
  // five transcendentals instead of six: (1 - z) n with z = 1 / (1 + e_z), n = tanh(a) = (1 - e_n) / (1 + e_n),
  // e_z = exp(-g_z), e_n = exp(-2 a) is e_z (1 - e_n) / ((1 + e_z)(1 + e_n)): one reciprocal for both.
  typedef float v2f __attribute__((ext_vector_type(2)));
  auto cell = [&](const float* x) -> float {
    v2f rz[2] = {{c_r[0], c_z[0]}, {c_r[1], c_z[1]}}, nn = {b_in[0], b_in[1]};
#pragma unroll
    for (int k = 0; k < IN; ++k) {
      const float xk = x[k];
      const v2f xx = {xk, xk};
      rz[0] = __builtin_elementwise_fma((v2f){w[0][0][k], w[0][1][k]}, xx, rz[0]);
      rz[1] = __builtin_elementwise_fma((v2f){w[1][0][k], w[1][1][k]}, xx, rz[1]);
      nn = __builtin_elementwise_fma((v2f){w[0][2][k], w[1][2][k]}, xx, nn);
    }
    float hs = 0.f;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      const float rg = fast_sigmoid(rz[dir].x);
      const float a = (dir ? nn.y : nn.x) + rg * b_hn[dir];
      // clamped exponents: exp(88) overflows float32 to inf and inf / inf is NaN; beyond +-30 the cell saturates anyway
      const float ez = __expf(-__builtin_fmaxf(__builtin_fminf(rz[dir].y, 30.f), -30.f));
      const float en = __expf(-2.0f * __builtin_fmaxf(__builtin_fminf(a, 15.f), -15.f));
      const float hd = ez * (1.0f - en) * __builtin_amdgcn_rcpf((1.0f + ez) * (1.0f + en));
      if (dir == 0 || bi) hs += hd;
    }
    return hs;
  };
  // A row whose VO row is all zeros - no velocity obstacle: nearly every row of a rollout - has the SAME hidden state
  // (the cell sees biases only): computed once, with the same instructions (bit-identical), and so are its two wave sums
  float h0, v0, q0;
  {
    float zero_in[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) zero_in[k] = 0.f;
    h0 = cell(zero_in);
    v0 = h0; q0 = h0 * h0;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) { v0 += __shfl_xor(v0, sh, 64); q0 += __shfl_xor(q0, sh, 64); }
  }
  const int64_t ngroups = (A.rows + kReaderRows - 1) / kReaderRows;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t row0 = grp * kReaderRows;
    __syncthreads();  // the previous trip's readers are done with the LDS buffers
    for (int i = u; i < kReaderRows * (SD + IN); i += H) {
      const int r = i / (SD + IN), k = i - r * (SD + IN);
      const int64_t row = row0 + r < A.rows ? row0 + r : A.rows - 1;
      s_in[r][k] = A.obs[row * A.obs_ld + k];
    }
    __syncthreads();
    float h[kReaderRows];
    unsigned zero_rows = 0;  // (workgroup-uniform: every thread looks at the same staged inputs)
#pragma unroll 2
    for (int r = 0; r < kReaderRows; ++r) {
      bool any = false;
#pragma unroll
      for (int k = 0; k < IN; ++k) any |= s_in[r][SD + k] != 0.f;
      if (any) {
        h[r] = cell(&s_in[r][SD]);
      } else {
        h[r] = h0;
        zero_rows |= 1u << r;
      }
    }
    // LayerNorm (biased variance, eps inside the root: torch.nn.LayerNorm): per row the sum and the sum of squares of
    // the hidden part (wave reduction, one pass: the features are O(1), 268 of them - var = E[x^2] - mean^2 loses
    // nothing visible at float32), then thread r < rows adds the waves' partial sums and the row's proprioceptive part
#pragma unroll
    for (int r = 0; r < kReaderRows; ++r) {
      float v = v0, q = q0;
      if (!((zero_rows >> r) & 1)) {
        v = h[r]; q = h[r] * h[r];
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) { v += __shfl_xor(v, sh, 64); q += __shfl_xor(q, sh, 64); }
      }
      if (lane == 0) { s_red[r][wv] = v; s_red2[r][wv] = q; }
    }
    __syncthreads();
    if (u < kReaderRows) {
      float t = 0.f, t2 = 0.f;
      for (int q = 0; q < nwv; ++q) { t += s_red[u][q]; t2 += s_red2[u][q]; }
      for (int k = 0; k < SD; ++k) { const float x = s_in[u][k]; t += x; t2 += x * x; }
      const float m = t / (float)D;
      const float var = __builtin_fmaxf(t2 / (float)D - m * m, 0.f);
      s_stat[0][u] = m;
      s_stat[1][u] = __builtin_amdgcn_rsqf(var + A.eps);
    }
    __syncthreads();
    if (A.feat_bf16 && (D & 3) == 0) {
      // bf16 rows leave through LDS: a thread storing its own 2 bytes per row makes 128-byte store instructions of
      // byte-masked dwords (the kernel spent half its time there); staged, the trip's eight rows go out as 8-byte
      // chunks, 512 B per store instruction (D * 2 bytes per row is a multiple of 8, the row stride of 16)
#pragma unroll
      for (int r = 0; r < kReaderRows; ++r) {
        const float mean = s_stat[0][r], rstd = s_stat[1][r];
        s_out[r][SD + u] = f32_to_bf16_rne((h[r] - mean) * rstd * lw + lb);
        if (u < SD) s_out[r][u] = f32_to_bf16_rne((s_in[r][u] - mean) * rstd * lw0 + lb0);
      }
      __syncthreads();
      const int cpr = D >> 2;  // 8-byte chunks per row
      for (int i = u; i < kReaderRows * cpr; i += H) {
        const int r = i / cpr, c = i - r * cpr;
        const int64_t row = row0 + r;
        if (row < A.rows)
          *reinterpret_cast<uint2*>(static_cast<uint16_t*>(A.feat) + row * A.feat_ld + 4 * c) =
              *reinterpret_cast<const uint2*>(&s_out[r][4 * c]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < kReaderRows; ++r) {
        const int64_t row = row0 + r;
        if (row >= A.rows) break;
        const float mean = s_stat[0][r], rstd = s_stat[1][r];
        const float y = (h[r] - mean) * rstd * lw + lb;
        const float y0 = u < SD ? (s_in[r][u] - mean) * rstd * lw0 + lb0 : 0.f;
        if (A.feat_bf16) {
          uint16_t* o = static_cast<uint16_t*>(A.feat) + row * A.feat_ld;
          o[SD + u] = f32_to_bf16_rne(y);
          if (u < SD) o[u] = f32_to_bf16_rne(y0);
        } else {
          float* o = static_cast<float*>(A.feat) + row * A.feat_ld;
          o[SD + u] = y;
          if (u < SD) o[u] = y0;
        }
      }
    }
  }
}

// ---- the reader's features of a row WITHOUT a velocity-obstacle row, collapsed ------------------------------------
// For such a row the GRU sees a zero input from h = 0: its hidden state h0 is the same for every row, and the
// LayerNorm of concat(p, h0) (p: the row's state_dim proprioceptive floats) depends on the row through two scalars
// only, mean and rstd:  f_p = (p - mean) rstd g_p + b_p,  f_h = rstd (h0 g_h) - mean rstd g_h + b_h.  A linear layer on
// those features is therefore  W_p f_p + rstd a - (mean rstd) b + c  with a = W_h (h0 g_h), b = W_h g_h, c = W_h b_h + bias
// - a product over state_dim + 3 inputs instead of state_dim + hidden.  This kernel writes, per row, the inputs of
// that product for rvo3d_policy_mlp_sample: f_p, then rstd and mean rstd each as bf16 head / head / tail (so that the
// bf16 products a_hi r_hi + a_lo r_hi + a_hi r_lo carry ~16 bits of each factor), then two ones (for c's head and tail).
struct ZeroFeatArgs {
  const float* obs; int64_t obs_ld, rows;
  int32_t state_dim, feat_dim;     // 12, state_dim + hidden
  const float* ln_w; const float* ln_b;  // LayerNorm affine: the first state_dim entries are used
  float sum_h0, sumsq_h0, eps;
  float* out; int64_t out_ld;      // [rows][>= state_dim + 8]
  const int32_t* cnt;              // optional [rows]: rows with cnt > 0 are appended to `list` (order: as the atomics fall)
  int32_t* list; int32_t* count;   // [rows], [1] (zero before the launch; policy_rows_kernel resets it)
};
__device__ __forceinline__ float bf16_head(float x) {
  return __builtin_bit_cast(float, (uint32_t)f32_to_bf16_rne(x) << 16);
}
__global__ void __launch_bounds__(256) reader_zero_features_kernel(const ZeroFeatArgs A) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= A.rows) return;
  const float* p = A.obs + row * A.obs_ld;
  float x[kReaderMaxSd];
  // (16 bytes per load instruction: a wave's lanes sit a whole observation row apart, so every load instruction
  // touches 64 cache lines - the fewer instructions the better: 51 -> 17 us at 262144 rows)
#pragma unroll
  for (int q = 0; q < kReaderMaxSd / 4; ++q)
    if (4 * q < A.state_dim) __builtin_memcpy(&x[4 * q], p + 4 * q, 16);  // (may read up to 3 floats past state_dim: still this row)
  float t = A.sum_h0, t2 = A.sumsq_h0;
#pragma unroll
  for (int k = 0; k < kReaderMaxSd; ++k)
    if (k < A.state_dim) { t += x[k]; t2 += x[k] * x[k]; }
  const float D = (float)A.feat_dim;
  const float mean = t / D;
  const float var = __builtin_fmaxf(t2 / D - mean * mean, 0.f);
  const float rstd = 1.0f / __builtin_sqrtf(var + A.eps);
  float* o = A.out + row * A.out_ld;
  const float m = mean * rstd;
  const float rh = bf16_head(rstd), mh = bf16_head(m);
  const float tail[8] = {rh, rh, rstd - rh, mh, mh, m - mh, 1.0f, 1.0f};
  if (A.state_dim == 12 && (A.out_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(A.out) & 15) == 0) {
    // (the env's shape: 20 floats per row leave as five 16-byte stores, not twenty 4-byte ones a row apart each)
    float y[20];
#pragma unroll
    for (int k = 0; k < 12; ++k) y[k] = (x[k] - mean) * rstd * A.ln_w[k] + A.ln_b[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) y[12 + k] = tail[k];
#pragma unroll
    for (int q = 0; q < 5; ++q) reinterpret_cast<float4*>(o)[q] = float4{y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]};
  } else {
#pragma unroll
    for (int k = 0; k < kReaderMaxSd; ++k)
      if (k < A.state_dim) o[k] = (x[k] - mean) * rstd * A.ln_w[k] + A.ln_b[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[A.state_dim + k] = tail[k];
  }
  if (A.cnt && A.cnt[row] > 0) A.list[atomicAdd(A.count, 1)] = (int32_t)row;
}

// ---- the policy step of single rows, exactly as the module computes it (float32) ----------------------------------
// The few rows of a rollout step that DO have velocity-obstacle rows (13 of 262 144 in the benchmark's world): two
// workgroups per listed row (one per network), thread u = hidden unit u: the (bi)GRU over the row's cnt VO rows (policy_rnn_ac.py:129-168;
// the recurrent product only from the second step on), direction sum, concat, LayerNorm, both (256, 256) heads stacks,
// tanh, sample, log-probability, np.round, stores.  Weights as the modules store them, read from L2; no attempt at
// speed - the list is short by construction (the caller takes the GEMM path when it is not).  The last workgroup to
// finish resets the list's counter for the next step: no host involvement at all.
struct PolicyRowsArgs {
  const float* obs; int64_t obs_ld;
  const int32_t* cnt; const int32_t* list; int32_t* count; int32_t* done_blocks;
  int32_t state_dim, in_dim, H, slots;          // 12, 9, reader hidden (<= 256), nm
  const float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];  // [3H][in_dim], [3H][H], [3H], [3H]; [1] = reverse direction or null
  const float *ln_w, *ln_b; float eps;
  const float *w1[2], *b1[2], *w2[2], *b2[2], *w3[2], *b3[2];  // actor, critic: [256][D], [256], [256][256], [256], [3|1][256], [3|1]
  PolicySampleArgs S;
};
__global__ void __launch_bounds__(256) policy_rows_kernel(const PolicyRowsArgs A) {
  __shared__ float s_x[kReaderMaxSd + 16 * kReaderMaxIn];
  __shared__ float s_h[2][256];      // the hidden state of the running direction (double-buffered across steps)
  __shared__ float s_f[kReaderMaxSd + 256], s_a[2][256], s_b[2][256];
  __shared__ float s_red[8][4];
  const int u = threadIdx.x, H = A.H, SD = A.state_dim, IN = A.in_dim, D = SD + H;
  const int n_rows = *A.count;
  const SampleConsts SC = sample_consts(A.S);
  // two workgroups per listed row, one per network (both compute the reader's features - that part is short): the
  // row's latency chain is as long as ONE stack's weight reads, not two
  for (int wi = blockIdx.x; wi < 2 * n_rows; wi += gridDim.x) {
    const int li = wi >> 1, my_net = wi & 1;
    const int64_t row = A.list[li];
    int n = A.cnt[row];
    n = n < 1 ? 1 : (n > A.slots ? A.slots : n);
    __syncthreads();
    for (int i = u; i < SD + n * IN; i += 256) s_x[i] = A.obs[row * A.obs_ld + i];
    __syncthreads();
    float hsum = 0.f;
    for (int dir = 0; dir < 2; ++dir) {
      if (!A.w_ih[dir]) break;
      float h = 0.f;
      for (int step = 0; step < n; ++step) {
        const int t = dir ? n - 1 - step : step;
        float g[3], gh[3];
        if (u < H) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            float acc = A.b_ih[dir][q * H + u];
            for (int k = 0; k < IN; ++k) acc += A.w_ih[dir][(size_t)(q * H + u) * IN + k] * s_x[SD + t * IN + k];
            g[q] = acc;
            float acc2 = A.b_hh[dir][q * H + u];
            if (step > 0) {
              const float* wr = A.w_hh[dir] + (size_t)(q * H + u) * H;
              const float* hp = s_h[(step - 1) & 1];
              for (int j = 0; j < H; ++j) acc2 += wr[j] * hp[j];
            }
            gh[q] = acc2;
          }
          const float rg = 1.0f / (1.0f + __expf(-(g[0] + gh[0])));
          const float zg = 1.0f / (1.0f + __expf(-(g[1] + gh[1])));
          const float ng = tanhf(g[2] + rg * gh[2]);
          h = (1.0f - zg) * ng + zg * h;
          s_h[step & 1][u] = h;
        }
        __syncthreads();
      }
      hsum += h;
    }
    // LayerNorm over concat(p, hsum)
    float v = u < H ? hsum : 0.f, q2 = v * v;
    if (u < SD) { v += s_x[u]; q2 += s_x[u] * s_x[u]; }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) { v += __shfl_xor(v, sh, 64); q2 += __shfl_xor(q2, sh, 64); }
    if ((u & 63) == 0) { s_red[0][u >> 6] = v; s_red[1][u >> 6] = q2; }
    __syncthreads();
    const float tot = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    const float tot2 = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    const float mean = tot / (float)D;
    const float rstd = 1.0f / __builtin_sqrtf(__builtin_fmaxf(tot2 / (float)D - mean * mean, 0.f) + A.eps);
    if (u < SD) s_f[u] = (s_x[u] - mean) * rstd * A.ln_w[u] + A.ln_b[u];
    if (u < H) s_f[SD + u] = (hsum - mean) * rstd * A.ln_w[SD + u] + A.ln_b[SD + u];
    __syncthreads();
    // the two (256, 256) stacks, thread u = unit u; 16 bytes per load where the row length allows (a wave's lanes sit
    // a whole weight row apart: every load instruction touches 64 cache lines, so few wide loads beat many narrow ones;
    // a wave per unit with a shuffle reduction was five times slower - one latency chain per unit)
    {
      const int net = my_net;
      float acc = A.b1[net][u];
      const float* wr = A.w1[net] + (size_t)u * D;
      if ((D & 3) == 0 && (reinterpret_cast<uintptr_t>(A.w1[net]) & 15) == 0) {
#pragma unroll 8
        for (int k = 0; k < D; k += 4) {  // (unrolled: eight weight loads in flight per thread, not one)
          const float4 w4 = *reinterpret_cast<const float4*>(wr + k);
          acc += w4.x * s_f[k] + w4.y * s_f[k + 1] + w4.z * s_f[k + 2] + w4.w * s_f[k + 3];
        }
      } else {
        for (int k = 0; k < D; ++k) acc += wr[k] * s_f[k];
      }
      s_a[net][u] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    {
      const int net = my_net;
      float acc = A.b2[net][u];
      const float* wr = A.w2[net] + (size_t)u * 256;
      if ((reinterpret_cast<uintptr_t>(A.w2[net]) & 15) == 0) {
#pragma unroll 8
        for (int k = 0; k < 256; k += 4) {
          const float4 w4 = *reinterpret_cast<const float4*>(wr + k);
          acc += w4.x * s_a[net][k] + w4.y * s_a[net][k + 1] + w4.z * s_a[net][k + 2] + w4.w * s_a[net][k + 3];
        }
      } else {
        for (int k = 0; k < 256; ++k) acc += wr[k] * s_a[net][k];
      }
      s_b[net][u] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (my_net == 0) { o[0] = s_b[0][u] * A.w3[0][u]; o[1] = s_b[0][u] * A.w3[0][256 + u]; o[2] = s_b[0][u] * A.w3[0][512 + u]; }
    else o[3] = s_b[1][u] * A.w3[1][u];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int sh = 32; sh >= 1; sh >>= 1) o[k] += __shfl_xor(o[k], sh, 64);
      if ((u & 63) == 0) s_red[4 + k][u >> 6] = o[k];
    }
    __syncthreads();
    if (u == 0) {
      float z[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) z[k] = s_red[4 + k][0] + s_red[4 + k][1] + s_red[4 + k][2] + s_red[4 + k][3];
      if (my_net == 0) finish_row(A.S, SC, row, z[0] + A.b3[0][0], z[1] + A.b3[0][1], z[2] + A.b3[0][2]);
      else A.S.val[row] = z[3] + A.b3[1][0];
    }
  }
  // the last workgroup out resets the list for the next step
  __syncthreads();
  if (u == 0) {
    __threadfence();
    if (atomicAdd(A.done_blocks, 1) == (int)gridDim.x - 1) { *A.count = 0; *A.done_blocks = 0; __threadfence(); }
  }
}

}  // namespace rvo3d
