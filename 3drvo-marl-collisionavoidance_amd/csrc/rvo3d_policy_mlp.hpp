// rvo3d_policy_mlp.hpp -- config 3's policy step in ONE kernel: MLP(256, 256) actor and critic on the env's
// observation rows (cast, two hidden layers, heads), tanh, Gaussian sample, log-probability, np.round(a, 2) and the
// buffer stores, on the matrix cores (v_mfma_f32_32x32x16_bf16) with every activation kept in registers.
// Reference: train/policy/policy_rnn_ac.py:57-69 (ac.step), :197-235 (GaussianActor), :238-257 (Critic) with the
// MLP(256, 256) of BASELINE config 3; train/policy/multi_ppo.py:193-197 (the rollout's policy call).
// Part of the gfx950 device code (see rvo3d_device.hpp for the overview).
//
// Orientation.  Everything is computed TRANSPOSED: H1^T = W1 X^T, H2^T = W2 H1^T, head^T = W3 H2^T, so that the
// weights are the A operand and the activations the B operand of every product.  A 32 x 32 result tile then has the
// batch row on the lane and the hidden unit in the 16 accumulator registers, and the next layer - which sums over
// hidden units - takes the converted accumulators as its B fragments as they are: no LDS, no lane movement between
// layers (the k order inside a 16-wide step is permuted, element j of lane half h is unit 8 (j >> 2) + 4 h + (j & 3):
// the packed weights of the following layer are stored in that order).  A wave takes 32 batch rows at a time through all layers (28 + 64 registers of bf16
// activations, 16 + 16 of accumulators) and finishes 64 rows - two passes - at once, one row per lane.
//
// Weights.  One workgroup (8 waves, one per CU: 2 per SIMD) serves ONE of the two networks and keeps in LDS, for its
// whole life, that network's first layer (8 KS1 KB of packed bf16 fragments, 1 KB per wave-load, lane-linear:
// conflict-free ds_read_b128), the second layer's bias table, the head rows and 12 of the 16 k-steps of every
// second-layer tile (96 KB); the other four k-steps per tile (32 KB per pass) every wave streams from L2 into registers
// one tile ahead.  After the one barrier behind that copy the waves run free.  The first layer's bias rides in the
// weight column k_in against a constant 1 in the activations; the second layer's bias is the C operand of the first
// MFMA of a tile; the heads are a third chained product (3 or 1 of 32 rows used).
// (An earlier version kept the whole second layer in LDS and streamed the first through a double-buffered LDS stage
// with a workgroup barrier per stage: the barriers kept the two waves of a SIMD in step - MFMA phases together,
// conversion / waits / sampling together - and the matrix pipe was busy 48 % of the time.)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "rvo3d_rollout_kernels.hpp"

namespace rvo3d {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMlpH = 256;                                   // hidden width of both layers
constexpr int kMlpW2Bytes = 8 * 16 * 1024;                   // [8 tiles][16 k-steps][64 lanes][8] bf16
constexpr int kMlpB2Bytes = 8 * 2 * 16 * 4;                  // [8 tiles][2 lane halves][16 registers] float
constexpr int kMlpW3Bytes = 16 * 4 * 2 * 16;                 // [16 k-steps][4 rows: 3 heads, zeros][2 lane halves][8] bf16
constexpr int kMlpHeadBiasBytes = 16;                        // float[4]
constexpr int kMlpResidentBytes = kMlpW2Bytes + kMlpB2Bytes + kMlpW3Bytes;
__host__ __device__ constexpr int mlp_ks1(int k_in) { return (k_in + 1 + 15) / 16; }
__host__ __device__ constexpr int64_t mlp_net_bytes(int ks1) {
  return (int64_t)8 * ks1 * 1024 + kMlpResidentBytes + kMlpHeadBiasBytes;
}
// hidden unit (within a 32-unit tile) that accumulator register i of lane half h holds
__host__ __device__ constexpr int mlp_acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---- packing: nn.Linear weights (float32, [out][in]) -> the fragments the kernel reads ----------------------------
struct MlpPackArgs {
  int32_t k_in, ks1;
  const float* w1[2];  // [256][k_in]
  const float* b1[2];  // [256]
  const float* w2[2];  // [256][256]
  const float* b2[2];  // [256]
  const float* w3[2];  // [3][256] actor, [1][256] critic
  const float* b3[2];  // [3] / [1]
  unsigned char* blob;  // 2 x mlp_net_bytes(ks1)
};
__global__ void __launch_bounds__(256) mlp_pack_kernel(const MlpPackArgs A) {
  const int net = blockIdx.y;
  const int64_t nb = mlp_net_bytes(A.ks1);
  unsigned char* const blob = A.blob + net * nb;
  const int n_w1 = 8 * A.ks1 * 512, n_w2 = 8 * 16 * 512, n_b2 = 256, n_w3 = 16 * 4 * 2 * 8, n_hb = 4;
  const int total = n_w1 + n_w2 + n_b2 + n_w3 + n_hb;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    int i = idx;
    if (i < n_w1) {  // [m][s][lane][j]: W1[32 m + r][16 s + 8 h + j], the bias in column k_in
      const int j = i & 7, lane = (i >> 3) & 63, ms = i >> 9, s = ms % A.ks1, m = ms / A.ks1;
      const int row = 32 * m + (lane & 31), k = 16 * s + 8 * (lane >> 5) + j;
      const float v = k < A.k_in ? A.w1[net][(int64_t)row * A.k_in + k] : (k == A.k_in ? A.b1[net][row] : 0.f);
      reinterpret_cast<uint16_t*>(blob)[i] = f32_to_bf16_rne(v);
      continue;
    }
    i -= n_w1;
    unsigned char* p = blob + (int64_t)8 * A.ks1 * 1024;
    if (i < n_w2) {  // [m2][t][lane][j]: W2[32 m2 + r][the unit the previous layer's fragment holds at (t, h, j)]
      const int j = i & 7, lane = (i >> 3) & 63, t = (i >> 9) & 15, m2 = i >> 13;
      const int row = 32 * m2 + (lane & 31), k = 32 * (t >> 1) + 16 * (t & 1) + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
      reinterpret_cast<uint16_t*>(p)[i] = f32_to_bf16_rne(A.w2[net][row * kMlpH + k]);
      continue;
    }
    i -= n_w2; p += kMlpW2Bytes;
    if (i < n_b2) {  // [m2][h][reg]
      const int reg = i & 15, h = (i >> 4) & 1, m2 = i >> 5;
      reinterpret_cast<float*>(p)[i] = A.b2[net][32 * m2 + mlp_acc_row(reg, h)];
      continue;
    }
    i -= n_b2; p += kMlpB2Bytes;
    if (i < n_w3) {  // [t][row][h][j]
      const int j = i & 7, h = (i >> 3) & 1, tr = i >> 4, row = tr & 3, t = tr >> 2;
      const int k = 32 * (t >> 1) + 16 * (t & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
      const int n_out = net == 0 ? 3 : 1;
      reinterpret_cast<uint16_t*>(p)[i] = f32_to_bf16_rne(row < n_out ? A.w3[net][row * kMlpH + k] : 0.f);
      continue;
    }
    i -= n_w3; p += kMlpW3Bytes;
    reinterpret_cast<float*>(p)[i] = i < (net == 0 ? 3 : 1) ? A.b3[net][i] : 0.f;
  }
}

// ---- the policy step ---------------------------------------------------------------------------------------------
struct PolicyMlpArgs {
  const unsigned char* blob;  // packed weights of the two nets (mlp_pack_kernel)
  int64_t net_bytes;
  const float* obs;           // [rows][ld_obs] float32: the env's observation rows
  int64_t ld_obs;
  int32_t k_in;               // observation width (12 + 9 nm)
  const int32_t* cnt;         // optional [rows]: the env's vo_count - a row holds state_dim + row_dim * cnt floats, zeros behind
  int32_t state_dim, row_dim; // 12, 9
  PolicySampleArgs S;         // tanh_out, log_std, std_factor, seed, step, rows, act / logp / val, dbg_*
};

__device__ __forceinline__ float relu_f32(float x) {
  // ONE instruction beside the MFMAs (v_max_i32: as integers, negative floats are negative, positive ones keep their
  // order).  fmaxf / fmed3 cost two - the compiler canonicalises the operand first -, and inline asm is out: the
  // compiler pads the MFMA -> VALU read hazard for its own instructions only.
  const int i = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, i > 0 ? i : 0);
}
#ifndef RVO3D_MLP_ABL
#define RVO3D_MLP_ABL 0  // (timing experiments only: 1 no sampling, 8 no observation loads)
#endif
#ifndef RVO3D_MLP_XPREFETCH
#define RVO3D_MLP_XPREFETCH 0  // 1: the next pass's rows are requested behind the last tile of layer 2
#endif
#ifndef RVO3D_MLP_CLOCK
#define RVO3D_MLP_CLOCK 0
#endif
// second-layer k-steps per tile whose fragments stay in LDS; the other 16 - TR are streamed from L2 per wave
// (as many as fit beside the first layer: all 16 for narrow inputs - nothing streamed -, 12 at config 3's width)
__host__ __device__ constexpr int mlp_tr(int ks1) {
  return (157 - 8 * ks1) / 8 >= 16 ? 16 : (157 - 8 * ks1) / 8;
}
__host__ __device__ constexpr int mlp_lds_bytes(int ks1) {
  return 8 * ks1 * 1024 + kMlpB2Bytes + kMlpW3Bytes + 8 * mlp_tr(ks1) * 1024;
}

template <int KS1, int NW>
__global__ void __launch_bounds__(64 * NW) policy_mlp_kernel(const PolicyMlpArgs A) {
  constexpr int TR = mlp_tr(KS1), NS = 16 - TR;
  static_assert(mlp_lds_bytes(KS1) <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: the whole first layer, the second layer's bias table, the head rows, TR of 16 k-steps of every second-layer tile
  const unsigned char* const w1s = smem;
  const float* const b2t = reinterpret_cast<const float*>(smem + 8 * KS1 * 1024);
  const unsigned char* const w3s = smem + 8 * KS1 * 1024 + kMlpB2Bytes;
  const unsigned char* const w2r = w3s + kMlpW3Bytes;

#if RVO3D_MLP_CLOCK
  const uint64_t clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the actor's and the critic's workgroup of one group of rows on the SAME XCD (workgroup b runs on XCD b % 8): they
  // read the same observation rows at about the same time, the second reader finds them in that XCD's L2
  const bool paired = (gridDim.x & 15) == 0;
  const int net = paired ? (blockIdx.x >> 3) & 1 : blockIdx.x & 1;
  const int g = paired ? (blockIdx.x >> 4) * 8 + (blockIdx.x & 7) : blockIdx.x >> 1, G = gridDim.x >> 1;
  const unsigned char* const blob = A.blob + net * A.net_bytes;
  const unsigned char* const gw2 = blob + (int64_t)8 * KS1 * 1024;  // [8][16] blocks of 1 KB
  const int64_t rows = A.S.rows;
  const int64_t nchunks = (rows + 63) >> 6;
  const int iters = (int)((nchunks + (int64_t)G * NW - 1) / ((int64_t)G * NW));

  {  // the resident weights: the only workgroup-wide step of the kernel
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const uint4* src = reinterpret_cast<const uint4*>(blob);
    for (int i = tid; i < 8 * KS1 * 64; i += 64 * NW) dst[i] = src[i];
    dst += 8 * KS1 * 64;
    src = reinterpret_cast<const uint4*>(gw2 + kMlpW2Bytes);
    for (int i = tid; i < (kMlpB2Bytes + kMlpW3Bytes) / 16; i += 64 * NW) dst[i] = src[i];
    dst += (kMlpB2Bytes + kMlpW3Bytes) / 16;
    src = reinterpret_cast<const uint4*>(gw2);
    for (int i = tid; i < 8 * TR * 64; i += 64 * NW) {
      const int blk = i >> 6, m2 = blk / TR, t = blk - m2 * TR;
      dst[i] = src[(m2 * 16 + t) * 64 + (i & 63)];
    }
  }
  __syncthreads();
  // From here on the waves run free: no barrier, nothing shared is written.  (Measured and left out: starting the second
  // half of the waves half a pass late, 110 vs 107 us; s_setprio 1 for the younger half - the half that gets the SIMD's
  // issue slots finishes after 68 us, the other after 90-100 us, whichever it is: 104 us either way.)
  const float4 head_bias = *reinterpret_cast<const float4*>(blob + A.net_bytes - kMlpHeadBiasBytes);
  const SampleConsts SC = sample_consts(A.S);

  // The observation rows are read through a buffer descriptor over exactly the bytes the caller owns: the 16-wide
  // k-steps run past a row's end (into the next row: masked below) and, for the last row, past the array's end,
  // where the hardware's range check returns zeros instead of touching memory.
  const __amdgpu_buffer_rsrc_t obs_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A.obs), 0, (int)(uint32_t)(((rows - 1) * A.ld_obs + A.k_in) * 4), 0x00020000);
  f32x8 Xraw[KS1];
  // Sparse rows.  The env writes an observation row as state_dim floats, then row_dim floats per kept velocity-obstacle
  // row, then zeros (in a rollout nearly every row has none or one: 21 of 102 floats); with the env's vo_count at hand a
  // wave knows how many leading 16-wide k-steps hold data for ANY of its 32 rows: the others are neither loaded nor
  // multiplied (a zero activation adds exactly nothing to a float32 sum; the last k-step always runs - it carries the
  // bias column).  n_data = KS1: no counts given, everything is loaded.
  auto data_steps = [&](int pass) -> int {
    if (!A.cnt) return KS1;
    const int64_t c = (int64_t)g * NW + wave + (int64_t)(pass >> 1) * G * NW;
    int64_t row = c * 64 + 32 * (pass & 1) + r;
    if (row >= rows) row = rows - 1;
    int cn = A.cnt[row];
    cn = cn < 0 ? 0 : cn;  // (count 0 = the single all-zero row: no data behind the state)
    const int mine = (A.state_dim + A.row_dim * cn + 15) >> 4;
    int n = 1;
#pragma unroll
    for (int q = 1; q < KS1; ++q) n = __builtin_amdgcn_ballot_w64(mine > q) != 0 ? q + 1 : n;
    return __builtin_amdgcn_readfirstlane(n);
  };
  auto request_rows = [&](int pass, int n_data) {
#if !(RVO3D_MLP_ABL & 8)
    const int64_t c = (int64_t)g * NW + wave + (int64_t)(pass >> 1) * G * NW;
    int64_t row = c * 64 + 32 * (pass & 1) + r;
    if (row >= rows) row = rows - 1;  // (a ragged tail / an idle wave re-reads the last row; nothing is stored)
    const uint32_t off = (uint32_t)((row * A.ld_obs + 8 * h) * 4);
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      if (s < n_data) {
        const float4 lo = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(obs_rs, off + 64 * s, 0, 0));
        const float4 hi = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(obs_rs, off + 64 * s + 16, 0, 0));
        Xraw[s] = f32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      } else {
        Xraw[s] = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      }
    }
#endif
  };
#if RVO3D_MLP_XPREFETCH
  request_rows(0, KS1);
#endif

  // Both layers are written as explicit software pipelines, one `sched_barrier` per MFMA: the A fragment of step
  // i + D is requested before the MFMA of step i, and the previous tile's epilogue (ReLU, conversion to the next
  // product's B fragments: 24 VALU instructions) is spread over the current tile's MFMAs, in whose shadow it runs -
  // two accumulators alternate.  (Left alone the scheduler sinks every LDS read to just before its MFMA and the
  // optimiser defers all epilogues of a layer to its end, with every accumulator live.)
  // epilogue step q = 0..7 of a finished tile: registers 2 q, 2 q + 1 -> ReLU -> one packed bf16 pair
  auto epi = [&](const f32x16& acc, int q) -> uint32_t {
    uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(
                                                  f32x2{relu_f32(acc[2 * q]), relu_f32(acc[2 * q + 1])}, bf16x2));
    asm volatile("" : "+v"(w));  // (pinned to this slot of the pipeline)
    return w;
  };
  auto stream_frag = [&](int m2, int u) {  // k-step TR + u of second-layer tile m2, from L2
    return *reinterpret_cast<const bf16x8*>(gw2 + ((int64_t)((m2 * 16 + TR + u) * 64 + lane)) * 16);
  };

  float zs0 = 0.f, zs1 = 0.f, zs2 = 0.f;
#if RVO3D_MLP_CLOCK
  uint64_t tk[4] = {0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define RVO3D_MLP_STAMP(i) { const uint64_t tn = __builtin_amdgcn_s_memtime(); tk[i] += tn - tprev; tprev = tn; }
#else
#define RVO3D_MLP_STAMP(i)
#endif
#pragma unroll 1
  for (int pass = 0; pass < 2 * iters; ++pass) {
    RVO3D_MLP_STAMP(3)
#if !RVO3D_MLP_XPREFETCH
    const int n_data = data_steps(pass);
    request_rows(pass, n_data);
#else
    const int n_data = KS1;
#endif
    // ---- 32 observation rows as the B fragments of the first product (cast to bf16 on the way) ----
    bf16x8 X[KS1];
#pragma unroll
    for (int s = 0; s < KS1 - 1; ++s) X[s] = __builtin_convertvector(Xraw[s], bf16x8);
    {  // the last step: the row's tail, the constant 1 that multiplies the bias column, zeros
      constexpr int s = KS1 - 1;
      f32x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * h + j;
        v[j] = k < A.k_in ? Xraw[s][j] : (k == A.k_in ? 1.0f : 0.0f);
      }
      X[s] = __builtin_convertvector(v, bf16x8);
    }
    asm volatile("" :: "v"(X[0]), "v"(X[KS1 - 1]));
    RVO3D_MLP_STAMP(0)
    u32x4 H1[16];  // H1^T [256][32] as the 16 B fragments of the second product
    f32x16 accs[2];
    bf16x8 S[NS > 0 ? NS : 1];  // the streamed fragments of the next second-layer tile (none when everything is resident)
    // ---- layer 1: H1^T = relu(W1 X^T): one flat stream of fragments out of LDS ----
    // Three straight-line versions, chosen per pass (wave-uniform): every k-step; the first ND_SPARSE k-steps (rows with
    // at most two kept VO rows) plus the bias step; the first k-step (rows without any: nearly all of a rollout) plus
    // the bias step.
    constexpr int ND_SPARSE = 2;
    auto layer1 = [&](auto nd_tag) {
      constexpr int ND = decltype(nd_tag)::value;                  // leading k-steps with data
      constexpr int NSTEP = ND < KS1 ? ND + 1 : KS1;               // MFMAs per tile: those + the last (bias) step
      constexpr int D1 = 4, kSteps = 8 * NSTEP;
      auto k_of = [](int j) constexpr { return j < ND ? j : KS1 - 1; };   // the k-step of a tile's j-th MFMA
      const unsigned char* const wb = w1s + lane * 16;
      auto frag = [&](int i) { return *reinterpret_cast<const bf16x8*>(wb + ((i / NSTEP) * KS1 + k_of(i % NSTEP)) * 1024); };
      bf16x8 ring[D1];
#pragma unroll
      for (int i = 0; i < D1; ++i) ring[i] = frag(i);
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int j = 0; j < NSTEP; ++j) {
        const int i = m * NSTEP + j, s2 = k_of(j);
        const bf16x8 a = ring[i % D1];
        if (i + D1 < kSteps) ring[i % D1] = frag(i + D1);
        if (i == kSteps - NSTEP) {
#pragma unroll
          for (int u = 0; u < NS; ++u) S[u] = stream_frag(0, u);
        }
        if (j == 0) {
          const f32x16 z = {0};
          accs[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, X[s2], z, 0, 0, 0);
        } else {
          accs[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, X[s2], accs[m & 1], 0, 0, 0);
        }
        if (m > 0) {  // the previous tile's epilogue: 8 steps over NSTEP MFMAs
#pragma unroll
          for (int q = (8 * j) / NSTEP; q < (8 * (j + 1)) / NSTEP; ++q)
            H1[2 * (m - 1) + (q >> 2)][q & 3] = epi(accs[(m - 1) & 1], q);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // (the last tile's epilogue is not overlapped: the second layer's first MFMAs - the streamed k-steps 12..15 -
      // need it at once)
#pragma unroll
      for (int q = 0; q < 8; ++q) H1[14 + (q >> 2)][q & 3] = epi(accs[1], q);
    };
    if (KS1 > ND_SPARSE + 1 && n_data <= 1) layer1(std::integral_constant<int, (KS1 > ND_SPARSE + 1 ? 1 : KS1)>{});
    else if (KS1 > ND_SPARSE + 1 && n_data <= ND_SPARSE) layer1(std::integral_constant<int, (KS1 > ND_SPARSE + 1 ? ND_SPARSE : KS1)>{});
    else layer1(std::integral_constant<int, KS1>{});
    RVO3D_MLP_STAMP(1)
    // ---- layer 2 + heads: H2^T = relu(W2 H1^T + b2), head^T += W3 H2^T ----
    f32x16 hd = {0};
    {
      constexpr int D2 = 4;
      auto read_bias = [&](int m2) {
        f32x16 b;
        const float4* bp = reinterpret_cast<const float4*>(b2t + (m2 * 2 + h) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = bp[q];
          b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
        }
        return b;
      };
      // per tile: first the NS streamed k-steps (their registers are then free for the next tile's request, which has
      // the TR resident steps to land), then the TR resident ones - one flat stream of 8 TR fragments in LDS, D2 in flight
      const unsigned char* const wb = w2r + lane * 16;
      bf16x8 ring[D2];
#pragma unroll
      for (int i = 0; i < D2; ++i) ring[i] = *reinterpret_cast<const bf16x8*>(wb + i * 1024);
      f32x16 bias = read_bias(0);
      const unsigned char* const w3l = w3s + ((r < 3 ? r : 3) * 2 + h) * 16;  // (row 3 is zeros)
      u32x4 h2[2];
      bf16x8 a3[2];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m2 = 0; m2 < 8; ++m2) {
        f32x16& cur = accs[m2 & 1];
        const f32x16& prev = accs[(m2 & 1) ^ 1];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          bf16x8 a;
          int t;  // the k-step this MFMA covers
          if (v < NS) {
            a = S[v < NS ? v : 0]; t = TR + v;
          } else {
            const int i = m2 * TR + (v - NS);
            a = ring[i % D2]; t = v - NS;
            if (i + D2 < 8 * TR) ring[i % D2] = *reinterpret_cast<const bf16x8*>(wb + (i + D2) * 1024);
          }
          cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, H1[t]), v == 0 ? bias : cur, 0, 0, 0);
          if (v == NS - 1 && m2 < 7) {
#pragma unroll
            for (int u = 0; u < NS; ++u) S[u] = stream_frag(m2 + 1, u);
          }
          if (m2 > 0 && v >= 1 && v <= 8) {  // the previous tile's epilogue, one packed pair per step
            const int q = v - 1;
            h2[q >> 2][q & 3] = epi(prev, q);
          }
          if (m2 > 0 && v == 6) {
            a3[0] = *reinterpret_cast<const bf16x8*>(w3l + (2 * (m2 - 1)) * 128);
            a3[1] = *reinterpret_cast<const bf16x8*>(w3l + (2 * (m2 - 1) + 1) * 128);
          }
          if (v == 8 && m2 < 7) bias = read_bias(m2 + 1);
          if (m2 > 0 && v == 10) hd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], __builtin_bit_cast(bf16x8, h2[0]), hd, 0, 0, 0);
          if (m2 > 0 && v == 12) hd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], __builtin_bit_cast(bf16x8, h2[1]), hd, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#if RVO3D_MLP_XPREFETCH
      if (pass + 1 < 2 * iters) request_rows(pass + 1, KS1);
#endif
      // the last tile's epilogue and head products
      a3[0] = *reinterpret_cast<const bf16x8*>(w3l + 14 * 128);
      a3[1] = *reinterpret_cast<const bf16x8*>(w3l + 15 * 128);
#pragma unroll
      for (int q = 0; q < 8; ++q) h2[q >> 2][q & 3] = epi(accs[1], q);
      hd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], __builtin_bit_cast(bf16x8, h2[0]), hd, 0, 0, 0);
      hd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], __builtin_bit_cast(bf16x8, h2[1]), hd, 0, 0, 0);
    }
    asm volatile("" :: "v"(hd));
    RVO3D_MLP_STAMP(2)
    // ---- rows 0..2 of a head tile sit in registers 0..2 of lanes 0..31: the lower half keeps the first pass's and
    // finishes those 32 rows after the second pass, when the upper half takes the second pass's ----
    if ((pass & 1) == 0) {
      zs0 = hd[0]; zs1 = hd[1]; zs2 = hd[2];
      continue;
    }
    const float o0 = __shfl_xor(hd[0], 32, 64), o1 = __shfl_xor(hd[1], 32, 64), o2 = __shfl_xor(hd[2], 32, 64);
    const float z0 = (h ? o0 : zs0) + head_bias.x, z1 = (h ? o1 : zs1) + head_bias.y, z2 = (h ? o2 : zs2) + head_bias.z;
    const int64_t c = (int64_t)g * NW + wave + (int64_t)(pass >> 1) * G * NW;
    const int64_t row = c * 64 + lane;
    if (row < rows) {
#if RVO3D_MLP_ABL & 1
      if (net == 0) A.S.logp[row] = z0 + z1 + z2;
#else
      if (net == 0) finish_row(A.S, SC, row, z0, z1, z2);
#endif
      else A.S.val[row] = z0;
    }
  }
#if RVO3D_MLP_CLOCK  // (experiment: shader cycles and 100 MHz ticks of one wave's life)
  if (tid == 0 && net == 1 && g == 0) {
    A.S.act[0] = (float)(__builtin_amdgcn_s_memtime() - clk0);
    A.S.act[1] = (float)(__builtin_amdgcn_s_memrealtime() - rt0);
  }
  if (lane == 0 && A.S.dbg_mu) {  // per wave: start / end in 100 MHz ticks (low 24 bits), XCC id
    float* o = A.S.dbg_mu + 4 * (blockIdx.x * NW + wave);
    o[0] = (float)(rt0 & 0xFFFFFF); o[1] = (float)(__builtin_amdgcn_s_memrealtime() & 0xFFFFFF);
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    o[2] = (float)(xcc & 15); o[3] = (float)((hw >> 8) & 0xff);  // cu_id[11:8] + sh/se bits
    float* o2 = A.S.dbg_mu + 4 * (gridDim.x * NW) + 4 * (blockIdx.x * NW + wave);
    o2[0] = (float)tk[0]; o2[1] = (float)tk[1]; o2[2] = (float)tk[2]; o2[3] = (float)tk[3];
  }
#endif
}

}  // namespace rvo3d
