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
// Weights.  One workgroup (8 waves, one per CU: 2 per SIMD) serves ONE of the two networks and keeps its second layer
// (128 KB of packed bf16 fragments, 1 KB per wave-load, lane-linear: conflict-free ds_read_b128) in LDS for its whole
// life; the first layer (8 tiles of KS1 KB) is streamed through a double-buffered 2-tile LDS stage that the workgroup
// fills in lockstep (global_load_lds: no registers; one barrier per stage, the next stage in flight behind the MFMAs).  The first
// layer's bias rides in the weight column k_in against a constant 1 in the activations; the second layer's bias is the
// C operand of the first MFMA of a tile; the heads are a third chained product (3 or 1 of 32 rows used).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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
  PolicySampleArgs S;         // tanh_out, log_std, std_factor, seed, step, rows, act / logp / val, dbg_*
};

__device__ __forceinline__ float relu_f32(float x) {
  // one instruction (fmaxf would canonicalise first); a builtin, not inline asm: the compiler pads the MFMA -> VALU
  // read hazard for its own instructions only
  return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff());
}
#ifndef RVO3D_MLP_ABL
#define RVO3D_MLP_ABL 0  // (timing experiments only: 1 no sampling, 2 no stage barriers, 8 no row prefetch)
#endif
template <int KS1, int NW>
__global__ void __launch_bounds__(64 * NW) policy_mlp_kernel(const PolicyMlpArgs A) {
  constexpr int TPS = KS1 <= 7 ? 2 : 1;                 // first-layer tiles per LDS stage
  constexpr int SPC = 8 / TPS;                          // stages per pass (even)
  constexpr int kStageBytes = TPS * KS1 * 1024;
  constexpr int kWaveLoads = TPS * KS1;                 // 1 KB pieces (one wave-wide 16-byte load each) of a stage
  static_assert(kMlpResidentBytes + 2 * kStageBytes <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const w2s = smem;
  const float* const b2t = reinterpret_cast<const float*>(smem + kMlpW2Bytes);
  const unsigned char* const w3s = smem + kMlpW2Bytes + kMlpB2Bytes;
  unsigned char* const stage = smem + kMlpResidentBytes;

  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: the staging loops are scalar loops)
  const int net = blockIdx.x & 1, g = blockIdx.x >> 1, G = gridDim.x >> 1;
  const unsigned char* const blob = A.blob + net * A.net_bytes;
  const int64_t rows = A.S.rows;
  const int64_t nchunks = (rows + 63) >> 6;
  const int iters = (int)((nchunks + (int64_t)G * NW - 1) / ((int64_t)G * NW));

  // resident part: second layer, its bias, the head
  {
    const uint4* src = reinterpret_cast<const uint4*>(blob + (int64_t)8 * KS1 * 1024);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = tid; i < kMlpResidentBytes / 16; i += 64 * NW) dst[i] = src[i];
  }
  // first-layer stage sc -> buffer b, global -> LDS without registers (lane-linear on both sides)
  auto dma_stage = [&](int sc, int b) {
    for (int i = wave; i < kWaveLoads; i += NW)
      __builtin_amdgcn_global_load_lds(
          (const void __attribute__((address_space(1)))*)(blob + ((int64_t)(sc * kWaveLoads + i) * 64 + lane) * 16),
          (void __attribute__((address_space(3)))*)(stage + b * kStageBytes + i * 1024), 16, 0, 0);
  };
  dma_stage(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();

  const float4 head_bias = *reinterpret_cast<const float4*>(blob + A.net_bytes - kMlpHeadBiasBytes);

  // The observation rows are read through a buffer descriptor over exactly the bytes the caller owns: the 16-wide
  // k-steps run past a row's end (into the next row: masked below) and, for the last row, past the array's end,
  // where the hardware's range check returns zeros instead of touching memory.
  const __amdgpu_buffer_rsrc_t obs_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(A.obs), 0, (int)(uint32_t)(((rows - 1) * A.ld_obs + A.k_in) * 4), 0x00020000);
  f32x8 Xraw[KS1];  // the NEXT pass's 32 rows, requested while the current pass is in its second layer
  auto request_rows = [&](int pass) {
    const int64_t c = (int64_t)g * NW + wave + (int64_t)(pass >> 1) * G * NW;
    int64_t row = c * 64 + 32 * (pass & 1) + r;
    if (row >= rows) row = rows - 1;  // (a ragged tail / an idle wave re-reads the last row; nothing is stored)
    const uint32_t off = (uint32_t)((row * A.ld_obs + 8 * h) * 4);
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      const float4 lo = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(obs_rs, off + 64 * s, 0, 0));
      const float4 hi = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(obs_rs, off + 64 * s + 16, 0, 0));
      Xraw[s] = f32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
  };
  request_rows(0);

  float zs0 = 0.f, zs1 = 0.f, zs2 = 0.f;
#pragma unroll 1
  for (int pass = 0; pass < 2 * iters; ++pass) {
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
    // Both layers are written as explicit software pipelines, one `sched_barrier` per MFMA: the A fragment of step
    // i + D is requested before the MFMA of step i, and the previous tile's epilogue (ReLU, conversion to the next
    // product's B fragments: 24 VALU instructions) is spread over the current tile's MFMAs, in whose shadow it runs -
    // two accumulators alternate.  (Left alone the scheduler sinks every LDS read to just before its MFMA and the
    // optimiser defers all epilogues of a layer to its end, with every accumulator live.)
    u32x4 H1[16];  // H1^T [256][32] as the 16 B fragments of the second product
    f32x16 accs[2];
    // epilogue step q = 0..7 of a finished tile: registers 2 q, 2 q + 1 -> ReLU -> one packed bf16 pair
    auto epi = [&](const f32x16& acc, int q) -> uint32_t {
      uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(
                                                    f32x2{relu_f32(acc[2 * q]), relu_f32(acc[2 * q + 1])}, bf16x2));
      asm volatile("" : "+v"(w));  // (pinned to this slot of the pipeline)
      return w;
    };
    // ---- layer 1: H1^T = relu(W1 X^T), tile by tile out of the stage buffers ----
#pragma unroll
    for (int sc = 0; sc < SPC; ++sc) {
      // the next stage lands in the other buffer (everybody finished reading it one barrier ago) meanwhile
      dma_stage((sc + 1) % SPC, (sc + 1) & 1);
      const unsigned char* const sb = stage + (sc & 1) * kStageBytes;
      constexpr int D1 = 4, kSteps = TPS * KS1;
      bf16x8 ring[D1];
#pragma unroll
      for (int i = 0; i < D1 && i < kSteps; ++i) ring[i] = *reinterpret_cast<const bf16x8*>(sb + (i * 64 + lane) * 16);
#pragma unroll
      for (int i = 0; i < kSteps; ++i) {
        const int ml = i / KS1, s2 = i % KS1, m = sc * TPS + ml;
        const bf16x8 a = ring[i % D1];
        if (i + D1 < kSteps) ring[i % D1] = *reinterpret_cast<const bf16x8*>(sb + ((i + D1) * 64 + lane) * 16);
        if (s2 == 0) {
          const f32x16 z = {0};
          accs[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, X[s2], z, 0, 0, 0);
        } else {
          accs[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, X[s2], accs[m & 1], 0, 0, 0);
        }
        if (m > 0) {  // the previous tile's epilogue: 8 steps over KS1 MFMAs
#pragma unroll
          for (int q = (8 * s2) / KS1; q < (8 * (s2 + 1)) / KS1; ++q)
            H1[2 * (m - 1) + (q >> 2)][q & 3] = epi(accs[(m - 1) & 1], q);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#if !(RVO3D_MLP_ABL & 2)
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's part of the next stage has landed
      __syncthreads();
#endif
    }
#if !(RVO3D_MLP_ABL & 8)
    if (pass + 1 < 2 * iters) request_rows(pass + 1);
#endif
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
      // one flat stream of 128 A fragments (8 tiles x 16 k-steps, consecutive in LDS), D2 of them in flight
      const unsigned char* const wb = w2s + lane * 16;
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
        const f32x16& prev = accs[(m2 & 1) ^ 1];  // layer 1's last tile when m2 == 0 (7 is odd)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int i = m2 * 16 + t;
          const bf16x8 a = ring[i % D2];
          if (i + D2 < 128) ring[i % D2] = *reinterpret_cast<const bf16x8*>(wb + (i + D2) * 1024);
          cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, H1[t]), t == 0 ? bias : cur, 0, 0, 0);
          if (t >= 1 && t <= 8) {  // the previous tile's epilogue, one packed pair per step
            const int q = t - 1;
            if (m2 == 0) H1[14 + (q >> 2)][q & 3] = epi(prev, q);
            else h2[q >> 2][q & 3] = epi(prev, q);
          }
          if (m2 > 0 && t == 6) {
            a3[0] = *reinterpret_cast<const bf16x8*>(w3l + (2 * (m2 - 1)) * 128);
            a3[1] = *reinterpret_cast<const bf16x8*>(w3l + (2 * (m2 - 1) + 1) * 128);
          }
          if (t == 8 && m2 < 7) bias = read_bias(m2 + 1);
          if (m2 > 0 && t == 10) hd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], __builtin_bit_cast(bf16x8, h2[0]), hd, 0, 0, 0);
          if (m2 > 0 && t == 12) hd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], __builtin_bit_cast(bf16x8, h2[1]), hd, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the last tile's epilogue and head products
      a3[0] = *reinterpret_cast<const bf16x8*>(w3l + 14 * 128);
      a3[1] = *reinterpret_cast<const bf16x8*>(w3l + 15 * 128);
#pragma unroll
      for (int q = 0; q < 8; ++q) h2[q >> 2][q & 3] = epi(accs[1], q);
      hd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0], __builtin_bit_cast(bf16x8, h2[0]), hd, 0, 0, 0);
      hd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[1], __builtin_bit_cast(bf16x8, h2[1]), hd, 0, 0, 0);
    }
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
      if (net == 0) finish_row(A.S, row, z0, z1, z2);
#endif
      else A.S.val[row] = z0;
    }
  }
}

}  // namespace rvo3d
