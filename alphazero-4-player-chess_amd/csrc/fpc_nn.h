// fpc_nn.h -- the policy/value ResNet of net.py:6-63 as hand-written gfx950 MFMA kernels.
//
//   ResNet(gameType, num_resBlocks, num_hidden): stem conv3x3(24->F)+BN+ReLU, N x ResBlock
//   (conv-BN-ReLU-conv-BN-+res-ReLU), policy head conv3x3(F->A_ch)+BN+ReLU+Flatten+Linear(A->A),
//   value head conv3x3(F->24)+BN+ReLU+Flatten+Linear(24*R*R->1)+Tanh.   eval() mode only
//   (mcts.py:15 @torch.no_grad, alphazero.py:262), so every BN is folded into its conv on export
//   (alphazero-4-player-chess_amd/weights.py).
//
// MI355X-first layout: activations are 16-bit NHWC on a zero-bordered (R+2)x(R+2) grid per game
// (16x16 = 256 rows per game at 14x14), so a 3x3 convolution is an implicit GEMM whose nine taps
// are nine row-shifted views of ONE matrix:  Y[m, co] = sum_t sum_ci X[m + off_t, ci] * W[t, co, ci]
// -- no im2col, no bounds checks in the inner loop.  M = games*256 rows, N = Cout, K = 9*Cin.
// One kernel template (k_gemm16) serves all convs and the policy Linear (ntaps = 1); tiles are
// 128x128xBK, 4 waves (2x2) of 64x64, v_mfma_f32_32x32x16_{bf16,f16}, LDS double-buffered with a
// 16-byte-chunk XOR swizzle that makes every ds_read_b128 fragment read conflict-free.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "fpc_tree_kernels.h"

namespace fpc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

template <int DT>
struct E16;
template <>
struct E16<0> {  // bf16
  static __device__ __forceinline__ uint16_t from_f32(float f) { __bf16 h = (__bf16)f; return __builtin_bit_cast(uint16_t, h); }
  static __device__ __forceinline__ float to_f32(uint16_t u) { return __builtin_bit_cast(float, (uint32_t)u << 16); }
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <>
struct E16<1> {  // fp16
  static __device__ __forceinline__ uint16_t from_f32(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
  static __device__ __forceinline__ float to_f32(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

struct GemmArgs {
  const uint16_t *A;     // [rows][lda] 16-bit; conv: base already advanced past the guard rows
  const uint16_t *B;     // [ntaps][N_pad][K_tap] 16-bit, K contiguous
  const float *bias;     // [N_pad]
  const uint16_t *Res;   // residual in the output's layout, or null
  void *out;
  int M, N_pad, K_tap, ntaps, lda, ldo;
  int P, R, PP;          // padded grid edge, board edge, rows per game
  int n_valid, m_valid;
  int mode;              // 0: conv -> padded grid, ReLU (+Res)   1: policy conv -> compact FC input
                         // 2: Linear -> f32 logits (+bias)
};

constexpr int GEMM_BM = 128, GEMM_BN = 128;

// byte offset of 16-B chunk j of tile row `row` (BK elements per row), XOR-swizzled so that the
// 16-lane groups of a ds_read_b128 touch 16 distinct 16-B slots of the 256-B bank row
template <int BK>
__device__ __forceinline__ int lds_off(int row, int j) {
  constexpr int CH = BK / 8;            // chunks per row
  constexpr int RPB = 256 / (BK * 2);   // tile rows per 256-B bank row
  return row * (BK * 2) + ((j ^ ((row / RPB) % CH)) << 4);
}

template <int DT, int BK>
__global__ void __launch_bounds__(256) k_gemm16(GemmArgs g) {
  constexpr int CH = BK / 8;
  constexpr int NLD = (GEMM_BM * CH) / 256;   // 16-B chunks per thread per operand per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *As = smem;                                  // [2][128*BK*2]
  unsigned char *Bs = smem + 2 * GEMM_BM * BK * 2;           // [2][128*BK*2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * GEMM_BM, n0 = blockIdx.y * GEMM_BN;
  const int ksteps = g.K_tap / BK, S = g.ntaps * ksteps;

  u32x4_t ra[NLD], rb[NLD];
#define FPC_GLOAD(S_)                                                                                   \
  {                                                                                                     \
    const int tap_ = (S_) / ksteps, k0_ = ((S_) % ksteps) * BK;                                         \
    const int off_ = g.ntaps == 9 ? (tap_ / 3 - 1) * g.P + (tap_ % 3 - 1) : 0;                          \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                   \
      const int c_ = tid + 256 * i, row_ = c_ / CH, j_ = c_ % CH;                                       \
      ra[i] = *reinterpret_cast<const u32x4_t *>(g.A + (long)(m0 + row_ + off_) * g.lda + k0_ + j_ * 8);  \
      rb[i] = *reinterpret_cast<const u32x4_t *>(g.B + ((long)tap_ * g.N_pad + n0 + row_) * g.K_tap + k0_ + j_ * 8); \
    }                                                                                                   \
  }
#define FPC_SSTORE(BUF_)                                                                                \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                   \
      const int c_ = tid + 256 * i, row_ = c_ / CH, j_ = c_ % CH;                                       \
      *reinterpret_cast<u32x4_t *>(As + (BUF_) * (GEMM_BM * BK * 2) + lds_off<BK>(row_, j_)) = ra[i];     \
      *reinterpret_cast<u32x4_t *>(Bs + (BUF_) * (GEMM_BN * BK * 2) + lds_off<BK>(row_, j_)) = rb[i];     \
    }                                                                                                   \
  }

  f32x16_t acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  FPC_GLOAD(0);
  FPC_SSTORE(0);
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    const int buf = s & 1;
    if (s + 1 < S) FPC_GLOAD(s + 1);
    const unsigned char *Ab = As + buf * (GEMM_BM * BK * 2), *Bb = Bs + buf * (GEMM_BN * BK * 2);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int j = ks * 2 + (lane >> 5);
      const u32x4_t fa0 = *reinterpret_cast<const u32x4_t *>(Ab + lds_off<BK>(wm * 64 + (lane & 31), j));
      const u32x4_t fa1 = *reinterpret_cast<const u32x4_t *>(Ab + lds_off<BK>(wm * 64 + 32 + (lane & 31), j));
      const u32x4_t fb0 = *reinterpret_cast<const u32x4_t *>(Bb + lds_off<BK>(wn * 64 + (lane & 31), j));
      const u32x4_t fb1 = *reinterpret_cast<const u32x4_t *>(Bb + lds_off<BK>(wn * 64 + 32 + (lane & 31), j));
      acc00 = E16<DT>::mfma(fa0, fb0, acc00);
      acc01 = E16<DT>::mfma(fa0, fb1, acc01);
      acc10 = E16<DT>::mfma(fa1, fb0, acc10);
      acc11 = E16<DT>::mfma(fa1, fb1, acc11);
    }
    if (s + 1 < S) FPC_SSTORE(buf ^ 1);
    __syncthreads();
  }
#undef FPC_GLOAD
#undef FPC_SSTORE

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = n0 + wn * 64 + b * 32 + (lane & 31);
      const float bias = g.bias[n];
      const f32x16_t accv = a == 0 ? (b == 0 ? acc00 : acc01) : (b == 0 ? acc10 : acc11);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = accv[r] + bias;
        if (g.mode == 2) {
          if (m < g.m_valid && n < g.n_valid) reinterpret_cast<float *>(g.out)[(long)m * g.ldo + n] = v;
        } else {
          const int game = m / g.PP, pos = m % g.PP, pi = pos / g.P, pj = pos % g.P;
          const bool interior = pi >= 1 && pi <= g.R && pj >= 1 && pj <= g.R && game < g.m_valid;
          if (interior && n < g.n_valid) {
            if (g.mode == 0) {
              if (g.Res) v += E16<DT>::to_f32(g.Res[(long)m * g.ldo + n]);
              v = v > 0.f ? v : 0.f;
              reinterpret_cast<uint16_t *>(g.out)[(long)m * g.ldo + n] = E16<DT>::from_f32(v);
            } else {
              v = v > 0.f ? v : 0.f;
              const int q = (pi - 1) * g.R + (pj - 1);
              reinterpret_cast<uint16_t *>(g.out)[(long)game * g.ldo + (long)q * g.n_valid + n] = E16<DT>::from_f32(v);
            }
          }
        }
      }
    }
  }
}

// ================================================================================================
// k_conv3x3: the 3x3 convolution proper.  One block = 256 consecutive rows of the bordered-grid
// matrix (at 14x14 exactly one game's 16x16 grid) x 128 output channels.  The block's input rows
// PLUS a halo of P+1 rows on each side are staged into LDS ONCE per 128-channel slab (74 kB) and all
// nine taps read them at row offsets (ky-1)*P+(kx-1); only the weights stream through a 3-slot LDS
// ring (16 kB per stage, register-prefetched two stages ahead).  8 waves (4 along M x 2 along N,
// 64x64 each, 2 waves per SIMD so one wave's LDS reads hide under the other's MFMAs).
// The XOR swizzle keys on (row mod 16), so a uniform row shift keeps every ds_read_b128 conflict-free.
// Epilogue: accumulators are staged as f32 through the (now free) image region, 64 columns at a
// time, then written row-major with bias + residual + ReLU fused and 16-byte stores; border rows of
// the grid are never written, so they stay zero for the next layer.
// ================================================================================================
struct ConvArgs {
  const uint16_t *X;     // input grid matrix, base already past the guard rows; row stride ldx
  const uint16_t *W;     // [9][cout_pad][cin] 16-bit
  const float *bias;     // [cout_pad]
  const uint16_t *Res;   // residual (output layout) or null
  uint16_t *out;
  int ldx, ldo, cin, cout_pad;
  int P, R, PP, n_valid, m_valid, mode;   // mode 0: grid output (ReLU, +Res); 1: compact policy layout
};

constexpr int CONV_BM = 256, CONV_BN = 128, CONV_THREADS = 512, CONV_HMAX = 17;

template <int DT, int CINC>
__global__ void __launch_bounds__(CONV_THREADS) k_conv3x3(ConvArgs g) {
  constexpr int BKS = CINC < 64 ? CINC : 64;           // K per weight stage
  constexpr int KC = CINC / BKS;                       // stages per (slab, tap)
  constexpr int CHA = CINC / 8;                        // 16-B chunks per image row
  constexpr int CHB = BKS / 8;
  constexpr int NB = (CONV_BN * CHB) / CONV_THREADS;   // weight chunks per thread per stage (1 or 2)
  constexpr int IMG_BYTES = (CONV_BM + 2 * CONV_HMAX) * CINC * 2;
  constexpr int STAGE_BYTES = CONV_BN * BKS * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *img = smem;
  unsigned char *ring = smem + (IMG_BYTES > CONV_BM * 64 * 4 ? IMG_BYTES : CONV_BM * 64 * 4);
  unsigned char *interior = ring + 3 * STAGE_BYTES;    // [256] row predicate
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * CONV_BM, n0 = blockIdx.y * CONV_BN;
  const int H = g.P + 1, rows_total = CONV_BM + 2 * H;
  const int nslab = g.cin / CINC, S = nslab * 9 * KC;

  if (tid < CONV_BM) {
    const int m = m0 + tid, game = m / g.PP, pos = m % g.PP, pi = pos / g.P, pj = pos % g.P;
    interior[tid] = (pi >= 1 && pi <= g.R && pj >= 1 && pj <= g.R && game < g.m_valid) ? 1 : 0;
  }

  u32x4_t rb0[NB], rb1[NB];
#define FPC_WLOAD(REGS, S_)                                                                          \
  {                                                                                                  \
    const int sl_ = (S_) / (9 * KC), tap_ = ((S_) / KC) % 9, kc_ = (S_) % KC;                        \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                 \
      const int c_ = tid + CONV_THREADS * i, row_ = c_ / CHB, j_ = c_ % CHB;                         \
      REGS[i] = *reinterpret_cast<const u32x4_t *>(g.W + ((long)tap_ * g.cout_pad + n0 + row_) * g.cin + \
                                                   sl_ * CINC + kc_ * BKS + j_ * 8);                 \
    }                                                                                                \
  }
#define FPC_WSTORE(REGS, SLOT_)                                                                      \
  {                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                 \
      const int c_ = tid + CONV_THREADS * i, row_ = c_ / CHB, j_ = c_ % CHB;                         \
      *reinterpret_cast<u32x4_t *>(ring + (SLOT_) * STAGE_BYTES + lds_off<BKS>(row_, j_)) = REGS[i]; \
    }                                                                                                \
  }
#define FPC_IMG_LOAD(SLAB_)                                                                          \
  for (int c_ = tid; c_ < rows_total * CHA; c_ += CONV_THREADS) {                                    \
    const int row_ = c_ / CHA, j_ = c_ % CHA;                                                        \
    *reinterpret_cast<u32x4_t *>(img + lds_off<CINC>(row_, j_)) =                                    \
        *reinterpret_cast<const u32x4_t *>(g.X + (long)(m0 - H + row_) * g.ldx + (SLAB_) * CINC + j_ * 8); \
  }

  f32x16_t acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

  FPC_WLOAD(rb0, 0);
  if (S > 1) FPC_WLOAD(rb1, 1);
  FPC_IMG_LOAD(0);
  FPC_WSTORE(rb0, 0);
  __syncthreads();

  // one pipeline step: weights of stage s are in ring slot s%3; REGS_NEXT holds stage s+1 (to be
  // written to its slot after the MFMAs), REGS_FREE receives stage s+2
#define FPC_STEP(S_, REGS_NEXT, REGS_FREE)                                                           \
  {                                                                                                  \
    const int s_ = (S_);                                                                             \
    const int tap_ = (s_ / KC) % 9, kc_ = s_ % KC;                                                   \
    if (s_ + 2 < S) FPC_WLOAD(REGS_FREE, s_ + 2);                                                    \
    const int arow_ = H + (tap_ / 3 - 1) * g.P + (tap_ % 3 - 1) + wm * 64 + (lane & 31);             \
    const unsigned char *wb_ = ring + (s_ % 3) * STAGE_BYTES;                                        \
    _Pragma("unroll") for (int ks = 0; ks < BKS / 16; ++ks) {                                        \
      const int ja_ = kc_ * CHB + ks * 2 + (lane >> 5), jb_ = ks * 2 + (lane >> 5);                  \
      const u32x4_t fa0 = *reinterpret_cast<const u32x4_t *>(img + lds_off<CINC>(arow_, ja_));       \
      const u32x4_t fa1 = *reinterpret_cast<const u32x4_t *>(img + lds_off<CINC>(arow_ + 32, ja_));  \
      const u32x4_t fb0 = *reinterpret_cast<const u32x4_t *>(wb_ + lds_off<BKS>(wn * 64 + (lane & 31), jb_));      \
      const u32x4_t fb1 = *reinterpret_cast<const u32x4_t *>(wb_ + lds_off<BKS>(wn * 64 + 32 + (lane & 31), jb_)); \
      acc00 = E16<DT>::mfma(fa0, fb0, acc00);                                                        \
      acc01 = E16<DT>::mfma(fa0, fb1, acc01);                                                        \
      acc10 = E16<DT>::mfma(fa1, fb0, acc10);                                                        \
      acc11 = E16<DT>::mfma(fa1, fb1, acc11);                                                        \
    }                                                                                                \
    if (s_ + 1 < S) {                                                                                \
      if ((s_ + 1) % (9 * KC) == 0) {   /* next stage starts a new 128-channel slab: restage image */ \
        __syncthreads();                                                                             \
        FPC_IMG_LOAD((s_ + 1) / (9 * KC));                                                           \
      }                                                                                              \
      FPC_WSTORE(REGS_NEXT, (s_ + 1) % 3);                                                           \
    }                                                                                                \
    __syncthreads();                                                                                 \
  }
  for (int s = 0; s < S; s += 2) {
    FPC_STEP(s, rb1, rb0);
    if (s + 1 < S) FPC_STEP(s + 1, rb0, rb1);
  }
#undef FPC_STEP
#undef FPC_WLOAD
#undef FPC_WSTORE
#undef FPC_IMG_LOAD

  // ---- epilogue: two column halves of 64 through an f32 [256][64] LDS stage (reuses the image)
  float *stage = reinterpret_cast<float *>(smem);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (wn == h) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const f32x16_t accv = a == 0 ? (b == 0 ? acc00 : acc01) : (b == 0 ? acc10 : acc11);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            stage[row * 64 + b * 32 + (lane & 31)] = accv[r];
          }
        }
    }
    __syncthreads();
    // 256 rows x 8 column groups of 8: 2048 items over 512 threads
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int item = tid + CONV_THREADS * i, row = item >> 3, cg = item & 7;
      const int n = n0 + h * 64 + cg * 8;
      if (interior[row] && n < g.n_valid) {
        const long m = (long)m0 + row;
        const float4 v0 = *reinterpret_cast<const float4 *>(stage + row * 64 + cg * 8);
        const float4 v1 = *reinterpret_cast<const float4 *>(stage + row * 64 + cg * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(g.bias + n);
        const float4 b1 = *reinterpret_cast<const float4 *>(g.bias + n + 4);
        float v[8] = {v0.x + b0.x, v0.y + b0.y, v0.z + b0.z, v0.w + b0.w, v1.x + b1.x, v1.y + b1.y, v1.z + b1.z, v1.w + b1.w};
        long o;
        if (g.mode == 0) {
          o = m * g.ldo + n;
          if (g.Res) {
            const u32x4_t rr = *reinterpret_cast<const u32x4_t *>(g.Res + o);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[2 * k] += E16<DT>::to_f32((uint16_t)(rr[k] & 0xffff));
              v[2 * k + 1] += E16<DT>::to_f32((uint16_t)(rr[k] >> 16));
            }
          }
        } else {
          const int game = (int)(m / g.PP), pos = (int)(m % g.PP);
          const int q = (pos / g.P - 1) * g.R + (pos % g.P - 1);
          o = (long)game * g.ldo + (long)q * g.n_valid + n;
        }
        u32x4_t pk;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float lo = v[2 * k] > 0.f ? v[2 * k] : 0.f, hi = v[2 * k + 1] > 0.f ? v[2 * k + 1] : 0.f;
          pk[k] = (uint32_t)E16<DT>::from_f32(lo) | ((uint32_t)E16<DT>::from_f32(hi) << 16);
        }
        *reinterpret_cast<u32x4_t *>(g.out + o) = pk;
      }
    }
    __syncthreads();
  }
}

// value head tail: Flatten + Linear(24*R*R -> 1) + Tanh (net.py:33-34) on the value conv output
template <int DT>
__global__ void __launch_bounds__(64) k_value_tail(const uint16_t *Y, const float *w, float bias, int P, int R, int PP,
                                                   int n, float *out) {
  const int g = blockIdx.x;
  if (g >= n) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int idx = lane; idx < R * R * 32; idx += 64) {
    const int q = idx >> 5, ch = idx & 31;
    const long m = (long)g * PP + (q / R + 1) * P + (q % R + 1);
    acc += E16<DT>::to_f32(Y[m * 32 + ch]) * w[idx];
  }
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) out[g] = tanhf(acc + bias);
}

// external-evaluator entry: f32 NCHW [n,24,R,R] -> 16-bit NHWC(32) on the padded grid
template <int DT>
__global__ void __launch_bounds__(64) k_nchw_to_grid(const float *x, int n, int R, int P, uint16_t *out) {
  const int g = blockIdx.x;
  if (g >= n) return;
  const int lane = threadIdx.x & 63, RR = R * R;
  for (int q = lane; q < RR; q += 64) {
    uint16_t *row = out + ((long)g * P * P + (q / R + 1) * P + (q % R + 1)) * 32;
    for (int c = 0; c < 32; ++c) row[c] = c < 24 ? E16<DT>::from_f32(x[((long)g * 24 + c) * RR + q]) : (uint16_t)0;
  }
}

// ------------------------------------------------------------------------------------------------
// weight blob (written by alphazero-4-player-chess_amd/weights.py):
//   header  : char magic[4]="FPCW"; int32 version=1, R, F, nblocks, dtype, A_ch, Np, Kp; pad to 64 B
//   sections, each 64-B aligned, in this order:
//     stem   w16[9][Fp][32]     b f32[Fp]          (Fp = F rounded up to 128)
//     block i: c1 w16[9][Fp][F] b f32[Fp] ; c2 w16[9][Fp][F] b f32[Fp]
//     policy conv w16[9][128][F] b f32[128]
//     value  conv w16[9][128][F] b f32[128]  (both heads: Cout zero-padded to the 128-wide tile)
//     policy fc   w16[Np][Kp]    b f32[Np]
//     value  fc   w f32[R*R][32] b f32[1]
// ------------------------------------------------------------------------------------------------
struct BlobHeader {
  char magic[4];
  int32_t version, R, F, nblocks, dtype, A_ch, Np, Kp;
  int32_t pad[7];
};
static_assert(sizeof(BlobHeader) == 64, "header is 64 bytes");

struct ConvW {
  uint16_t *w = nullptr;
  float *b = nullptr;
  int cin = 0, cout_pad = 0;
};

struct NN {
  bool loaded = false;
  DevCfg dc{};
  int Gmax = 0, dtype = 0, F = 0, nblocks = 0, P = 0, PP = 0, Np = 0, Kp = 0, Mrows = 0, guard = 64, Gpad = 0;
  hipStream_t stream = nullptr;
  std::vector<void *> allocs;
  uint16_t *in16 = nullptr;      // [guard + Mrows + guard][32]
  uint16_t *act[3] = {nullptr, nullptr, nullptr};   // [guard + Mrows + guard][F]
  uint16_t *yv = nullptr;        // value conv out [guard + Mrows + guard][32] (ld 32)
  uint16_t *xfc = nullptr;       // [Gpad][Kp]
  float *d_logits = nullptr;     // [Gmax][A]
  float *d_value = nullptr;      // [Gmax]
  ConvW stem, pconv, vconv;
  std::vector<ConvW> c1, c2;
  uint16_t *fcw = nullptr;
  float *fcb = nullptr, *vw = nullptr;
  float vb = 0.f;
  bool attr_set[2] = {false, false};

  template <class T>
  int dmalloc(T **p, size_t count, std::string *err) {
    void *q = nullptr;
    if (hipMalloc(&q, count * sizeof(T)) != hipSuccess) { *err = "hipMalloc failed in NN (" + std::to_string(count * sizeof(T)) + " bytes)"; return FPC_ENOMEM; }
    (void)hipMemset(q, 0, count * sizeof(T));
    allocs.push_back(q);
    *p = (T *)q;
    return 0;
  }

  int init(const DevCfg &c, int max_games, int nn_dtype, hipStream_t s, std::string *) {
    dc = c; Gmax = max_games; dtype = nn_dtype ? 1 : 0; stream = s;
    P = c.R + 2; PP = P * P;
    Mrows = ((max_games * PP + CONV_BM - 1) / CONV_BM) * CONV_BM;
    Gpad = ((max_games + GEMM_BM - 1) / GEMM_BM) * GEMM_BM;
    return 0;
  }
  void destroy() {
    for (void *p : allocs) (void)hipFree(p);
    allocs.clear();
    loaded = false;
  }
  uint16_t *input16() { return in16 + (size_t)guard * 32; }
  uint16_t one16() const { return dtype ? 0x3C00 : 0x3F80; }
  float *logits() { return d_logits; }
  float *value() { return d_value; }

  int load(const void *blob, uint64_t nbytes, std::string *err) {
    if (nbytes < sizeof(BlobHeader)) { *err = "weight blob too small"; return FPC_EWEIGHTS; }
    BlobHeader h;
    memcpy(&h, blob, sizeof(h));
    if (memcmp(h.magic, "FPCW", 4) || h.version != 1) { *err = "bad weight blob magic/version"; return FPC_EWEIGHTS; }
    if (h.R != dc.R || h.A_ch != dc.A_ch) { *err = "weight blob is for a different board size"; return FPC_EWEIGHTS; }
    if (h.dtype != dtype) { *err = "weight blob dtype differs from engine nn_dtype"; return FPC_EWEIGHTS; }
    if (h.F % 64 || h.F < 64 || h.F > 512 || h.nblocks < 0 || h.Np % GEMM_BN || h.Kp % 64 || h.Np < dc.A || h.Kp < dc.A) {
      *err = "unsupported network shape in weight blob (hidden must be a multiple of 64)";
      return FPC_EWEIGHTS;
    }
    destroy();
    F = h.F; nblocks = h.nblocks; Np = h.Np; Kp = h.Kp;
    const unsigned char *base = (const unsigned char *)blob;
    uint64_t off = sizeof(BlobHeader);
    int rc = 0;
    auto take = [&](void **dev, uint64_t bytes) -> int {
      off = (off + 63) & ~63ull;
      if (off + bytes > nbytes) { *err = "weight blob truncated"; return FPC_EWEIGHTS; }
      unsigned char *d = nullptr;
      if ((rc = dmalloc(&d, bytes, err))) return rc;
      if (hipMemcpy(d, base + off, bytes, hipMemcpyHostToDevice) != hipSuccess) { *err = "weight upload failed"; return FPC_ENODEVICE; }
      off += bytes;
      *dev = d;
      return 0;
    };
    auto conv = [&](ConvW &cw, int cin, int cout_pad) -> int {
      cw.cin = cin; cw.cout_pad = cout_pad;
      if ((rc = take((void **)&cw.w, (uint64_t)9 * cout_pad * cin * 2))) return rc;
      return take((void **)&cw.b, (uint64_t)cout_pad * 4);
    };
    const int Fp = (F + GEMM_BN - 1) / GEMM_BN * GEMM_BN;   // Cout rows are zero-padded to the 128-wide tile
    if ((rc = conv(stem, 32, Fp))) return rc;
    c1.assign(nblocks, ConvW()); c2.assign(nblocks, ConvW());
    for (int i = 0; i < nblocks; ++i) { if ((rc = conv(c1[i], F, Fp)) || (rc = conv(c2[i], F, Fp))) return rc; }
    if ((rc = conv(pconv, F, 128))) return rc;
    if ((rc = conv(vconv, F, 128))) return rc;
    if ((rc = take((void **)&fcw, (uint64_t)Np * Kp * 2)) || (rc = take((void **)&fcb, (uint64_t)Np * 4))) return rc;
    if ((rc = take((void **)&vw, (uint64_t)dc.RR * 32 * 4))) return rc;
    off = (off + 63) & ~63ull;
    if (off + 4 > nbytes) { *err = "weight blob truncated"; return FPC_EWEIGHTS; }
    memcpy(&vb, base + off, 4);
    // activations
    const size_t rows = (size_t)Mrows + 2 * guard;
    if ((rc = dmalloc(&in16, rows * 32, err))) return rc;
    for (auto &a : act) if ((rc = dmalloc(&a, rows * F, err))) return rc;
    if ((rc = dmalloc(&yv, rows * 32, err))) return rc;
    if ((rc = dmalloc(&xfc, (size_t)Gpad * Kp, err))) return rc;
    if ((rc = dmalloc(&d_logits, (size_t)Gmax * dc.A, err))) return rc;
    if ((rc = dmalloc(&d_value, (size_t)Gmax, err))) return rc;
    loaded = true;
    return 0;
  }

  template <int DT>
  int launch_gemm(const GemmArgs &g, int bk, std::string *err) {
    dim3 grid(g.M / GEMM_BM, g.N_pad / GEMM_BN), block(256);
    const size_t lds = (size_t)2 * (GEMM_BM + GEMM_BN) * bk * 2;
    if (!attr_set[DT]) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm16<DT, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (GEMM_BM + GEMM_BN) * 64 * 2);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm16<DT, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (GEMM_BM + GEMM_BN) * 32 * 2);
      attr_set[DT] = true;
    }
    if (bk == 64) hipLaunchKernelGGL((k_gemm16<DT, 64>), grid, block, lds, stream, g);
    else hipLaunchKernelGGL((k_gemm16<DT, 32>), grid, block, lds, stream, g);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
      *err = std::string("k_gemm16 launch failed: ") + hipGetErrorString(le) + " (grid " + std::to_string(grid.x) + "x" + std::to_string(grid.y) + ", lds " + std::to_string(lds) + ")";
      return FPC_ENODEVICE;
    }
    return 0;
  }

  template <int DT, int CINC>
  int launch_conv_c(const ConvArgs &g, int M, std::string *err) {
    constexpr int BKS = CINC < 64 ? CINC : 64;
    constexpr int IMG = (CONV_BM + 2 * CONV_HMAX) * CINC * 2;
    constexpr int lds = (IMG > CONV_BM * 64 * 4 ? IMG : CONV_BM * 64 * 4) + 3 * CONV_BN * BKS * 2 + CONV_BM;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3x3<DT, CINC>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr = true;
    }
    dim3 grid(M / CONV_BM, g.cout_pad / CONV_BN), block(CONV_THREADS);
    hipLaunchKernelGGL((k_conv3x3<DT, CINC>), grid, block, lds, stream, g);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) { *err = std::string("k_conv3x3 launch failed: ") + hipGetErrorString(le); return FPC_ENODEVICE; }
    return 0;
  }
  template <int DT>
  int launch_conv(const ConvArgs &g, int M, std::string *err) {
    if (g.P + 1 > CONV_HMAX) { *err = "board too large for the conv halo"; return FPC_EINVAL; }
    if (g.cin == 32) return launch_conv_c<DT, 32>(g, M, err);
    if (g.cin == 64) return launch_conv_c<DT, 64>(g, M, err);
    if (g.cin % 128 == 0) return launch_conv_c<DT, 128>(g, M, err);
    *err = "unsupported conv input width";
    return FPC_EINVAL;
  }

  template <int DT>
  int forward_t(int n, float *logits_out, float *value_out, std::string *err) {
    int rc;
    const int M = ((n * PP + CONV_BM - 1) / CONV_BM) * CONV_BM;
    auto conv = [&](const ConvW &cw, const uint16_t *in, int in_ld, const uint16_t *res, uint16_t *out, int out_ld, int n_valid, int mode) -> int {
      ConvArgs g{};
      g.X = in + (size_t)guard * in_ld; g.W = cw.w; g.bias = cw.b;
      g.Res = res ? res + (size_t)guard * out_ld : nullptr;
      g.out = mode == 1 ? out : out + (size_t)guard * out_ld;
      g.ldx = in_ld; g.ldo = out_ld; g.cin = cw.cin; g.cout_pad = cw.cout_pad;
      g.P = P; g.R = dc.R; g.PP = PP; g.n_valid = n_valid; g.m_valid = n; g.mode = mode;
      return launch_conv<DT>(g, M, err);
    };
    if ((rc = conv(stem, in16, 32, nullptr, act[0], F, F, 0))) return rc;
    int cur = 0;
    for (int i = 0; i < nblocks; ++i) {
      const int t1 = (cur + 1) % 3, t2 = (cur + 2) % 3;
      if ((rc = conv(c1[i], act[cur], F, nullptr, act[t1], F, F, 0))) return rc;
      if ((rc = conv(c2[i], act[t1], F, act[cur], act[t2], F, F, 0))) return rc;
      cur = t2;
    }
    if ((rc = conv(pconv, act[cur], F, nullptr, xfc, Kp, dc.A_ch, 1))) return rc;
    if ((rc = conv(vconv, act[cur], F, nullptr, yv, 32, 24, 0))) return rc;
    {
      GemmArgs g{};
      g.A = xfc; g.B = fcw; g.bias = fcb; g.Res = nullptr; g.out = logits_out;
      g.M = ((n + GEMM_BM - 1) / GEMM_BM) * GEMM_BM; g.N_pad = Np; g.K_tap = Kp; g.ntaps = 1; g.lda = Kp; g.ldo = dc.A;
      g.P = P; g.R = dc.R; g.PP = PP; g.n_valid = dc.A; g.m_valid = n; g.mode = 2;
      if ((rc = launch_gemm<DT>(g, 64, err))) return rc;
    }
    hipLaunchKernelGGL((k_value_tail<DT>), dim3(n), dim3(64), 0, stream, (const uint16_t *)(yv + (size_t)guard * 32),
                       (const float *)vw, vb, P, dc.R, PP, n, value_out);
    if (hipGetLastError() != hipSuccess) { *err = "k_value_tail launch failed"; return FPC_ENODEVICE; }
    return 0;
  }

  int forward(int n, std::string *err) {
    return dtype ? forward_t<1>(n, d_logits, d_value, err) : forward_t<0>(n, d_logits, d_value, err);
  }
  int forward_external(const float *enc, int n, float *logits_out, float *value_out, std::string *err) {
    if (dtype) hipLaunchKernelGGL((k_nchw_to_grid<1>), dim3(n), dim3(64), 0, stream, enc, n, dc.R, P, input16());
    else hipLaunchKernelGGL((k_nchw_to_grid<0>), dim3(n), dim3(64), 0, stream, enc, n, dc.R, P, input16());
    int rc = dtype ? forward_t<1>(n, logits_out, value_out, err) : forward_t<0>(n, logits_out, value_out, err);
    if (rc) return rc;
    if (hipStreamSynchronize(stream) != hipSuccess) { *err = "NN forward failed"; return FPC_ENODEVICE; }
    return 0;
  }
};

}  // namespace fpc
