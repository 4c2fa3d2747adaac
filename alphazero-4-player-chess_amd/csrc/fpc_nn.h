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
// Kernels: k_tower (fpc_tower.h: whole residual tower + head convs, activations LDS-resident, hidden = 128),
// k_conv3x3 (one conv per launch, any hidden width), k_fc (fpc_fc.h: weight-streaming policy Linear).
// All use v_mfma_f32_32x32x16_{bf16,f16} and a 16-byte-chunk XOR swizzle in LDS that makes every
// ds_read_b128 fragment read conflict-free.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fpc_tree_kernels.h"
#include "fpc_tower.h"
#include "fpc_towerw.h"

namespace fpc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int DT>
struct E16;
template <>
struct E16<0> {  // bf16
  static __device__ __forceinline__ uint16_t from_f32(float f) { __bf16 h = (__bf16)f; return __builtin_bit_cast(uint16_t, h); }
  static __device__ __forceinline__ float to_f32(uint16_t u) { return __builtin_bit_cast(float, (uint32_t)u << 16); }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return (uint32_t)from_f32(lo) | ((uint32_t)from_f32(hi) << 16); }
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <>
struct E16<1> {  // fp16
  static __device__ __forceinline__ uint16_t from_f32(float f) { _Float16 h = (_Float16)f; return __builtin_bit_cast(uint16_t, h); }
  static __device__ __forceinline__ float to_f32(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return (uint32_t)from_f32(lo) | ((uint32_t)from_f32(hi) << 16); }
  static __device__ __forceinline__ f32x16_t mfma(u32x4_t a, u32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

constexpr int GEMM_BM = 128, GEMM_BN = 128;
constexpr int FC_SPLITK = 4;   // K-splits of a long policy-Linear block; short blocks use 2x (see k_fc)

// byte offset of 16-B chunk j of tile row `row` (BK elements per row), XOR-swizzled so that the
// 16-lane groups of a ds_read_b128 touch 16 distinct 16-B slots of the 256-B bank row
template <int BK>
__device__ __forceinline__ int lds_off(int row, int j) {
  constexpr int CH = BK / 8;            // chunks per row
  constexpr int RPB = 256 / (BK * 2);   // tile rows per 256-B bank row
  return row * (BK * 2) + ((j ^ ((row / RPB) % CH)) << 4);
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

}  // namespace fpc
#include "fpc_fc.h"   // k_fc + k_fc_reduce: the weight-streaming policy Linear
namespace fpc {

// ================================================================================================
// LEGAL-ONLY policy head (opt-in, fpc_set_policy_mode(FPC_POLICY_LEGAL)).
// The search consumes the policy Linear's output only at the leaf's legal moves (mask-multiply +
// renormalise, mcts.py:74-76; the softmax denominator cancels), ~40 of 23 520 columns per game.
// k_policy_gemv evaluates exactly those: for every (game, legal move) pair one wave streams the
// move's weight row (Kp 16-bit values, contiguous in a row-major copy of the weights made once by
// k_fc_unfrag) against the game's activation row staged in LDS -- 0.5 GB of weight rows per step at
// 14x14 instead of the whole 1.1 GB matrix and its 283 GFLOP.  Pairs are dealt to the blocks in equal
// shares (a game with 90 legal moves does not hold one CU three times longer than one with 30).
// ================================================================================================
template <int DT>
__global__ void __launch_bounds__(256) k_fc_unfrag(const uint16_t *Wf, uint16_t *W2, int Np, int Kp) {
  // one thread per 16-byte chunk of the row-major matrix: (n, k8) <- fragment order [k-step of 32][tile of 16 columns][lane = 16 q + c][8]
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  const long chunks = (long)Np * (Kp / 8);
  if (c >= chunks) return;
  const int n = (int)(c / (Kp / 8)), k8 = (int)(c % (Kp / 8));
  const long src = (((long)(k8 >> 2) * (Np / 16) + (n >> 4)) * 64 + ((k8 & 3) * 16 + (n & 15))) * 8;
  *reinterpret_cast<u32x4_t *>(W2 + (long)n * Kp + (long)k8 * 8) = *reinterpret_cast<const u32x4_t *>(Wf + src);
}

template <int DT>
struct Dot16;
template <>
struct Dot16<0> {
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    typedef __attribute__((ext_vector_type(2))) __bf16 v2;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
  }
};
template <>
struct Dot16<1> {
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    typedef __attribute__((ext_vector_type(2))) _Float16 v2;
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
  }
};

constexpr int GEMV_THREADS = 256, GEMV_MAXG = 2048;

template <int DT>
__global__ void __launch_bounds__(GEMV_THREADS) k_policy_gemv(DevCfg c, Tree t, int G, const uint16_t *X, const uint16_t *W2,
                                                              const float *bias, int Kp, float *ll) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xs[];     // Kp * 2 bytes: one game's activation row
  __shared__ int cum[GEMV_MAXG + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // pair offsets: cum[g] = number of (live game, legal move) pairs before game g
  if (wave == 0) {
    int run = 0;
    for (int base = 0; base < G; base += 64) {
      const int g = base + lane;
      int v = (g < G && t.leaf_node[g] >= 0) ? t.nlegal[g] : 0;
      int inc = v;
      for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (lane >= off) inc += o; }
      if (g < G) cum[g] = run + inc - v;
      run += __shfl(inc, 63);
    }
    if (lane == 0) cum[G] = run;
  }
  __syncthreads();
  const int total = cum[G];
  const int p0 = (int)((long)total * blockIdx.x / gridDim.x), p1 = (int)((long)total * (blockIdx.x + 1) / gridDim.x);
  if (p0 >= p1) return;
  const int turn_batch = first_leaf_turn(t.leaf_node, t.leaf_turn, G);
  // first game of the range: the last g with cum[g] <= p0 among games that own pairs
  int g = 0;
  { int lo = 0, hi = G; while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (cum[mid] <= p0) lo = mid; else hi = mid - 1; } g = lo; }
  const int chunks = Kp / 8;                       // 16-byte chunks per row; a multiple of 64 (Kp % 512 == 0)
  for (int p = p0; p < p1;) {
    while (cum[g + 1] <= p) ++g;                   // skip games without pairs
    const int pend = cum[g + 1] < p1 ? cum[g + 1] : p1;
    __syncthreads();                               // the previous game's row is no longer being read
    for (int q = tid; q < chunks; q += GEMV_THREADS)
      reinterpret_cast<u32x4_t *>(xs)[q] = *reinterpret_cast<const u32x4_t *>(X + (long)g * Kp + (long)q * 8);
    __syncthreads();
    const uint16_t *legal = t.legal + (size_t)g * FPC_MAX_MOVES;
    for (int pp = p + wave; pp < pend; pp += GEMV_THREADS / 64) {
      const int j = pp - cum[g];
      const int fl = legal[j];
      const int plane = fl / c.RR, pos = fl % c.RR;
      const int turn0 = (c.rules & FPC_RULES_ROTATION) ? t.leaf_turn[g] : turn_batch;
      const int nsrc = plane * c.RR + rot90_src(c.R, -turn0, pos / c.R, pos % c.R);   // ParseActionspace's inverse rotation
      const u32x4_t *wrow = reinterpret_cast<const u32x4_t *>(W2 + (long)nsrc * Kp) + lane;
      const u32x4_t *xrow = reinterpret_cast<const u32x4_t *>(xs) + lane;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int q0 = 0; q0 < chunks; q0 += 64 * 8) {          // 8 row chunks per lane in flight
        u32x4_t w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = q0 + 64 * u < chunks ? wrow[q0 + 64 * u] : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (q0 + 64 * u < chunks) {
            const u32x4_t x = xrow[q0 + 64 * u];
            a0 = Dot16<DT>::dot2(w[u][0], x[0], a0); a1 = Dot16<DT>::dot2(w[u][1], x[1], a1);
            a2 = Dot16<DT>::dot2(w[u][2], x[2], a2); a3 = Dot16<DT>::dot2(w[u][3], x[3], a3);
          }
        }
      }
      float a = (a0 + a1) + (a2 + a3);
      for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
      if (lane == 0) ll[(size_t)g * FPC_MAX_MOVES + j] = bias[nsrc] + a;
    }
    p = pend;
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
//   header  : char magic[4]="FPCW"; int32 version (3; version 2 = the retired 32x32x16 fragment order of rounds 1-4: refused), R, F, nblocks, dtype, A_ch, Np, Kp, fc_layout; pad to 64 B
//   sections, each 64-B aligned, in this order:
//     stem   w16[9][Fp][32]     b f32[Fp]          (Fp = F rounded up to 128)
//     block i: c1 w16[9][Fp][F] b f32[Fp] ; c2 w16[9][Fp][F] b f32[Fp]
//     policy conv w16[9][128][F] b f32[128]
//     value  conv w16[9][128][F] b f32[128]  (both heads: Cout zero-padded to the 128-wide tile)
//     policy fc   w16 in MFMA fragment order [Kp/32][Np/16][64 lanes][8] (lane = 16*(k/8%4) + n%16; Np = 256 * groups for fc_layout 1, 384 * groups for 2)   b f32[Np]
//     value  fc   w f32[R*R][32] b f32[1]
// ------------------------------------------------------------------------------------------------
struct BlobHeader {
  char magic[4];
  int32_t version, R, F, nblocks, dtype, A_ch, Np, Kp;
  int32_t fc_layout;     // policy Linear: 1 = 16x16x32 fragments over 256-column groups (k_fc16), 2 = over 384-column groups (k_fcw); 0 (32x32x16, k_fc) retired in round 5
  int32_t pad[6];
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
  float *d_stats = nullptr;      // [Gmax][SM_MAXCH][SM_REC] softmax chunk statistics of d_logits' rows (k_fc_reduce)
  float *d_value = nullptr;      // [Gmax]
  ConvW stem, pconv, vconv;
  std::vector<ConvW> c1, c2;
  uint16_t *fcw = nullptr;
  float *fcb = nullptr, *vw = nullptr;
  float *fc_part = nullptr;      // [slabs][Gpad][256 | 384] partial sums of the policy Linear (sized for 2*FC_SPLITK slabs per column)
  int fc_G1 = 0, fc_s1 = FC_SPLITK, fc_s2 = FC_SPLITK;   // work decomposition chosen at load time (plan_fc)
  int fc_layout = 1;                 // the loaded blob's policy-Linear layout: 1 -> k_fc16 (256 x 256 block tiles), 2 -> k_fcw (256 x 384)
  int fc_gw = 256;                   // columns per column group (block tile width): 256, or FCW_COLS for layout 2
  unsigned char *towerW = nullptr;   // k_tower's weight stream: [(2*nblocks + 2) * 9 taps][32 KiB] in LDS-image order (F == 128)
  unsigned char *stemW = nullptr;    // [9 taps][8 KiB], same order
  float *towerB = nullptr;           // [2*nblocks + 2][256] (128 used)
  bool use_tower = false;            // hidden == 128: k_tower
  int tower_waves = 8;               // developer knob FPC_TOWER_WAVES (A/B): 4 = one wave per SIMD
  bool tower_compact = true;         // 14x14: k_towerc (k_tower's skeleton on the compact image, 13 row tiles); developer knob FPC_TOWER_COMPACT=0: k_tower
  int towerw_rows = 1;               // developer knob FPC_TOWERW_ROWS=2: hidden 256 on two wave rows x four (A/B)
  bool use_towerw = false;           // k_towerw runs the tower (hidden 256; hidden 128 on every board but 14x14)
  void (*mark_fn)(void *, int) = nullptr;   // stage-timing hook of the engine (tag 2 = policy Linear starts)
  void *mark_ctx = nullptr;
  float vb = 0.f;
  // dynamic-LDS opt-ins (hipFuncSetAttribute) already made for this engine's device
  bool attr_conv[2][3] = {{false, false, false}, {false, false, false}}, attr_tower[2] = {false, false}, attr_towerw[2] = {false, false}, attr_fc[2] = {false, false}, attr_gemv = false;

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
    Gpad = ((max_games + 255) / 256) * 256;
    return 0;
  }
  void destroy() {
    for (void *p : allocs) (void)hipFree(p);
    allocs.clear();
    loaded = false;
    fcw2 = nullptr; d_ll = nullptr;
  }
  uint16_t *input16() { return in16 + (size_t)guard * 32; }
  // the search's leaf positions as the next forward's input (tower path only; cleared by the forward)
  const fpc_board *in_boards = nullptr;
  const int *in_leaf_slot = nullptr, *in_leaf_turn = nullptr;
  int in_board_stride = 0;
  bool takes_boards() const { return use_tower || use_towerw; }
  void set_board_input(const fpc_board *b, int stride, const int *slot, const int *turn) { in_boards = b; in_board_stride = stride; in_leaf_slot = slot; in_leaf_turn = turn; }
  uint16_t one16() const { return dtype ? 0x3C00 : 0x3F80; }
  float *logits() { return d_logits; }
  float *stats() { return d_stats; }
  float *value() { return d_value; }
  uint16_t *fcw2 = nullptr;      // row-major [Np][Kp] copy of the policy weights (legal-only head), made on first use
  float *d_ll = nullptr;         // [Gmax][FPC_MAX_MOVES] logits of the leaves' legal moves
  float *legal_logits() { return d_ll; }

  // Policy-Linear work decomposition (see k_fc): how many column groups get FC_SPLITK long blocks,
  // the rest getting twice as many half-length ones, so that the last round of blocks is full.
  // Blocks are dispatched in id order, one per CU: simulate that and keep the shortest makespan.
  // Layout 2 (k_fcw): every column group of 384 gets the same number of K-splits, chosen so that all blocks run in ONE
  // round (groups x splits <= CUs) with whole stages of 64 per block; plan_fcw() says whether such a split exists.
  static int plan_fcw(int Np, int Kp, int cus) {
    const int groups = Np / FCW_COLS, stages = Kp / 64;
    int best = 0;
    for (int sk = 1; sk <= 2 * FC_SPLITK && groups * sk <= cus; ++sk)
      if (stages % sk == 0 && stages / sk >= 4) best = sk;
    return best;
  }
  void plan_fc() {
    if (fc_layout == 2) { fc_G1 = Np / FCW_COLS; fc_s2 = fc_s1; return; }     // fc_s1 set by load()
    fc_s1 = FC_SPLITK;
    const int groups = Np / 256, ks = Kp / 16, mt = (Gmax + 255) / 256;
    fc_s2 = (ks % (2 * FC_SPLITK * 8) == 0 && ks / (2 * FC_SPLITK) / 4 >= 4) ? 2 * FC_SPLITK : FC_SPLITK;
    fc_G1 = groups;
    if (fc_s2 == FC_SPLITK) return;
    int cus = 256;
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    long best = -1;
    std::vector<long> fin(cus);
    for (int g1 = groups; g1 >= 0; --g1) {
      std::fill(fin.begin(), fin.end(), 0L);
      for (int y = 0; y < mt; ++y)
        for (int id = 0, nb = g1 * FC_SPLITK + (groups - g1) * fc_s2; id < nb; ++id) {
          auto it = std::min_element(fin.begin(), fin.end());
          *it += id < g1 * FC_SPLITK ? 2 : 1;
        }
      const long span = *std::max_element(fin.begin(), fin.end());
      if (best < 0 || span < best) { best = span; fc_G1 = g1; }
    }
  }

  int load(const void *blob, uint64_t nbytes, std::string *err) {
    if (nbytes < sizeof(BlobHeader)) { *err = "weight blob too small"; return FPC_EWEIGHTS; }
    BlobHeader h;
    memcpy(&h, blob, sizeof(h));
    if (memcmp(h.magic, "FPCW", 4) || h.version != 3) { *err = "bad weight blob magic/version (this engine reads version 3; version 2 carried the retired 32x32x16 policy-Linear order: export again)"; return FPC_EWEIGHTS; }
    if (h.R != dc.R || h.A_ch != dc.A_ch) { *err = "weight blob is for a different board size"; return FPC_EWEIGHTS; }
    if (h.dtype != dtype) { *err = "weight blob dtype differs from engine nn_dtype"; return FPC_EWEIGHTS; }
    if (h.fc_layout < 1 || h.fc_layout > 2) { *err = "unknown policy-Linear weight layout in weight blob (1 = k_fc16, 2 = k_fcw)"; return FPC_EWEIGHTS; }
    if (h.F % 64 || h.F < 64 || h.F > 512 || h.nblocks < 0 || h.Np % (h.fc_layout == 2 ? FCW_COLS : 256) || h.Kp % 512 || h.Kp < 1024 || h.Np < dc.A || h.Kp < dc.A) {
      *err = "unsupported network shape in weight blob (hidden must be a multiple of 64, Np of 256 -- 384 for fc_layout 2 --, Kp of 512)";
      return FPC_EWEIGHTS;
    }
    int fcw_split = 0;
    if (h.fc_layout == 2) {
      int cus = 256, dev = 0;
      hipDeviceProp_t prop;
      (void)hipGetDevice(&dev);
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
      fcw_split = plan_fcw(h.Np, h.Kp, cus);
      if (!fcw_split) { *err = "weight blob fc_layout 2 (k_fcw): no one-round K-split for this shape on this device; export with fc_layout 1"; return FPC_EWEIGHTS; }
    }
    destroy();
    F = h.F; nblocks = h.nblocks; Np = h.Np; Kp = h.Kp;
    fc_layout = h.fc_layout;
    fc_gw = fc_layout == 2 ? FCW_COLS : 256;
    if (fc_layout == 2) fc_s1 = fcw_split;
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
    if ((rc = dmalloc(&fc_part, (size_t)2 * FC_SPLITK * Gpad * Np, err))) return rc;
    plan_fc();
    if ((rc = dmalloc(&d_logits, (size_t)Gmax * dc.A, err))) return rc;
    if ((rc = dmalloc(&d_stats, (size_t)Gmax * SM_MAXCH * SM_REC, err))) return rc;
    if ((rc = dmalloc(&d_value, (size_t)Gmax, err))) return rc;
    use_tower = use_towerw = false;
    // Developer knobs (same-box A/Bs, bit-identity tests): consulted ONLY when FPC_DEV_KNOBS=1 is in the environment, so a
    // product run never changes kernels because of a stray variable.
    const bool dev = getenv("FPC_DEV_KNOBS") && atoi(getenv("FPC_DEV_KNOBS")) != 0;
    auto knob = [dev](const char *name, int dflt) { const char *v = dev ? getenv(name) : nullptr; return v ? atoi(v) : dflt; };
    const bool no_tower = knob("FPC_NO_TOWER", 0) != 0;
    tower_waves = knob("FPC_TOWER_WAVES", 8);
    tower_compact = knob("FPC_TOWER_COMPACT", 1) != 0;
    // Which megakernel runs the tower (one launch, activations LDS-resident; everything else: k_conv3x3 per layer):
    //   hidden 256: k_towerw (two waves per SIMD, weights L2 -> registers; any board size).
    //   hidden 128: 14x14: k_tower (LDS-DMA weight ring, loader / staggered wave roles; its grid rows ARE the 16-position row
    //               tiles there: 0.257 ms per 256 leaves against k_towerw's 0.274); every other board size: k_towerw, whose
    //               compact image computes ceil(R^2 / 16) row tiles where k_tower's bordered grid needs 6 / 10 / 14
    //               (8x8: 0.122 against 0.147 ms, 10x10: 0.171 / 0.208, 13x13: 0.242 / 0.285).  Developer knob
    //               FPC_TOWERW=0 / 1 forces k_tower / k_towerw.
    if (F == 256 && !no_tower) {
      use_towerw = true;
      towerw_rows = knob("FPC_TOWERW_ROWS", 1);
    } else if (F == 128 && !no_tower) {
      const int kw = knob("FPC_TOWERW", -1);          // -1: by board size
      use_towerw = kw < 0 ? dc.R != 14 : kw != 0;
      use_tower = !use_towerw;
    }
    const int layers = 2 * nblocks + 2;
    if (use_towerw) {
      // weights in MFMA fragment order [layer][tap][k-step of 32][cout tile of 16][lane][8] (fpc_towerw.h); one slab =
      // one k-step of one tap = F * 64 bytes; TWW_PAD_SLABS slabs of padding behind the last layer (the prefetch runs up
      // to three k-steps ahead without a branch); head convolutions zero-padded to F output channels
      const int slab = tww_slab(F), ksn = F / 32, tiles = F / 16;
      if ((rc = dmalloc(&towerW, ((size_t)layers * 9 * ksn + TWW_PAD_SLABS) * slab, err)) || (rc = dmalloc(&stemW, (size_t)9 * slab, err)) ||
          (rc = dmalloc(&towerB, (size_t)layers * 256, err))) return rc;
      auto prep = [&](const ConvW &cw, int layer) {
        hipLaunchKernelGGL(k_towerw_prep, dim3((9 * ksn * tiles * 64 + 255) / 256), dim3(256), 0, stream, (const uint16_t *)cw.w,
                           towerW + (size_t)layer * 9 * ksn * slab, 9, cw.cout_pad, F, tiles);
        (void)hipMemcpyAsync(towerB + (size_t)layer * 256, cw.b, (size_t)std::min(cw.cout_pad, F) * 4, hipMemcpyDeviceToDevice, stream);
      };
      for (int i = 0; i < nblocks; ++i) { prep(c1[i], 2 * i); prep(c2[i], 2 * i + 1); }
      prep(vconv, 2 * nblocks);
      prep(pconv, 2 * nblocks + 1);
      hipLaunchKernelGGL(k_towerw_prep, dim3((9 * tiles * 64 + 255) / 256), dim3(256), 0, stream, (const uint16_t *)stem.w, stemW, 9, stem.cout_pad, 32, tiles);
      if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) { *err = "k_towerw_prep failed"; return FPC_ENODEVICE; }
    } else if (use_tower) {
      // k_tower streams every tap as one 32 KiB block already laid out as its LDS image (fpc_tower.h);
      // layers: c1[0], c2[0], ..., then the value conv and the policy conv
      if ((rc = dmalloc(&towerW, (size_t)(layers * 9 + 3) * TW_TAP, err)) || (rc = dmalloc(&stemW, (size_t)9 * TW_STEM_TAP, err)) ||
          (rc = dmalloc(&towerB, (size_t)layers * 256, err))) return rc;
      auto prep = [&](const ConvW &cw, int layer) {
        hipLaunchKernelGGL(k_tower_prep, dim3((9 * 128 * 16 + 255) / 256), dim3(256), 0, stream, (const uint16_t *)cw.w,
                           towerW + (size_t)layer * 9 * TW_TAP, 9, 128);
        (void)hipMemcpyAsync(towerB + (size_t)layer * 256, cw.b, 128 * 4, hipMemcpyDeviceToDevice, stream);
      };
      for (int i = 0; i < nblocks; ++i) { prep(c1[i], 2 * i); prep(c2[i], 2 * i + 1); }
      prep(vconv, 2 * nblocks);
      prep(pconv, 2 * nblocks + 1);
      hipLaunchKernelGGL(k_tower_prep, dim3((9 * 128 * 4 + 255) / 256), dim3(256), 0, stream, (const uint16_t *)stem.w, stemW, 9, 32);
      if (hipStreamSynchronize(stream) != hipSuccess) { *err = "k_tower_prep failed"; return FPC_ENODEVICE; }
    }
    loaded = true;
    return 0;
  }

  template <int DT, int CINC>
  int launch_conv_c(const ConvArgs &g, int M, std::string *err) {
    constexpr int BKS = CINC < 64 ? CINC : 64;
    constexpr int IMG = (CONV_BM + 2 * CONV_HMAX) * CINC * 2;
    constexpr int lds = (IMG > CONV_BM * 64 * 4 ? IMG : CONV_BM * 64 * 4) + 3 * CONV_BN * BKS * 2 + CONV_BM;
    bool &attr = attr_conv[DT][CINC == 32 ? 0 : CINC == 64 ? 1 : 2];    // per engine, hence per device
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
    int cur = 0;
    if (use_tower || use_towerw) {
      TowerArgs t{};
      t.boards = in_boards; t.leaf_slot = in_leaf_slot; t.leaf_turn = in_leaf_turn; t.board_stride = in_board_stride; t.one16 = one16();
      in_boards = nullptr;
      t.in16 = in16 + (size_t)guard * 32; t.Wstem = stemW; t.bstem = stem.b; t.Wt = towerW; t.bt = towerB;
      t.xfc = xfc; t.vw = vw; t.value = value_out; t.vb = vb;
      t.L = 2 * nblocks; t.P = P; t.R = dc.R; t.PP = PP; t.NR = PP - P; t.T0 = (P + 1) / 16; t.n_games = n; t.Kp = Kp; t.A_ch = dc.A_ch; t.rules = dc.rules;
#if defined(TW_STAMPS) || defined(TWW_STAMPS)
      static unsigned long long *d_stamps = nullptr;
      if (!d_stamps) { (void)hipMalloc(&d_stamps, 2 * 8 * 16 * 8); }
      (void)hipMemsetAsync(d_stamps, 0, 2 * 8 * 16 * 8, stream);
      t.stamps = d_stamps;
#endif
      // row tiles of 16 grid positions per wave (two waves along M): 8x8 -> 3, 9..11 -> 5, 12..14 -> 7
      const int mt = dc.R <= 8 ? 3 : dc.R <= 11 ? 5 : 7;
      bool &attr = attr_tower[DT];
      if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tower<DT, 3, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tower<DT, 5, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tower<DT, 7, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tower<DT, 7, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tower<DT, 7, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tower<DT, 3, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tower<DT, 5, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tower<DT, 7, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_towerc<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS);
        attr = true;
      }
      if (use_towerw) {
        bool &aw = attr_towerw[DT];
        if (!aw) {
#define FPC_TWW_ATTR(F_, MT_) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_towerw<DT, F_, MT_>), hipFuncAttributeMaxDynamicSharedMemorySize, tww_lds(F_))
#define FPC_TWW_ATTRS(F_) FPC_TWW_ATTR(F_, 2); FPC_TWW_ATTR(F_, 3); FPC_TWW_ATTR(F_, 4); FPC_TWW_ATTR(F_, 5); FPC_TWW_ATTR(F_, 6); FPC_TWW_ATTR(F_, 7)
          FPC_TWW_ATTRS(256); FPC_TWW_ATTRS(128);
#define FPC_TWW_ATTR1(F_, MT_) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_towerw<DT, F_, MT_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, tww_lds(F_))
          FPC_TWW_ATTR1(128, 4); FPC_TWW_ATTR1(256, 4); FPC_TWW_ATTR1(256, 6); FPC_TWW_ATTR1(256, 7); FPC_TWW_ATTR1(256, 8);
          FPC_TWW_ATTR1(256, 9); FPC_TWW_ATTR1(256, 11); FPC_TWW_ATTR1(256, 13);
#undef FPC_TWW_ATTR1
          aw = true;
        }
        // row tiles of 16 SQUARES for the first wave row (compact image): 8x8 -> 2, 9 -> 3, 10, 11 -> 4, 12 -> 5, 13 -> 6, 14 -> 7
        const int mtw = tww_mt(dc.R);
        if (mtw < 2 || mtw > 7) { *err = "k_towerw: board size outside 8..14"; return FPC_EINVAL; }
#define FPC_TWW_GO(F_, MT_) hipLaunchKernelGGL((k_towerw<DT, F_, MT_>), dim3(n), dim3(TWW_THREADS), tww_lds(F_), stream, t)
#define FPC_TWW_GOS(F_) switch (mtw) { case 2: FPC_TWW_GO(F_, 2); break; case 3: FPC_TWW_GO(F_, 3); break; case 4: FPC_TWW_GO(F_, 4); break; \
                                       case 5: FPC_TWW_GO(F_, 5); break; case 6: FPC_TWW_GO(F_, 6); break; default: FPC_TWW_GO(F_, 7); break; }
        // One wave row (eight waves side by side along the output channels, each over ALL row tiles): hidden 256 at every
        // board size, hidden 128 at 8x8.  Two wave rows x four: hidden 128 elsewhere.  FPC_TOWERW_ROWS=2 forces the latter.
        const int tiles = tww_tiles(dc.R);
        const bool onerow = tiles <= 4 || (F == 256 && towerw_rows != 2);
#define FPC_TWW_GO1(F_, MT_) hipLaunchKernelGGL((k_towerw<DT, F_, MT_, true>), dim3(n), dim3(TWW_THREADS), tww_lds(F_), stream, t)
        if (onerow && F == 128) FPC_TWW_GO1(128, 4);
        else if (onerow) {
          switch (tiles) {
            case 4: FPC_TWW_GO1(256, 4); break;   case 6: FPC_TWW_GO1(256, 6); break;   case 7: FPC_TWW_GO1(256, 7); break;
            case 8: FPC_TWW_GO1(256, 8); break;   case 9: FPC_TWW_GO1(256, 9); break;   case 11: FPC_TWW_GO1(256, 11); break;
            default: FPC_TWW_GO1(256, 13); break;
          }
        } else if (F == 256) { FPC_TWW_GOS(256) } else { FPC_TWW_GOS(128) }
#undef FPC_TWW_GO1
#undef FPC_TWW_GOS
#undef FPC_TWW_GO
      } else if (mt == 3 && tower_waves == 8) hipLaunchKernelGGL((k_tower<DT, 3, false, 8>), dim3(n), dim3(512), TW_LDS, stream, t);
      else if (mt == 3) hipLaunchKernelGGL((k_tower<DT, 3, false, 4>), dim3(n), dim3(256), TW_LDS, stream, t);
      else if (mt == 5 && tower_waves == 8) hipLaunchKernelGGL((k_tower<DT, 5, false, 8>), dim3(n), dim3(512), TW_LDS, stream, t);
      else if (mt == 5) hipLaunchKernelGGL((k_tower<DT, 5, false, 4>), dim3(n), dim3(256), TW_LDS, stream, t);
      else if (P != 16 && tower_waves == 8) hipLaunchKernelGGL((k_tower<DT, 7, false, 8>), dim3(n), dim3(512), TW_LDS, stream, t);
      else if (P != 16) hipLaunchKernelGGL((k_tower<DT, 7, false, 4>), dim3(n), dim3(256), TW_LDS, stream, t);
      // 14x14: the compact image (13 row tiles: loaders 6, staggered half 7) ...
      else if (tower_waves == 8 && tower_compact && dc.R == 14) hipLaunchKernelGGL((k_towerc<DT>), dim3(n), dim3(512), TW_LDS, stream, t);
      // ... or the bordered grid (grid pitch == tile height): two waves per SIMD, 7 x 2 tiles each
      else if (tower_waves == 8) hipLaunchKernelGGL((k_tower<DT, 7, true, 8>), dim3(n), dim3(512), TW_LDS, stream, t);
      else hipLaunchKernelGGL((k_tower<DT, 7, true, 4>), dim3(n), dim3(256), TW_LDS, stream, t);
      const hipError_t le = hipGetLastError();
      if (le != hipSuccess) { *err = std::string("k_tower launch failed: ") + hipGetErrorString(le); return FPC_ENODEVICE; }
#if defined(TW_STAMPS) || defined(TWW_STAMPS)
      if (getenv("FPC_TW_STAMPS_FILE")) {     // diagnostic build: dump the stamps of this launch
        unsigned long long h[2 * 8 * 16];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, t.stamps, sizeof(h), hipMemcpyDeviceToHost);
        if (FILE *f = fopen(getenv("FPC_TW_STAMPS_FILE"), "w")) {
          for (int i = 0; i < 2 * 8 * 16; ++i) fprintf(f, "%llu%c", h[i], i % 16 == 15 ? '\n' : ' ');
          fclose(f);
        }
      }
#endif
    } else if ((rc = conv(stem, in16, 32, nullptr, act[0], F, F, 0))) return rc;
    const bool fused = use_tower || use_towerw;
    for (int i = 0; i < nblocks && !fused; ++i) {
      const int t1 = (cur + 1) % 3, t2 = (cur + 2) % 3;
      if ((rc = conv(c1[i], act[cur], F, nullptr, act[t1], F, F, 0))) return rc;
      if ((rc = conv(c2[i], act[t1], F, act[cur], act[t2], F, F, 0))) return rc;
      cur = t2;
    }
    if (!fused) {
      if ((rc = conv(pconv, act[cur], F, nullptr, xfc, Kp, dc.A_ch, 1))) return rc;
      if ((rc = conv(vconv, act[cur], F, nullptr, yv, 32, 24, 0))) return rc;
    }
    if (mark_fn) mark_fn(mark_ctx, 2);
#if defined(TW_STRIP) && (TW_STRIP & 128)
    if (false) {                 // timing-only diagnostic build: the tower alone, no policy Linear behind it
#else
    if (logits_out) {            // null: legal-only policy head, the caller runs k_policy_gemv instead
#endif
      FcArgs f{};
      f.X = xfc; f.Wf = fcw; f.part = fc_part; f.Kp = Kp; f.Np = Np; f.ksteps = Kp / 16; f.Mtot = Gpad;
      f.G1 = fc_G1; f.s1 = fc_s1; f.s2 = fc_s2;
      const int mtiles = (n + 255) / 256;
      const int blocks = fc_G1 * fc_s1 + (Np / fc_gw - fc_G1) * fc_s2;
      bool &fattr = attr_fc[DT];
      if (!fattr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fc16<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, FC_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fcw<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, FCW_LDS);
        fattr = true;
      }
      if (fc_layout == 2) hipLaunchKernelGGL((k_fcw<DT>), dim3(blocks, mtiles), dim3(FC_THREADS), FCW_LDS, stream, f);
      else hipLaunchKernelGGL((k_fc16<DT>), dim3(blocks, mtiles), dim3(FC_THREADS), FC_LDS, stream, f);
      hipLaunchKernelGGL(k_fc_reduce, dim3((dc.A / 4 + 255) / 256, n), dim3(256), 0, stream, (const float *)fc_part, (const float *)fcb,
                         fc_G1, fc_s1, fc_s2, fc_gw, Gpad, dc.A, n, skip_dense ? (float *)nullptr : logits_out, d_stats);
      const hipError_t le = hipGetLastError();
      if (le != hipSuccess) { *err = std::string("k_fc launch failed: ") + hipGetErrorString(le); return FPC_ENODEVICE; }
    }
    if (!fused) hipLaunchKernelGGL((k_value_tail<DT>), dim3(n), dim3(64), 0, stream, (const uint16_t *)(yv + (size_t)guard * 32),
                       (const float *)vw, vb, P, dc.R, PP, n, value_out);
    if (hipGetLastError() != hipSuccess) { *err = "k_value_tail launch failed"; return FPC_ENODEVICE; }
    return 0;
  }

  // dense_logits false (the fused search): k_fc_reduce leaves only the softmax records; the expansion adds up the
  // split-K slabs at the leaf's legal moves itself (logit_src(), fpc_tree_kernels.h: LogitSrc)
  bool skip_dense = false;
  int forward(int n, bool dense_logits, std::string *err) {
    skip_dense = !dense_logits;
    const int rc = dtype ? forward_t<1>(n, d_logits, d_value, err) : forward_t<0>(n, d_logits, d_value, err);
    skip_dense = false;
    return rc;
  }
  static_assert(2 * FC_SPLITK <= LOGIT_MAX_SLABS, "logit_at() requests at most LOGIT_MAX_SLABS slabs per column");
  LogitSrc logit_src(bool dense) const {
    if (dense) return LogitSrc{d_logits, nullptr, nullptr, 0, 0, 0, 0, 256};
    return LogitSrc{nullptr, fc_part, fcb, fc_G1, fc_s1, fc_s2, Gpad, fc_gw};
  }
  // ---- legal-only policy head --------------------------------------------------------------
  // one-time: row-major copy of the policy weights + the legal-logit buffer
  int ensure_legal_head(std::string *err) {
    if (fcw2) return 0;
    if (Gmax > GEMV_MAXG) { *err = "legal-only policy head supports at most " + std::to_string(GEMV_MAXG) + " games per engine"; return FPC_EINVAL; }
    int rc;
    if ((rc = dmalloc(&fcw2, (size_t)Np * Kp, err)) || (rc = dmalloc(&d_ll, (size_t)Gmax * FPC_MAX_MOVES, err))) return rc;
    const long chunks = (long)Np * (Kp / 8);
    if (dtype) hipLaunchKernelGGL((k_fc_unfrag<1>), dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, (const uint16_t *)fcw, fcw2, Np, Kp);   // layouts 1 and 2: one fragment order
    else hipLaunchKernelGGL((k_fc_unfrag<0>), dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, stream, (const uint16_t *)fcw, fcw2, Np, Kp);
    if (hipGetLastError() != hipSuccess) { *err = "k_fc_unfrag launch failed"; return FPC_ENODEVICE; }
    return 0;
  }
  // tower + heads only, then the policy Linear at the legal moves of the G leaves in `t`
  int forward_legal(int n, const Tree &t, std::string *err) {
    int rc = ensure_legal_head(err);
    if (rc) return rc;
    if ((rc = dtype ? forward_t<1>(n, nullptr, d_value, err) : forward_t<0>(n, nullptr, d_value, err))) return rc;
    const size_t lds = (size_t)Kp * 2;
    bool &attr = attr_gemv;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_gemv<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_policy_gemv<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      attr = true;
    }
    if (lds > 64 * 1024) { *err = "legal-only policy head: activation row does not fit LDS"; return FPC_EINVAL; }
    const int blocks = 768;      // 3 per CU (47 KiB of LDS each at 14x14); pairs are dealt out in equal shares
    if (dtype) hipLaunchKernelGGL((k_policy_gemv<1>), dim3(blocks), dim3(GEMV_THREADS), lds, stream, dc, t, n, (const uint16_t *)xfc, (const uint16_t *)fcw2, (const float *)fcb, Kp, d_ll);
    else hipLaunchKernelGGL((k_policy_gemv<0>), dim3(blocks), dim3(GEMV_THREADS), lds, stream, dc, t, n, (const uint16_t *)xfc, (const uint16_t *)fcw2, (const float *)fcb, Kp, d_ll);
    if (hipGetLastError() != hipSuccess) { *err = "k_policy_gemv launch failed"; return FPC_ENODEVICE; }
    return 0;
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
