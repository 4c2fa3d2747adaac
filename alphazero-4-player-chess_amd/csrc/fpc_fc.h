// fpc_fc.h -- k_fc: the policy Linear (net.py:25: A -> A, 553 M weights at 14x14 = 1.1 GB in 16 bit)
// as a weight-streaming GEMM for M = 256 rows: every weight byte is used once per forward, so the
// kernel is bound by the HBM stream (1.11 GB per launch; 283 GFLOP ride on it).
//
//   * W is stored by the exporter in MFMA FRAGMENT ORDER  [kstep16][n_tile32][lane 64][8 elems]
//     (k-step major: all waves advance through K together, so what the chip reads at any moment is a
//     few contiguous regions spread over every HBM channel): one v_mfma_f32_32x32x16 B-operand of a
//     wave is one contiguous, perfectly coalesced 1-KiB read that goes straight from HBM into VGPRs --
//     the weights never touch LDS and are never shared between waves.
//   * A block is 4 waves, one per SIMD, each with the whole register file: a wave owns 64 output columns
//     for all 256 rows = 8 x 2 accumulator tiles of 32x32 (256 AGPRs).  Every activation fragment read
//     from LDS feeds two MFMAs (0.5 KiB of LDS per MFMA; the 8-wave / 32-column version of round 1 paid
//     1 KiB and stalled on the LDS queue), and the fragment reads run a whole k-step (16 MFMAs, 512
//     cycles) ahead of their use.
//   * Only the activations X[256][K] (12 MB, L2 resident, re-read by every column group) go through
//     LDS (2 x 32 KiB double buffer of BK = 64 stages, XOR-swizzled), fetched in full 128-byte lines
//     two stages ahead through registers.
//   * Work decomposition: column group j (256 columns) x K-split i.  Groups [0, G1) are cut into s1
//     K-splits, the remaining groups into s2 = 2*s1 half-length ones, G1 chosen by the host so that
//     the short blocks fill the tail of the last round (368 equal blocks on 256 CUs would idle 28 %).
//     Block ids put the K-split in the low bits, so one XCD (id mod 8) only ever walks one K window
//     of X and keeps it in its own L2.
//   * Every block writes its f32 partial slab [Mtot][256]; k_fc_reduce adds a group's slabs and the
//     bias in a fixed order (deterministic, no atomics).
//   * No runtime conditionals surround memory operations in the steady-state loop (hipcc would fall back
//     to s_waitcnt vmcnt(0) at the join); the first and the last two stages are separate copies of the
//     stage body.  The first MFMA on every accumulator takes C = 0 as an inline constant.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace fpc {

struct FcArgs {
  const uint16_t *X;      // [Mpad][Kp]
  const uint16_t *Wf;     // fragment order
  float *part;            // [slabs][Mtot][256]
  int Kp, Np, ksteps, Mtot;
  int G1, s1, s2;         // groups [0,G1): s1 splits; groups [G1, Np/256): s2 splits
};

constexpr int FC_THREADS = 256;

template <int DT>
__global__ void __launch_bounds__(FC_THREADS, 1) k_fc(FcArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char As[2][256 * 64 * 2];   // 2 x 32 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block id -> (column group, K-split, slab)
  const int nbig = g.G1 * g.s1;
  const int id = blockIdx.x;
  const bool big = id < nbig;
  const int idr = big ? id : id - nbig;
  const int sk = big ? g.s1 : g.s2;
  const int group = big ? idr / sk : g.G1 + idr / sk;
  const int split = idr % sk;
  const int ntile = group * 8 + wave * 2;         // this wave's two 32-column tiles: ntile, ntile + 1
  const int KS = g.ksteps / sk;                   // k-steps (of 16) handled by this block; multiple of 8
  const int ks0 = split * KS;
  const int S = KS / 4;                           // stages of BK = 64 (even, >= 4)
  const long wstride = (long)(g.Np / 32) * 64;    // u32x4 units between consecutive k-steps
  const u32x4_t *wsrc = reinterpret_cast<const u32x4_t *>(g.Wf) + ((long)ks0 * (g.Np / 32) + ntile) * 64 + lane;
  const long mrow0 = (long)blockIdx.y * 256;
  // activation staging: 256 threads cover 32 rows (8 lanes x 16 B = one 128-byte line each) per pass, 8 passes
  const uint16_t *xsrc = g.X + (mrow0 + (tid >> 3)) * g.Kp + (long)ks0 * 16 + (tid & 7) * 8;
  const long xrow32 = 32L * g.Kp;

  f32x16_t acc[8][2];
  u32x4_t wq[2][4][2];    // weight fragments of stages s, s+1 (slot = stage parity) [k-step][column tile]
  u32x4_t ra[8];          // the activation tile of the next stage on its way global -> LDS
  u32x4_t xf[2][8];       // activation fragments of the current / the next k-step
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 1> c1{};
  const std::integral_constant<int, 2> c2{};
  const std::integral_constant<int, 3> c3{};

  auto xload = [&](int i, int s) { ra[i] = *reinterpret_cast<const u32x4_t *>(xsrc + i * xrow32 + (long)s * 64); };
  auto xstore = [&](int i, int buf) {
    *reinterpret_cast<u32x4_t *>(As[buf] + lds_off<64>((tid >> 3) + 32 * i, tid & 7)) = ra[i];
  };
  auto wload = [&](int p, int ks, int n, int s) { wq[p][ks][n] = wsrc[(long)(4 * s + ks) * wstride + n * 64]; };
  auto xfrag = [&](int b, int t, const unsigned char *ab, int ks) {
    xf[b][t] = *reinterpret_cast<const u32x4_t *>(ab + lds_off<64>(t * 32 + (lane & 31), ks * 2 + (lane >> 5)));
  };

  // One k-step of stage s (parity P): 16 MFMAs on xf[KS & 1] in 8 groups of 2 (one row tile x two column
  // tiles), the activation fragments of the NEXT k-step read between the groups, and this k-step's share
  // of the stage's memory traffic placed behind its groups -- order pinned with sched_barrier:
  //   LW: the two weight fragments of this k-step are re-requested for stage s + 2 right after their last use;
  //   SX (k-steps 0, 1): the staged activations of stage s + 1 go to the other LDS buffer, four 16-byte writes each;
  //   LX: each of those registers is re-requested for stage s + 2 right behind its write (a whole stage of flight);
  //   k-step 3 opens with THE stage barrier: every wave has written its share of stage s + 1 and has all
  //   its fragments of stage s in registers, so the next k-step's fragments come from the other buffer
  //   and the buffer of stage s may be overwritten during stage s + 1.
  auto kstep = [&](auto p_c, auto ks_c, auto z_c, auto lw_c, auto sx_c, auto lx_c, int s) {
    constexpr int P = decltype(p_c)::value, KSI = decltype(ks_c)::value;
    constexpr bool Z = decltype(z_c)::value != 0, LW = decltype(lw_c)::value != 0, SX = decltype(sx_c)::value != 0,
                   LX = decltype(lx_c)::value != 0;
    constexpr int CUR = KSI & 1, NXT = CUR ^ 1;
    if (KSI == 3) __syncthreads();
    const unsigned char *ab = KSI == 3 ? As[P ^ 1] : As[P];
    constexpr int KSN = KSI == 3 ? 0 : KSI + 1;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      acc[t][0] = E16<DT>::mfma(xf[CUR][t], wq[P][KSI][0], Z ? zero16 : acc[t][0]);
      acc[t][1] = E16<DT>::mfma(xf[CUR][t], wq[P][KSI][1], Z ? zero16 : acc[t][1]);
      __builtin_amdgcn_sched_barrier(0);
      xfrag(NXT, t, ab, KSN);
      if (SX && KSI < 2 && (t & 1)) {
        xstore(KSI * 4 + (t >> 1), P ^ 1);
        if (LX) xload(KSI * 4 + (t >> 1), s + 2);
      }
      if (LW && t == 7) { wload(P, KSI, 0, s + 2); wload(P, KSI, 1, s + 2); }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto stage = [&](auto p_c, auto z_c, auto lw_c, auto sx_c, auto lx_c, int s) {
    kstep(p_c, c0, z_c, lw_c, sx_c, lx_c, s);
    kstep(p_c, c1, c0, lw_c, sx_c, lx_c, s);
    kstep(p_c, c2, c0, lw_c, sx_c, lx_c, s);
    kstep(p_c, c3, c0, lw_c, sx_c, lx_c, s);
  };

  // prologue: stage 0 of X into LDS, stage 1 into registers, weights of stages 0 and 1, fragments of (0, 0)
#pragma unroll
  for (int i = 0; i < 8; ++i) xload(i, 0);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { wload(0, ks, 0, 0); wload(0, ks, 1, 0); }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { wload(1, ks, 0, 1); wload(1, ks, 1, 1); }
#pragma unroll
  for (int i = 0; i < 8; ++i) xstore(i, 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) xload(i, 1);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 8; ++t) xfrag(0, t, As[0], 0);

  stage(c0, c1, c1, c1, c1, 0);                   // first stage: C = 0 on the first k-step
  stage(c1, c0, c1, c1, c1, 1);
  int s = 2;
#pragma unroll 1
  for (; s + 4 <= S; s += 2) {
    stage(c0, c0, c1, c1, c1, s);
    stage(c1, c0, c1, c1, c1, s + 1);
  }
  stage(c0, c0, c0, c1, c0, s);                   // last two stages: nothing left to fetch
  stage(c1, c0, c0, c0, c0, s + 1);

  const int slab = big ? id : nbig + idr;
  float *out = g.part + ((long)slab * g.Mtot + mrow0) * 256;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = wave * 64 + n * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[(long)m * 256 + col] = acc[t][n][r];
      }
    }
}

// logits[m][n] = bias[n] + slab[base][m][n%256] + slab[base+1][m][n%256] + ...   (fixed order)
__global__ void __launch_bounds__(256) k_fc_reduce(const float *part, const float *bias, int G1, int s1, int s2, int Mtot, int A,
                                                   int n_rows, float *logits) {
  const int q = blockIdx.x * 256 + threadIdx.x;         // float4 index within a row
  const int m = blockIdx.y;
  if (m >= n_rows || q * 4 >= A) return;
  const int j = q >> 6;                                 // column group of 256
  const int base = j < G1 ? j * s1 : G1 * s1 + (j - G1) * s2, cnt = j < G1 ? s1 : s2;
  float4 v = *reinterpret_cast<const float4 *>(bias + q * 4);
  for (int k = 0; k < cnt; ++k) {
    const float4 p = *reinterpret_cast<const float4 *>(part + ((long)(base + k) * Mtot + m) * 256 + (q & 63) * 4);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
  }
  *reinterpret_cast<float4 *>(logits + (long)m * A + q * 4) = v;
}

}  // namespace fpc
