// fpc_fc.h -- the policy Linear (net.py:25: A -> A, 553 M weights at 14x14 = 1.1 GB in 16 bit) as a weight-streaming
// GEMM for M = 256 rows: every weight byte is used once per forward, so the kernels are bound by the HBM stream
// (1.11 GB per launch; 283 GFLOP ride on it) and by whatever else travels the same L1 / L2 / LDS-DMA path.
//   k_fcw   (round 5; blob fc_layout 2; the default wherever its one-round K-split exists: every board of 8..14 a side on
//           a 256-CU part): 256 x 384 block tiles, one round of blocks -- see below.
//   k_fc16  (round 3; fc_layout 1; the fallback): 256 x 256 block tiles, long and short blocks in two rounds.
//   (k_fc, rounds 1-4: the same on v_mfma_f32_32x32x16 with its own fragment order -- retired in round 5; what it
//   established is kept in the notes below.)
// Common to both:
//   * W is stored by the exporter in MFMA FRAGMENT ORDER  [k-step of 32][column tile of 16][lane = 16 q + c][8]
//     (element (k32, nt, q, c, e) = W'[16 nt + c][32 k32 + 8 q + e]; k-step major: all waves advance through K together,
//     so what the chip reads at any moment is a few contiguous regions spread over every HBM channel): one A operand of a
//     wave is one contiguous, perfectly coalesced 1-KiB piece.
//   * A block is 4 waves, one per SIMD, each with the whole register file; a wave owns 64 (k_fc16) or 96 (k_fcw) output
//     columns for all 256 rows, issued as W x X^T: a lane owns ONE activation row and four consecutive output columns
//     per accumulator, so a partial tile leaves as one 16-byte store per lane.
//   * The weight pieces carry the non-temporal hint: 1.1 GB read once per launch would otherwise push the
//     activations, the tower's weights and the tree out of L2 / MALL (same-box A/B: +2.6 % on the whole step).
//   * Both operands arrive by LDS-DMA (global_load_lds_dwordx4, no VGPR hop): the activations X[256][K]
//     (12 MB, L2 resident, re-read by every column group) in full 128-byte lines into 32 KiB stage buffers shared by
//     the block (XOR-swizzled through the source address), the weights into a private ring per wave from which each
//     lane reads back exactly the 16 bytes it stored.
//   * Work = column group x K-split; block ids put the K-split in the low bits, so one XCD (id mod 8) only ever walks
//     one K window of X and keeps it in its own L2.  Every block writes its f32 partial slab [Mtot][group width];
//     k_fc_reduce adds a group's slabs and the bias in a fixed order (deterministic, no atomics).
//   * The first MFMA on every accumulator takes C = 0 as an inline constant.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fpc_tower.h"   // M16: the 16x16x32 MFMA wrappers

namespace fpc {

struct FcArgs {
  const uint16_t *X;      // [Mpad][Kp]
  const uint16_t *Wf;     // fragment order
  float *part;            // [slabs][Mtot][256]
  int Kp, Np, ksteps, Mtot;
  int G1, s1, s2;         // groups [0,G1): s1 splits; groups [G1, Np/256): s2 splits
};

constexpr int FC_THREADS = 256;
constexpr int FC_XBUF = 256 * 64 * 2;             // one BK = 64 stage of activations: 32 KiB
constexpr int FC_WRING = 16 * 1024;               // per wave: 2 stages x 4 k-steps x 2 column tiles x 1 KiB
constexpr int FC_LDS = 3 * FC_XBUF + 4 * FC_WRING;   // 163840: the whole LDS of a CU

// one 1-KiB LDS-DMA piece (64 lanes x 16 B): global (SGPR base + per-lane byte offset) -> LDS, no VGPR
// destination.  Inline asm on purpose: every wait for these pieces is an explicit counted s_waitcnt in
// k_fc.  M0 carries the wave-uniform LDS byte address and is restored inside the statement.
__device__ __forceinline__ void fc_dma(const void *gsrc_uniform, uint32_t lane_off, uint32_t lds_addr_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_addr_uniform) : "memory");
}
// two consecutive pieces (source and destination both advance by 1 KiB: the instruction's immediate moves BOTH
// addresses, tools/micro/dma_imm_check.cpp) behind ONE write of M0: an s_mov to M0 waits for the wave's LDS reads
// in flight (~20 cycles beside a fragment stream, tools/micro/dma_stagger.cpp), the piece itself costs ~6
__device__ __forceinline__ void fc_dma2_nt(const void *gsrc_uniform, uint32_t lane_off, uint32_t lds_addr_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_addr_uniform) : "memory");
}
// the same with the non-temporal hint: for bytes that are read once per launch (the weight stream)
__device__ __forceinline__ void fc_dma_nt(const void *gsrc_uniform, uint32_t lane_off, uint32_t lds_addr_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_addr_uniform) : "memory");
}

// ------------------------------------------------------------------------------------------------------------
// k_fc16: the Linear on v_mfma_f32_16x16x32 (the shape the chip holds the higher clock on under load,
// MI355X_MICROARCH.md "DVFS give-back" item 7; k_tower uses it for the same reason) with 256 x 256 block tiles:
//   * a wave owns 64 columns x 256 rows = 16 x 4 accumulator tiles (256 AGPRs); every activation fragment read from LDS
//     feeds four MFMAs, and the fragment reads run a whole k-step ahead of their use (two sets of 16);
//   * three 32 KiB activation stages + a 16 KiB weight ring per wave (2 stages x 2 k-steps x 4 tiles); BOTH streams run
//     the same two stages ahead: vmcnt retires in issue order, so waiting for a young activation piece would otherwise
//     force every older, deeper weight fragment to have landed and cut the weight prefetch to a fraction of its ring;
//   * column groups [0, G1) are cut into s1 K-splits, the remaining groups into s2 = 2 * s1 half-length ones, G1
//     chosen by the host so that the short blocks fill the tail of the last round (plan_fc);
//   * a k-step is 32 deep: 64 MFMAs on 16 activation fragments x 4 weight fragments; 8 DMA pieces per k-step in the
//     fixed order  X, W, W, X, X, W, W, X  (the weight pieces in pairs behind one M0 write).  Counted waits:
//     first k-step of a stage: the weights of its second k-step were issued three k-steps ago; behind their last
//     piece came 1 + 8 + 8 pieces -> vmcnt(17); second k-step: the activations of the next stage (last piece: two
//     k-steps ago) and everything older -> vmcnt(8), then THE stage barrier.
template <int DT>
__global__ void __launch_bounds__(FC_THREADS, 1) k_fc16(FcArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [3 X buffers][4 per-wave weight rings]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbig = g.G1 * g.s1;
  const int id = blockIdx.x;
  const bool big = id < nbig;
  const int idr = big ? id : id - nbig;
  const int sk = big ? g.s1 : g.s2;
  const int group = big ? idr / sk : g.G1 + idr / sk;
  const int split = idr % sk;
  const int ntile = group * 16 + wave * 4;        // this wave's four 16-column tiles
  const int KS = g.ksteps / sk;                   // k-steps of 16 handled by this block; multiple of 8
  const int ks0 = split * KS;
  const int S = KS / 4;                           // stages of BK = 64 (even, >= 4)
  const long mrow0 = (long)blockIdx.y * 256;
  // weight stream of this wave: fragment (k32-step k, tile ntile + n) = 1 KiB at wbase + (k * Np/16 + n) KiB
  const unsigned char *wbase = reinterpret_cast<const unsigned char *>(g.Wf) + ((long)(ks0 / 2) * (g.Np / 16) + ntile) * 1024;
  const long wk32 = (long)(g.Np / 16) * 1024;
  const uint32_t wlane = (uint32_t)lane * 16u;
  const unsigned char *xbase = reinterpret_cast<const unsigned char *>(g.X) + ((mrow0 + wave * 64) * g.Kp + (long)ks0 * 16) * 2;
  const long xpiece = 8L * g.Kp * 2;
  uint32_t xlane[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int j = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
    xlane[par] = (uint32_t)((lane >> 3) * g.Kp * 2 + j * 16);
  }
  const uint32_t lds_x = 0, lds_w = (uint32_t)(3 * FC_XBUF + wave * FC_WRING);
  const unsigned char *wr = smem + 3 * FC_XBUF + wave * FC_WRING + lane * 16;

  f32x4_t acc[16][4];
  u32x4_t xf[2][16], wf[2][4];
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 1> c1{};

  auto issue_x = [&](int s, int p) {              // piece p (8 rows) of activation stage s
    const int sc = s < S ? s : S - 1;
    fc_dma(xbase + (long)p * xpiece + (long)sc * 128, xlane[p & 1],
           (uint32_t)__builtin_amdgcn_readfirstlane(lds_x + (s % 3) * FC_XBUF + (wave * 64 + p * 8) * 128));
  };
  auto issue_w2 = [&](int s, int j, int n0) {      // column tiles n0, n0 + 1 of k32-step j of stage s
    const int sc = s < S ? s : S - 1;
    fc_dma2_nt(wbase + (long)(2 * sc + j) * wk32 + n0 * 1024, wlane,
               (uint32_t)__builtin_amdgcn_readfirstlane(lds_w + ((s & 1) * 8 + j * 4 + n0) * 1024));
  };
  auto xfrag = [&](int b, int t, const unsigned char *ab, int j) {
    xf[b][t] = *reinterpret_cast<const u32x4_t *>(ab + lds_off<64>(t * 16 + (lane & 15), j * 4 + (lane >> 4)));
  };
  auto wfrag = [&](int b, int s, int j) {
#pragma unroll
    for (int n = 0; n < 4; ++n) wf[b][n] = *reinterpret_cast<const u32x4_t *>(wr + ((s & 1) * 8 + j * 4 + n) * 1024);
  };
  auto kstep = [&](auto j_c, auto z_c, int s) {
    constexpr int J = decltype(j_c)::value;
    constexpr bool Z = decltype(z_c)::value != 0;
    constexpr int CUR = J, NXT = J ^ 1;
    if (J == 0) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    if (J == 1) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); __syncthreads(); }
    const unsigned char *ab = smem + ((J == 1 ? s + 1 : s) % 3) * FC_XBUF;
    constexpr int JN = J ^ 1;
    __builtin_amdgcn_sched_barrier(0);
    wfrag(NXT, J == 1 ? s + 1 : s, JN);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[t][n] = M16<DT>::mfma(wf[CUR][n], xf[CUR][t], Z ? zero4 : acc[t][n]);
      __builtin_amdgcn_sched_barrier(0);
      xfrag(NXT, t, ab, JN);
      if (t == 1) issue_x(s + 2, J * 4 + 0);
      if (t == 5) issue_w2(s + 2, J, 0);
      if (t == 7) issue_x(s + 2, J * 4 + 1);
      if (t == 9) issue_x(s + 2, J * 4 + 2);
      if (t == 13) issue_w2(s + 2, J, 2);
      if (t == 15) issue_x(s + 2, J * 4 + 3);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: stages 0 and 1 of both streams, then the fragments of (0, 0)
#pragma unroll
  for (int st = 0; st < 2; ++st) {
#pragma unroll
    for (int p = 0; p < 8; ++p) issue_x(st, p);
#pragma unroll
    for (int j = 0; j < 2; ++j) { issue_w2(st, j, 0); issue_w2(st, j, 2); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 16; ++t) xfrag(0, t, smem, 0);
  wfrag(0, 0, 0);

  kstep(c0, c1, 0);                               // (0, 0): C = 0
  kstep(c1, c0, 0);
#pragma unroll 1
  for (int s = 1; s < S; ++s) {
    kstep(c0, c0, s);
    kstep(c1, c0, s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail pieces: nothing may still target LDS at exit

  const int slab = big ? id : nbig + idr;
  float *out = g.part + ((long)slab * g.Mtot + mrow0) * 256;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int m = t * 16 + (lane & 15);
#pragma unroll
    for (int n = 0; n < 4; ++n)
      *reinterpret_cast<f32x4_t *>(out + (long)m * 256 + wave * 64 + n * 16 + (lane >> 4) * 4) = acc[t][n];
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_fcw (round 5): the same Linear on 256 x 384 block tiles -- the default (blob header fc_layout = 2).
//
// Why.  tools/micro/fc_stream.cpp replays k_fc16's memory pattern with the arithmetic taken out or left in (register
// operands, no LDS reads) and reproduces its 250 us; taken apart on one box: the weight stream alone 168 us (6.6 TB/s, the
// HBM rate), + the activation re-reads 214 (X is L2-resident, but its 1.1 GB -- every column group reads all of X --
// go through the same L1 / L2 / LDS-DMA path as the weights, at the core clock), + the MFMAs 252 (the clock drops under
// them and that path with it), + the 126 MB of split-K slabs 279.  Row pitch, a tiled X, rotated K order, deeper
// run-ahead, store flavours: within 3 %.  What does move it is the BYTES on that path:
//   * a block owns 384 columns instead of 256: 62 column groups instead of 92 read X -> 0.74 GB of re-reads (-1/3);
//   * 62 groups x 4 K-splits = 248 blocks = ONE round on 256 CUs (k_fc16: 256 long + 224 short blocks, two rounds
//     with a second ramp-up), every block the same length;
//   * 248 slabs of 256 x 384 f32 = 97 MB instead of 126 MB, four per column, all groups alike.
//   Same-box micro-benchmark with slabs and MFMAs: 308 -> 243 us.
// How.  A wave owns 96 columns x 256 rows = 16 x 6 accumulator tiles of 16 x 16 (384 registers), so the operand
// registers had to shrink: the activation fragments are a RING OF FOUR quads (fragment t + 3 is read while tile row t
// multiplies; k_fc16 holds two whole k-steps of 16), the six weight fragments are double-buffered as before.
// LDS: two activation stages (64 KiB; k_fc16: three) + a 24 KiB weight ring per wave (2 stages x 2 k-steps x 6 tiles)
// = 160 KiB.  The activations therefore run ONE stage ahead and the weights two; vmcnt retires in issue order, so
// within a stage the activation pieces go out FIRST (k-step 0, tile rows 0..7) and the weight pieces behind them:
// waiting for the youngest activation piece then leaves the twelve weight pieces issued behind it in flight.
// Issue pattern per stage and wave: k-step 0: 8 X pieces, then 3 weight pairs; k-step 1: 3 weight pairs.  Counted waits:
//   start of k-step (s, 0): the weights of (s, 1), issued in (s - 2, 1); behind them a whole stage       -> vmcnt(20)
//   start of k-step (s, 1): the weights of (s + 1, 0), issued in (s - 1, 0); behind them 6 + 8 + 6       -> vmcnt(20)
//   k-step (s, 1) behind tile row 13: the activations of stage s + 1, issued in (s, 0); behind them 6 + 6 -> vmcnt(12),
//   then THE stage barrier: every wave's pieces of stage s + 1 have landed, and every wave has read its last fragment
//   of stage s (fragment 15 of k-step 1 is read behind tile row 12), so the buffer of stage s may be refilled.
// Weight order in memory: as k_fc16's, [k-step of 32][column tile of 16][lane][8] over Np = 384 * groups columns.
// Summation order: a column's K quarter in ascending k inside the MFMA accumulators, then bias + the four slabs
// ascending (k_fc_reduce / logit_at): other last bits than k_fc16's where that used eight slabs.
// One v_mfma_f32_16x16x32 with its accumulator PINNED to a register class: AG = true -> an AGPR quad, false -> a VGPR quad.
// k_fcw keeps 384 accumulator registers per lane; left to itself hipcc parks all of them in the 256 AGPRs by turns and
// moves 66 tiles through v_accvgpr_read / _write on every stage (528 moves + 142 s_nop per 192 MFMAs).  With the class
// in the constraint, 64 tiles live in a0..a255, 32 in VGPRs, and nothing moves.  (Inline asm: the compiler adds the
// lgkmcnt wait for the fragment registers it loaded itself; there is no MFMA -> MFMA dependency closer than 95 MFMAs.)
template <int DT, bool AG, bool Z>
__device__ __forceinline__ void fcw_mfma(f32x4_t &acc, const u32x4_t &a, const u32x4_t &b) {
  if (DT == 1) {
    if (AG) { if (Z) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }
    else    { if (Z) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
  } else {
    if (AG) { if (Z) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }
    else    { if (Z) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b)); else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
  }
}

constexpr int FCW_COLS = 384;
constexpr int FCW_WRING = 24 * 1024;
constexpr int FCW_LDS = 2 * FC_XBUF + 4 * FCW_WRING;   // 163840

template <int DT>
__global__ void __launch_bounds__(FC_THREADS, 1) k_fcw(FcArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2 X buffers][4 per-wave weight rings]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sk = g.s1;                            // K-splits per column group (every group alike)
  const int id = blockIdx.x;
  const int group = id / sk, split = id % sk;     // the K-split in the low bits: an XCD walks one K window of X
  const int ntile = group * 24 + wave * 6;        // this wave's six 16-column tiles
  const int KS = g.ksteps / sk;                   // k-steps of 16 handled by this block
  const int ks0 = split * KS;
  const int S = KS / 4;                           // stages of BK = 64
  const long mrow0 = (long)blockIdx.y * 256;
  const unsigned char *wbase = reinterpret_cast<const unsigned char *>(g.Wf) + ((long)(ks0 / 2) * (g.Np / 16) + ntile) * 1024;
  const long wk32 = (long)(g.Np / 16) * 1024;
  const uint32_t wlane = (uint32_t)lane * 16u;
  const unsigned char *xbase = reinterpret_cast<const unsigned char *>(g.X) + ((mrow0 + wave * 64) * g.Kp + (long)ks0 * 16) * 2;
  const long xpiece = 8L * g.Kp * 2;
  uint32_t xlane[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int j = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
    xlane[par] = (uint32_t)((lane >> 3) * g.Kp * 2 + j * 16);
  }
  const uint32_t lds_w = (uint32_t)(2 * FC_XBUF + wave * FCW_WRING);
  const unsigned char *wr = smem + 2 * FC_XBUF + wave * FCW_WRING + lane * 16;

  f32x4_t acc[16][6];
  u32x4_t xr[4], wf[2][6];
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 1> c1{};

  auto issue_x = [&](int s, int p) {              // piece p (8 rows) of activation stage s
    const int sc = s < S ? s : S - 1;
    fc_dma(xbase + (long)p * xpiece + (long)sc * 128, xlane[p & 1],
           (uint32_t)__builtin_amdgcn_readfirstlane((s & 1) * FC_XBUF + (wave * 64 + p * 8) * 128));
  };
  auto issue_w2 = [&](int s, int j, int n0) {      // column tiles n0, n0 + 1 of k32-step j of stage s
    const int sc = s < S ? s : S - 1;
    fc_dma2_nt(wbase + (long)(2 * sc + j) * wk32 + n0 * 1024, wlane,
               (uint32_t)__builtin_amdgcn_readfirstlane(lds_w + ((s & 1) * 12 + j * 6 + n0) * 1024));
  };
  auto xfrag = [&](int slot, int t, const unsigned char *ab, int j) {
    xr[slot] = *reinterpret_cast<const u32x4_t *>(ab + lds_off<64>(t * 16 + (lane & 15), j * 4 + (lane >> 4)));
  };
  auto wfrag = [&](int b, int s, int j) {
#pragma unroll
    for (int n = 0; n < 6; ++n) wf[b][n] = *reinterpret_cast<const u32x4_t *>(wr + ((s & 1) * 12 + j * 6 + n) * 1024);
  };
  auto kstep = [&](auto j_c, auto z_c, int s) {
    constexpr int J = decltype(j_c)::value;
    constexpr bool Z = decltype(z_c)::value != 0;
    constexpr int CUR = J, NXT = J ^ 1;
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    const unsigned char *ab = smem + (s & 1) * FC_XBUF;                                  // this stage's activations
    const unsigned char *abn = J == 1 ? smem + ((s + 1) & 1) * FC_XBUF : ab;             // the next k-step's
    __builtin_amdgcn_sched_barrier(0);
    wfrag(NXT, J == 1 ? s + 1 : s, NXT);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        if (n < 4) fcw_mfma<DT, true, Z>(acc[t][n], wf[CUR][n], xr[t & 3]);
        else fcw_mfma<DT, false, Z>(acc[t][n], wf[CUR][n], xr[t & 3]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (J == 1 && t == 13) { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); __syncthreads(); }
      if (t + 3 < 16) xfrag((t + 3) & 3, t + 3, ab, J);
      else xfrag((t + 3) & 3, t + 3 - 16, abn, NXT);
      if (J == 0 && t < 8) issue_x(s + 1, t);
      if (J == 0 && (t == 8 || t == 10 || t == 12)) issue_w2(s + 2, 0, t - 8);
      if (J == 1 && (t == 2 || t == 6 || t == 10)) issue_w2(s + 2, 1, (t - 2) / 2);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: the activations of stage 0, the weights of stages 0 and 1, then the first fragments
#pragma unroll
  for (int p = 0; p < 8; ++p) issue_x(0, p);
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int j = 0; j < 2; ++j) { issue_w2(st, j, 0); issue_w2(st, j, 2); issue_w2(st, j, 4); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 3; ++t) xfrag(t, t, smem, 0);
  wfrag(0, 0, 0);

  kstep(c0, c1, 0);                               // (0, 0): C = 0
  kstep(c1, c0, 0);
#pragma unroll 1
  for (int s = 1; s < S; ++s) {
    kstep(c0, c0, s);
    kstep(c1, c0, s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail pieces: nothing may still target LDS at exit
  // The MFMAs above are inline asm: hipcc's hazard recogniser does not see them, so the wait states the ISA asks for between
  // an 8-pass MFMA writing a register and a store / v_accvgpr_read reading it (11) are spelt out here, once, behind the loop.
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");

  float *out = g.part + ((long)id * g.Mtot + mrow0) * FCW_COLS;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int m = t * 16 + (lane & 15);
#pragma unroll
    for (int n = 0; n < 6; ++n)
      *reinterpret_cast<f32x4_t *>(out + (long)m * FCW_COLS + wave * 96 + n * 16 + (lane >> 4) * 4) = acc[t][n];
  }
}

// ------------------------------------------------------------------------------------------------------------
// Round 4, built, measured and NOT kept (same-box A/Bs, both variants bit-identical to k_fc16): the recipe that took the hidden-256 tower from 0.36 to 0.49 of peak (fpc_towerw.h: two waves
// per SIMD, weights straight from memory into registers three k-steps ahead by counted-vmcnt inline-asm loads, no
// weight ring in LDS, activations by LDS-DMA as here) does NOT carry over to this Linear:
//   * k_fc16w, wave = 128 rows x 64 columns (two waves share every weight fragment): 0.364 ms with the non-temporal hint
//     on the weight loads (the second wave's request misses too: HBM sees the weights twice), 0.307 ms without it;
//   * k_fc16n, wave = 256 rows x 32 columns (every fragment has one owner, nt kept, twice the activation reads per
//     MFMA): 0.277-0.292 ms against k_fc16's 0.272-0.280 ms, and k_tower beside it 2 % slower (power).
// So the issue port is not what holds this kernel (k_fc16's MFMA pipe is idle 48 % of the time WAITING FOR DATA): the
// bound is the memory side -- 1.17 GB of weights from HBM + 1.1 GB of activation re-reads and 0.13 GB of slab writes
// through the same L2s.
// A lesson worth keeping from the one GPU fault of that work: an inline-asm load into a register the compiler considers
// DEAD (the clamped tail prefetches behind the last k-step) is still in flight when hipcc hands that register to later
// code -- the epilogue's address arithmetic sat in front of the plain `s_waitcnt vmcnt(0)` statement, a late-landing
// fragment overwrote a store address, the kernel faulted.  A wait that must cover asm loads has to be TIED to their
// destination registers ("+v" operands of the s_waitcnt statement), also for loads whose data nobody reads.
// ------------------------------------------------------------------------------------------------------------

// logits[m][n] = bias[n] + slab[base][m][n%256] + slab[base+1][m][n%256] + ...   (fixed order)
// Sums a column group's K-split slabs and the bias in fixed order, writes the logits, and -- while the
// block still holds its 1024 logits of the row in registers -- leaves that chunk's softmax statistics
// (softmax_chunk_stats, fpc_tree_kernels.h) for the expansion, which then never sweeps the row again.
__global__ void __launch_bounds__(256) k_fc_reduce(const float *part, const float *bias, int G1, int s1, int s2, int gw, int Mtot, int A,
                                                   int n_rows, float *logits, float *stats) {
  __shared__ float red[8];
  const int q = blockIdx.x * 256 + threadIdx.x;         // float4 index within a row
  const int m = blockIdx.y;
  if (m >= n_rows) return;
  const bool valid = q * 4 < A;
  float4 v{0.f, 0.f, 0.f, 0.f};
  if (valid) {
    const int j = gw == 256 ? q >> 6 : (q * 4) / gw;    // column group of gw columns (256: k_fc / k_fc16; 384: k_fcw)
    const int col = q * 4 - j * gw;
    const int base = j < G1 ? j * s1 : G1 * s1 + (j - G1) * s2, cnt = j < G1 ? s1 : s2;
    v = *reinterpret_cast<const float4 *>(bias + q * 4);
    for (int k = 0; k < cnt; ++k) {
      const float4 p = *reinterpret_cast<const float4 *>(part + ((long)(base + k) * Mtot + m) * gw + col);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (logits) *reinterpret_cast<float4 *>(logits + (long)m * A + q * 4) = v;     // null: the fused search keeps only the records
  }
  softmax_chunk_stats(v, valid, red, stats + ((long)m * SM_MAXCH + blockIdx.x) * SM_REC);
}

}  // namespace fpc
