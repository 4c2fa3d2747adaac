// fpc_tower.h -- k_tower: the whole residual tower of net.py:6-63 (stem + 2*Nb residual convs + both
// head convs, hidden = 128) in ONE launch, one game per workgroup, activations never leaving the CU.
//
// Shape of the kernel (gfx950):
//   * 8 waves = two per SIMD, each with half the register file: wave (wm, wn) owns MT row tiles of 16 grid
//     positions x 2 column tiles of 16 output channels (112 x 32 outputs at 14x14) as v_mfma_f32_16x16x32
//     accumulators, plus the residual x_l of the same outputs packed in registers.  The two waves of a SIMD (w and
//     w + 4) have ROLES: the older one carries the whole weight DMA of its pair, the other one passes every tap
//     barrier three k-steps early, so that one of them always has MFMAs to issue (TW_LOADERS / TW_STAGGER below).
//     (NW = 4, round 2's one wave per SIMD with 7 x 4 tiles, is kept for same-box A/Bs: FPC_TOWER_WAVES=4.)
//   * the MFMA is issued as W x X^T: a lane then owns ONE grid position and four consecutive output
//     channels per accumulator -- one interior predicate and one 8-byte LDS write per tile.
//   * only tiles that contain interior squares are computed (14 of the 16 grid rows at 14x14).
//   * LDS image layout  [row/8][chunk/2][chunk%2][row%8] x 16 B  (row = grid position or output
//     channel, chunk = 8 input channels): a 16x16x32 operand fragment is one ds_read_b128 whose four
//     16-lane groups each cover all 16 slots of the 256-byte bank row for ANY row shift, so the nine
//     taps read row-shifted views of one image without bank conflicts.
//   * weights: one tap = [128 cout][128 cin] = 32 KiB, stored in HBM already in LDS-image order and
//     brought in by LDS-DMA (global_load_lds_dwordx4, no VGPR hop, no ds_write) into a 3-slot ring;
//     one barrier per tap publishes it.
//   * the conv zero padding is the image's zero border; the bottom border row is aliased onto the top
//     one (row index mod NR), which makes room for the ring in the 160 KiB of LDS.
//   * the instruction stream is kept lean: fragment addresses are one VGPR + immediates (14x14: every row
//     tile but the last is a constant 4 KiB apart), the bias enters as the accumulators' initial value,
//     ReLU runs on packed 16-bit pairs, the DMA uses the SGPR-base form and (streamed form) immediates.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fpc_tree_kernels.h"

namespace fpc {

typedef __attribute__((ext_vector_type(8))) __bf16 t_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 t_f16x8;
typedef __attribute__((ext_vector_type(4))) float t_f32x4;
typedef __attribute__((ext_vector_type(2))) float t_f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t t_u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t t_u32x2;
typedef __attribute__((ext_vector_type(2))) short t_i16x2;

template <int DT>
struct M16;
template <>
struct M16<0> {  // bf16
  static __device__ __forceinline__ t_f32x4 mfma(t_u32x4 a, t_u32x4 b, t_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(t_bf16x8, a), __builtin_bit_cast(t_bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {      // v_cvt_pk_bf16_f32, round to nearest even
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((t_f32x2{lo, hi}), b2));
  }
  static __device__ __forceinline__ float lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
  static __device__ __forceinline__ float hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
};
template <>
struct M16<1> {  // fp16
  static __device__ __forceinline__ t_f32x4 mfma(t_u32x4 a, t_u32x4 b, t_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(t_f16x8, a), __builtin_bit_cast(t_f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {      // v_cvt_pk_f16_f32, round to nearest even
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((t_f32x2{lo, hi}), h2));
  }
  static __device__ __forceinline__ float lo(uint32_t p) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(p & 0xffffu)); }
  static __device__ __forceinline__ float hi(uint32_t p) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(p >> 16)); }
};
// ReLU of a packed pair of bf16 or fp16 values: as signed 16-bit integers every negative float (and -0)
// is negative and every positive float keeps its order, so max(x, 0) per half is the ReLU (v_pk_max_i16)
__device__ __forceinline__ uint32_t tw_relu2(uint32_t p) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(t_i16x2, p), (t_i16x2{0, 0})));
}

// byte offset of 16-byte chunk c of row `row` in an image whose rows have CH chunks
__host__ __device__ constexpr int tw_lay(int CH, int row, int c) {
  return (row >> 3) * (CH * 128) + (c >> 1) * 256 + (c & 1) * 128 + (row & 7) * 16;
}

// LDS map (160 KiB exactly).  The 4 KiB in front of the image also absorb the one fragment row that a
// border position of the first row tile reads at index -1 (its output is never stored).
constexpr int TW_THREADS = 256;
constexpr int TW_BIAS = 0;                       // [2][256] f32, filled by LDS-DMA (128 used per layer)
constexpr int TW_DUMMY = 2048;                   // 512 B: where non-interior lanes send their epilogue writes
constexpr int TW_VRED = 2560;                    // [4] f32 value-head partials
constexpr int TW_BOARD = 2624;                   // the game's leaf board (288 B)
constexpr int TW_PACE = 2944;                    // k_towerw: [8] k-step counters, one per wave (pacing of the two waves of a SIMD)
constexpr int TW_IMG0 = 4096;
constexpr int TW_IMG = 61440;                    // 240 rows x 256 B (14x14: 15 of the 16 grid rows)
constexpr int TW_TAP = 32768;                    // [128 cout][128 cin] x 2 B
constexpr int TW_RING = TW_IMG0 + TW_IMG;        // 3 slots
constexpr int TW_LDS = TW_RING + 3 * TW_TAP;     // 163840
constexpr int TW_STEM_TAP = 8192;                // [128 cout][32 cin] x 2 B
constexpr int TW_ENC = TW_RING + 9 * TW_STEM_TAP;   // the stem's input image (<= 15 KiB) behind its 72 KiB of weights
static_assert(TW_LDS == 163840, "k_tower uses the whole LDS of a CU");

struct TowerArgs {
  // input: either the search's leaf boards (fused GetEncodedStates, board.cpp:305-356) or an encoded grid
  const fpc_board *boards;
  const int *leaf_slot, *leaf_turn;
  int board_stride;
  uint16_t one16;
  const uint16_t *in16;          // [game][PP rows][32] 16-bit (boards == null)
  const unsigned char *Wstem;    // [9][8 KiB]  LDS-image order
  const float *bstem;            // [128]
  const unsigned char *Wt;       // [(L + 2) * 9][32 KiB]  LDS-image order; layer L = value conv, L + 1 = policy conv
  const float *bt;               // [L + 2][256] (128 used)
  uint16_t *xfc;                 // policy-Linear input [game][Kp]: index q * A_ch + ch
  const float *vw;               // value Linear weights [R*R][32]
  float *value;                  // [n_games] = tanh(vb + sum relu(vconv) * vw)
  float vb;
  int L, P, R, PP, NR, T0, n_games, Kp, A_ch;
  int rules;                     // FPC_RULES_* (rotation / plane numbering of the fused encode)
  unsigned long long *stamps;    // diagnostic builds only (-DTW_STAMPS): [2 taps][8 waves][16] s_memtime values of block 0; else null
};

// LDS-DMA: `PIECES` consecutive 1-KiB pieces (64 lanes x 16 B each) global -> LDS with no VGPR
// destination, source = SGPR base + per-lane byte offset.  Issued from inline asm on purpose: hipcc
// would otherwise order every later LDS access behind a vmcnt(0) for the pending LDS write.  All waits
// for these pieces are the explicit `s_waitcnt vmcnt(0)` statements in k_tower (each followed by the
// barrier that publishes the data to the other waves).  M0 carries the wave-uniform LDS byte address
// and is restored inside the same statement.
__device__ __forceinline__ void tw_dma_8k(const unsigned char *gsrc_uniform, uint32_t lane_off, uint32_t lds_addr_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "+v"(lane_off) : "s"(gsrc_uniform), "s"(lds_addr_uniform) : "memory", "scc");
}
__device__ __forceinline__ void tw_dma_4k(const unsigned char *gsrc_uniform, uint32_t lane_off, uint32_t lds_addr_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x400\n\tv_add_u32 %1, 0x400, %1\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "+v"(lane_off) : "s"(gsrc_uniform), "s"(lds_addr_uniform) : "memory", "scc");
}
// one 1-KiB piece at immediate offset OFF from the wave's stream base: M0 (set once per tap by tw_set_m0) + OFF is
// the LDS destination and gsrc + OFF the source -- the instruction's immediate moves BOTH addresses
// (tools/micro/dma_imm_check.cpp).  No M0 write here on purpose: an s_mov to M0 waits for the wave's LDS reads
// in flight (~20 cycles each beside a fragment stream, tools/micro/dma_stagger.cpp), the piece itself costs ~6.
template <int OFF>
__device__ __forceinline__ void tw_dma_piece(const unsigned char *gsrc_uniform, uint32_t lane_off) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(lane_off), "s"(gsrc_uniform), "i"(OFF) : "memory");
}
__device__ __forceinline__ void tw_set_m0(uint32_t lds_addr_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(lds_addr_uniform) : "memory");
}
__device__ __forceinline__ void tw_dma_256(const unsigned char *gsrc_uniform, uint32_t lane_off, uint32_t lds_addr_uniform) {   // 64 x 4 B
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_addr_uniform) : "memory");
}

// FAST: the 14x14 geometry (grid pitch 16 = tile height), where the row tiles of a wave are a constant
// 4 KiB apart in the image and only the last one can cross the aliased bottom border.
// diagnostic builds (-DTW_STAMPS=<tap index>; tools/tower_stamps.py): s_memtime at nine points of two consecutive taps, block 0
#ifdef TW_STAMPS
#define TW_STAMP(GT, I)                                                                                       \
  do {                                                                                                        \
    if (g.stamps && blockIdx.x == 0 && ((GT) == TW_STAMPS || (GT) == TW_STAMPS + 1)) {                        \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                             \
      if (lane == 0) g.stamps[(((GT) - TW_STAMPS) * 8 + wave) * 16 + (I)] = t_;                               \
    }                                                                                                         \
  } while (0)
#else
#define TW_STAMP(GT, I) do {} while (0)
#endif
// The weight stream (32 KiB per tap = 32 LDS-DMA pieces of 1 KiB per CU).
// TW_AHEAD 1 (round 2): the tap two ahead... of the tap being read is issued as ONE burst behind each tap barrier;
//   the CU's vector-memory path takes 1 KiB per 16 cycles, so the burst holds its waves ~520 cycles per tap
//   (s_memtime stamps, tools/tower_stamps.py).
// TW_AHEAD 2: the stream runs one tap further ahead (the slot of the tap whose last fragments went into registers
//   before the barrier is free) and its pieces go out BETWEEN the MFMAs of the four k-steps behind the barrier,
//   2 * LOAD... per k-step.  What makes that cheap is that M0 is written once per tap (right behind the barrier,
//   no LDS read in flight) and every piece addresses its destination through the instruction's immediate:
//   an s_mov to M0 in the middle of the fragment stream waits for the reads in flight (~20 cycles; with it the
//   spread stream was 6-10 % SLOWER than the burst).  The wait in front of a barrier leaves exactly the youngest
//   tap's pieces in flight (`s_waitcnt vmcnt` retires in issue order).  The stream runs 3 taps past the end of the
//   weights (padding allocated by the host) so that no k-step carries a branch.
// TW_LOADERS (8-wave form): 0 = every wave carries 1 piece per k-step, 1 = waves 0-3 carry 2, waves 4-7 none
//   (code specialised per half: no branch in the k-steps).
#ifndef TW_AHEAD
#define TW_AHEAD 2
#endif
#ifndef TW_LOADERS
#define TW_LOADERS 1
#endif
// TW_STAGGER = S (8-wave form with TW_LOADERS 1): the waves that carry no DMA (4-7) pass every tap barrier S k-steps
// EARLIER in their k-step sequence than the loader waves (S = 3: in front of a tap's k-step 0 instead of behind its
// k-step 2).  A barrier releases all waves at the same instant; what the stagger changes is where in its program each
// wave of a SIMD pair then stands: the loader goes into its DMA burst and address arithmetic while its partner runs
// four k-steps of MFMAs, instead of both standing in the same transition.  At a layer's end the early half still has
// S + 1 k-steps to go while the loader half converts its accumulators -- legal because the last tap looks down-right
// and therefore reads only image rows of the late half's own tiles and below.  Needs the burst form of the stream
// (behind barrier t the staggered waves still read slot t, so tap t + 3 cannot go there yet).
// Same-box A/B, k_tower per launch, logits bit-identical: S = 0 / 1 / 2 / 3: 0.2874 / 0.2771 / 0.2757 / 0.2653 ms.
// On top of S = 3, measured and not kept: static s_setprio for either half (0 ... +1.3 %), the burst behind k-step 3
// instead of behind the barrier (+6 %), and the halves not meeting between layers (the loader half waits for its own
// four epilogues on an LDS counter and goes on; correct, bit-identical, +0.5 %), and the staggered partner carrying
// 2 or 4 of its pair's 8 pieces one k-step behind its barrier (+8 % / +7 %).
#ifndef TW_STAGGER
#define TW_STAGGER 3
#endif
// TIMING-ONLY diagnostic builds (-DTW_STRIP=<bits>; results are wrong on purpose): take the tower apart the way
// tools/micro/fc_stream.cpp took the Linear apart.  1: no MFMAs in the k-steps; 2: no fragment reads in the k-steps;
// 4: no weight DMA; 8: no tap barriers; 16: no epilogue conversions / stores; 32: return behind the stem; 64: no copy-out
// of the policy rows; 128 (fpc_nn.h): no policy Linear behind the tower (its power draw depends on what the tower wrote).
// The product build defines nothing.
#ifndef TW_STRIP
#define TW_STRIP 0
#endif
// k_towerc: how many of the 13 row tiles the loader half (waves 0-3, which also carries the weight DMA) owns
#ifndef TWC_LOADER_TILES
#define TWC_LOADER_TILES 6
#endif

// NW: waves per workgroup.  4 = one wave per SIMD, wave (wm, wn) owns MT row tiles x 4 column tiles (64 output
// channels).  8 = two waves per SIMD, each with half the register file: wave (wm, wn) owns MT row tiles x 2 column
// tiles (32 channels).  The 8-wave form pays 18 instead of 11 fragment reads per SIMD and k-step, but what one
// wave of a SIMD cannot issue while it waits -- for an LDS-DMA piece to leave the issue port (~60 cycles), for a
// fragment, at the epilogue's conversions -- the other one's MFMAs fill: with one wave per SIMD those costs were
// serial with the MFMA stream.  Every output element sees the same MFMAs on the same operands in
// the same order in both forms, so the logits are bit-identical.
// LOAD: weight-DMA pieces this wave carries per k-step (0, 1 or 2), i.e. 4 * LOAD KiB of every tap
// CMP (round 5; 14x14 only, NW = 8 with loader / stagger roles): the COMPACT image of k_towerw in this kernel's skeleton --
// image row 16 + p is square p = 14 i + j of the board, no border columns, 16 zero rows in front and 28 behind.  A tap is
// the row shift 14 dy + dx; where j + dx leaves the board (column 0 under dx = -1, column 13 under dx = +1) the lane reads
// a zero row of the same LDS bank instead (one v_cndmask per row tile and tap on a compile-time lane mask).  13 row tiles
// of 16 squares instead of 14 grid rows: the loader half (waves 0-3, which also carries the weight DMA) owns tiles 0..5,
// the staggered half tiles 6..12 -- 7 % fewer MFMAs and fragment reads per layer, same MFMAs per output element in the
// same order (bit-identical logits; the value head sums the same terms in another order).
template <int DT, int MT, bool FAST, int NW, int LOAD, bool CMP = false>
__device__ __forceinline__ void tw_body(const TowerArgs &g) {
  static_assert(!CMP || (!FAST && NW == 8 && TW_LOADERS == 1 && MT == (LOAD == 2 ? TWC_LOADER_TILES : 13 - TWC_LOADER_TILES)), "compact form: 8 waves, 13 row tiles");
  constexpr int CR = 14, CZ = 16;                   // CMP: board side, zero rows in front of square 0
  constexpr int TB0 = LOAD == 2 ? 0 : TWC_LOADER_TILES;   // CMP: first row tile of this wave half
  constexpr int NT = NW * 64;                      // threads
  constexpr int WN = NW / 2;                       // waves along the output channels
  constexpr int CT = 8 / WN;                       // column tiles (16 channels) per wave: 4 or 2
  constexpr int PIECES = 4 * LOAD;                 // 1-KiB pieces of a tap this wave brings in
  constexpr int DMA_PER_WAVE = PIECES * 1024;
  constexpr int MID = PIECES == 8 ? 4096 : 0;      // the stream base sits in the middle of an 8-KiB share (immediates reach -4096 .. 4095)
  constexpr bool STREAM = TW_AHEAD == 2 && !(TW_STAGGER != 0 && NW == 8 && TW_LOADERS == 1);   // pieces between the MFMAs, two taps ahead
  constexpr int STAG = (NW == 8 && TW_LOADERS == 1 && LOAD == 0) ? TW_STAGGER : 0;        // k-steps this wave passes the tap barrier early (0..3)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *const img = smem + TW_IMG0;
  unsigned char *const ring = smem + TW_RING;
  const float *const biasbuf = reinterpret_cast<const float *>(smem + TW_BIAS);
  float *const vred = reinterpret_cast<float *>(smem + TW_VRED);
  fpc_board *const lboard = reinterpret_cast<fpc_board *>(smem + TW_BOARD);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, lq = lane >> 4;
  const int game = blockIdx.x;
  const int P = g.P, NR = g.NR;

  // a game that has left the search (Q5): nothing to evaluate
  int slot = 0, rot_k = 0;
  if (g.boards) {
    slot = g.leaf_slot[game];
    if (slot < 0) return;
    rot_k = first_leaf_turn(g.leaf_slot, g.leaf_turn, g.n_games);
  }

  // ---- per-lane geometry -------------------------------------------------------------------------
  const int rbase = CMP ? CZ + TB0 * 16 + li : (g.T0 + wm * MT) * 16 + li;       // grid position (image row) of this lane in its first row tile
  uint32_t inmask = 0;                                // bit mt: the lane's position in row tile mt is an interior square
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = rbase + 16 * mt;
    if (CMP) {
      if (r - CZ < CR * CR) inmask |= 1u << mt;       // only the board's last tile (196 = 12 * 16 + 4) has lanes past the last square
    } else {
      const int pi = r / P, pj = r - pi * P;
      if (r < g.PP && pi >= 1 && pi <= g.R && pj >= 1 && pj <= g.R) inmask |= 1u << mt;
    }
  }
  // CMP: the lanes of row tile mt whose square sits in board column 0 (dx = -1 reads off the board) / 13 (dx = +1): compile-time
  auto colmask = [](int mt, int dx) -> uint64_t {
    uint32_t m16 = 0;
    for (int i = 0; i < 16; ++i)
      if (dx != 0 && ((TB0 + mt) * 16 + i) % CR == (dx < 0 ? 0 : CR - 1)) m16 |= 1u << i;
    const uint32_t m32 = m16 * 0x10001u;
    return ((uint64_t)m32 << 32) | m32;
  };
  auto lane_select = [](uint32_t a, uint32_t b, uint64_t m) -> uint32_t {   // m's lanes: b; the others: a
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
  };
  const int bq = (lq >> 1) * 256 + (lq & 1) * 128;                 // chunk-within-k-step part of a fragment address
  const int apart = (li >> 3) * 2048 + (li & 7) * 16 + bq;         // weight fragment: row li of a 16-row tile (CH = 16)
  const int apart_s = (li >> 3) * 512 + (li & 7) * 16 + bq;        // same for the stem's 4-chunk rows
  const int cb64 = wn * (CT * 16);                                 // first output channel of this wave (conv layers)
  const int cb16 = (wn & 1) * 16;                                  // value conv: 32 live channels, 16 per wave; NW == 8: the waves wn >= 2 sit it out
  const bool vactive = wn < 2;

  // ---- zero the image, build the stem's input image, fetch the stem's weights -----------------------
  for (int c = tid; c < (TW_IMG0 + TW_IMG) / 16; c += NT) reinterpret_cast<t_u32x4 *>(smem)[c] = t_u32x4{0u, 0u, 0u, 0u};
  unsigned char *const enc = smem + TW_ENC;
  const int enc_bytes = CMP ? 240 * 64 : ((NR + 7) >> 3) * 512;
  for (int c = tid; c < enc_bytes / 16; c += NT) reinterpret_cast<t_u32x4 *>(enc)[c] = t_u32x4{0u, 0u, 0u, 0u};
  {  // stem weights: 72 KiB by plain 16-byte copies (once per launch)
    const t_u32x4 *src = reinterpret_cast<const t_u32x4 *>(g.Wstem);
    t_u32x4 *dst = reinterpret_cast<t_u32x4 *>(ring);
    for (int c = tid; c < 9 * TW_STEM_TAP / 16; c += NT) dst[c] = src[c];
  }
  t_f32x4 bst[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bst[ct] = *reinterpret_cast<const t_f32x4 *>(g.bstem + cb64 + ct * 16 + 4 * lq);
  __syncthreads();
  if (g.boards) {
    constexpr int WPB = (int)(sizeof(fpc_board) / 4);
    if (tid < WPB) reinterpret_cast<uint32_t *>(lboard)[tid] =
        reinterpret_cast<const uint32_t *>(g.boards + (size_t)game * g.board_stride + slot)[tid];
    __syncthreads();
    // GetEncodedStates: plane = 6*((colour - turn) & 3) + type - 1, -1 wrapping to 23 (Q7); the whole
    // batch is rotated by the turn of the first live leaf (Q6) -- unless the non-strict rules say otherwise
    if (g.rules & FPC_RULES_ROTATION) rot_k = lboard->turn;
    for (int r = tid; r < (CMP ? CR * CR : g.PP); r += NT) {
      const int pi = CMP ? r / CR + 1 : r / P, pj = CMP ? r % CR + 1 : r - pi * P;      // 1-based square of grid position / square r
      if (pi < 1 || pi > g.R || pj < 1 || pj > g.R) continue;
      const uint8_t p = lboard->sq[rot90_src(g.R, rot_k, pi - 1, pj - 1)];
      if (!present(p)) continue;
      const int plane = piece_plane(p, lboard->turn, g.rules);
      *reinterpret_cast<uint16_t *>(enc + tw_lay(4, CMP ? CZ + r : r, plane >> 3) + (plane & 7) * 2) = g.one16;
    }
  } else if (CMP) {
    const uint16_t *src = g.in16 + (size_t)game * g.PP * 32;      // bordered grid [(R + 2)^2][32]
    for (int c = tid; c < CR * CR * 4; c += NT) {
      const int q = c >> 2, j = c & 3;
      *reinterpret_cast<t_u32x4 *>(enc + tw_lay(4, CZ + q, j)) =
          *reinterpret_cast<const t_u32x4 *>(src + ((size_t)(q / CR + 1) * P + q % CR + 1) * 32 + j * 8);
    }
  } else {
    const uint16_t *src = g.in16 + (size_t)game * g.PP * 32;
    for (int c = tid; c < NR * 4; c += NT) {
      const int r = c >> 2, j = c & 3;
      *reinterpret_cast<t_u32x4 *>(enc + tw_lay(4, r, j)) = *reinterpret_cast<const t_u32x4 *>(src + (size_t)r * 32 + j * 8);
    }
  }
  __syncthreads();

  // Accumulators are never zeroed: the first MFMA of every layer takes the layer's bias as C
  // (a zero-initialised accumulator carried into the tap loop makes hipcc rotate 100+ registers per tap).
  t_f32x4 acc[MT][CT];
  t_u32x2 res[MT][CT];           // residual x_l of this lane's outputs, packed 16-bit channel pairs

  // image row of this lane for row tile mt under a tap shift, bottom border aliased onto the top one
  auto brow = [&](int mt, int shift) -> int {
    int r = rbase + 16 * mt + shift;
    r = r < 0 ? r + NR : r;
    r = r >= NR ? r - NR : r;
    return r;
  };

  // ---- stem: conv3x3(24 -> 128) on the 32-channel input image, one 32-deep k-step per tap -----------
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int shift = (tap / 3 - 1) * (CMP ? CR : P) + (tap % 3 - 1);
    t_u32x4 fa[CT], fb[MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      fa[ct] = *reinterpret_cast<const t_u32x4 *>(ring + tap * TW_STEM_TAP + (cb64 + ct * 16) * 64 + apart_s);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = CMP ? (int)lane_select((uint32_t)(rbase + 16 * mt + shift), (uint32_t)((rbase + shift) & 7), colmask(mt, tap % 3 - 1)) : brow(mt, shift);
      fb[mt] = *reinterpret_cast<const t_u32x4 *>(enc + (r >> 3) * 512 + (r & 7) * 16 + bq);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = M16<DT>::mfma(fa[ct], fb[mt], tap == 0 ? bst[ct] : acc[mt][ct]);
  }
  __syncthreads();               // every wave is done with the stem's weights and input image

  // ---- weight ring ------------------------------------------------------------------------------------
  const int total = (g.L + 2) * 9;
  const uint32_t dma_lane = (uint32_t)lane * 16u;
  const int lw = LOAD == 2 ? (wave & 3) : wave;    // index among the waves that carry the stream
  auto issue_bias = [&](int T) {                   // with a layer's first tap: its biases (waves 0-3)
    if (T % 9 == 0 && T < total && wave < 4) {
      const int l = T / 9;
      tw_dma_256(reinterpret_cast<const unsigned char *>(g.bt + (size_t)l * 256 + wave * 64), (uint32_t)lane * 4u,
                 (uint32_t)__builtin_amdgcn_readfirstlane(TW_BIAS + ((l & 1) * 256 + wave * 64) * 4));
    }
  };
  auto issue_tap = [&](int T) {   // the whole tap T -> ring slot T % 3 in one burst
    issue_bias(T);
    if (LOAD == 0 || (TW_STRIP & 4)) return;
    const unsigned char *src = g.Wt + (size_t)T * TW_TAP + lw * DMA_PER_WAVE;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane(TW_RING + (T % 3) * TW_TAP + lw * DMA_PER_WAVE);   // smem starts at LDS byte 0
    if (PIECES == 8) tw_dma_8k(src, dma_lane, dst);
    else tw_dma_4k(src, dma_lane, dst);
  };
  // TW_AHEAD 2: the stream.  open_tap(T) right behind a tap barrier (no LDS read in flight): M0 and the source
  // base of tap T; piece<I>() anywhere after that.
  const unsigned char *dsrc = g.Wt;
  auto open_tap = [&](int T) {
    issue_bias(T);                                 // saves / restores M0 itself, so: before M0 is set
    if (LOAD == 0) return;
    dsrc = g.Wt + (size_t)T * TW_TAP + lw * DMA_PER_WAVE + MID;
    tw_set_m0((uint32_t)__builtin_amdgcn_readfirstlane(TW_RING + (T % 3) * TW_TAP + lw * DMA_PER_WAVE + MID));
  };
  auto piece = [&](auto i_c) {
    constexpr int I = decltype(i_c)::value;
    if (LOAD != 0) tw_dma_piece<I * 1024 - MID>(dsrc, dma_lane);
  };
  issue_tap(0);
  issue_tap(1);
  if (STREAM) {                                    // first quarter of tap 2; the k-steps carry on from there
    open_tap(2);
    piece(std::integral_constant<int, 0>{});
    if (LOAD == 2) piece(std::integral_constant<int, 1>{});
  }

  // epilogue: the accumulators hold conv + bias; (+ residual, f32); 16-bit; ReLU on the packed pairs;
  // written IN PLACE into the image at interior squares (4 consecutive channels = one 8-byte write; the
  // other lanes write to a dummy strip, no branch)
  unsigned char *const dummy = smem + TW_DUMMY + lane * 8;
  unsigned char *const wbase = img + (rbase >> 3) * 2048 + (rbase & 7) * 16 + wn * (CT * 256) + (lane >> 5) * 128 + ((lane >> 4) & 1) * 8;
  auto epilogue = [&](auto res_c) {
    constexpr int RES = decltype(res_c)::value;      // 0: plain; 1: keep as residual (stem); 2: add the residual, keep
    if (TW_STRIP & 16) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      unsigned char *dst = ((inmask >> mt) & 1u) ? wbase + mt * 4096 : dummy;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        t_f32x4 v = acc[mt][ct];
        if (RES == 2) {
          v[0] += M16<DT>::lo(res[mt][ct][0]); v[1] += M16<DT>::hi(res[mt][ct][0]);
          v[2] += M16<DT>::lo(res[mt][ct][1]); v[3] += M16<DT>::hi(res[mt][ct][1]);
        }
        const t_u32x2 pk = t_u32x2{tw_relu2(M16<DT>::pack2(v[0], v[1])), tw_relu2(M16<DT>::pack2(v[2], v[3]))};
        if (RES != 0) res[mt][ct] = pk;
        *reinterpret_cast<t_u32x2 *>(((inmask >> mt) & 1u) ? dst + ct * 256 : dst) = pk;
      }
      __builtin_amdgcn_sched_barrier(0);     // one row tile at a time: keeps the accumulator reads from piling up in VGPRs
    }
  };
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 1> c1{};
  const std::integral_constant<int, 2> c2{};
  const std::integral_constant<int, 3> c3{};
  epilogue(c1);                  // stem: x_0 = relu(conv + b)
  if (TW_STRIP & 32) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }      // timing only: prologue + stem alone
  if (STREAM) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES + LOAD) : "memory");   // tap 1 and the first quarter of tap 2 stay in flight
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();               // x_0 complete, tap 0 and layer 0's biases landed

  // ---- the tower: L residual convs + value conv + policy conv, 9 taps x 4 k-steps each -------------
  // Fragments are double-buffered one k-step ahead of the MFMAs that consume them; every load below is
  // unconditional and lands in a statically named register set.  The k-step loop is rotated by one:
  // a layer opens with (tap 0, k-step 0) on C = bias, and the tap loop body is k-steps 1, 2, 3 of its
  // tap followed by k-step 0 of the next one.
  t_u32x4 fa[2][CT], fb[2][MT];
  int gt = 0;                                      // running tap index over all layers (ring slot = gt % 3)
  const unsigned char *wslot;                      // this wave's weight rows in the current tap's ring slot
  const unsigned char *bbase;                      // FAST: this lane's image row in its first row tile under the current shift
  const unsigned char *blast;                      // FAST: the same for the last row tile (the only one that can cross the border alias)
  int boff[FAST ? 1 : MT];                         // !FAST: image byte offset per row tile
  auto set_tap = [&](int tap, int cb) {
    wslot = ring + (gt % 3) * TW_TAP + cb * 256 + apart;
    const int shift = (tap / 3 - 1) * P + (tap % 3 - 1);
    if (FAST) {
      const int r = rbase + shift;                 // >= -1; row -1 lands in the 4 KiB in front of the image
      bbase = img + (r >> 3) * 2048 + (r & 7) * 16 + bq;
      int rl = r + 16 * (MT - 1);
      rl = rl >= NR ? rl - NR : rl;
      blast = img + (rl >> 3) * 2048 + (rl & 7) * 16 + bq;
    } else if (CMP) {
      const int dx = tap % 3 - 1;                    // wave-uniform
      const int r = rbase + (tap / 3 - 1) * CR + dx; // >= 16 - 15
      const uint32_t zoff = (r & 7) * 16 + bq;       // the zero row with this row's low three bits: same LDS bank
      const uint32_t b0 = (r >> 3) * 2048 + zoff;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const uint64_t m = dx < 0 ? colmask(mt, -1) : dx > 0 ? colmask(mt, 1) : 0ull;
        boff[mt] = (int)lane_select(b0 + mt * 4096, zoff, m);
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int r = brow(mt, shift);
        boff[mt] = (r >> 3) * 2048 + (r & 7) * 16 + bq;
      }
    }
  };
  auto bptr = [&](int mt) -> const unsigned char * {
    if (FAST) return mt == MT - 1 ? blast : bbase + mt * 4096;
    return img + boff[mt];
  };
  // One k-step: the MFMAs on buffer B, with the fragment reads of k-step KSN (of the current wslot / image
  // rows) into the other buffer spread between them -- order pinned, hipcc would otherwise sink every
  // read down to just in front of its first use and expose the LDS latency.
  // MODE 0: 64 output channels per wave (4 column tiles); MODE 1 (value conv, 32 live channels): 16 per wave.
  // BIAS: this is the layer's first k-step: C = the layer's bias instead of the accumulators.
  // Q: the k-step's position behind the tap barrier (k-step 3 = 0, next tap's k-steps 0, 1, 2 = 1, 2, 3): with
  // TW_AHEAD 2 it carries pieces LOAD * Q ... of the open tap between its MFMAs.
  auto kstep = [&](auto mode_c, auto buf_c, auto ksn_c, auto bias_c, const t_f32x4 *b4, auto q_c) {
    constexpr int MODE = decltype(mode_c)::value, B = decltype(buf_c)::value, KSN = decltype(ksn_c)::value;
    constexpr bool BIAS = decltype(bias_c)::value != 0;
    constexpr int N = B ^ 1, Q = decltype(q_c)::value;
    if (!(TW_STRIP & 2)) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) fa[N][ct] = *reinterpret_cast<const t_u32x4 *>(wslot + ct * 4096 + KSN * 512);
      fb[N][0] = *reinterpret_cast<const t_u32x4 *>(bptr(0) + KSN * 512);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (TW_STRIP & 1) {
        if (BIAS) {
#pragma unroll
          for (int ct = 0; ct < (MODE == 0 ? CT : 1); ++ct) acc[mt][ct] = b4[ct];
        }
        asm volatile("" : "+v"(fa[B][0]), "+v"(fb[B][mt]));      // the fragments stay "used"
      } else if (MODE == 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = M16<DT>::mfma(fa[B][ct], fb[B][mt], BIAS ? b4[ct] : acc[mt][ct]);
      } else if (NW == 4 || vactive) {             // wave-uniform
        acc[mt][0] = M16<DT>::mfma(fa[B][0], fb[B][mt], BIAS ? b4[0] : acc[mt][0]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (mt + 1 < MT && !(TW_STRIP & 2)) {
        fb[N][mt + 1] = *reinterpret_cast<const t_u32x4 *>(bptr(mt + 1) + KSN * 512);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (STREAM && LOAD != 0) {
        if (mt == (LOAD == 2 ? MT / 3 : MT / 2)) piece(std::integral_constant<int, LOAD * Q>{});
        if (LOAD == 2 && mt == (2 * MT) / 3) piece(std::integral_constant<int, LOAD * Q + 1>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto load_b0 = [&]() {                           // image fragments of k-step 0 of the current tap (after an epilogue)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fb[0][mt] = *reinterpret_cast<const t_u32x4 *>(bptr(mt));
  };
  // One conv layer = 9 taps.  On entry fa[0] / fb[0] hold (tap 0, k-step 0) and wslot / the image row
  // addresses are set for tap 0; on exit the same holds for the NEXT layer (whose first weight row of
  // this wave is cb_next), except that fb[0] was read from the image the epilogue is about to rewrite
  // (load_b0 after it).
  auto run_layer = [&](auto mode_c, const int layer, const int cb, const int cb_next) {
    constexpr int MODE = decltype(mode_c)::value;
    t_f32x4 b4[CT];
    {
      const float *bl = biasbuf + (layer & 1) * 256 + cb + 4 * lq;
#pragma unroll
      for (int ct = 0; ct < (MODE == 0 ? CT : 1); ++ct) b4[ct] = *reinterpret_cast<const t_f32x4 *>(bl + ct * 16);
    }
    if (STAG == 3 && !(TW_STRIP & 8)) __syncthreads();                // (a staggered wave issued no DMA: nothing to wait for)
    kstep(mode_c, c0, c1, c1, b4, c1);             // (tap 0, k-step 0), C = bias
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      if (STAG == 2) {
        TW_STAMP(gt, 3);
        __syncthreads();
        TW_STAMP(gt, 4);
      }
      TW_STAMP(gt, 0);
      kstep(mode_c, c1, c2, c0, b4, c2);           // k-step 1
      TW_STAMP(gt, 1);
      if (STAG == 1) __syncthreads();
      kstep(mode_c, c0, c3, c0, b4, c3);           // k-step 2
      TW_STAMP(gt, 2);
      // tap gt + 1's weights: every wave's pieces of it have landed (STREAM: tap gt + 2's stay in flight), and
      // nobody reads slot gt % 3 any more -- the fragments of this tap's last k-step are in registers -- so
      // (STREAM) tap gt + 3 goes there.  Also: all fragment reads of this tap are complete, so after tap 8 the
      // epilogue may rewrite the image in place (the staggered waves read on behind barrier 8, but only image
      // rows of their own half and below: tap 8 looks down-right).
      if (!STAG) {
        if (STREAM) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TW_STAMP(gt, 3);
        if (!(TW_STRIP & 8)) __syncthreads();
        TW_STAMP(gt, 4);
        if (STREAM) open_tap(gt + 3);
        else if (gt + 2 < total) issue_tap(gt + 2);
        TW_STAMP(gt, 5);
      }
      ++gt;
      set_tap(tap == 8 ? 0 : tap + 1, tap == 8 ? cb_next : cb);
      TW_STAMP(gt - 1, 6);
      kstep(mode_c, c1, c0, c0, b4, c0);           // k-step 3, reading (next tap, k-step 0)
      TW_STAMP(gt - 1, 7);
      if (STAG == 3 && tap < 8 && !(TW_STRIP & 8)) __syncthreads();
      if (tap < 8) kstep(mode_c, c0, c1, c0, b4, c1);  // (next tap, k-step 0)
      TW_STAMP(gt - 1, 8);
    }
  };

  set_tap(0, cb64);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) fa[0][ct] = *reinterpret_cast<const t_u32x4 *>(wslot + ct * 4096);
  load_b0();
  const int nblocks = g.L / 2;
#pragma unroll 1
  for (int blk = 0; blk < nblocks; ++blk) {
    run_layer(c0, 2 * blk, cb64, cb64);                               // conv1 + BN + ReLU
    epilogue(c0);
    __syncthreads();                                                  // conv2's input image is complete
    load_b0();
    run_layer(c0, 2 * blk + 1, cb64, blk + 1 == nblocks ? cb16 : cb64);   // conv2 + BN, + x_l, ReLU
    epilogue(c2);
    __syncthreads();
    load_b0();
  }
  float vpart = 0.f;
  {
    // value head (net.py:28-35): relu(conv + b)[pos][ch] . vw[pos][ch], ch < 24 (weights, biases and vw zero-padded to 32)
    run_layer(c1, g.L, cb16, cb64);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = rbase + 16 * mt;
      const int pi = r / P, pj = r - pi * P;
      const bool in = ((inmask >> mt) & 1u) && (NW == 4 || vactive);
      const int qp = !in ? 0 : CMP ? r - CZ : (pi - 1) * g.R + (pj - 1);
      const t_f32x4 w4 = *reinterpret_cast<const t_f32x4 *>(g.vw + qp * 32 + cb16 + 4 * lq);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[mt][0][j];
        v = v > 0.f ? v : 0.f;
        vpart += in ? v * w4[j] : 0.f;
      }
    }
    // (the value conv leaves the image as it was: fb[0], read during its last k-step, is valid)
  }
  run_layer(c0, g.L + 1, cb64, cb64);                                 // policy conv + BN + ReLU, 16-bit rows in place
  epilogue(c0);
  __syncthreads();

  // ---- heads: policy-conv rows -> Linear input (position-major), value -> tanh ---------------------------
  if (!(TW_STRIP & 64)) {
    const int cpr = g.A_ch / 8;                     // 16-byte chunks per position
    for (int c = tid; c < (CMP ? CR * CR : g.PP) * 16; c += NT) {
      const int j = c & 15;
      if (j >= cpr) continue;
      int r = c >> 4, q = r;
      if (CMP) r += CZ;
      else {
        const int pi = r / P, pj = r - pi * P;
        if (pi < 1 || pi > g.R || pj < 1 || pj > g.R) continue;
        q = (pi - 1) * g.R + (pj - 1);
      }
      *reinterpret_cast<t_u32x4 *>(g.xfc + (size_t)game * g.Kp + (size_t)q * g.A_ch + j * 8) =
          *reinterpret_cast<const t_u32x4 *>(img + tw_lay(16, r, j));
    }
  }
  for (int off = 32; off >= 1; off >>= 1) vpart += __shfl_xor(vpart, off);
  if (lane == 0) vred[wave] = vpart;
  __syncthreads();
  // partials in the 4-wave order (wm, wn & 1); NW == 8: the waves with wn >= 2 hold zero and are not read
  if (tid == 0) g.value[game] = tanhf(g.vb + ((vred[0] + vred[1]) + (vred[WN] + vred[WN + 1])));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stream's tail (pieces past the last tap): nothing may still target LDS at exit
}

// The two wave halves of the 8-wave form may carry different shares of the weight stream (TW_LOADERS); they run the
// same sequence of barriers.
template <int DT, int MT, bool FAST, int NW>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_tower(TowerArgs g) {
  if (NW == 8 && TW_LOADERS == 1) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) tw_body<DT, MT, FAST, NW, 2>(g);
    else tw_body<DT, MT, FAST, NW, 0>(g);
  } else {
    tw_body<DT, MT, FAST, NW, NW == 4 ? 2 : 1>(g);
  }
}

// the compact form at 14x14 (CMP): loaders = row tiles 0 .. TWC_LOADER_TILES - 1, staggered half = the rest of the 13
template <int DT>
__global__ void __launch_bounds__(512, 2) k_towerc(TowerArgs g) {
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4) tw_body<DT, TWC_LOADER_TILES, false, 8, 2, true>(g);
  else tw_body<DT, 13 - TWC_LOADER_TILES, false, 8, 0, true>(g);
}

// weights [taps][128 rows][cin] 16-bit row-major -> per tap the LDS image (tw_lay with cin/8 chunks per row)
__global__ void __launch_bounds__(256) k_tower_prep(const uint16_t *W, unsigned char *out, int taps, int cin) {
  const int CH = cin / 8;
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c >= (long)taps * 128 * CH) return;
  const int tap = (int)(c / (128 * CH)), rem = (int)(c % (128 * CH)), row = rem / CH, j = rem % CH;
  *reinterpret_cast<t_u32x4 *>(out + (size_t)tap * 128 * cin * 2 + tw_lay(CH, row, j)) =
      *reinterpret_cast<const t_u32x4 *>(W + ((size_t)tap * 128 + row) * cin + j * 8);
}

}  // namespace fpc
