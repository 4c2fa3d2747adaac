// fpc_towerw.h -- k_towerw<DT, F, MT, ONEROW> (round 4): the residual tower of net.py:6-63 (stem + 2*Nb residual convs +
// both head convs) in ONE launch, one game per workgroup, activations resident in LDS, for hidden F = 128 or 256 and
// any board of 8..14 squares a side -- on TWO WAVES PER SIMD with the weights going straight from L2 into registers:
// no weight ring in LDS, no LDS-DMA, no barrier inside a layer.
//
// Where it comes from.  k_tower256 (fpc_tower256.h, round 2: one wave per SIMD, 2-slab LDS weight ring) sat at 0.36 of
// the MFMA peak.  Its inner loop was not the problem (hipcc's k-step there is 56 MFMAs + 15 fragment reads, clean);
// what stood between two k-steps was: `s_waitcnt vmcnt(0)`, the workgroup barrier and a burst of four LDS-DMA pieces
// (60-185 cycles of issue port each) 72 times per layer, with nothing to run in their shadow -- ~2 000 cycles per
// k-step for 896 cycles of matrix pipe.  Here:
//   * 8 waves.  F = 256 (and F = 128 at 8x8): ONE wave row, the waves side by side along the output channels, each over
//     all MT = ceil(R^2 / 16) row tiles of 16 SQUARES x F/128 column tiles of 16 channels -- at 14x14: 104 accumulator
//     registers + double-buffered weight fragments (16) + a ring of four image fragments (16) + 13 row offsets, of the
//     256 a wave of a two-waves-per-SIMD kernel may hold.  F = 128 elsewhere: two wave rows x four, wave (wm, wn) = up to
//     MT = ceil(tiles / 2) row tiles x 2 column tiles (one wave row would read one image fragment per MFMA there).
//     Two wave rows pull every weight fragment through the CU's L1 twice: with few row tiles that bounds the k-step.  The residual is
//     NOT among them at F = 256 (RESIN below): carried in registers across a residual block (56 more) it ended up parked
//     in scratch by hipcc, 0.6 GB written per launch.
//   * COMPACT image: image row 16 + p is square p of the board (no border columns); a tap whose column shift leaves
//     the board reads a zero row instead (lane masks built on the scalar unit).  ceil(R^2 / 16) row tiles per layer
//     instead of one per row of the bordered (R + 2)^2 grid: 13 for 14 at 14x14, 4 for 6 at 8x8, 7 for 10 at 10x10.
//   * a wave's weight operand of a k-step is [16 CT cout][32 cin] = CT A fragments; the host stores the weights in
//     FRAGMENT ORDER  [layer][tap][k-step][cout tile of 16][lane][8]  (k_towerw_prep), so one fragment is one perfectly
//     coalesced 1-KiB `global_load_dwordx4` of the wave, issued PD - 1 k-steps ahead of its use (PD = 2 where the
//     registers are full, 4 where few row tiles leave them free and a k-step is shorter than an L2 round trip:
//     tww_depth) and waited for by the counted vmcnt hipcc places in front of the first MFMA that needs it.  The two
//     waves that share a cout range (wm = 0 / 1) fetch the same lines within a few hundred cycles of each other: the
//     second fetch is an L1 hit.  Nothing is shared through LDS, so nothing has to be published: a layer's k-steps run
//     without a single barrier.
//   * PACING: the two waves of a SIMD keep within a tap of each other by giving the one that is behind s_setprio 1
//     (the matrix pipe otherwise serves the older wave first and the younger one runs the end of every layer alone).
//   * two barriers per layer remain: in front of the layer's last k-step (every wave has read its last image
//     fragments: the epilogues may rewrite the image in place) and behind the epilogues.
//   * every convolution runs the same MFMAs on the same operands in the same order as in k_tower256 / k_tower.  At
//     F = 128 the logits are BIT-IDENTICAL to k_tower's (the value head sums the same terms in another order, the compact
//     image dealing the squares to other lanes).  At F = 256 the residual enters conv2's accumulators in front of its
//     MFMAs instead of behind them: the same terms in another order, outputs within 2e-4 (fp16) of k_tower256's
//     (tests/test_nn_gpu.py); what counts is the 1e-3 against the reference's fp32 network, unchanged.
//   * LDS: 4 KiB front strip + 240 image rows x 2F bytes = 124 KiB (F = 256, one workgroup per CU) / 64 KiB (F = 128,
//     two); hidden = 256 runs as ONE launch at every board size, incl. the reference's shipped ResNet(15, 256) on its
//     8x8 board (alphazero.py:288); hidden = 128 runs here on every board but 14x14 (k_tower's home: fpc_nn.h).
// What the stamped timeline (tools/towerw_stamps.py, -DTWW_STAMPS) showed on the way, ResNet(20,256) 14x14, cycles per
// layer: k-steps 75 k + last k-step 1.2 k + epilogue 3 k (bordered grid, 14 tiles); the compact image alone made it
// SLOWER (75 k + 4.4 k + 13 k): hipcc turned the epilogue's per-lane dummy-address select into MT x CT address
// registers parked in scratch and a scratch-load wait in front of every store -- one lane address + immediates +
// an EXEC mask: 1.7 k; pacing: 75 k -> 71 k.  An s_load prefetch of the stream three k-steps ahead through the scalar
// cache bought nothing (the 32 workgroups of an XCD walk the stream together; a slab's first touch is not what a
// k-step waits for), splitting the row tiles 7 + 6 costs nothing against 7 + 7.
// Then (same day): the image fragments as a ring of four (RING: 12 registers) 1.845 -> 1.794 ms, and the residual out
// of the registers altogether (RESIN) 1.772 -> 1.684 ms, no scratch access left inside a residual block.
// One wave row instead of two: 1.72 -> 1.65 ms, and 0.56 -> 0.36 ms for ResNet(15,256) on the 8x8 board.
// Measured (same box, stage timers of bench.py): ResNet(20,256) 14x14, 256 leaves: 1.65 ms per launch against 1.94 for
// round 4's first (bordered) form and k_tower256's 2.74; ResNet(15,256) 8x8, 100 leaves: 0.36 ms against 0.62;
// ResNet(10,128) 8x8 / 10x10 / 13x13: 0.120 / 0.171 / 0.242 ms against k_tower's 0.147 / 0.208 / 0.285.
#pragma once
#include "fpc_tower.h"

namespace fpc {

constexpr int TWW_THREADS = 512;
constexpr int TWW_IMG0 = 4096;                   // [0, 4096): dummy strip, value partials, leaf board (k_tower's offsets)
// diagnostic builds (-DTWW_STAMPS=<layer index>; tools/towerw_stamps.py): s_memtime (shader cycles) at 15 points of one conv
// layer, every wave of blocks 0 and 131, parked in LDS and copied out at the end
#ifdef TWW_STAMPS
#define TWW_STAMP(LAYER, I)                                                                                   \
  do {                                                                                                        \
    if ((LAYER) == TWW_STAMPS) {                                                                              \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                             \
      if (lane == 0) reinterpret_cast<unsigned long long *>(smem + 3072)[wave * 16 + (I)] = t_;               \
    }                                                                                                         \
  } while (0)
#else
#define TWW_STAMP(LAYER, I) do {} while (0)
#endif
constexpr int TWW_ZR = 16;                       // zero rows in front of the first square (>= R + 1: every tap of row 0 reads zeros)
constexpr int TWW_ROWS = 240;                    // image rows: 16 zero rows, up to 196 squares, zero rows behind them (a tap reaches 15 rows past a tile)
__host__ __device__ constexpr int tww_lds(int F) { return TWW_IMG0 + TWW_ROWS * F * 2; }
__host__ __device__ constexpr int tww_slab(int F) { return F * 64; }             // one 32-deep k-step of one tap: [F cout][32 cin] x 2 B
// Depth of the weight ring in registers: k-step k reads buffer k mod PD while slab k + PD - 1 is on its way from L2.
// Two buffers (one k-step of cover: its own MFMAs and its partner's) are enough where a k-step is long (14x14 at
// F = 256: 2 x 26 MFMAs outlast an L2 round trip; measured 1.646 ms with two against 1.665 with four); with fewer row
// tiles they are not, and the registers are free: four buffers.
#ifndef FPC_TWW_BUFLOAD
#define FPC_TWW_BUFLOAD 1
#endif
#ifndef FPC_TWW_RESIN
#define FPC_TWW_RESIN 1
#endif
#ifndef FPC_TWW_RING
#define FPC_TWW_RING 1
#endif
#ifndef FPC_TWW_PD
#define FPC_TWW_PD 0
#endif
__host__ __device__ constexpr int tww_depth(int F, int MT, bool onerow) {
  return FPC_TWW_PD ? FPC_TWW_PD : onerow ? (MT >= 12 ? 2 : 4) : (F == 128 || MT <= 4) ? 4 : 2;
}
constexpr int TWW_PAD_SLABS = 3;                 // slabs the weight stream is padded with behind the last layer (deepest ring - 1)
// row tiles (16 squares) of a board and the share of the first wave row (the second gets the rest, at most as many)
__host__ __device__ constexpr int tww_tiles(int R) { return (R * R + 15) / 16; }
__host__ __device__ constexpr int tww_mt(int R) { return (tww_tiles(R) + 1) / 2; }

// COMPACT image (round 4): image row 16 + p holds square p = R * i + j of the board -- no border columns, 16 zero rows
// in front of square 0 and zero rows behind the last one.  A 3x3 tap (dy, dx) is then the row shift R * dy + dx for every
// square EXCEPT where j + dx leaves the board (the shifted row is the neighbouring board row's far end): those lanes
// read one of the zero rows instead (an address select per row tile and tap, `set_tap`; the conv's zero padding in the
// row direction is the zero rows).  Against the bordered (R + 2)^2 grid of k_tower / k_tower256 this computes
// ceil(R^2 / 16) row tiles instead of one per grid row: 13 instead of 14 at 14x14 (-7 % MFMAs), 4 instead of 6 at 8x8,
// 7 instead of 10 at 10x10 -- and every tile is 16 CONSECUTIVE image rows at every board size, so the conflict-free
// fragment reads and the constant tile-to-tile address step hold everywhere (no generic addressing path).
// TowerArgs as for k_tower, with: Wstem = 9 slabs, Wt = (L + 2) * 9 * (F / 32) slabs + TWW_PAD_SLABS of padding, both in
// fragment order; bt = [L + 2][256]; bstem = [F]; in16 (external input) in k_tower's bordered-grid layout.
// Two wave rows x four (ONEROW = false; F = 128 off the 8x8 board): MT = tww_mt(R) row tiles for the waves wm = 0, the
// waves wm = 1 own tww_tiles(R) - MT (MT or MT - 1) of them.
// ONEROW (F = 256 at every board size, F = 128 at 8x8; MT = tww_tiles(R)): all eight waves side by side along the output
// channels, wave = all MT row tiles x F/128 column tiles.  Two wave rows pull every weight fragment through the CU's
// L1 twice, which bounds the k-step when there are few row tiles; at F = 128 with many row tiles one wave row would
// read one image fragment per MFMA instead (LDS-bound).
template <int DT, int F, int MT, bool ONEROW = false>
__global__ void __launch_bounds__(TWW_THREADS, 2) k_towerw(TowerArgs g) {
  constexpr int NT = TWW_THREADS;
  constexpr int WN = ONEROW ? 8 : 4;               // waves along the output channels
  constexpr int CT = F / 16 / WN;                  // column tiles (16 output channels) per wave: 4 or 2 (ONEROW: 2 or 1)
  constexpr int KSN = F / 32;                      // k-steps per tap: 8 or 4
  constexpr int GRP = F * 16;                      // bytes of 8 image rows
  constexpr int SLAB = tww_slab(F);
  constexpr int PD = tww_depth(F, MT, ONEROW);     // weight fragments of PD - 1 k-steps in flight
  // Image fragments: with few row tiles one register quad per tile, reloaded in place for the next k-step behind its
  // MFMAs.  With MT >= 5 a RING of four: the row tiles of a layer are numbered through (c = k-step * MT + tile), tile c
  // sits in quad c mod 4, and behind its MFMAs the quad is reloaded with tile c + 4 -- same k-step, next k-step or next
  // tap.  Twelve registers less at MT = 7 (two wave rows), 36 at MT = 13 (one).
  constexpr bool RING = FPC_TWW_RING != 0 && F == 256 && MT >= 5;     // (F = 128 has the registers: measured 1 % slower there)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *const img = smem + TWW_IMG0;
  float *const vred = reinterpret_cast<float *>(smem + TW_VRED);        // [8]
  fpc_board *const lboard = reinterpret_cast<fpc_board *>(smem + TW_BOARD);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = ONEROW ? 0 : wave >> 2, wn = ONEROW ? wave : wave & 3;   // waves w and w + 4 share a SIMD: 7 + 6 row tiles at 14x14
  const int li = lane & 15, lq = lane >> 4;
  const int game = blockIdx.x;
  const int R = g.R, RR = R * R;

  int slot = 0, rot_k = 0;
  if (g.boards) {
    slot = g.leaf_slot[game];
    if (slot < 0) return;                      // the game has left the search (Q5)
    rot_k = first_leaf_turn(g.leaf_slot, g.leaf_turn, g.n_games);
  }

  // ---- per-lane geometry: this lane's square in row tile mt is p0 + 16 mt ------------------------------------
  const int p0 = wm * MT * 16 + li;
  const bool last_on = ONEROW || wm == 0 || tww_tiles(R) == 2 * MT;   // wave-uniform: does this wave own an MT-th row tile (13 = 7 + 6 at 14x14)
  // Which lanes of row tile mt sit in board column 0 (a tap with dx = -1 reads off the board there) or R - 1 (dx = +1)?
  // Square p0 + 16 mt = tile_base + li is in column 0 iff li == colk[mt] (mod R), colk[mt] := (-tile_base) mod R, and in
  // column R - 1 iff li == colk[mt] - 1 (mod R): the lane mask of a tap is built on the SCALAR unit from MT wave-uniform
  // numbers (at most two of 16 values of li match; the four lq groups repeat them), costs no lane register -- the
  // register file is full (256 per lane at F = 256, MT = 7), and a spilled per-lane mask costs every tap a scratch load
  // and, vmcnt being in-order, a drain of the weight prefetch.
  int colk[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int tb = (wm * MT + mt) * 16;
    colk[mt] = __builtin_amdgcn_readfirstlane((R - tb % R) % R);
  }
  const int bq = (lq >> 1) * 256 + (lq & 1) * 128;
  const int tileW = wn * CT;                          // first 16-cout tile of this wave in a conv layer
  const int tile16 = wn & 1;                          // value conv: 32 live channels = tiles 0, 1; the waves wn >= 2 sit it out
  const bool half_active = wn < 2;                    // value conv (32 live rows) and, at F = 256, policy conv (128 of 256)
  const uint32_t wlane = (uint32_t)lane * 16u;
  const int rbase = TWW_ZR + p0;                      // image row of this lane in its first row tile

  // ---- zero the front strip and the image, build the stem's 32-channel input image inside it -----------
  for (int c = tid; c < tww_lds(F) / 16; c += NT) reinterpret_cast<t_u32x4 *>(smem)[c] = t_u32x4{0u, 0u, 0u, 0u};
  unsigned char *const enc = img;                     // rows x 64 B, same row numbering
  t_f32x4 bst[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bst[ct] = *reinterpret_cast<const t_f32x4 *>(g.bstem + (tileW + ct) * 16 + 4 * lq);
  __syncthreads();
  if (g.boards) {
    constexpr int WPB = (int)(sizeof(fpc_board) / 4);
    if (tid < WPB) reinterpret_cast<uint32_t *>(lboard)[tid] =
        reinterpret_cast<const uint32_t *>(g.boards + (size_t)game * g.board_stride + slot)[tid];
    __syncthreads();
    // GetEncodedStates (board.cpp:305-356): plane = 6*((colour - turn) & 3) + type - 1, -1 wrapping to 23 (Q7); the
    // whole batch is rotated by the turn of the first live leaf (Q6) -- unless the non-strict rules say otherwise
    if (g.rules & FPC_RULES_ROTATION) rot_k = lboard->turn;
    for (int p = tid; p < RR; p += NT) {
      const int pi = p / R, pj = p - pi * R;
      const uint8_t pc = lboard->sq[rot90_src(R, rot_k, pi, pj)];
      if (!present(pc)) continue;
      const int plane = piece_plane(pc, lboard->turn, g.rules);
      *reinterpret_cast<uint16_t *>(enc + tw_lay(4, TWW_ZR + p, plane >> 3) + (plane & 7) * 2) = g.one16;
    }
  } else {
    const uint16_t *src = g.in16 + (size_t)game * g.PP * 32;      // bordered grid [(R + 2)^2][32]
    for (int c = tid; c < RR * 4; c += NT) {
      const int p = c >> 2, j = c & 3;
      const int pi = p / R, pj = p - pi * R;
      *reinterpret_cast<t_u32x4 *>(enc + tw_lay(4, TWW_ZR + p, j)) =
          *reinterpret_cast<const t_u32x4 *>(src + ((size_t)(pi + 1) * g.P + pj + 1) * 32 + j * 8);
    }
  }
  __syncthreads();

  // Accumulators are never zeroed: the first MFMA of every layer takes the layer's bias as C.
  t_f32x4 acc[MT][CT];
  // The residual x_l of this lane's outputs.  F = 128: packed 16-bit channel pairs in registers across the block's two
  // convolutions, added to conv2's accumulators in its epilogue (k_tower's order of operations: bit-identical logits).
  // F = 256 (RESIN): the 56 registers that takes are what hipcc parked in scratch (0.6 GB written per launch), so there
  // the residual never lives in registers across a k-step: conv1's epilogue reads x_l back from the image -- at the very
  // address it is about to overwrite with conv1's output -- and leaves  bias2 + x_l  in the accumulators, which conv2's
  // MFMAs then accumulate on: the same terms summed in another order (last-bit differences against k_tower256).
  constexpr bool RESIN = FPC_TWW_RESIN != 0 && F == 256;
  t_u32x2 res[RESIN ? 1 : MT][CT];
  // Where a tap's column shift leaves the board the lane reads a ZERO row instead (rows 0 .. 7, the one with its own
  // row's low three bits: same LDS bank as the read it replaces, so the conflict-free pattern survives): one address
  // select in front of the load, nothing behind it.
  // lane mask (a scalar register pair) of the lanes whose row tile mt reads off the board under tap
  auto off_board = [&](int tap, int mt) -> uint64_t {     // tap: wave-uniform
    const int dx = tap % 3 - 1;
    const int k = dx < 0 ? colk[mt] : colk[mt] == 0 ? R - 1 : colk[mt] - 1;     // li == k or k + R (R >= 8: no third)
    const uint32_t m16 = dx == 0 ? 0u : ((1u << k) | (1u << (k + R))) & 0xffffu;
    const uint32_t m32 = m16 * 0x10001u;
    return ((uint64_t)m32 << 32) | m32;
  };
  auto lane_select = [](uint32_t a, uint32_t b, uint64_t m) -> uint32_t {   // m's lanes: b; the others: a
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
  };

  // ---- stem: conv3x3(24 -> F) on the 32-channel input image: one 32-deep k-step per tap -----------------
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int shift = (tap / 3 - 1) * R + (tap % 3 - 1);
    t_u32x4 fa[CT], fb[MT];
    const unsigned char *wsrc = g.Wstem + (size_t)tap * SLAB + tileW * 1024;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) fa[ct] = *reinterpret_cast<const t_u32x4 *>(wsrc + ct * 1024 + wlane);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (mt == MT - 1 && !last_on) { fb[mt] = t_u32x4{0u, 0u, 0u, 0u}; continue; }
      const int r = (int)lane_select((uint32_t)(rbase + 16 * mt + shift), 0u, off_board(tap, mt));
      fb[mt] = *reinterpret_cast<const t_u32x4 *>(enc + (r >> 3) * 512 + (r & 7) * 16 + bq);
    }
    if (tap == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = M16<DT>::mfma(fa[ct], fb[mt], bst[ct]);
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = M16<DT>::mfma(fa[ct], fb[mt], acc[mt][ct]);
    }
  }
  __syncthreads();                                    // every wave is done with the stem's input image
  {                                                   // wipe it: x_0 is about to be written there and the zero rows must read as zero
    const int enc_bytes = TWW_ROWS * 64;
    for (int c = tid; c < enc_bytes / 16; c += NT) reinterpret_cast<t_u32x4 *>(enc)[c] = t_u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();

  // epilogue: the accumulators hold conv + bias; (+ residual, f32); 16-bit; ReLU on the packed pairs; written IN
  // PLACE into the image at the board's squares (4 consecutive channels = one 8-byte write at ONE lane address plus an
  // immediate; the lanes past the last square -- the board's last row tile only -- are masked off.  Not a per-lane
  // select of a dummy address: hipcc turns that into MT x CT address registers, parks them in scratch and waits out a
  // scratch load in front of every store)
  unsigned char *const wbase = img + (rbase >> 3) * GRP + (rbase & 7) * 16 + wn * (CT * 256) + (lane >> 5) * 128 + ((lane >> 4) & 1) * 8;
  // RES: 0: plain; 1: keep as residual (stem); 2: add the residual, keep (1, 2: F = 128 only); 3 (RESIN, conv1 of a
  // residual block, `layer`): pick x_l up from the image and leave  bias(layer + 1) + x_l  in the accumulators
  auto epilogue = [&](auto res_c, const int layer = 0) {
    constexpr int RES = decltype(res_c)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (mt == MT - 1 && !last_on) break;            // wave-uniform: the waves wm = 1 own one row tile less on some boards
      t_u32x2 pk[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        t_f32x4 v = acc[mt][ct];
        if (RES == 2) {
          v[0] += M16<DT>::lo(res[mt][ct][0]); v[1] += M16<DT>::hi(res[mt][ct][0]);
          v[2] += M16<DT>::lo(res[mt][ct][1]); v[3] += M16<DT>::hi(res[mt][ct][1]);
        }
        pk[ct] = t_u32x2{tw_relu2(M16<DT>::pack2(v[0], v[1])), tw_relu2(M16<DT>::pack2(v[2], v[3]))};
        if (RES == 1 || RES == 2) res[mt][ct] = pk[ct];
      }
      if (RES == 3) {
        const float *bn = g.bt + (size_t)(layer + 1) * 256 + tileW * 16 + 4 * lq;
        t_u32x2 x[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) x[ct] = t_u32x2{0u, 0u};
        if (p0 + 16 * mt < RR) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            t_u32x2 *const at = reinterpret_cast<t_u32x2 *>(wbase + mt * (2 * GRP) + ct * 256);
            x[ct] = *at;                              // x_l, then conv1's output over it (LDS keeps a wave's accesses in order)
            *at = pk[ct];
          }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const t_f32x4 b = *reinterpret_cast<const t_f32x4 *>(bn + ct * 16);
          acc[mt][ct] = t_f32x4{b[0] + M16<DT>::lo(x[ct][0]), b[1] + M16<DT>::hi(x[ct][0]),
                                b[2] + M16<DT>::lo(x[ct][1]), b[3] + M16<DT>::hi(x[ct][1])};
        }
      } else if (p0 + 16 * mt < RR) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) *reinterpret_cast<t_u32x2 *>(wbase + mt * (2 * GRP) + ct * 256) = pk[ct];
      }
      __builtin_amdgcn_sched_barrier(0);              // one row tile at a time: keeps the accumulator reads from piling up in VGPRs
    }
  };
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 1> c1{};
  const std::integral_constant<int, 2> c2{};
  if constexpr (RESIN) epilogue(c0); else epilogue(c1);   // stem: x_0 = relu(conv + b)
  __syncthreads();                                    // x_0 complete

  // ---- 9 taps x KSN k-steps per layer ---------------------------------------------------------------------
  t_u32x4 fa[PD][CT], fb[RING ? 4 : MT];
  int gk = 0;                                         // running slab index over all layers
  uint32_t brow[MT];                                  // image byte offset of this lane's row per row tile under the current tap's shift -- or of its zero row
  auto set_tap = [&](int tap) {
    const int r = rbase + (tap / 3 - 1) * R + (tap % 3 - 1);   // >= 16 - 15
    const uint32_t zoff = (r & 7) * 16 + bq;
    const uint32_t boff = (r >> 3) * GRP + zoff;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) brow[mt] = lane_select(boff + mt * (2 * GRP), zoff, off_board(tap, mt));
  };
  // ks: k-step within the tap (64 B per k-step along a row; the zero rows are zero at every k)
  auto bptr = [&](int mt, int ks) -> const unsigned char * { return img + brow[mt] + ks * 512; };
  // The weight stream through a buffer resource: scalar base + scalar byte offset (slab, first tile) + ONE 32-bit lane
  // offset + an immediate per column tile -- no 64-bit address arithmetic in vector registers, half the address data
  // per load (`buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen offset:imm`).
#if FPC_TWW_BUFLOAD
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(g.Wt), 0, 0x7fffffff, 0x00020000);
#endif
  auto wload = [&](auto buf_c, int slab, int tile0) {  // the wave's CT A fragments of slab `slab`, straight into registers
    constexpr int B = decltype(buf_c)::value;
#if FPC_TWW_BUFLOAD
    const int soff = slab * SLAB + tile0 * 1024;      // wave-uniform; the stream is < 2 GiB
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      fa[B][ct] = __builtin_bit_cast(t_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)wlane + ct * 1024, soff, 0));
#else
    const unsigned char *wsrc = g.Wt + (size_t)slab * SLAB + tile0 * 1024;      // wave-uniform
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) fa[B][ct] = *reinterpret_cast<const t_u32x4 *>(wsrc + ct * 1024 + wlane);
#endif
  };
  auto load_b = [&]() {                               // all image fragments of (tap 0, k-step 0), after an epilogue
    set_tap(0);                                       // here, not in front of the epilogue: MT row offsets less to carry through it
#pragma unroll
    for (int mt = 0; mt < (RING ? 4 : MT); ++mt)
      if (mt < MT - 1 || last_on) fb[mt] = *reinterpret_cast<const t_u32x4 *>(bptr(mt, 0));
  };
  // One k-step (slab gk, fragments in fa[KS mod PD] / fb): first the loads of slab gk + PD - 1 into the buffer k-step
  // gk - 1 has just left (first tile tile_next where that slab is the next layer's: LAST = the layer's last tap; the
  // stream is padded by TWW_PAD_SLABS slabs behind the last layer),
  // then MT groups of MFMAs, each followed by the in-place reload of its image fragment for the NEXT k-step (NEXT = 1:
  // 64 B further along the row; NEXT = 2: the caller has already moved the row address and the column mask to the next
  // tap: offset 0; NEXT = 0: the layer's last k-step: no reload -- the epilogue rewrites the image, load_b() follows).
  // The MT-th row tile exists only for the waves with `last_on` (wave-uniform): 13 = 7 + 6 tiles at 14x14.
  // MODE 0: CT column tiles per wave; 1 (value conv): one.  KS: the k-step's index within its tap (reload offset).
  // RING: NEXT = 2 moves the row offsets to tap `tap_next` itself, behind the last read of this tap's rows (the caller
  // does not); NEXT = 0 carries the layer's barrier, behind the layer's last image read (tile MT - 1 of this k-step,
  // issued behind tile MT - 5): every fragment this wave still needs is on its way, the epilogues may rewrite the image.
  auto kstep = [&](auto mode_c, auto ks_c, auto bias_c, auto next_c, auto last_c, const t_f32x4 *b4, const int tile0, const int tile_next,
                   const int tap_next = 0) {
    constexpr int MODE = decltype(mode_c)::value, KS = decltype(ks_c)::value, B = KS % PD;
    constexpr bool BIAS = decltype(bias_c)::value != 0;
    constexpr int NEXT = decltype(next_c)::value;
    constexpr bool CROSS = decltype(last_c)::value != 0 && KS + PD - 1 >= KSN;     // the slab fetched here is the next layer's
    wload(std::integral_constant<int, (KS + PD - 1) % PD>{}, gk + PD - 1, CROSS ? tile_next : tile0);
    ++gk;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int q = RING ? (KS * MT + mt) & 3 : mt;   // the register quad of this tile
      if (mt < MT - 1 || last_on) {
        if (MODE == 1) {
          acc[mt][0] = M16<DT>::mfma(fa[B][0], fb[q], BIAS ? b4[0] : acc[mt][0]);
        } else {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = M16<DT>::mfma(fa[B][ct], fb[q], BIAS ? b4[ct] : acc[mt][ct]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!RING && NEXT != 0) {
          fb[q] = *reinterpret_cast<const t_u32x4 *>(bptr(mt, NEXT == 1 ? KS + 1 : 0));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (RING) {                           // tile c + 4 into the quad tile c has just left (also behind a tile this wave skips)
        const int tm = mt + 4;
        if (tm < MT) {                                // ... of this k-step
          if (tm < MT - 1 || last_on) fb[q] = *reinterpret_cast<const t_u32x4 *>(bptr(tm, KS));
        } else if (NEXT != 0) {                       // ... of the next one
          if (NEXT == 2 && tm == MT) set_tap(tap_next);
          fb[q] = *reinterpret_cast<const t_u32x4 *>(bptr(tm - MT, NEXT == 1 ? KS + 1 : 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (NEXT == 0 && mt == MT - 5) {
          __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  // k-steps 1 .. KSN - 2 of a tap.  A macro, not a lambda: hipcc does not inline a lambda that calls `kstep` six times
  // into `run_layer`, and an out-of-line call sends every captured register array (accumulators, fragments) through memory.
#define FPC_TWW_MID_KSTEPS(LAST)                                                               \
  do {                                                                                         \
    kstep(mode_c, std::integral_constant<int, 1>{}, c0, c1, LAST, b4, tile0, tile_next);       \
    kstep(mode_c, std::integral_constant<int, 2>{}, c0, c1, LAST, b4, tile0, tile_next);       \
    if (KSN == 8) {                                                                            \
      kstep(mode_c, std::integral_constant<int, 3>{}, c0, c1, LAST, b4, tile0, tile_next);     \
      kstep(mode_c, std::integral_constant<int, 4>{}, c0, c1, LAST, b4, tile0, tile_next);     \
      kstep(mode_c, std::integral_constant<int, 5>{}, c0, c1, LAST, b4, tile0, tile_next);     \
      kstep(mode_c, std::integral_constant<int, 6>{}, c0, c1, LAST, b4, tile0, tile_next);     \
    }                                                                                          \
  } while (0)
  // PACING of the two waves of a SIMD.  Waves w and w ^ 4 share a SIMD (a workgroup's waves are dealt to the four SIMDs
  // cyclically) and its matrix pipe, which arbitrates by priority, then AGE: left alone, the older wave runs a layer at
  // its own full speed (~23 cycles per MFMA: its loads, LDS reads and waits sit between its MFMAs), the younger one gets
  // the leftover third -- and then runs the rest of the layer ALONE, again at 23 cycles per MFMA, where the two
  // together sustain ~17.5 (stamped timeline: tools/towerw_stamps.py).  So at every tap each wave publishes its
  // k-step counter in LDS, reads its partner's, and the one that is BEHIND takes s_setprio 1 until the next tap: the two
  // stay within a tap of each other and the pipe sees both streams for the whole layer.
  int *const pace = reinterpret_cast<int *>(smem + TW_PACE);              // [8] k-step counters
  auto pace_exchange = [&]() -> int {
#ifdef FPC_TWW_NOPACE
    return 0;
#else
    const int seen = pace[wave ^ 4];
    if (lane == 0) pace[wave] = gk;
    return seen;
#endif
  };
  auto pace_set = [&](int seen) {
#ifndef FPC_TWW_NOPACE
    if (__builtin_amdgcn_readfirstlane(seen) > gk) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#endif
  };
  // the first PD - 1 slabs of a layer (gk: its first slab) -- the kernel's prologue and the waves that sat a layer out
  auto wload_head = [&](int tile) {
    wload(c0, gk, tile);
    if constexpr (PD == 4) { wload(c1, gk + 1, tile); wload(c2, gk + 2, tile); }
  };
  // One conv layer.  On entry fa[0 .. PD - 2] hold the fragments of slabs gk .. gk + PD - 2 of this wave (first tile tile0), fb the image fragments
  // of (tap 0, k-step 0) and the row address / column mask are tap 0's.  `active`: false for the waves that own none of
  // the layer's live output channels (value conv; policy conv at F = 256): they only keep the layer's barrier and fetch
  // what the layer's last k-step leaves behind.
  auto run_layer = [&](auto mode_c, auto pre_c, const bool active, const int layer, const int tile0, const int tile_next) {
    if (!active) {                                    // wave-uniform
      gk += 9 * KSN;
      __syncthreads();                                // the layer's one barrier
      wload_head(tile_next);
      return;
    }
    constexpr int MODE = decltype(mode_c)::value;
    constexpr bool PRE = decltype(pre_c)::value != 0;  // the accumulators already hold the layer's bias (+ the residual): epilogue mode 3
    t_f32x4 b4[CT];
    if (!PRE) {
      const float *bl = g.bt + (size_t)layer * 256 + tile0 * 16 + 4 * lq;
#pragma unroll
      for (int ct = 0; ct < (MODE == 1 ? 1 : CT); ++ct) b4[ct] = *reinterpret_cast<const t_f32x4 *>(bl + ct * 16);
    }
    TWW_STAMP(layer, 0);
    kstep(mode_c, c0, std::integral_constant<int, PRE ? 0 : 1>{}, c1, c0, b4, tile0, tile_next);      // (tap 0, k-step 0), C = bias
#pragma unroll 1
    for (int tap = 0; tap < 8; ++tap) {
      FPC_TWW_MID_KSTEPS(c0);
      if constexpr (!RING) set_tap(tap + 1);          // every read of this tap's rows has been issued
      TWW_STAMP(layer, 1 + tap);
      const int seen = pace_exchange();               // issued in front of the next tap's fragment reads: done when the first of them is
      kstep(mode_c, std::integral_constant<int, KSN - 1>{}, c0, c2, c0, b4, tile0, tile_next, tap + 1);   // reloads (next tap, k-step 0)
      pace_set(seen);
      kstep(mode_c, c0, c0, c1, c0, b4, tile0, tile_next);      // (next tap, k-step 0)
    }
    FPC_TWW_MID_KSTEPS(c1);                           // tap 8: its late k-steps fetch the next layer's first slabs
    __builtin_amdgcn_s_setprio(0);
    TWW_STAMP(layer, 9);
    if constexpr (!RING) __syncthreads();             // every wave holds its last image fragments: the epilogues may rewrite the image
    TWW_STAMP(layer, 10);
    kstep(mode_c, std::integral_constant<int, KSN - 1>{}, c0, c0, c1, b4, tile0, tile_next);   // the layer's last k-step
    TWW_STAMP(layer, 11);
  };
#undef FPC_TWW_MID_KSTEPS

  wload_head(tileW);
  load_b();
  const int nblocks = g.L / 2;
#pragma unroll 1
  for (int blk = 0; blk < nblocks; ++blk) {
    run_layer(c0, c0, true, 2 * blk, tileW, tileW);                           // conv1 + BN + ReLU
    if constexpr (RESIN) epilogue(std::integral_constant<int, 3>{}, 2 * blk); else epilogue(c0);
    TWW_STAMP(2 * blk, 12);
    __syncthreads();
    TWW_STAMP(2 * blk, 13);
    load_b();
    TWW_STAMP(2 * blk, 14);
    run_layer(c0, std::integral_constant<int, RESIN ? 1 : 0>{}, true, 2 * blk + 1, tileW, blk + 1 == nblocks ? tile16 : tileW);   // conv2 + BN, + x_l, ReLU
    if constexpr (RESIN) epilogue(c0); else epilogue(c2);
    __syncthreads();
    load_b();
  }
  float vpart = 0.f;
  {
    // value head (net.py:28-35): relu(conv + b)[pos][ch] . vw[pos][ch], ch < 24 (weights, biases and vw zero-padded to 32)
    run_layer(c1, c0, half_active, g.L, tile16, tileW);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int p = p0 + 16 * mt;
      const bool in = p < RR && half_active && (mt < MT - 1 || last_on);
      const t_f32x4 w4 = *reinterpret_cast<const t_f32x4 *>(g.vw + (in ? p : 0) * 32 + tile16 * 16 + 4 * lq);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[mt][0][j];
        v = v > 0.f ? v : 0.f;
        vpart += in ? v * w4[j] : 0.f;
      }
    }
    load_b();                                         // the value conv leaves the image as it was
  }
  const bool pol_active = tileW < 8;                  // policy conv: 128 output rows (120 live) = 8 column tiles
  run_layer(c0, c0, pol_active, g.L + 1, tileW, tileW);                           // policy conv + BN + ReLU, 16-bit rows in place
  if (pol_active) epilogue(c0);
  __syncthreads();

  // ---- heads: policy-conv rows -> Linear input (position-major), value -> tanh ---------------------------
  {
    const int cpr = g.A_ch / 8;                       // 16-byte chunks per position
    for (int c = tid; c < RR * 16; c += NT) {
      const int q = c >> 4, j = c & 15;
      if (j >= cpr) continue;
      *reinterpret_cast<t_u32x4 *>(g.xfc + (size_t)game * g.Kp + (size_t)q * g.A_ch + j * 8) =
          *reinterpret_cast<const t_u32x4 *>(img + tw_lay(F / 8, TWW_ZR + q, j));
    }
  }
  for (int off = 32; off >= 1; off >>= 1) vpart += __shfl_xor(vpart, off);
  if (lane == 0) vred[wave] = vpart;
  __syncthreads();
#ifdef TWW_STAMPS
  if (g.stamps && (blockIdx.x == 0 || blockIdx.x == 131) && tid < 128)
    g.stamps[(blockIdx.x ? 128 : 0) + tid] = reinterpret_cast<unsigned long long *>(smem + 3072)[tid];
#endif
  // partials in the order (wm, wn & 1): waves 0, 1, 4, 5 (the others hold zero and are not read)
  if (tid == 0) g.value[game] = tanhf(g.vb + ((vred[0] + vred[1]) + (vred[4] + vred[5])));
}

// weights [taps][cout_pad rows][cin] 16-bit row-major -> [tap][k-step of 32][cout tile of 16][lane = 16 q + c][8]:
// element (tap, ks, t, q, c, e) = W[tap][16 t + c][32 ks + 8 q + e]; `tiles` cout tiles per slab, rows >= cout_pad zero
__global__ void __launch_bounds__(256) k_towerw_prep(const uint16_t *W, unsigned char *out, int taps, int cout_pad, int cin, int tiles) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;             // one 16-byte chunk of the output
  const int ksn = cin / 32;
  if (c >= (long)taps * ksn * tiles * 64) return;
  const int ln = (int)(c & 63), t = (int)((c >> 6) % tiles), ks = (int)((c / (64 * tiles)) % ksn), tap = (int)(c / ((long)64 * tiles * ksn));
  const int row = t * 16 + (ln & 15), q = ln >> 4;
  t_u32x4 v = t_u32x4{0u, 0u, 0u, 0u};
  if (row < cout_pad) v = *reinterpret_cast<const t_u32x4 *>(W + ((size_t)tap * cout_pad + row) * cin + ks * 32 + q * 8);
  *reinterpret_cast<t_u32x4 *>(out + (size_t)c * 16) = v;
}

}  // namespace fpc
