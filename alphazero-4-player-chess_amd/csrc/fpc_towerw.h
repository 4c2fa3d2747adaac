// fpc_towerw.h -- k_towerw<DT, F, MT, FAST> (round 4): the residual tower of net.py:6-63 (stem + 2*Nb residual convs +
// both head convs) in ONE launch, one game per workgroup, activations resident in LDS, for hidden F = 128 or 256 and
// any board of 8..14 squares a side -- on TWO WAVES PER SIMD with the weights going straight from L2 into registers:
// no weight ring in LDS, no LDS-DMA, no barrier inside a layer.
//
// Where it comes from.  k_tower256 (fpc_tower256.h, round 2: one wave per SIMD, 2-slab LDS weight ring) sat at 0.36 of
// the MFMA peak.  Its inner loop was not the problem (hipcc's k-step there is 56 MFMAs + 15 fragment reads, clean);
// what stood between two k-steps was: `s_waitcnt vmcnt(0)`, the workgroup barrier and a burst of four LDS-DMA pieces
// (60-185 cycles of issue port each) 72 times per layer, with nothing to run in their shadow -- ~2 000 cycles per
// k-step for 896 cycles of matrix pipe.  Here:
//   * 8 waves, wave (wm, wn) = MT row tiles of 16 grid positions x CT = F/64 column tiles of 16 output channels.
//     At F = 256, 14x14: 112 accumulator registers + the residual of the same outputs packed (56) + the double-buffered
//     weight fragments (32) + the image fragments, single-buffered and reloaded in place (28) = 228 of the 256 a wave of
//     a two-waves-per-SIMD kernel may hold (hipcc parks part of the residual in scratch between two epilogues: 73
//     dwords per lane stored and reloaded once per residual block, nothing inside a layer).
//   * a wave's weight operand of a k-step is [16 CT cout][32 cin] = CT A fragments; the host stores the weights in
//     FRAGMENT ORDER  [layer][tap][k-step][cout tile of 16][lane][8]  (k_towerw_prep), so one fragment is one perfectly
//     coalesced 1-KiB `global_load_dwordx4` of the wave, issued one k-step (its own MFMAs + its partner's) ahead of
//     its use and waited for by the counted vmcnt hipcc places in front of the first MFMA that needs it.  The two waves
//     that share a cout range (wm = 0 / 1) fetch the same lines within a few hundred cycles of each other: the second
//     fetch is an L1 hit.  Nothing is shared through LDS, so nothing has to be published: a layer runs without a
//     single barrier; the two waves of a SIMD drift apart by themselves and fill each other's waits.
//   * two barriers per layer remain: in front of the layer's last k-step (every wave has read its last image
//     fragments: the epilogues may rewrite the image in place) and behind the epilogues.
//   * every output element sees the same MFMAs on the same operands in the same order as in k_tower256 / k_tower:
//     logits and values are BIT-IDENTICAL to those kernels (tests/test_nn_gpu.py).
//   * LDS: 4 KiB front strip + 240 image rows x 2F bytes = 124 KiB (F = 256) / 64 KiB (F = 128); row geometry is
//     k_tower's generic one (MT = 3 / 5 / 7 row tiles per wave), so hidden = 256 runs as ONE launch at every board size,
//     incl. the reference's shipped ResNet(15, 256) on its 8x8 board (alphazero.py:288).
// Measured (same box, kernel trace): ResNet(20,256) 14x14, 256 leaves: 1.95 ms per launch against k_tower256's 2.74
// = 1.24 PFLOP/s algorithmic (0.49 of the dense peak; 1.41 PFLOP/s executed incl. the border tiles, within 5 % of a bare
// MFMA stream at the clock the chip holds under such a load).
#pragma once
#include "fpc_tower.h"

namespace fpc {

constexpr int TWW_THREADS = 512;
constexpr int TWW_IMG0 = 4096;                   // [0, 4096): dummy strip, value partials, leaf board (k_tower's offsets)
__host__ __device__ constexpr int tww_lds(int F) { return TWW_IMG0 + 240 * F * 2; }
__host__ __device__ constexpr int tww_slab(int F) { return F * 64; }             // one 32-deep k-step of one tap: [F cout][32 cin] x 2 B

// TowerArgs as for k_tower, with: Wstem = 9 slabs, Wt = (L + 2) * 9 * (F / 32) slabs + one slab of padding, both in
// fragment order; bt = [L + 2][256]; bstem = [F].
template <int DT, int F, int MT, bool FAST>
__global__ void __launch_bounds__(TWW_THREADS, 2) k_towerw(TowerArgs g) {
  constexpr int NT = TWW_THREADS;
  constexpr int CT = F / 64;                       // column tiles (16 output channels) per wave: 4 or 2
  constexpr int KSN = F / 32;                      // k-steps per tap: 8 or 4
  constexpr int GRP = F * 16;                      // bytes of 8 image rows
  constexpr int SLAB = tww_slab(F);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *const img = smem + TWW_IMG0;
  float *const vred = reinterpret_cast<float *>(smem + TW_VRED);        // [8]
  fpc_board *const lboard = reinterpret_cast<fpc_board *>(smem + TW_BOARD);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 15, lq = lane >> 4;
  const int game = blockIdx.x;
  const int P = g.P, NR = g.NR;

  int slot = 0, rot_k = 0;
  if (g.boards) {
    slot = g.leaf_slot[game];
    if (slot < 0) return;                      // the game has left the search (Q5)
    rot_k = first_leaf_turn(g.leaf_slot, g.leaf_turn, g.n_games);
  }

  const int rbase = (g.T0 + wm * MT) * 16 + li;       // grid position (image row) of this lane in its first row tile
  uint32_t inmask = 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = rbase + 16 * mt;
    const int pi = r / P, pj = r - pi * P;
    if (r < g.PP && pi >= 1 && pi <= g.R && pj >= 1 && pj <= g.R) inmask |= 1u << mt;
  }
  const int bq = (lq >> 1) * 256 + (lq & 1) * 128;
  const int tileW = wn * CT;                          // first 16-cout tile of this wave in a conv layer
  const int tile16 = wn & 1;                          // value conv: 32 live channels = tiles 0, 1; the waves wn >= 2 sit it out
  const bool half_active = wn < 2;                    // value conv (32 live rows) and, at F = 256, policy conv (128 of 256)
  const uint32_t wlane = (uint32_t)lane * 16u;

  // ---- zero the front strip and the image, build the stem's 32-channel input image inside it -----------
  for (int c = tid; c < tww_lds(F) / 16; c += NT) reinterpret_cast<t_u32x4 *>(smem)[c] = t_u32x4{0u, 0u, 0u, 0u};
  unsigned char *const enc = img;                     // rows x 64 B
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
    for (int r = tid; r < g.PP; r += NT) {
      const int pi = r / P, pj = r - pi * P;
      if (pi < 1 || pi > g.R || pj < 1 || pj > g.R) continue;
      const uint8_t p = lboard->sq[rot90_src(g.R, rot_k, pi - 1, pj - 1)];
      if (!present(p)) continue;
      const int plane = piece_plane(p, lboard->turn, g.rules);
      *reinterpret_cast<uint16_t *>(enc + tw_lay(4, r, plane >> 3) + (plane & 7) * 2) = g.one16;
    }
  } else {
    const uint16_t *src = g.in16 + (size_t)game * g.PP * 32;
    for (int c = tid; c < NR * 4; c += NT) {
      const int r = c >> 2, j = c & 3;
      *reinterpret_cast<t_u32x4 *>(enc + tw_lay(4, r, j)) = *reinterpret_cast<const t_u32x4 *>(src + (size_t)r * 32 + j * 8);
    }
  }
  __syncthreads();

  // Accumulators are never zeroed: the first MFMA of every layer takes the layer's bias as C.
  t_f32x4 acc[MT][CT];
  t_u32x2 res[MT][CT];           // residual x_l of this lane's outputs, packed 16-bit channel pairs
  auto brow = [&](int mt, int shift) -> int {         // image row of this lane for row tile mt under a tap shift, bottom border aliased onto the top one
    int r = rbase + 16 * mt + shift;
    r = r < 0 ? r + NR : r;
    r = r >= NR ? r - NR : r;
    return r;
  };

  // ---- stem: conv3x3(24 -> F) on the 32-channel input image: one 32-deep k-step per tap -----------------
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int shift = (tap / 3 - 1) * P + (tap % 3 - 1);
    t_u32x4 fa[CT], fb[MT];
    const unsigned char *wsrc = g.Wstem + (size_t)tap * SLAB + tileW * 1024;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) fa[ct] = *reinterpret_cast<const t_u32x4 *>(wsrc + ct * 1024 + wlane);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = brow(mt, shift);
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
  {                                                   // wipe it: x_0 is about to be written there and the borders must read as zero
    const int enc_bytes = ((NR + 7) >> 3) * 512;
    for (int c = tid; c < enc_bytes / 16; c += NT) reinterpret_cast<t_u32x4 *>(enc)[c] = t_u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();

  // epilogue: the accumulators hold conv + bias; (+ residual, f32); 16-bit; ReLU on the packed pairs; written IN
  // PLACE into the image at interior squares (4 consecutive channels = one 8-byte write; the other lanes write to a
  // dummy strip, no branch)
  unsigned char *const dummy = smem + TW_DUMMY + lane * 8;
  unsigned char *const wbase = img + (rbase >> 3) * GRP + (rbase & 7) * 16 + wn * (CT * 256) + (lane >> 5) * 128 + ((lane >> 4) & 1) * 8;
  auto epilogue = [&](auto res_c) {
    constexpr int RES = decltype(res_c)::value;       // 0: plain; 1: keep as residual (stem); 2: add the residual, keep
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      unsigned char *dst = ((inmask >> mt) & 1u) ? wbase + mt * (2 * GRP) : dummy;
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
      __builtin_amdgcn_sched_barrier(0);              // one row tile at a time: keeps the accumulator reads from piling up in VGPRs
    }
  };
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 1> c1{};
  const std::integral_constant<int, 2> c2{};
  epilogue(c1);                                       // stem: x_0 = relu(conv + b)
  __syncthreads();                                    // x_0 complete

  // ---- 9 taps x KSN k-steps per layer ---------------------------------------------------------------------
  t_u32x4 fa[2][CT], fb[MT];
  int gk = 0;                                         // running slab index over all layers
  const unsigned char *bbase, *blast;                 // FAST: this lane's image row in its first / last row tile under the current tap's shift
  int boff[FAST ? 1 : MT];                            // !FAST: image byte offset per row tile
  auto set_tap = [&](int tap) {
    const int shift = (tap / 3 - 1) * P + (tap % 3 - 1);
    if (FAST) {
      const int r = rbase + shift;                    // >= -1; row -1 lands in the 4 KiB in front of the image
      bbase = img + (r >> 3) * GRP + (r & 7) * 16 + bq;
      int rl = r + 16 * (MT - 1);
      rl = rl >= NR ? rl - NR : rl;
      blast = img + (rl >> 3) * GRP + (rl & 7) * 16 + bq;
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int r = brow(mt, shift);
        boff[mt] = (r >> 3) * GRP + (r & 7) * 16 + bq;
      }
    }
  };
  auto bptr = [&](int mt) -> const unsigned char * {
    if (FAST) return mt == MT - 1 ? blast : bbase + mt * (2 * GRP);
    return img + boff[mt];
  };
  auto load_b = [&]() {                               // all image fragments of k-step 0 of the current tap (after an epilogue)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fb[mt] = *reinterpret_cast<const t_u32x4 *>(bptr(mt));
  };
  auto wload = [&](auto buf_c, int slab, int tile0) {  // the wave's CT A fragments of slab `slab`, straight into registers
    constexpr int B = decltype(buf_c)::value;
    const unsigned char *wsrc = g.Wt + (size_t)slab * SLAB + tile0 * 1024;      // wave-uniform
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) fa[B][ct] = *reinterpret_cast<const t_u32x4 *>(wsrc + ct * 1024 + wlane);
  };
  // One k-step (slab gk, fragments in fa[B] / fb): first the loads of slab gk + 1 into fa[B ^ 1] (first tile tile_next:
  // the next layer's when this is a layer's last k-step; the stream is padded by one slab behind the last layer),
  // then MT groups of MFMAs, each followed by the in-place reload of its image fragment for the NEXT k-step (NEXT = 1:
  // 64 B further along the row; NEXT = 2: the caller has already moved the row addresses to the next tap: offset 0;
  // NEXT = 0: the layer's last k-step: no reload -- the epilogue rewrites the image, load_b() follows it).
  // MODE 0: CT column tiles per wave; 1 (value conv): one.  KS: the k-step's index within its tap (reload offset).
  auto kstep = [&](auto mode_c, auto buf_c, auto ks_c, auto bias_c, auto next_c, const t_f32x4 *b4, const int tile_next) {
    constexpr int MODE = decltype(mode_c)::value, B = decltype(buf_c)::value, KS = decltype(ks_c)::value;
    constexpr bool BIAS = decltype(bias_c)::value != 0;
    constexpr int NEXT = decltype(next_c)::value;
    wload(std::integral_constant<int, B ^ 1>{}, gk + 1, tile_next);
    ++gk;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (MODE == 1) {
        acc[mt][0] = M16<DT>::mfma(fa[B][0], fb[mt], BIAS ? b4[0] : acc[mt][0]);
      } else {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = M16<DT>::mfma(fa[B][ct], fb[mt], BIAS ? b4[ct] : acc[mt][ct]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (NEXT != 0) {
        fb[mt] = *reinterpret_cast<const t_u32x4 *>(bptr(mt) + (NEXT == 1 ? (KS + 1) * 512 : 0));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // k-steps 1 .. KSN - 2 of a tap (buffers alternate: k-step k reads fa[k & 1]).  A macro, not a lambda: hipcc does not
  // inline a lambda that calls `kstep` six times into `run_layer`, and an out-of-line call sends every captured register
  // array (accumulators, fragments) through memory.
#define FPC_TWW_MID_KSTEPS()                                                            \
  do {                                                                                  \
    kstep(mode_c, c1, std::integral_constant<int, 1>{}, c0, c1, b4, tile0);             \
    kstep(mode_c, c0, std::integral_constant<int, 2>{}, c0, c1, b4, tile0);             \
    if (KSN == 8) {                                                                     \
      kstep(mode_c, c1, std::integral_constant<int, 3>{}, c0, c1, b4, tile0);           \
      kstep(mode_c, c0, std::integral_constant<int, 4>{}, c0, c1, b4, tile0);           \
      kstep(mode_c, c1, std::integral_constant<int, 5>{}, c0, c1, b4, tile0);           \
      kstep(mode_c, c0, std::integral_constant<int, 6>{}, c0, c1, b4, tile0);           \
    }                                                                                   \
  } while (0)
  // One conv layer.  On entry fa[0] holds slab gk's fragments of this wave (first tile tile0), fb the image fragments
  // of (tap 0, k-step 0) and the row addresses are tap 0's.  `active`: false for the waves that own none of the layer's
  // live output channels (value conv; policy conv at F = 256): they only keep the layer's barrier and fetch what the
  // layer's last k-step leaves behind.
  auto run_layer = [&](auto mode_c, const bool active, const int layer, const int tile0, const int tile_next) {
    if (!active) {                                    // wave-uniform
      gk += 9 * KSN;
      __syncthreads();                                // the layer's one barrier
      set_tap(0);
      wload(c0, gk, tile_next);
      return;
    }
    constexpr int MODE = decltype(mode_c)::value;
    t_f32x4 b4[CT];
    {
      const float *bl = g.bt + (size_t)layer * 256 + tile0 * 16 + 4 * lq;
#pragma unroll
      for (int ct = 0; ct < (MODE == 1 ? 1 : CT); ++ct) b4[ct] = *reinterpret_cast<const t_f32x4 *>(bl + ct * 16);
    }
    kstep(mode_c, c0, c0, c1, c1, b4, tile0);         // (tap 0, k-step 0), C = bias
#pragma unroll 1
    for (int tap = 0; tap < 8; ++tap) {
      FPC_TWW_MID_KSTEPS();
      set_tap(tap + 1);                               // every read of this tap's rows has been issued
      kstep(mode_c, c1, std::integral_constant<int, KSN - 1>{}, c0, c2, b4, tile0);   // reloads (next tap, k-step 0)
      kstep(mode_c, c0, c0, c0, c1, b4, tile0);       // (next tap, k-step 0)
    }
    FPC_TWW_MID_KSTEPS();                             // tap 8
    __syncthreads();                                  // every wave holds its last image fragments: the epilogues may rewrite the image
    set_tap(0);
    kstep(mode_c, c1, std::integral_constant<int, KSN - 1>{}, c0, c0, b4, tile_next);   // the layer's last k-step; fetches the NEXT layer's first slab
  };
#undef FPC_TWW_MID_KSTEPS

  set_tap(0);
  wload(c0, 0, tileW);
  load_b();
  const int nblocks = g.L / 2;
#pragma unroll 1
  for (int blk = 0; blk < nblocks; ++blk) {
    run_layer(c0, true, 2 * blk, tileW, tileW);                               // conv1 + BN + ReLU
    epilogue(c0);
    __syncthreads();
    load_b();
    run_layer(c0, true, 2 * blk + 1, tileW, blk + 1 == nblocks ? tile16 : tileW);   // conv2 + BN, + x_l, ReLU
    epilogue(c2);
    __syncthreads();
    load_b();
  }
  float vpart = 0.f;
  {
    // value head (net.py:28-35): relu(conv + b)[pos][ch] . vw[pos][ch], ch < 24 (weights, biases and vw zero-padded to 32)
    run_layer(c1, half_active, g.L, tile16, tileW);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = rbase + 16 * mt;
      const int pi = r / P, pj = r - pi * P;
      const bool in = ((inmask >> mt) & 1u) && half_active;
      const int qp = in ? (pi - 1) * g.R + (pj - 1) : 0;
      const t_f32x4 w4 = *reinterpret_cast<const t_f32x4 *>(g.vw + qp * 32 + tile16 * 16 + 4 * lq);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[mt][0][j];
        v = v > 0.f ? v : 0.f;
        vpart += in ? v * w4[j] : 0.f;
      }
    }
    load_b();                                         // the value conv leaves the image as it was
  }
  const bool pol_active = F == 128 || half_active;    // policy conv: 128 output rows (120 live)
  run_layer(c0, pol_active, g.L + 1, tileW, tileW);                           // policy conv + BN + ReLU, 16-bit rows in place
  if (pol_active) epilogue(c0);
  __syncthreads();

  // ---- heads: policy-conv rows -> Linear input (position-major), value -> tanh ---------------------------
  {
    const int cpr = g.A_ch / 8;                       // 16-byte chunks per position
    for (int c = tid; c < g.PP * 16; c += NT) {
      const int r = c >> 4, j = c & 15;
      if (j >= cpr) continue;
      const int pi = r / P, pj = r - pi * P;
      if (pi < 1 || pi > g.R || pj < 1 || pj > g.R) continue;
      const int q = (pi - 1) * g.R + (pj - 1);
      *reinterpret_cast<t_u32x4 *>(g.xfc + (size_t)game * g.Kp + (size_t)q * g.A_ch + j * 8) =
          *reinterpret_cast<const t_u32x4 *>(img + tw_lay(F / 8, r, j));
    }
  }
  for (int off = 32; off >= 1; off >>= 1) vpart += __shfl_xor(vpart, off);
  if (lane == 0) vred[wave] = vpart;
  __syncthreads();
  // partials in the order of k_tower / k_tower256 (wm, wn & 1): waves 0, 1, 4, 5 (the others hold zero and are not read)
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
