// fpc_tower256.h -- k_tower256: the residual tower of net.py:6-63 at hidden = 256 on the 14x14 board
// (BASELINE configs[3]: ResNet(20,256), fp16) in ONE launch, one game per workgroup, activations resident
// in LDS.  Same construction as k_tower (fpc_tower.h: 4 waves = one per SIMD, v_mfma_f32_16x16x32 issued
// as W x X^T, only the 14 row tiles with interior squares, LDS image [row/8][chunk/2][chunk%2][row%8] x 16 B
// that is conflict-free under any row shift, bottom border aliased onto the top one, weights by LDS-DMA),
// re-proportioned for 256 channels:
//   * image 240 rows x 512 B = 120 KiB; what is left of the 160 KiB holds TWO 16-KiB weight slabs
//     (one 32-deep k-step of one tap: [256 cout][32 cin]), so the ring turns once per k-step: the barrier
//     at the top of k-step k publishes slab k + 1 (whose fragments are read during k-step k, one step
//     ahead of their MFMAs) and frees the buffer of slab k for the DMA of slab k + 2.
//   * wave (wm, wn) owns 7 row tiles x 8 column tiles (112 positions x 128 channels): 56 MFMAs per 15
//     fragment reads, 224 accumulator registers (in place) + the residual of the same outputs packed
//     (112).  The image fragments are single-buffered and reloaded in place, each right behind the 8
//     MFMAs that consumed it, for the next k-step; only the weight fragments are double-buffered.
//   * the stem (24 -> 256, one k-step per tap) streams its nine slabs through the same two buffers; its
//     32-channel input image lives inside the (still unused) main image and is wiped before the stem's
//     epilogue writes x_0.
//   * head convolutions ride the same stream with their output channels zero-padded to 256.
#pragma once
#include "fpc_tower.h"

namespace fpc {

constexpr int T2_IMG0 = 4096;                    // [0, 4096): biases 2 x 1 KiB, dummy strip, value partials, leaf board (as k_tower)
constexpr int T2_IMG = 240 * 512;                // 122880
constexpr int T2_SLAB = 16384;                   // [256 cout][32 cin] x 2 B
constexpr int T2_RING = T2_IMG0 + T2_IMG;        // 126976: 2 slabs
constexpr int T2_LDS = T2_RING + 2 * T2_SLAB;    // 159744
constexpr int T2_KS = 8;                         // k-steps per tap (256 / 32)

// tw_dma_4k (4 consecutive 1-KiB LDS-DMA pieces): fpc_tower.h

#ifndef T2_STREAM
#define T2_STREAM 0              // 0: a slab's four DMA pieces per wave go out as a burst behind the k-step barrier.  1 (k_tower's
                                 // scheme: one M0 write, pieces between the first MFMA groups): measured 4.5 % SLOWER on the same box
                                 // (2.78 -> 2.91 ms): with only two slab buffers a slab has ONE k-step to land, and every cycle a
                                 // piece is issued later is a cycle the next barrier waits longer
#endif
// TowerArgs as for k_tower, with: Wstem = 9 slabs, Wt = (L + 2) * 72 slabs, bt = [L + 2][256].  14x14 only.
template <int DT>
__global__ void __launch_bounds__(TW_THREADS, 1) k_tower256(TowerArgs g) {
  constexpr int MT = 7;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *const img = smem + T2_IMG0;
  unsigned char *const ring = smem + T2_RING;
  const float *const biasbuf = reinterpret_cast<const float *>(smem + TW_BIAS);
  float *const vred = reinterpret_cast<float *>(smem + TW_VRED);
  fpc_board *const lboard = reinterpret_cast<fpc_board *>(smem + TW_BOARD);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lq = lane >> 4;
  const int game = blockIdx.x;
  const int P = 16, NR = 240, R = 14;

  int slot = 0, rot_k = 0;
  if (g.boards) {
    slot = g.leaf_slot[game];
    if (slot < 0) return;                      // the game has left the search (Q5)
    rot_k = first_leaf_turn(g.leaf_slot, g.leaf_turn, g.n_games);
  }

  const int rbase = (1 + wm * MT) * 16 + li;   // grid position of this lane in its first row tile (row tiles 1 .. 14)
  uint32_t inmask = 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int r = rbase + 16 * mt, pi = r >> 4, pj = r & 15;
    if (pi >= 1 && pi <= R && pj >= 1 && pj <= R) inmask |= 1u << mt;
  }
  const int bq = (lq >> 1) * 256 + (lq & 1) * 128;
  const int apart_s = (li >> 3) * 512 + (li & 7) * 16 + bq;        // weight fragment: row li of a 16-row tile of a slab (4 chunks per row)
  const int cb128 = wn * 128, cb16 = wn * 16;
  const uint32_t dma_lane = (uint32_t)lane * 16u;

  // slab DMA: 16 KiB = 4 pieces per wave
  auto issue = [&](const unsigned char *src_slab, int buf) {
    tw_dma_4k(src_slab + wave * 4096, dma_lane, (uint32_t)__builtin_amdgcn_readfirstlane(T2_RING + buf * T2_SLAB + wave * 4096));
  };

  // ---- zero everything in front of the ring, build the stem's input image inside the main image ----
  for (int c = tid; c < T2_RING / 16; c += TW_THREADS) reinterpret_cast<t_u32x4 *>(smem)[c] = t_u32x4{0u, 0u, 0u, 0u};
  unsigned char *const enc = img;              // 240 rows x 64 B
  t_f32x4 bst[8];
#pragma unroll
  for (int ct = 0; ct < 8; ++ct) bst[ct] = *reinterpret_cast<const t_f32x4 *>(g.bstem + cb128 + ct * 16 + 4 * lq);
  __syncthreads();
  issue(g.Wstem, 0);
  if (g.boards) {
    constexpr int WPB = (int)(sizeof(fpc_board) / 4);
    if (tid < WPB) reinterpret_cast<uint32_t *>(lboard)[tid] =
        reinterpret_cast<const uint32_t *>(g.boards + (size_t)game * g.board_stride + slot)[tid];
    __syncthreads();
    if (g.rules & FPC_RULES_ROTATION) rot_k = lboard->turn;
    for (int r = tid; r < 256; r += TW_THREADS) {
      const int pi = r >> 4, pj = r & 15;
      if (pi < 1 || pi > R || pj < 1 || pj > R) continue;
      const uint8_t p = lboard->sq[rot90_src(R, rot_k, pi - 1, pj - 1)];
      if (!present(p)) continue;
      const int plane = piece_plane(p, lboard->turn, g.rules);
      *reinterpret_cast<uint16_t *>(enc + tw_lay(4, r, plane >> 3) + (plane & 7) * 2) = g.one16;
    }
  } else {
    const uint16_t *src = g.in16 + (size_t)game * 256 * 32;
    for (int c = tid; c < NR * 4; c += TW_THREADS) {
      const int r = c >> 2, j = c & 3;
      *reinterpret_cast<t_u32x4 *>(enc + tw_lay(4, r, j)) = *reinterpret_cast<const t_u32x4 *>(src + (size_t)r * 32 + j * 8);
    }
  }

  t_f32x4 acc[MT][8];
  t_u32x2 res[MT][8];
  auto brow = [&](int mt, int shift) -> int {
    int r = rbase + 16 * mt + shift;
    r = r < 0 ? r + NR : r;
    r = r >= NR ? r - NR : r;
    return r;
  };

  // ---- stem: one slab (= one tap, K = 32) per step through the two buffers ------------------------
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    if (tap + 1 < 9) { issue(g.Wstem + (size_t)(tap + 1) * T2_SLAB, (tap + 1) & 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                           // slab `tap` landed (every wave waited for its pieces); the input image is complete
    const int shift = (tap / 3 - 1) * P + (tap % 3 - 1);
    const unsigned char *sl = ring + (tap & 1) * T2_SLAB + cb128 * 64 + apart_s;
    t_u32x4 fa[8], fb[MT];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) fa[ct] = *reinterpret_cast<const t_u32x4 *>(sl + ct * 1024);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = brow(mt, shift);
      fb[mt] = *reinterpret_cast<const t_u32x4 *>(enc + (r >> 3) * 512 + (r & 7) * 16 + bq);
    }
    if (tap == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[mt][ct] = M16<DT>::mfma(fa[ct], fb[mt], bst[ct]);
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[mt][ct] = M16<DT>::mfma(fa[ct], fb[mt], acc[mt][ct]);
    }
    __syncthreads();                           // the buffer of this slab may be refilled two taps on
  }
  // wipe the stem's input image (it sits where x_0 is about to be written; borders must read as zero)
  for (int c = tid; c < NR * 64 / 16; c += TW_THREADS) reinterpret_cast<t_u32x4 *>(enc)[c] = t_u32x4{0u, 0u, 0u, 0u};

  // ---- the main weight stream -----------------------------------------------------------------------
  const int total = (g.L + 2) * 9 * T2_KS;     // slabs
  auto issue_slab = [&](int gk) {              // slab gk -> buffer gk % 2; first slab of a layer: + its biases.  Clamped past the end.
    const int gc = gk < total ? gk : total - 1;
    issue(g.Wt + (size_t)gc * T2_SLAB, gk & 1);
    if (gk % (9 * T2_KS) == 0 && gk < total) {
      const int l = gk / (9 * T2_KS);
      tw_dma_256(reinterpret_cast<const unsigned char *>(g.bt + (size_t)l * 256 + wave * 64), (uint32_t)lane * 4u,
                 (uint32_t)__builtin_amdgcn_readfirstlane(TW_BIAS + ((l & 1) * 256 + wave * 64) * 4));
    }
  };
  __syncthreads();                             // wipe complete, stem buffers free
  issue_slab(0);
  issue_slab(1);

  unsigned char *const dummy = smem + TW_DUMMY + lane * 8;
  unsigned char *const wbase = img + (rbase >> 3) * 4096 + (rbase & 7) * 16 + wn * 2048 + (lane >> 5) * 128 + ((lane >> 4) & 1) * 8;
  // epilogue (see k_tower): accumulators hold conv + bias; (+ residual); 16-bit; ReLU on packed pairs; in place
  auto epilogue = [&](auto res_c) {
    constexpr int RES = decltype(res_c)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      unsigned char *dst = ((inmask >> mt) & 1u) ? wbase + mt * 8192 : dummy;
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        t_f32x4 v = acc[mt][ct];
        if (RES == 2) {
          v[0] += M16<DT>::lo(res[mt][ct][0]); v[1] += M16<DT>::hi(res[mt][ct][0]);
          v[2] += M16<DT>::lo(res[mt][ct][1]); v[3] += M16<DT>::hi(res[mt][ct][1]);
        }
        const t_u32x2 pk = t_u32x2{tw_relu2(M16<DT>::pack2(v[0], v[1])), tw_relu2(M16<DT>::pack2(v[2], v[3]))};
        if (RES != 0) res[mt][ct] = pk;
        *reinterpret_cast<t_u32x2 *>(((inmask >> mt) & 1u) ? dst + ct * 256 : dst) = pk;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const std::integral_constant<int, 0> c0{};
  const std::integral_constant<int, 1> c1{};
  const std::integral_constant<int, 2> c2{};
  epilogue(c1);                                // stem: x_0 = relu(conv + b)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                             // x_0 complete, slabs 0 and 1 (and layer 0's biases) landed

  // ---- 9 taps x 8 k-steps per layer ---------------------------------------------------------------------
  t_u32x4 fa[2][8], fb[MT];
  int gk = 0;                                  // running slab index
  const unsigned char *bbase, *blast;          // this lane's image row in its first / last row tile under the current tap's shift
  const unsigned char *bbase_n, *blast_n;      // ... under the next tap's shift
  auto tap_addr = [&](int tap, const unsigned char *&b0, const unsigned char *&bl) {
    const int shift = (tap / 3 - 1) * P + (tap % 3 - 1);
    const int r = rbase + shift;               // >= -1; row -1 lands in the 4 KiB in front of the image
    b0 = img + (r >> 3) * 4096 + (r & 7) * 16 + bq;
    int rl = r + 16 * (MT - 1);
    rl = rl >= NR ? rl - NR : rl;
    bl = img + (rl >> 3) * 4096 + (rl & 7) * 16 + bq;
  };
  auto load_b = [&]() {                        // all image fragments of k-step 0 of the current tap (after an epilogue)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fb[mt] = *reinterpret_cast<const t_u32x4 *>(mt == MT - 1 ? blast : bbase + mt * 8192);
  };
  // One k-step (slab gk): barrier that publishes slab gk + 1 and frees the buffer of slab gk (whose
  // fragments every wave already holds in registers); DMA of slab gk + 2; then 7 groups of 8 MFMAs, each
  // followed by the in-place reload of its image fragment for the NEXT k-step and by one or two weight
  // fragments of slab gk + 1.  MODE 0: 128 output channels per wave; MODE 1 (value conv): 16.
  //   KS: k-step within the tap (7: the next k-step belongs to the next tap -> its row addresses)
  auto kstep = [&](auto mode_c, auto buf_c, auto ks_c, auto bias_c, const t_f32x4 *b8, const int cb_next) {
    constexpr int MODE = decltype(mode_c)::value, B = decltype(buf_c)::value, KS = decltype(ks_c)::value;
    constexpr bool BIAS = decltype(bias_c)::value != 0;
    constexpr int N = B ^ 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#if T2_STREAM
    // slab gk + 2: M0 and the source base once, right behind the barrier (no LDS read in flight: an s_mov to M0
    // waits for those); its four pieces then go out between the first MFMA groups through the instruction's
    // immediate (fpc_tower.h: tw_dma_piece), where they cost ~6 cycles each instead of ~64 as a burst
    const unsigned char *dsrc;
    {
      const int gn = gk + 2, gc = gn < total ? gn : total - 1;
      if (gn % (9 * T2_KS) == 0 && gn < total) {
        const int l = gn / (9 * T2_KS);
        tw_dma_256(reinterpret_cast<const unsigned char *>(g.bt + (size_t)l * 256 + wave * 64), (uint32_t)lane * 4u,
                   (uint32_t)__builtin_amdgcn_readfirstlane(TW_BIAS + ((l & 1) * 256 + wave * 64) * 4));
      }
      dsrc = g.Wt + (size_t)gc * T2_SLAB + wave * 4096;
      tw_set_m0((uint32_t)__builtin_amdgcn_readfirstlane(T2_RING + (gn & 1) * T2_SLAB + wave * 4096));
    }
#else
    issue_slab(gk + 2);
#endif
    const unsigned char *sl = ring + ((gk + 1) & 1) * T2_SLAB + cb_next * 64 + apart_s;
    ++gk;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (MODE == 0) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[mt][ct] = M16<DT>::mfma(fa[B][ct], fb[mt], BIAS ? b8[ct] : acc[mt][ct]);
      } else {
        acc[mt][0] = M16<DT>::mfma(fa[B][0], fb[mt], BIAS ? b8[0] : acc[mt][0]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (KS < T2_KS - 1) fb[mt] = *reinterpret_cast<const t_u32x4 *>((mt == MT - 1 ? blast : bbase + mt * 8192) + (KS + 1) * 512);
      else fb[mt] = *reinterpret_cast<const t_u32x4 *>(mt == MT - 1 ? blast_n : bbase_n + mt * 8192);
      fa[N][mt] = *reinterpret_cast<const t_u32x4 *>(sl + mt * 1024);
      if (mt == MT - 1) fa[N][7] = *reinterpret_cast<const t_u32x4 *>(sl + 7 * 1024);
      __builtin_amdgcn_sched_barrier(0);
#if T2_STREAM
      if (mt == 0) tw_dma_piece<0>(dsrc, dma_lane);
      if (mt == 1) tw_dma_piece<1024>(dsrc, dma_lane);
      if (mt == 2) tw_dma_piece<2048>(dsrc, dma_lane);
      if (mt == 3) tw_dma_piece<3072>(dsrc, dma_lane);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
  const std::integral_constant<int, 3> c3{};
  const std::integral_constant<int, 4> c4{};
  const std::integral_constant<int, 5> c5{};
  const std::integral_constant<int, 6> c6{};
  const std::integral_constant<int, 7> c7{};
  // One conv layer.  On entry fa[0] holds slab gk's fragments for this wave (cb), fb the image fragments
  // of (tap 0, k-step 0), bbase / blast the tap-0 addresses.  cb_next: first weight row of the NEXT layer.
  auto run_layer = [&](auto mode_c, const int layer, const int cb, const int cb_next) {
    constexpr int MODE = decltype(mode_c)::value;
    t_f32x4 b8[8];
    {
      const float *bl = biasbuf + (layer & 1) * 256 + cb + 4 * lq;
#pragma unroll
      for (int ct = 0; ct < (MODE == 0 ? 8 : 1); ++ct) b8[ct] = *reinterpret_cast<const t_f32x4 *>(bl + ct * 16);
    }
    tap_addr(1, bbase_n, blast_n);
    kstep(mode_c, c0, c0, c1, b8, cb);         // (tap 0, k-step 0), C = bias
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      kstep(mode_c, c1, c1, c0, b8, cb);
      kstep(mode_c, c0, c2, c0, b8, cb);
      kstep(mode_c, c1, c3, c0, b8, cb);
      kstep(mode_c, c0, c4, c0, b8, cb);
      kstep(mode_c, c1, c5, c0, b8, cb);
      kstep(mode_c, c0, c6, c0, b8, cb);
      kstep(mode_c, c1, c7, c0, b8, tap == 8 ? cb_next : cb);     // reads (next tap, k-step 0)
      bbase = bbase_n; blast = blast_n;                            // after tap 8: the next layer's tap 0
      tap_addr((tap + 2) % 9, bbase_n, blast_n);
      if (tap < 8) kstep(mode_c, c0, c0, c0, b8, cb);             // (next tap, k-step 0)
    }
  };

  tap_addr(0, bbase, blast);
  {
    const unsigned char *sl = ring + cb128 * 64 + apart_s;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) fa[0][ct] = *reinterpret_cast<const t_u32x4 *>(sl + ct * 1024);
  }
  load_b();
  const int nblocks = g.L / 2;
#pragma unroll 1
  for (int blk = 0; blk < nblocks; ++blk) {
    run_layer(c0, 2 * blk, cb128, cb128);                               // conv1 + BN + ReLU
    epilogue(c0);
    __syncthreads();
    load_b();
    run_layer(c0, 2 * blk + 1, cb128, blk + 1 == nblocks ? cb16 : cb128);   // conv2 + BN, + x_l, ReLU
    epilogue(c2);
    __syncthreads();
    load_b();
  }
  float vpart = 0.f;
  {
    run_layer(c1, g.L, cb16, cb128);                                    // value conv: 32 live channels, 16 per (wn)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int r = rbase + 16 * mt, pi = r >> 4, pj = r & 15;
      const bool in = (inmask >> mt) & 1u;
      const int qp = in ? (pi - 1) * R + (pj - 1) : 0;
      const t_f32x4 w4 = *reinterpret_cast<const t_f32x4 *>(g.vw + qp * 32 + cb16 + 4 * lq);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc[mt][0][j];
        v = v > 0.f ? v : 0.f;
        vpart += in ? v * w4[j] : 0.f;
      }
    }
    // (the value conv leaves the image as it was: the fragments reloaded during its last k-step are valid)
  }
  run_layer(c0, g.L + 1, cb128, cb128);                                 // policy conv + BN + ReLU
  epilogue(c0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the clamped tail slabs: nothing may still target LDS at exit
  __syncthreads();

  {
    const int cpr = g.A_ch / 8;
    for (int c = tid; c < 256 * 16; c += TW_THREADS) {
      const int r = c >> 4, j = c & 15;
      if (j >= cpr) continue;
      const int pi = r >> 4, pj = r & 15;
      if (pi < 1 || pi > R || pj < 1 || pj > R) continue;
      const int q = (pi - 1) * R + (pj - 1);
      *reinterpret_cast<t_u32x4 *>(g.xfc + (size_t)game * g.Kp + (size_t)q * g.A_ch + j * 8) =
          *reinterpret_cast<const t_u32x4 *>(img + tw_lay(32, r, j));
    }
  }
  for (int off = 32; off >= 1; off >>= 1) vpart += __shfl_xor(vpart, off);
  if (lane == 0) vred[wave] = vpart;
  __syncthreads();
  if (tid == 0) g.value[game] = tanhf(g.vb + ((vred[0] + vred[1]) + (vred[2] + vred[3])));
}

// weights [taps][cout_pad rows][256 cin] 16-bit row-major -> slabs [tap][k-step][256 rows][4 chunks] in LDS-image
// order (rows >= cout_pad zero)
__global__ void __launch_bounds__(256) k_tower256_prep(const uint16_t *W, unsigned char *out, int taps, int cout_pad) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;             // one 16-byte chunk of the output
  if (c >= (long)taps * T2_KS * 256 * 4) return;
  const int j = (int)(c & 3), row = (int)((c >> 2) & 255), ks = (int)((c >> 10) % T2_KS), tap = (int)(c / (1024 * T2_KS));
  t_u32x4 v = t_u32x4{0u, 0u, 0u, 0u};
  if (row < cout_pad) v = *reinterpret_cast<const t_u32x4 *>(W + ((size_t)tap * cout_pad + row) * 256 + ks * 32 + j * 8);
  *reinterpret_cast<t_u32x4 *>(out + ((size_t)tap * T2_KS + ks) * T2_SLAB + tw_lay(4, row, j)) = v;
}
// stem weights [9][256][32] -> 9 slabs
__global__ void __launch_bounds__(256) k_tower256_prep_stem(const uint16_t *W, unsigned char *out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 9 * 256 * 4) return;
  const int j = c & 3, row = (c >> 2) & 255, tap = c >> 10;
  *reinterpret_cast<t_u32x4 *>(out + (size_t)tap * T2_SLAB + tw_lay(4, row, j)) =
      *reinterpret_cast<const t_u32x4 *>(W + ((size_t)tap * 256 + row) * 32 + j * 8);
}

}  // namespace fpc
