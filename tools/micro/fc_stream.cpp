// micro-benchmark (round 5): what the memory system delivers for the policy Linear's ACCESS PATTERN, with the
// arithmetic taken out or left in.  k_fc16 (csrc/fpc_fc.h) moves 1.11 GB of weights per launch at 4.5 TB/s while
// LDS-DMA streams reach 6.5 TB/s elsewhere (MI355X_MICROARCH.md, ldsdma-fill); this program runs the same DMA
// issue pattern -- 1-KiB pieces, a 16 KiB ring per wave, four k32-steps in flight -- over
//   layout 0: the exporter's order [k32-step][column tile of 16][lane][8]  (a block's 16 KiB per k-step, stride Np/16 KiB)
//   layout 1: block-contiguous     [column group of 256][k32-step][16 KiB]
//   decomp 0: k_fc16's 64 x 4 long + 28 x 8 short blocks (480), id order as in the kernel
//   decomp 1: 256 persistent blocks, each 23/64 of a column group's K (stream-K: 92 groups / 256 CUs)
//   decomp 2: 92 x 2 (184 blocks, half the slabs; 72 % of the CUs)
//   x 0/1:    the activation stream beside it (X[256][Kp], L2 resident), same piece shapes as the kernel
//   mfma 0/1: 64 v_mfma_f32_16x16x32_f16 per k32-step and wave on register operands (no LDS reads)
//   slab 0/1: the f32 partial tile [256][256] written at the end of every segment
// usage: fc_stream [reps]   -> one line per combination
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void dma2(const void *g, uint32_t lane_off, uint32_t lds, bool nt) {
  uint32_t keep;
  if (nt)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(g), "s"(lds) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void dma1(const void *g, uint32_t lane_off, uint32_t lds) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off), "s"(g), "s"(lds) : "memory");
}

__device__ __forceinline__ const unsigned char *uni(const unsigned char *p) {   // tell the compiler the pointer is wave-uniform
  const unsigned long long v = (unsigned long long)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const unsigned char *)(((unsigned long long)hi << 32) | lo);
}
struct Seg { long long w_off; long long w_stride; int nk; int k0; int slab; int pad; };   // weight bytes of wave 0 at k32-step j: w_off + j * w_stride
struct Work { Seg s[2]; int nseg; int pad[3]; };

constexpr int XBUF = 256 * 64 * 2, WRING = 16 * 1024, LDS = 3 * XBUF + 4 * WRING;

template <int WITHX, int MFMA, int NT>
__global__ void __launch_bounds__(256, 1) k_stream(const unsigned char *W, const unsigned char *X, float *part, const Work *work, int Kp, int slabs_on, int xtiled, int depth = 4, int withw = 1, int regx = 0, int regw = 0, int xthird = 0, int rot = 0, int wide = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Work wk = work[blockIdx.x];
  reinterpret_cast<volatile uint32_t *>(smem)[threadIdx.x] = 0;      // the kernel USES its dynamic LDS (the DMA writes are invisible to the compiler)
  const uint32_t wlane = lane * 16u;
  const uint32_t lds_w = 3 * XBUF + wave * WRING;
  // row-major X[256][pitch Kp]: a piece = 8 rows x 128 B; tiled X[k64][256][64]: a piece = 1 KiB contiguous
  const uint32_t xlane = xtiled ? (uint32_t)lane * 16u : (uint32_t)((lane >> 3) * Kp * 2 + (lane & 7) * 16);
  f32x4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t fa = {0x3c003c00u + lane, 0x3c013c00u, 0x3c003c02u, 0x3c003c00u}, fb = {0x3c003c00u, 0x3c103c00u + lane, 0x3c003c00u, 0x3c203c00u};
  for (int si = 0; si < wk.nseg; ++si) {
    const Seg sg = wk.s[si];
    const unsigned char *wb = W + sg.w_off + wave * (wide ? 6144 : 4096);
    const unsigned char *xb = xtiled ? X + (long)(sg.k0 >> 1) * 32768 + wave * 8192 : X + ((long)(wave * 64) * Kp + (long)sg.k0 * 32) * 2;
    const int r0 = rot ? (int)((blockIdx.x * 97u) % (unsigned)sg.nk) : 0;
    auto issue = [&](int j, auto u_c) {
      constexpr int U = decltype(u_c)::value;      // j & 3, compile-time: the ring slot of the register-load variants
      int jc = j < sg.nk ? j : sg.nk - 1;
      jc += r0; jc -= jc >= sg.nk ? sg.nk : 0;
      const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane(lds_w + (j & 3) * 4096);   // (a deeper run-ahead than 4 reuses ring slots early: the data is never read here)
      const unsigned char *src = uni(wb + (long)jc * sg.w_stride);
      if (withw) {
        dma2(src, wlane, dst, NT);
        dma2(src + 2048, wlane, dst + 2048, NT);
        if (wide) dma2(src + 4096, wlane, dst + 4096 - 2048, NT);      // (ring slots alias in this variant: the data is never read)
      }
      if (WITHX && !(xthird && j % 3 == 2)) {        // 4 pieces of 8 rows x 128 B per k32-step (a BK = 64 stage is 8 pieces over two k32-steps)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int pp = (j & 1) * 4 + p;
          dma1(uni(xtiled ? xb + (long)(jc >> 1) * 32768 + pp * 1024 : xb + (long)pp * 8 * Kp * 2 + (long)(jc >> 1) * 128), xlane,
               (uint32_t)__builtin_amdgcn_readfirstlane(((j >> 1) % 3) * XBUF + (wave * 64 + pp * 8) * 128));
        }
      }
    };
    const std::integral_constant<int, 0> c0{}; const std::integral_constant<int, 1> c1{}; const std::integral_constant<int, 2> c2{}; const std::integral_constant<int, 3> c3{};
    auto issue_rt = [&](int j) { switch (j & 3) { case 0: issue(j, c0); break; case 1: issue(j, c1); break; case 2: issue(j, c2); break; default: issue(j, c3); break; } };
    for (int j = 0; j < depth; ++j) issue_rt(j);
#pragma unroll 1
    for (int j = 0; j < sg.nk; ++j) {
      {   // pieces per step: 4 W (if any) + 4 X (if any); allow depth - 1 steps outstanding
        const int per = (withw ? (wide ? 6 : 4) : 0) + (WITHX ? 4 : 0), allow = per * (depth - 1);
        if (allow >= 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
        else if (allow >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (allow >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        else if (allow >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (allow >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (allow >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (allow >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (allow >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (MFMA) {
#pragma unroll
        for (int m = 0; m < 96; ++m) {
          if (m >= 64 && !wide) break;
          acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa), __builtin_bit_cast(f16x8_t, fb), acc[m & 15], 0, 0, 0);
        }
      }
      issue_rt(j + depth);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (slabs_on) {
      float *out = part + (long)sg.slab * 256 * (wide ? 384 : 256);
      if (wide) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int m = t * 16 + (lane & 15);
#pragma unroll
          for (int n = 0; n < 6; ++n)
            *reinterpret_cast<f32x4_t *>(out + (long)m * 384 + wave * 96 + n * 16 + (lane >> 4) * 4) = acc[(t + n) & 15];
        }
      } else
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int m = t * 16 + (lane & 15);
#pragma unroll
        for (int n = 0; n < 4; ++n)
        {
          f32x4_t *dst = reinterpret_cast<f32x4_t *>(out + (long)m * 256 + wave * 64 + n * 16 + (lane >> 4) * 4);
          if (slabs_on == 2) __builtin_nontemporal_store(acc[(t + n) & 15], dst);
          else if (slabs_on == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(acc[(t + n) & 15]) : "memory");
          else *dst = acc[(t + n) & 15];
        }
      }
    }
  }
  if ((MFMA && acc[0][0] == 123.456f) || reinterpret_cast<volatile uint32_t *>(smem)[LDS / 4 - 1 - threadIdx.x] == 0x12345u) part[0] = acc[1][1];
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const int Np = 23552, Kp = 23552, groups = Np / 256, nk = Kp / 32;   // 92 groups, 736 k32-steps
  unsigned char *W, *X; float *part; Work *dwork;
  const size_t wbytes = (size_t)Np * Kp * 2;
  CK(hipMalloc(&W, wbytes + (32 << 20))); CK(hipMalloc(&X, (size_t)256 * (Kp + 512) * 2 + (1 << 20))); CK(hipMalloc(&part, (size_t)1024 * 256 * 256 * 4)); CK(hipMalloc(&dwork, 1024 * sizeof(Work)));
  CK(hipMemset(W, 0x3c, wbytes)); CK(hipMemset(X, 0x3c, (size_t)256 * Kp * 2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto seg = [&](int layout, int group, int k0, int n, int slab) {
    Seg s{};
    if (layout == 0) { s.w_off = ((long long)k0 * (Np / 16) + group * 16) * 1024; s.w_stride = (long long)(Np / 16) * 1024; }
    else { s.w_off = ((long long)group * nk + k0) * 16384; s.w_stride = 16384; }
    s.nk = n; s.k0 = k0; s.slab = slab;
    return s;
  };
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  if (mode == 2) {
    // sweep 3: run-ahead depth, and the X stream alone
    std::vector<Work> work; int slabs = 0;
    const int G1 = 64, s1 = 4, s2 = 8;
    for (int id = 0; id < G1 * s1 + (groups - G1) * s2; ++id) {
      const bool big = id < G1 * s1; const int idr = big ? id : id - G1 * s1, sk = big ? s1 : s2;
      const int group = big ? idr / sk : G1 + idr / sk, split = idr % sk, KS = nk / sk;
      Work w{}; w.nseg = 1; w.s[0] = seg(0, group, split * KS, KS, slabs++); work.push_back(w);
    }
    CK(hipMemcpy(dwork, work.data(), work.size() * sizeof(Work), hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<0, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<1, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    struct C { int x, w, mf, depth; };
    const C cs[] = {{1, 1, 1, 4}, {1, 1, 1, 2}, {1, 1, 1, 3}, {1, 1, 1, 6}, {1, 1, 1, 7}, {0, 1, 1, 2}, {0, 1, 1, 4}, {0, 1, 1, 8}, {0, 1, 1, 12},
                    {1, 0, 1, 4}, {1, 0, 1, 8}, {1, 0, 0, 4}, {1, 0, 0, 8}, {1, 1, 0, 4}, {1, 1, 1, 4}};
    for (const C &c : cs) {
      auto launch = [&]() {
        if (!c.x) hipLaunchKernelGGL((k_stream<0, 1, 1>), dim3((unsigned)work.size()), dim3(256), LDS, 0, W, X, part, dwork, Kp + 64, 0, 0, c.depth, c.w);
        else if (c.mf) hipLaunchKernelGGL((k_stream<1, 1, 1>), dim3((unsigned)work.size()), dim3(256), LDS, 0, W, X, part, dwork, Kp + 64, 0, 0, c.depth, c.w);
        else hipLaunchKernelGGL((k_stream<1, 0, 1>), dim3((unsigned)work.size()), dim3(256), LDS, 0, W, X, part, dwork, Kp + 64, 0, 0, c.depth, c.w);
      };
      launch(); launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      printf("X %d W %d mfma %d depth %2d | %.1f us\n", c.x, c.w, c.mf, c.depth, ms * 1e3);
      fflush(stdout);
    }
    return 0;
  }
  if (mode == 3) {
    // sweep 4: register loads instead of LDS-DMA, X at 2/3 of the rate (384-column tiles), rotated K start (no two blocks on one line at a time)
    std::vector<Work> work; int slabs = 0;
    const int G1 = 64, s1 = 4, s2 = 8;
    for (int id = 0; id < G1 * s1 + (groups - G1) * s2; ++id) {
      const bool big = id < G1 * s1; const int idr = big ? id : id - G1 * s1, sk = big ? s1 : s2;
      const int group = big ? idr / sk : G1 + idr / sk, split = idr % sk, KS = nk / sk;
      Work w{}; w.nseg = 1; w.s[0] = seg(0, group, split * KS, KS, slabs++); work.push_back(w);
    }
    CK(hipMemcpy(dwork, work.data(), work.size() * sizeof(Work), hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<1, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    struct C { int mf, w, regx, regw, xthird, rot, tiled; };
    const C cs[] = {{0, 1, 0, 0, 0, 0, 0}, {1, 1, 0, 0, 0, 0, 0},      // baseline: DMA both
                    {0, 1, 0, 0, 1, 0, 0}, {1, 1, 0, 0, 1, 0, 0},      // X at 2/3 rate
                    {0, 1, 0, 0, 0, 1, 0}, {1, 1, 0, 0, 0, 1, 0},      // rotated K start
                    {0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 1, 0},      // X alone: aligned / rotated
                    {0, 0, 0, 0, 0, 0, 1}, {0, 0, 0, 0, 0, 1, 1}, {1, 1, 0, 0, 0, 0, 0}};
    for (const C &c : cs) {
      auto launch = [&]() {
        if (c.mf) hipLaunchKernelGGL((k_stream<1, 1, 1>), dim3((unsigned)work.size()), dim3(256), LDS, 0, W, X, part, dwork, Kp + 64, 0, c.tiled, 4, c.w, c.regx, c.regw, c.xthird, c.rot);
        else hipLaunchKernelGGL((k_stream<1, 0, 1>), dim3((unsigned)work.size()), dim3(256), LDS, 0, W, X, part, dwork, Kp + 64, 0, c.tiled, 4, c.w, c.regx, c.regw, c.xthird, c.rot);
      };
      for (int r = 0; r < reps / 3 + 2; ++r) launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      printf("mfma %d W %d | X by %s%s, W by %s | X rate %s | K start %s | %.1f us\n", c.mf, c.w, c.regx ? "regs" : "dma", c.tiled ? " (tiled)" : "", c.regw ? "regs" : "dma",
             c.xthird ? "2/3" : "1", c.rot ? "rotated" : "aligned", ms * 1e3);
      fflush(stdout);
    }
    return 0;
  }
  if (mode == 4) {
    // sweep 5: 256 x 384 tiles (62 column groups x 4 K-splits = 248 blocks, ONE round) against the 256 x 256 plan, slabs written
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<1, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    for (int rep = 0; rep < 2; ++rep)
    for (int wide = 0; wide < 2; ++wide) {
      std::vector<Work> work; int slabs = 0;
      if (!wide) {
        const int G1 = 64, s1 = 4, s2 = 8;
        for (int id = 0; id < G1 * s1 + (groups - G1) * s2; ++id) {
          const bool big = id < G1 * s1; const int idr = big ? id : id - G1 * s1, sk = big ? s1 : s2;
          const int group = big ? idr / sk : G1 + idr / sk, split = idr % sk, KS = nk / sk;
          Work w{}; w.nseg = 1; w.s[0] = seg(0, group, split * KS, KS, slabs++); work.push_back(w);
        }
      } else {
        const int NpW = 62 * 384;       // 23808: the weight matrix padded to whole 384-column groups ([k32][NpW / 16 tiles][1 KiB])
        for (int id = 0; id < 62 * 4; ++id) {
          const int group = id / 4, split = id % 4, KS = nk / 4;
          Work w{}; w.nseg = 1;
          Seg sg{}; sg.w_off = ((long long)split * KS * (NpW / 16) + group * 24) * 1024; sg.w_stride = (long long)(NpW / 16) * 1024; sg.nk = KS; sg.k0 = split * KS; sg.slab = slabs++;
          w.s[0] = sg; work.push_back(w);
        }
      }
      CK(hipMemcpy(dwork, work.data(), work.size() * sizeof(Work), hipMemcpyHostToDevice));
      for (int combo = 0; combo < 6; ++combo) {
        const int mf = combo < 4 ? (combo & 1) : 1, sl = combo < 4 ? (combo >> 1) : 1, tiled = combo == 4, pitch0 = combo == 5;
        auto launch = [&]() {
          if (mf) hipLaunchKernelGGL((k_stream<1, 1, 1>), dim3((unsigned)work.size()), dim3(256), LDS, 0, W, X, part, dwork, pitch0 ? Kp : Kp + 64, sl, tiled, 4, 1, 0, 0, 0, 0, wide);
          else hipLaunchKernelGGL((k_stream<1, 0, 1>), dim3((unsigned)work.size()), dim3(256), LDS, 0, W, X, part, dwork, Kp + 64, sl, 0, 4, 1, 0, 0, 0, 0, wide);
        };
        for (int r = 0; r < reps / 3 + 2; ++r) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("tile 256 x %d  blocks %zu | mfma %d slabs %d X %s | %.1f us\n", wide ? 384 : 256, work.size(), mf, sl, tiled ? "tiled" : pitch0 ? "rows pitch Kp" : "rows pitch Kp+64", ms * 1e3);
        fflush(stdout);
      }
    }
    return 0;
  }
  if (mode == 1) {
    // sweep 2: decomp 0, layout 0, mfma on; X pitch / tiling and slab store flavour
    std::vector<Work> work; int slabs = 0;
    const int G1 = 64, s1 = 4, s2 = 8;
    for (int id = 0; id < G1 * s1 + (groups - G1) * s2; ++id) {
      const bool big = id < G1 * s1; const int idr = big ? id : id - G1 * s1, sk = big ? s1 : s2;
      const int group = big ? idr / sk : G1 + idr / sk, split = idr % sk, KS = nk / sk;
      Work w{}; w.nseg = 1; w.s[0] = seg(0, group, split * KS, KS, slabs++); work.push_back(w);
    }
    CK(hipMemcpy(dwork, work.data(), work.size() * sizeof(Work), hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<0, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int pitches[] = {Kp, Kp + 32, Kp + 64, Kp + 128, Kp + 192, Kp + 320, -1 /* tiled */, 0 /* no X */};
    for (int pi = 0; pi < 8; ++pi)
      for (int sl = 0; sl < 4; ++sl) {
        const int pitch = pitches[pi];
        auto launch = [&]() {
          if (pitch == 0) hipLaunchKernelGGL((k_stream<0, 1, 1>), dim3((unsigned)work.size()), dim3(256), LDS, 0, W, X, part, dwork, Kp, sl, 0);
          else hipLaunchKernelGGL((k_stream<1, 1, 1>), dim3((unsigned)work.size()), dim3(256), LDS, 0, W, X, part, dwork, pitch < 0 ? Kp : pitch, sl, pitch < 0 ? 1 : 0);
        };
        launch(); launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("X %s pitch %d | slabs %s | %.1f us  W %.2f TB/s\n", pitch == 0 ? "none" : pitch < 0 ? "tiled" : "rows", pitch,
               sl == 0 ? "off" : sl == 1 ? "plain" : sl == 2 ? "nt" : "sc0sc1", ms * 1e3, wbytes / (ms * 1e-3) / 1e12);
        fflush(stdout);
      }
    return 0;
  }
  for (int decomp = 0; decomp < 3; ++decomp)
    for (int layout = 0; layout < 2; ++layout) {
      std::vector<Work> work;
      int slabs = 0;
      if (decomp == 0) {
        const int G1 = 64, s1 = 4, s2 = 8;
        for (int id = 0; id < G1 * s1 + (groups - G1) * s2; ++id) {
          const bool big = id < G1 * s1; const int idr = big ? id : id - G1 * s1, sk = big ? s1 : s2;
          const int group = big ? idr / sk : G1 + idr / sk, split = idr % sk, KS = nk / sk;
          Work w{}; w.nseg = 1; w.s[0] = seg(layout, group, split * KS, KS, slabs++); work.push_back(w);
        }
      } else if (decomp == 1) {
        // linear space of 92 * 736 k32-steps cut into 256 equal pieces of 264.5 -> alternate 264 / 265
        long long pos = 0; const long long total = (long long)groups * nk;
        for (int b = 0; b < 256; ++b) {
          const long long end = total * (b + 1) / 256;
          Work w{}; w.nseg = 0;
          while (pos < end) {
            const int g = (int)(pos / nk), k0 = (int)(pos % nk);
            const int n = (int)std::min<long long>(end - pos, nk - k0);
            w.s[w.nseg++] = seg(layout, g, k0, n, slabs++); pos += n;
          }
          work.push_back(w);
        }
      } else {
        for (int id = 0; id < groups * 2; ++id) { Work w{}; w.nseg = 1; w.s[0] = seg(layout, id / 2, (id % 2) * (nk / 2), nk / 2, slabs++); work.push_back(w); }
      }
      CK(hipMemcpy(dwork, work.data(), work.size() * sizeof(Work), hipMemcpyHostToDevice));
      for (int combo = 0; combo < 6; ++combo) {
        // combos: 0 W only | 1 W + X | 2 W + mfma | 3 W + X + mfma | 4 W + X + mfma + slabs | 5 as 4 without nt
        const int withx = combo == 1 || combo >= 3, mf = combo >= 2, sl = combo >= 4, nt = combo != 5;
        auto launch = [&]() {
#define GO(X_, M_, N_) do { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stream<X_, M_, N_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
            hipLaunchKernelGGL((k_stream<X_, M_, N_>), dim3((unsigned)work.size()), dim3(256), LDS, 0, W, X, part, dwork, Kp, sl, 0); } while (0)
          if (!nt) GO(1, 1, 0);
          else if (withx && mf) GO(1, 1, 1); else if (withx) GO(1, 0, 1); else if (mf) GO(0, 1, 1); else GO(0, 0, 1);
#undef GO
        };
        launch(); launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("decomp %d layout %d blocks %3zu slabs %3d | X %d mfma %d slabs %d nt %d | %.1f us  W %.2f TB/s\n", decomp, layout, work.size(), slabs, withx, mf, sl, nt,
               ms * 1e3, wbytes / (ms * 1e-3) / 1e12);
        fflush(stdout);
      }
    }
  return 0;
}
