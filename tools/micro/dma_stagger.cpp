// micro-benchmark: what an LDS-DMA piece (global_load_lds_dwordx4, 1 KiB) costs the issuing wave inside an MFMA
// stream (one wave per SIMD, 4 waves per CU), as a function of WHEN the four waves of the CU issue theirs:
//   mode 0: no DMA at all (the MFMA stream alone)
//   mode 1: every wave issues its P pieces at the same places of the k-step (what k_fc / k_tower do: same program)
//   mode 2: the places are rotated by wave number (code specialised per wave: no branch in the stream), so that
//           the CU's vector-memory path sees one wave's piece at a time
//   mode 3: all P pieces of a k-step in one burst at its start
// Every block streams the same 4 x 512 KiB (one window per wave number): L2 resident, like k_tower's weights.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

__device__ __forceinline__ void dma1k_m0set(const void *gsrc_uniform, uint32_t lane_off) {   // M0 already holds the LDS address
  asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(gsrc_uniform) : "memory");
}
__device__ __forceinline__ void dma1k(const void *gsrc_uniform, uint32_t lane_off, uint32_t lds_addr_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off), "s"(gsrc_uniform), "s"(lds_addr_uniform) : "memory");
}

constexpr int NM = 16;   // MFMAs (32x32x16) per k-step: 512 matrix-pipe cycles

template <int MODE, int P, int W, int RD, int M0ONCE>
__device__ __forceinline__ void body(const unsigned char *src, float *out, int iters, unsigned char *smem) {
  const int lane = threadIdx.x & 63;
  f32x16_t acc[8];
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u32x4_t fa = {0x3f003f00u + lane, 0x3f013f00u, 0x3f003f02u, 0x3f003f00u}, fb = {0x3f003f00u, 0x3f103f00u + lane, 0x3f003f00u, 0x3f203f00u};
  const uint32_t lane_off = lane * 16u;
  const unsigned char *base = src + (size_t)W * (512u << 10);   // 512 KiB window per wave number, shared by all blocks
  const uint32_t lds0 = W * 32768u;                                                // 32 KiB ring per wave
  int piece = 0;
  u32x4_t rd[4] = {fa, fb, fa, fb};
  const unsigned char *rbase = smem + 4 * 32768 + W * 4096 + lane * 16;
  if (M0ONCE) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0) : "memory");
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, fb), acc[m & 7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bool here = false;
      if (MODE == 1) here = (m % (NM / P)) == 0;                               // same places for every wave
      if (MODE == 2) here = ((m + NM - W * (NM / P / 4 > 0 ? NM / P / 4 : 1)) % (NM / P)) == 0 && true;   // rotated by wave
      if (MODE == 3) here = false;
      if (MODE == 3 && m == 0) {
#pragma unroll
        for (int p = 0; p < P; ++p) { dma1k(base + (size_t)((piece + p) & 511) * 1024, lane_off, (uint32_t)__builtin_amdgcn_readfirstlane(lds0 + ((piece + p) & 31) * 1024)); }
        piece += P;
      }
      if (here) {
        if (M0ONCE) dma1k_m0set(base + (size_t)(piece & 511) * 1024, lane_off);
        else dma1k(base + (size_t)(piece & 511) * 1024, lane_off, (uint32_t)__builtin_amdgcn_readfirstlane(lds0 + (piece & 31) * 1024));
        ++piece;
      }
      if (RD && (m % (NM / RD)) == 0) {          // RD fragment reads per k-step, consumed one k-step later (kept in flight)
        const int j = (m / (NM / RD)) & 3;
        fa[j & 3] ^= rd[j][0] & 1u;               // use the OLD value (forces the previous read's wait here, a k-step later)
        rd[j] = *reinterpret_cast<const u32x4_t *>(rbase + j * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");   // two k-steps of pieces stay in flight
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
  for (int a = 0; a < 8; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s + smem[threadIdx.x];
}

template <int MODE, int P, int RD, int M0ONCE>
__global__ void __launch_bounds__(256, 1) k(const unsigned char *src, float *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (MODE != 2) {
    // one code path; W only selects the wave's window / ring
    if (wave == 0) body<MODE, P, 0, RD, M0ONCE>(src, out, iters, smem);
    else if (wave == 1) body<MODE, P, 1, RD, M0ONCE>(src, out, iters, smem);
    else if (wave == 2) body<MODE, P, 2, RD, M0ONCE>(src, out, iters, smem);
    else body<MODE, P, 3, RD, M0ONCE>(src, out, iters, smem);
  } else {
    if (wave == 0) body<2, P, 0, RD, M0ONCE>(src, out, iters, smem);
    else if (wave == 1) body<2, P, 1, RD, M0ONCE>(src, out, iters, smem);
    else if (wave == 2) body<2, P, 2, RD, M0ONCE>(src, out, iters, smem);
    else body<2, P, 3, RD, M0ONCE>(src, out, iters, smem);
  }
}

template <int MODE, int P, int RD = 0, int M0ONCE = 0>
void run(const unsigned char *d, float *o, const char *name) {
  const int blocks = 256, iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE, P, RD, M0ONCE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, P, RD, M0ONCE>), dim3(blocks), dim3(256), 131072, 0, d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double fl = (double)blocks * 4 * iters * NM * 32768.0;
  printf("%-34s RD=%d M0once=%d P=%d  %.3f ms  %.1f ns per k-step  %.1f TFLOP/s  %.2f TB/s DMA\n", name, RD, M0ONCE, P, best, best * 1e6 / iters, fl / best / 1e9,
         MODE ? (double)blocks * 4 * iters * P * 1024.0 / best / 1e9 : 0.0);
}

int main() {
  const size_t bytes = (size_t)4 * (512u << 10);   // 2 MiB
  unsigned char *d; float *o;
  hipMalloc(&d, bytes); hipMalloc(&o, 256 * 256 * 4);
  hipMemset(d, 0x3c, bytes);
  run<0, 4>(d, o, "MFMA stream alone");
  run<0, 4, 8>(d, o, "MFMA + 8 fragment reads");
  run<0, 4, 16>(d, o, "MFMA + 16 fragment reads");
  run<1, 4>(d, o, "same places, every wave");
  run<1, 4, 8>(d, o, "same places, every wave");
  run<1, 4, 16>(d, o, "same places, every wave");
  run<1, 4, 16, 1>(d, o, "same places, every wave");
  run<2, 4, 16>(d, o, "places rotated by wave");
  run<3, 4, 16>(d, o, "burst at the k-step's start");
  run<1, 8>(d, o, "same places, every wave");
  run<1, 8, 16>(d, o, "same places, every wave");
  run<1, 8, 16, 1>(d, o, "same places, every wave");
  run<2, 8, 16>(d, o, "places rotated by wave");
  run<3, 8, 16>(d, o, "burst at the k-step's start");
  run<3, 8, 16, 1>(d, o, "burst at the k-step's start");
  return 0;
}
