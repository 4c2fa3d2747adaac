// micro-benchmark (round 5): what v_mfma_f32_16x16x32_f16 SUSTAINS on this part once the power management has settled --
// the ceiling the tower kernels can be priced against beside the 2.5 PFLOP/s of the data sheet.  8 waves per CU (two per
// SIMD), operands in registers, 16 independent accumulators per wave; successive launches of ~10 ms each are timed one
// by one, so the settling of the clock shows.  Operand patterns: dense (every element a different non-zero value), relu
// (every second 8-element group of the B operand zero: what a post-ReLU activation fragment looks like to the multipliers),
// zero (all-zero B: switching power at its minimum).
//    mfma_sustained [seconds per pattern]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// the tower kernels' operand pattern: per k-step 2 weight fragments x 7 image fragments (14 MFMAs), the fragments of two
// k-steps alternating -- 4 A and 14 B registers quads, every MFMA sees operands other than its predecessor's
__global__ void __launch_bounds__(512, 2) k2(const u32x4_t *in, float *out, int iters) {
  u32x4_t a[4], b[14];
  for (int i = 0; i < 4; ++i) a[i] = in[(threadIdx.x + 512 * i) & 2047];
  for (int i = 0; i < 14; ++i) b[i] = in[(threadIdx.x * 3 + 131 * i + 1024) & 2047];
  f32x4_t c[14];
  for (int i = 0; i < 14; ++i) c[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mt = 0; mt < 7; ++mt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          c[mt * 2 + ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a[(ks & 1) * 2 + ct]), __builtin_bit_cast(f16x8_t, b[(ks & 1) * 7 + mt]), c[mt * 2 + ct], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 14; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(512, 2) k(const u32x4_t *in, float *out, int iters) {
  u32x4_t a[2], b[2];
  a[0] = in[threadIdx.x]; a[1] = in[threadIdx.x + 512]; b[0] = in[threadIdx.x + 1024]; b[1] = in[threadIdx.x + 1536];
  f32x4_t c[16];
  for (int i = 0; i < 16; ++i) c[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u)
      c[u & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a[u & 1]), __builtin_bit_cast(f16x8_t, b[(u >> 1) & 1]), c[u & 15], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 0.6;
  const int blocks = 512, iters = 1500;        // 2 workgroups per CU
  u32x4_t *d; float *o;
  CK(hipMalloc(&d, 2048 * 16)); CK(hipMalloc(&o, blocks * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *names[6] = {"2+2 fixed operands, dense [0.5,1)", "2+2 fixed, relu (half of B's groups zero)", "2+2 fixed, zero B",
                          "tower pattern (4 A x 14 B), N(0,1)-like signed values", "tower pattern, B relu-like (half zero, rest positive)", "tower pattern, zero B"};
  for (int pat = 0; pat < 6; ++pat) {
    std::vector<uint32_t> h(2048 * 4);
    for (size_t i = 0; i < h.size(); ++i) {
      const uint32_t x = (uint32_t)(i * 2654435761u);
      uint32_t v = (x & 0x03ff03ffu) | 0x38003800u;                 // fp16 pairs in [0.5, 1)
      const bool isB = i >= 1024 * 4;
      if (pat >= 3) {                                                // random signs, exponents over 2^-4 .. 2^1, random mantissas
        const uint32_t y = x ^ (x >> 13) ^ (uint32_t)(i * 40503u);
        auto half = [](uint32_t r) { return ((r & 1u) << 15) | ((11u + ((r >> 1) % 6u)) << 10) | ((r >> 5) & 0x3ffu); };
        v = half(y) | (half(y >> 16 | y << 16) << 16);
        if (isB && pat == 4) v = ((i >> 2) & 1) ? 0u : (v & 0x7fff7fffu);
        if (isB && pat == 5) v = 0u;
      } else {
        if (isB && pat == 1 && ((i >> 2) & 1)) v = 0u;               // every second 16-byte group of B
        if (isB && pat == 2) v = 0u;
      }
      h[i] = v;
    }
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double fl = (double)blocks * 8 * iters * (pat >= 3 ? 56 : 64) * 16384.0;
    double t = 0.0; int n = 0; float first = 0.f, last = 0.f, minms = 1e9f;
    while (t < secs) {
      CK(hipEventRecord(e0));
      if (pat >= 3) hipLaunchKernelGGL(k2, dim3(blocks), dim3(512), 0, 0, d, o, iters);
      else hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, o, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (n == 0) first = ms;
      last = ms; if (ms < minms) minms = ms;
      t += ms * 1e-3; ++n;
    }
    printf("%-58s first launch %.1f TFLOP/s, best %.1f, settled (last of %d launches, %.2f s) %.1f TFLOP/s = %.3f of 2500\n", names[pat],
           fl / first / 1e9, fl / minms / 1e9, n, t, fl / last / 1e9, fl / last / 1e9 / 2500.0);
    fflush(stdout);
  }
  return 0;
}
