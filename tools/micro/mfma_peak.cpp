// micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate, 8 waves/CU (2 per SIMD), operands in registers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__global__ void __launch_bounds__(512) k(const u32x4_t *in, float *out, int iters) {
  u32x4_t a0 = in[threadIdx.x], a1 = in[threadIdx.x + 512], b0 = in[threadIdx.x + 1024], b1 = in[threadIdx.x + 1536];
  f32x16_t c00, c01, c10, c11;
  for (int r = 0; r < 16; ++r) { c00[r] = 0; c01[r] = 0; c10[r] = 0; c11[r] = 0; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a0), __builtin_bit_cast(bf16x8_t, b0), c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a0), __builtin_bit_cast(bf16x8_t, b1), c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, b0), c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, b1), c11, 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c00[r] + c01[r] + c10[r] + c11[r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = 4000;
  std::vector<uint32_t> h(2048 * 4);
  for (size_t i = 0; i < h.size(); ++i) { uint32_t x = (uint32_t)(i * 2654435761u); h[i] = (argc > 2 && argv[2][0] == 'z') ? 0u : ((x & 0x807f807fu) | 0x3f003f00u); }
  u32x4_t *d; float *o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, blocks * 512 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 8 * iters * 32 * 32768.0;
    printf("blocks %d: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4GHz)\n", blocks, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 32.0 * 2));
  }
  return 0;
}
