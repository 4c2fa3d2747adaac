// fpc_platform.h -- the one place that knows whether the tree kernels are being built by hipcc for
// gfx950 (the product) or by g++ against tests/emul/wave_emul.h (a lock-step 64-lane wavefront
// emulator used ONLY by the CPU test-suite to exercise the very same kernel source without a GPU;
// the product library never contains it).
#pragma once

#ifdef FPC_EMUL
#include "wave_emul.h"
#else
#include <hip/hip_runtime.h>
#define FPC_LAUNCH(kernel, grid, block, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif

#include <stdint.h>

// wave_read(v, lane): the value lane `lane` holds, for a lane index that is the SAME in every lane of the wave
// (v_readlane_b32: a few cycles through the scalar unit, against ~100 for the LDS-crossbar shuffle a general
// __shfl costs on a lone wave).  The emulator models it as the shuffle it is.
#ifdef FPC_EMUL
template <class T>
inline T wave_read(T v, int lane_uniform) { return __shfl(v, lane_uniform); }
#else
__device__ __forceinline__ int wave_read(int v, int lane_uniform) { return __builtin_amdgcn_readlane(v, lane_uniform); }
__device__ __forceinline__ float wave_read(float v, int lane_uniform) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_uniform));
}
__device__ __forceinline__ double wave_read(double v, int lane_uniform) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane_uniform);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane_uniform);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
#endif
