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
