// tests/emul/wave_emul.h -- TEST INFRASTRUCTURE ONLY.
//
// A lock-step emulator of ONE 64-lane CDNA wavefront per block, so that the CPU test-suite
// (pytest -m "not gpu", no GPU in the build container) can execute the very same tree-kernel source
// that hipcc compiles for gfx950 (alphazero-4-player-chess_amd/csrc/fpc_tree_kernels.h).  Each lane
// is a ucontext fibre; wave collectives (__ballot, __shfl*) rendezvous the 64 fibres of a wave and
// __syncthreads the whole block (generation barriers that tolerate exited lanes, as the hardware
// barrier does), which reproduces the SIMT semantics the kernels rely on.  Blocks of up to 16 waves
// run one after another.
//
// This is a model of the HARDWARE, not of the algorithm, and it is never part of the product:
// libfpc_engine.so is built by hipcc only and has no CPU path.  The emulator build
// (tests/emul/libfpc_emul.so) exists so host logic + kernel logic get CPU coverage.
#pragma once

#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct wemu_dim3 {
  unsigned x, y, z;
};
struct float4 {
  float x, y, z, w;
};

namespace wemu {
constexpr int WAVE = 64;
constexpr int MAX_THREADS = 1024;
struct Bar {
  int count = 0, gen = 0;
};
struct State {
  ucontext_t main_ctx;
  ucontext_t lane_ctx[MAX_THREADS];
  char *stacks[MAX_THREADS];
  bool done[MAX_THREADS];
  int nthreads;
  int cur;                       // running fibre == threadIdx.x
  wemu_dim3 block_idx;
  // Collectives alternate between two exchange buffers, so ONE rendezvous each is enough (a lane can only reach
  // a buffer's next reuse through the rendezvous in between).  Every lane counts its collectives and stamps
  // what it posts: a source lane took part iff its stamp equals the reader's count -- lanes that left the
  // kernel earlier (or right after posting) are told apart without a second rendezvous.
  uint64_t slot[2][MAX_THREADS];
  uint32_t stamp[2][MAX_THREADS];
  uint32_t cnt[MAX_THREADS];
  Bar block_bar, wave_bar[MAX_THREADS / WAVE];
  int block_live, wave_live[MAX_THREADS / WAVE];
  std::function<void()> body;
};
State &st();
void yield();
void run_grid(int grid, int block, const std::function<void()> &body);

// generation barrier over a (possibly shrinking) set of live fibres
inline void wait(Bar &b, const int &live) {
  const int my = b.gen;
  b.count++;
  while (b.gen == my) {
    if (b.count >= live) { b.count = 0; b.gen++; break; }
    yield();
  }
}
inline void block_barrier() { State &s = st(); wait(s.block_bar, s.block_live); }
inline void wave_barrier() { State &s = st(); wait(s.wave_bar[s.cur / WAVE], s.wave_live[s.cur / WAVE]); }

template <class T>
inline uint64_t to_bits(T v) {
  uint64_t b = 0;
  static_assert(sizeof(T) <= 8, "shuffle payload too wide");
  memcpy(&b, &v, sizeof(T));
  return b;
}
template <class T>
inline T from_bits(uint64_t b) {
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}
// src: lane index within the caller's wave
template <class T>
inline T exchange(T v, int src) {
  State &s = st();
  const int base = (s.cur / WAVE) * WAVE;
  const uint32_t c = ++s.cnt[s.cur];
  const int p = (int)(c & 1u);
  s.slot[p][s.cur] = to_bits(v);
  s.stamp[p][s.cur] = c;
  wave_barrier();
  return (src >= 0 && src < WAVE && base + src < s.nthreads && s.stamp[p][base + src] == c) ? from_bits<T>(s.slot[p][base + src]) : v;
}
}  // namespace wemu

struct wemu_tid_proxy {
  struct X {
    operator unsigned() const { return (unsigned)wemu::st().cur; }
  } x;
};
struct wemu_bid_proxy {
  struct X {
    operator unsigned() const { return wemu::st().block_idx.x; }
  } x;
};
[[maybe_unused]] static wemu_tid_proxy threadIdx;
[[maybe_unused]] static wemu_bid_proxy blockIdx;

inline void __syncthreads() { wemu::block_barrier(); }
inline unsigned long long __ballot(bool p) {
  wemu::State &s = wemu::st();
  const int base = (s.cur / wemu::WAVE) * wemu::WAVE;
  const uint32_t c = ++s.cnt[s.cur];
  const int b = (int)(c & 1u);
  s.slot[b][s.cur] = p ? 1 : 0;
  s.stamp[b][s.cur] = c;
  wemu::wave_barrier();
  unsigned long long m = 0;
  for (int i = 0; i < wemu::WAVE && base + i < s.nthreads; ++i)
    if (s.stamp[b][base + i] == c && s.slot[b][base + i]) m |= 1ull << i;
  return m;
}
template <class T>
inline T __shfl(T v, int lane) { return wemu::exchange(v, lane); }
template <class T>
inline T __shfl_xor(T v, int mask) { return wemu::exchange(v, (wemu::st().cur % wemu::WAVE) ^ mask); }
template <class T>
inline T __shfl_up(T v, int delta) {
  const int l = wemu::st().cur % wemu::WAVE;
  return wemu::exchange(v, l - delta >= 0 ? l - delta : l);
}
// LDS atomics: fibres are cooperative (one runs at a time), so a plain read-modify-write is atomic
inline int atomicMax(int *p, int v) { const int o = *p; if (v > o) *p = v; return o; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }

// ---- the sliver of the HIP runtime API the host engine uses, mapped onto host memory ------------
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void *p) { free(p); return 0; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return 0; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }

#define FPC_LAUNCH(kernel, grid, block, stream, ...) \
  wemu::run_grid((int)(grid), (int)(block), [&]() { kernel(__VA_ARGS__); })
