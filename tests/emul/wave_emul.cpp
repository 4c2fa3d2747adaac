// tests/emul/wave_emul.cpp -- TEST INFRASTRUCTURE ONLY: fibre scheduler of the wavefront emulator.
#include "wave_emul.h"

// AddressSanitizer build (make SAN=1): every stack switch is announced, otherwise ASan takes the fibres' frames
// for stack-buffer overflows of the main stack.
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#define WEMU_ASAN 1
#else
#define WEMU_ASAN 0
#endif

namespace wemu {

#if WEMU_ASAN
static void *g_main_fake = nullptr, *g_lane_fake[MAX_THREADS];
static const void *g_main_bottom = nullptr;
static size_t g_main_size = 0;
#endif
// lane -> scheduler
static void to_main(State &s, bool last) {
#if WEMU_ASAN
  __sanitizer_start_switch_fiber(last ? nullptr : &g_lane_fake[s.cur], g_main_bottom, g_main_size);
#endif
  const int me = s.cur;
  swapcontext(&s.lane_ctx[me], &s.main_ctx);
#if WEMU_ASAN
  __sanitizer_finish_switch_fiber(g_lane_fake[me], &g_main_bottom, &g_main_size);
#endif
  (void)last;
}

State &st() {
  static State s;
  return s;
}

static void trampoline() {
  State &s = st();
#if WEMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &g_main_bottom, &g_main_size);
#endif
  s.body();
  s.done[s.cur] = true;
  s.block_live--;
  s.wave_live[s.cur / WAVE]--;
  to_main(s, true);
}

void yield() {
  to_main(st(), false);
}

void run_grid(int grid, int block, const std::function<void()> &body) {
  if (block <= 0 || block > MAX_THREADS || block % WAVE) {
    fprintf(stderr, "wave_emul: block size must be a multiple of 64 up to %d, got %d\n", MAX_THREADS, block);
    abort();
  }
  State &s = st();
  constexpr size_t STK = 256 * 1024;
  for (int l = 0; l < block; ++l)
    if (!s.stacks[l]) s.stacks[l] = (char *)malloc(STK);
  s.body = body;
  s.nthreads = block;
  for (int b = 0; b < grid; ++b) {
    s.block_idx.x = (unsigned)b;
    s.block_bar = Bar();
    s.block_live = block;
    for (int w = 0; w < block / WAVE; ++w) { s.wave_bar[w] = Bar(); s.wave_live[w] = WAVE; }
    for (int l = 0; l < block; ++l) {
      getcontext(&s.lane_ctx[l]);
      s.lane_ctx[l].uc_stack.ss_sp = s.stacks[l];
      s.lane_ctx[l].uc_stack.ss_size = STK;
      s.lane_ctx[l].uc_link = &s.main_ctx;
      makecontext(&s.lane_ctx[l], trampoline, 0);
      s.done[l] = false;
      s.cnt[l] = 0;
      s.stamp[0][l] = s.stamp[1][l] = 0;
    }
    for (;;) {
      bool any = false;
      for (int l = 0; l < block; ++l) {
        if (s.done[l]) continue;
        any = true;
        s.cur = l;
#if WEMU_ASAN
        __sanitizer_start_switch_fiber(&g_main_fake, s.stacks[l], STK);
#endif
        swapcontext(&s.main_ctx, &s.lane_ctx[l]);
#if WEMU_ASAN
        __sanitizer_finish_switch_fiber(g_main_fake, nullptr, nullptr);
#endif
      }
      if (!any) break;
    }
  }
}

}  // namespace wemu
