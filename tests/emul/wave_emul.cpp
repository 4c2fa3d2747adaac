// tests/emul/wave_emul.cpp -- TEST INFRASTRUCTURE ONLY: fibre scheduler of the wavefront emulator.
#include "wave_emul.h"

namespace wemu {

static int g_tag[WAVE];

State &st() {
  static State s;
  return s;
}

static void trampoline() {
  State &s = st();
  s.body();
  s.done[s.cur] = true;
  swapcontext(&s.lane_ctx[s.cur], &s.main_ctx);
}

void barrier() {
  State &s = st();
  swapcontext(&s.lane_ctx[s.cur], &s.main_ctx);
}

void run_grid(int grid, int block, const std::function<void()> &body) {
  if (block != WAVE) {
    fprintf(stderr, "wave_emul: block size must be 64, got %d\n", block);
    abort();
  }
  State &s = st();
  constexpr size_t STK = 256 * 1024;
  if (!s.stacks[0])
    for (int l = 0; l < WAVE; ++l) s.stacks[l] = (char *)malloc(STK);
  s.body = body;
  for (int b = 0; b < grid; ++b) {
    s.block_idx.x = (unsigned)b;
    for (int l = 0; l < WAVE; ++l) {
      getcontext(&s.lane_ctx[l]);
      s.lane_ctx[l].uc_stack.ss_sp = s.stacks[l];
      s.lane_ctx[l].uc_stack.ss_size = STK;
      s.lane_ctx[l].uc_link = &s.main_ctx;
      makecontext(&s.lane_ctx[l], trampoline, 0);
      s.done[l] = false;
    }
    for (;;) {
      bool any = false;
      for (int l = 0; l < WAVE; ++l) {
        if (s.done[l]) continue;
        any = true;
        s.cur = l;
        swapcontext(&s.main_ctx, &s.lane_ctx[l]);
      }
      if (!any) break;
    }
  }
  (void)g_tag;
}

}  // namespace wemu
