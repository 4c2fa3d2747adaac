// tests/emul/wave_emul.cpp -- TEST INFRASTRUCTURE ONLY: fibre scheduler of the wavefront emulator.
#include "wave_emul.h"

namespace wemu {

State &st() {
  static State s;
  return s;
}

static void trampoline() {
  State &s = st();
  s.body();
  s.done[s.cur] = true;
  s.block_live--;
  s.wave_live[s.cur / WAVE]--;
  swapcontext(&s.lane_ctx[s.cur], &s.main_ctx);
}

void yield() {
  State &s = st();
  swapcontext(&s.lane_ctx[s.cur], &s.main_ctx);
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
        swapcontext(&s.main_ctx, &s.lane_ctx[l]);
      }
      if (!any) break;
    }
  }
}

}  // namespace wemu
