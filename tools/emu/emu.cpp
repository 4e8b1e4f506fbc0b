// TEST INFRASTRUCTURE ONLY -- see emu.h.
#include "emu.h"

namespace emu {

thread_local BlockState* g_bs = nullptr;
thread_local emu_dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

void fiber_entry() {
  BlockState* bs = g_bs;
  int me = bs->cur;
  bs->body();
  bs->fibers[me].done = true;
  swapcontext(&bs->fibers[me].ctx, &bs->sched);
}

void run_block(BlockState& bs) {
  bs.bar_count = 0; bs.bar_gen = 0;
  const unsigned nwaves = (bs.nthreads + 63) / 64;
  bs.dma_log.assign(nwaves, {}); bs.dma_wait_log.assign(nwaves, {});
  bs.dma_pending.assign(bs.nthreads, {}); bs.dma_cursor.assign(bs.nthreads, 0); bs.dma_waits.assign(bs.nthreads, 0);
  memset(bs.wbar_count, 0, sizeof(bs.wbar_count));
  memset(bs.wbar_gen, 0, sizeof(bs.wbar_gen));
  for (unsigned i = 0; i < bs.nthreads; ++i) {
    Fiber& f = bs.fibers[i];
    f.done = false;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = 128 * 1024;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  unsigned remaining = bs.nthreads;
  while (remaining) {
    for (unsigned i = 0; i < bs.nthreads; ++i) {
      Fiber& f = bs.fibers[i];
      if (f.done) continue;
      bs.cur = (int)i;
      set_tid((int)i);
      swapcontext(&bs.sched, &f.ctx);
      if (f.done) --remaining;
    }
  }
}

}  // namespace emu
