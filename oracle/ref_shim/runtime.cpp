// runtime.cpp -- fiber scheduler behind cuda_on_cpu.h (one fiber per CUDA thread of a block).
#include "cuda_on_cpu.h"

uint3_ threadIdx, blockIdx;
dim3 blockDim, gridDim;

float atomicAdd(float *p, float v) { const float old = *p; *p = old + v; return old; }
int atomicAdd(int *p, int v) { const int old = *p; *p = old + v; return old; }

namespace pvref {
namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
  ucontext_t ctx;
  char *stack = nullptr;
  bool done = false;
  uint3_ tid;
};
ucontext_t g_sched;
Fiber *g_cur = nullptr;
const std::function<void()> *g_body = nullptr;
bool g_descending = false;

void trampoline() {
  (*g_body)();
  g_cur->done = true;
  swapcontext(&g_cur->ctx, &g_sched);
}
}  // namespace

void yield_to_scheduler() { swapcontext(&g_cur->ctx, &g_sched); }
void set_descending_schedule(bool on) { g_descending = on; }

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body) {
  gridDim = grid;
  blockDim = block;
  g_body = &body;
  const unsigned nthreads = block.x * block.y * block.z;
  std::vector<Fiber> fibers(nthreads);
  for (auto &f : fibers) f.stack = static_cast<char *>(malloc(kStack));
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        unsigned t = 0;
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx, ++t) {
              Fiber &f = fibers[t];
              f.done = false;
              f.tid = {tx, ty, tz};
              getcontext(&f.ctx);
              f.ctx.uc_stack.ss_sp = f.stack;
              f.ctx.uc_stack.ss_size = kStack;
              f.ctx.uc_link = nullptr;
              makecontext(&f.ctx, trampoline, 0);
            }
        bool alive = true;
        while (alive) {   // one pass = run every live thread up to its next __syncthreads()
          alive = false;
          for (unsigned q = 0; q < nthreads; ++q) {
            Fiber &f = fibers[g_descending ? nthreads - 1 - q : q];
            if (f.done) continue;
            g_cur = &f;
            threadIdx = f.tid;
            swapcontext(&g_sched, &f.ctx);
            alive = alive || !f.done;
          }
        }
      }
  for (auto &f : fibers) free(f.stack);
}
}  // namespace pvref
