// TEST INFRASTRUCTURE ONLY -- scheduler of the HIP emulation (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <ucontext.h>

#include <cstdlib>
#include <vector>

namespace hipemu {
thread_local State st;
namespace {
struct Fiber { ucontext_t ctx; char* stack = nullptr; bool done = true; dim3 tid; };
constexpr size_t kStack = 256 * 1024;
thread_local std::vector<Fiber> fibers;
thread_local ucontext_t sched;
thread_local Fiber* cur = nullptr;
thread_local const std::function<void()>* body = nullptr;
void entry() {
    (*body)();
    cur->done = true;
    swapcontext(&cur->ctx, &sched);
}
}  // namespace
void sync() { swapcontext(&cur->ctx, &sched); }
// Cross-lane shift by one lane over a 64-lane wave.  The fibres of a workgroup run round-robin between
// yields, so "deposit, yield, read the neighbour, yield" is a correct exchange as long as all lanes of the
// wave execute the same sequence of yields (wave-uniform control flow, as DPP requires on the hardware).
int dpp_wave_shift(int old, int src, int ctrl) {
    static thread_local int slot[1024];
    const unsigned t = st.tid.x + st.bdim.x * (st.tid.y + st.bdim.y * st.tid.z);
    slot[t] = src;
    sync();
    const unsigned lane = t & 63u;
    int r = old;
    if (ctrl == 0x138) { if (lane > 0) r = slot[t - 1]; }
    else if (ctrl == 0x130) { if (lane < 63) r = slot[t + 1]; }
    else std::abort();
    sync();
    return r;
}
void launch(const std::function<void()>& fn, dim3 grid, dim3 block) {
    const unsigned n = block.x * block.y * block.z;
    if (fibers.size() < n) {
        const size_t old = fibers.size();
        fibers.resize(n);
        for (size_t i = old; i < n; ++i) fibers[i].stack = (char*)std::malloc(kStack);
    }
    body = &fn;
    st.gdim = grid; st.bdim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                st.bid = dim3(bx, by, bz);
                for (unsigned t = 0; t < n; ++t) {
                    Fiber& f = fibers[t];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, entry, 0);
                    f.done = false;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                }
                unsigned alive = n;
                while (alive) {
                    alive = 0;
                    for (unsigned t = 0; t < n; ++t) {
                        Fiber& f = fibers[t];
                        if (f.done) continue;
                        cur = &f;
                        st.tid = f.tid;
                        swapcontext(&sched, &f.ctx);
                        if (!f.done) ++alive;
                    }
                }
            }
}
}  // namespace hipemu
