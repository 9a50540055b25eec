// TEST INFRASTRUCTURE ONLY -- scheduler of the HIP emulation (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#if defined(__SANITIZE_ADDRESS__) || !defined(__x86_64__)
#define HIPEMU_UCONTEXT 1          // AddressSanitizer knows swapcontext; everything else switches stacks itself
#include <ucontext.h>
#endif

#include <sys/mman.h>

#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#ifdef HIPEMU_POISON_LDS
#include <cstring>
extern "C" char __start_lds_emu, __stop_lds_emu;
#endif
namespace hipemu {
namespace {
std::mutex g_guard_mu;
std::map<void*, std::pair<void*, size_t>> g_guard;      // user pointer -> (mapping, length)
bool guard_on() { static const bool on = [] { const char* e = std::getenv("HIPEMU_GUARD"); return e && std::atoi(e) != 0; }(); return on; }
}  // namespace
// page-locked host allocations (hipHostMalloc): base -> size
namespace { std::mutex g_host_mu; std::map<char*, size_t> g_host; }
void host_track(void* p, size_t n) { std::lock_guard<std::mutex> lk(g_host_mu); g_host[(char*)p] = n; }
void host_untrack(void* p) { std::lock_guard<std::mutex> lk(g_host_mu); g_host.erase((char*)p); }
void* host_lookup(const void* p) {
    std::lock_guard<std::mutex> lk(g_host_mu);
    auto it = g_host.upper_bound((char*)p);
    if (it == g_host.begin()) return nullptr;
    --it;
    return ((const char*)p < it->first + it->second) ? const_cast<void*>(p) : nullptr;
}
void* guard_malloc(size_t n) {
    if (!guard_on()) return nullptr;
    constexpr size_t kMargin = 16u << 20, kPage = 4096;
    const size_t body = (n + kPage - 1) / kPage * kPage, total = body + 2 * kMargin;
    char* base = (char*)mmap(nullptr, total, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (base == (char*)MAP_FAILED) return nullptr;
    if (mprotect(base + kMargin, body, PROT_READ | PROT_WRITE) != 0) { munmap(base, total); return nullptr; }
    char* user = base + kMargin + body - (n + 255) / 256 * 256;          // 256-byte aligned like hipMalloc, as close to the end as that allows
    if (user < base + kMargin) user = base + kMargin;
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard[user] = {base, total};
    return user;
}
bool guard_free(void* p) {
    std::lock_guard<std::mutex> lk(g_guard_mu);
    auto it = g_guard.find(p);
    if (it == g_guard.end()) return false;
    munmap(it->second.first, it->second.second);
    g_guard.erase(it);
    return true;
}
thread_local State st;
#ifndef HIPEMU_UCONTEXT
// swapcontext() makes a system call per switch (the signal mask); a wave shift is two switches per lane.  This switch saves the
// callee-saved registers + MXCSR / x87 control word on the old stack and resumes the new one: same semantics, ~50 x cheaper.
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch, .-hipemu_switch
)");
#endif
namespace {
constexpr size_t kStack = 256 * 1024;
#ifdef HIPEMU_UCONTEXT
struct Fiber { ucontext_t ctx; char* stack = nullptr; bool done = true; dim3 tid; };
thread_local ucontext_t sched;
#else
struct Fiber { void* sp = nullptr; char* stack = nullptr; bool done = true; dim3 tid; };
thread_local void* sched = nullptr;
#endif
thread_local std::vector<Fiber> fibers;
thread_local Fiber* cur = nullptr;
thread_local const std::function<void()>* body = nullptr;
#ifdef HIPEMU_UCONTEXT
void to_sched() { swapcontext(&cur->ctx, &sched); }
void to_fiber(Fiber& f) { swapcontext(&sched, &f.ctx); }
#else
void to_sched() { hipemu_switch(&cur->sp, sched); }
void to_fiber(Fiber& f) { hipemu_switch(&sched, f.sp); }
#endif
void entry() {
    (*body)();
    cur->done = true;
    to_sched();
    std::abort();          // a finished fibre is never resumed
}
void arm(Fiber& f) {
#ifdef HIPEMU_UCONTEXT
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, entry, 0);
#else
    // the frame hipemu_switch pops: MXCSR | x87 CW, r15 r14 r13 r12 rbx rbp, return address = entry; entry() then sees the stack
    // alignment of a called function (rsp = 16 n + 8)
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                       // entry's (never used) return address
    *--sp = (void*)&entry;
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    --sp;
    unsigned* cw = (unsigned*)sp;
    cw[0] = 0x1F80u;                       // MXCSR default: all exceptions masked, round to nearest
    cw[1] = 0x037Fu;                       // x87 default control word
    f.sp = sp;
#endif
}
}  // namespace
void sync() { to_sched(); }
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
// ds_bpermute_b32: lane i receives the value of lane (byte_addr / 4) % 64 of its wave (same deposit / yield / read scheme;
// every lane of the wave must execute it)
int wave_bpermute(int byte_addr, int src) {
    static thread_local int slot[1024];
    const unsigned t = st.tid.x + st.bdim.x * (st.tid.y + st.bdim.y * st.tid.z);
    slot[t] = src;
    sync();
    const int r = slot[(t & ~63u) + (((unsigned)byte_addr >> 2) & 63u)];
    sync();
    return r;
}
// v_cmp into an SGPR pair: bit i = predicate of lane i of the caller's wave (same deposit / yield / read scheme)
unsigned long long wave_ballot(int pred) {
    static thread_local int slot[1024];
    const unsigned n = st.bdim.x * st.bdim.y * st.bdim.z;
    const unsigned t = st.tid.x + st.bdim.x * (st.tid.y + st.bdim.y * st.tid.z);
    slot[t] = pred;
    sync();
    unsigned long long m = 0;
    for (unsigned i = 0; i < 64 && (t & ~63u) + i < n; ++i) if (slot[(t & ~63u) + i]) m |= 1ull << i;
    sync();
    return m;
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
    // HIPEMU_ORDER=reverse: workgroups from the last to the first, and inside a workgroup the work-items from the last to the first
    // between barriers -- a kernel whose result depends on the order (a missing barrier, a workgroup that reads what another one of the
    // same launch writes) computes something else then.  Default: ascending.
    static const bool reverse = [] { const char* e = std::getenv("HIPEMU_ORDER"); return e && std::string(e) == "reverse"; }();
    for (unsigned bz0 = 0; bz0 < grid.z; ++bz0)
        for (unsigned by0 = 0; by0 < grid.y; ++by0)
            for (unsigned bx0 = 0; bx0 < grid.x; ++bx0) {
                const unsigned bx = reverse ? grid.x - 1 - bx0 : bx0, by = reverse ? grid.y - 1 - by0 : by0, bz = reverse ? grid.z - 1 - bz0 : bz0;
                st.bid = dim3(bx, by, bz);
#ifdef HIPEMU_POISON_LDS
                std::memset(&__start_lds_emu, 0xFF, (size_t)(&__stop_lds_emu - &__start_lds_emu));
#endif
                for (unsigned t = 0; t < n; ++t) {
                    Fiber& f = fibers[t];
                    arm(f);
                    f.done = false;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                }
                unsigned alive = n;
                while (alive) {
                    alive = 0;
                    for (unsigned t0 = 0; t0 < n; ++t0) {
                        const unsigned t = reverse ? n - 1 - t0 : t0;
                        Fiber& f = fibers[t];
                        if (f.done) continue;
                        cur = &f;
                        st.tid = f.tid;
                        to_fiber(f);
                        if (!f.done) ++alive;
                    }
                }
            }
}
}  // namespace hipemu
