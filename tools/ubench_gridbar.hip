// What would ONE persistent kernel with grid-wide barriers buy the per-frame Laplace surface (T = 1, one 1080p stream)?
// The middle of that frame is six dependent launches (level-1 pyrDown, two two-level pyrDowns, the IIR of levels 2..5, their collapse,
// the fused level-1 step): 54 us of its 73 (profiles/README.md), each moving between 0.1 and 28 MB.  This benchmark runs SIX PHASES
// with those byte counts (streaming copies standing in for the arithmetic)
//   (a) as six dependent launches on one stream -- the structure that ships,
//   (b) as ONE persistent launch, 1 or 2 workgroups per CU, with an XCD-hierarchical grid barrier between the phases
//       (MI355X_MICROARCH.md "barrier-xcd": per-XCD arrival counter -> top counter -> generation word; agent-scope release before the
//       arrival, agent-scope acquire after the wait; every spin bounded),
// and prints both, plus the barrier alone.   hipcc --offload-arch=gfx950 -O3 tools/ubench_gridbar.hip -o tools/ubench_gridbar
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Phase { const float4* src; float4* dst; unsigned n_in, n_out; };     // float4 counts: reads n_in, writes n_out (n_out <= n_in)
struct Phases { Phase p[6]; int n; };
struct Bar { unsigned* xcc; unsigned* top; unsigned* gen; unsigned* timeout; };      // xcc[8] (one 128-byte line each), top, generation

__device__ __forceinline__ void phase_work(const Phase& ph, unsigned wg, unsigned nwg) {
    // reads n_in float4, writes n_out: every output is the sum of n_in / n_out consecutive inputs (a reduction like a pyrDown)
    const unsigned ratio = ph.n_in / ph.n_out;
    for (unsigned o = wg * 256 + threadIdx.x; o < ph.n_out; o += nwg * 256) {
        float4 acc = make_float4(0, 0, 0, 0);
        for (unsigned k = 0; k < ratio; ++k) { const float4 v = ph.src[(size_t)k * ph.n_out + o]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        ph.dst[o] = acc;
    }
}
__global__ __launch_bounds__(256) void k_phase(Phase ph) { phase_work(ph, blockIdx.x, gridDim.x); }

// XCD-hierarchical grid barrier; group = blockIdx & 7 (the placement the hardware uses today: for speed only, any grouping is correct)
__device__ __forceinline__ bool grid_barrier(const Bar& b, unsigned epoch, unsigned nwg) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned g = blockIdx.x & 7u, per = (nwg + 7u - g) / 8u;          // workgroups of this group
        const unsigned a = __hip_atomic_fetch_add(b.xcc + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == per * epoch) {
            const unsigned t = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == (nwg < 8u ? nwg : 8u) * epoch) __hip_atomic_store(b.gen, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        while (__hip_atomic_load(b.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { __hip_atomic_store(b.timeout, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
__global__ __launch_bounds__(256) void k_persistent(Phases ps, Bar b, int with_work) {
    for (int i = 0; i < ps.n; ++i) {
        if (with_work) phase_work(ps.p[i], blockIdx.x, gridDim.x);
        if (i + 1 < ps.n && !grid_barrier(b, (unsigned)(i + 1), gridDim.x)) return;
    }
}

int main() {
    const size_t cap = (size_t)64 << 20;
    float4 *a, *b2; unsigned* barmem;
    CK(hipMalloc(&a, cap)); CK(hipMalloc(&b2, cap)); CK(hipMalloc(&barmem, 4096 + 8 * 128));
    CK(hipMemset(a, 0, cap)); CK(hipMemset(b2, 0, cap));
    // bytes of the six launches of a 1080p frame (3 float planes): in -> out
    const double mb_in[6] = {6.2, 1.55, 0.1, 2.1, 1.0, 21.6}, mb_out[6] = {1.55, 0.5, 0.03, 1.0, 1.55, 6.2};
    Phases ps{}; ps.n = 6;
    for (int i = 0; i < 6; ++i) {
        unsigned n_out = (unsigned)(mb_out[i] * 1e6 / 16), ratio = (unsigned)(mb_in[i] / mb_out[i] + 0.5); if (ratio < 1) ratio = 1;
        ps.p[i] = Phase{(i & 1) ? b2 : a, (i & 1) ? a : b2, n_out * ratio, n_out};
    }
    Bar bar{barmem + 1024, barmem, barmem + 32, barmem + 64};
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, int reps, auto fn) {
        for (int r = 0; r < 20; ++r) fn();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) fn();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned tmo = 0; CK(hipMemcpy(&tmo, bar.timeout, 4, hipMemcpyDeviceToHost));
        printf("%-86s %7.2f us per frame%s\n", name, ms * 1e3 / reps, tmo ? "  (BARRIER TIMEOUT)" : "");
    };
    for (int grid : {256, 512, 1024}) {
        char nm[160];
        snprintf(nm, sizeof nm, "six dependent launches, %d workgroups each", grid);
        timeit(nm, 200, [&] { for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(k_phase, dim3(grid), dim3(256), 0, s, ps.p[i]); });
    }
    timeit("six dependent launches, grid sized to the phase (one float4 per thread, <= 2048)", 200, [&] {
        for (int i = 0; i < 6; ++i) { unsigned g = (ps.p[i].n_out + 255) / 256; if (g > 2048) g = 2048; hipLaunchKernelGGL(k_phase, dim3(g), dim3(256), 0, s, ps.p[i]); } });
    for (int grid : {256, 512}) {
        char nm[160];
        snprintf(nm, sizeof nm, "ONE persistent launch, %d workgroups, 5 XCD-hierarchical grid barriers", grid);
        timeit(nm, 200, [&] { CK(hipMemsetAsync(barmem, 0, 4096 + 8 * 128, s)); hipLaunchKernelGGL(k_persistent, dim3(grid), dim3(256), 0, s, ps, bar, 1); });
        snprintf(nm, sizeof nm, "   the same without the phases' work (5 barriers + launch + memset)");
        timeit(nm, 200, [&] { CK(hipMemsetAsync(barmem, 0, 4096 + 8 * 128, s)); hipLaunchKernelGGL(k_persistent, dim3(grid), dim3(256), 0, s, ps, bar, 0); });
    }
    timeit("one empty launch (256 workgroups) + memset", 200, [&] { Phases one = ps; one.n = 1; CK(hipMemsetAsync(barmem, 0, 4096 + 8 * 128, s)); hipLaunchKernelGGL(k_persistent, dim3(256), dim3(256), 0, s, one, bar, 0); });
    return 0;
}
