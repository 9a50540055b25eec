// labconv.hip -- u8 BGR frames -> integer Lab planes through OpenCV 4's interpolated 33^3 table (lab_lut.h).
//
// Replaces, for the default flavour, the float conversion the reference does once per frame
// (convertTo + cv::cvtColor(COLOR_BGR2Lab), MagnifyCore.hpp:89-90 and :218-219).  The table look-up is the expensive
// part of a frame (lab_lut.h), so every frame is converted exactly ONCE, here, and the kernels that need Lab(in) -- the
// first pyramid kernel and the output kernel of the Laplace path, the L plane and the output kernel of the Riesz path --
// read the integers back: iL as uint16 (Laplace) or already as the float L plane (Riesz), (ia, ib) as one dword.
// 6 bytes per pixel written + read instead of a second conversion.
//
// One persistent 1024-thread workgroup per CU holds the (a, b) node table in LDS (144 KB); a lane converts 4 adjacent
// pixels per step (12-byte load, 8 + 16-byte stores) or single pixels for frames whose rows are not dword-aligned.
#include "lvm_internal.h"

namespace lvm {

constexpr int LC_THREADS = 1024;

struct __attribute__((packed, aligned(4))) LcPx4 { uint32_t a, b, c; };

// VEC: w % 4 == 0 and rows / frames dword aligned.  LFLOAT: L is stored as float (the Riesz L plane) instead of uint16.
template <bool VEC, bool LFLOAT>
__global__ __launch_bounds__(LC_THREADS) void k_lab_planes(const uint8_t* __restrict__ in, long in_stride, long in_sstride, int w, int h,
                                                           int nframes, LabLut lut, uint16_t* __restrict__ iLp, float* __restrict__ Lfp,
                                                           uint32_t* __restrict__ iabp, int per_block) {
    __shared__ uint32_t s_ab[kLabAbWords];
    for (int i = threadIdx.x; i < kLabAbWords; i += LC_THREADS) s_ab[i] = lut.ab[i];
    __syncthreads();
    const int upr = VEC ? (w >> 2) : w;                   // work units (4-pixel groups / pixels) per row
    const long total = (long)upr * h * nframes;
    const long u0 = (long)blockIdx.x * per_block;
    const long uend = u0 + per_block < total ? u0 + per_block : total;
    const int upf = upr * h;                              // units per frame
    // (frame, row, unit in row) of this lane's first unit by division, of the following ones by stepping: a lane moves
    // LC_THREADS units per step
    long u = u0 + threadIdx.x;
    int f = (int)(u / upf);
    int y, xu;
    { const int r = (int)(u - (long)f * upf); y = r / upr; xu = r - y * upr; }
    const int step_y = LC_THREADS / upr, step_x = LC_THREADS - step_y * upr;
    auto advance = [&]() __attribute__((always_inline)) {          // the lane's next unit
        u += LC_THREADS; xu += step_x; y += step_y;
        if (xu >= upr) { xu -= upr; ++y; }
        while (y >= h) { y -= h; ++f; }
    };
    if (VEC) {
        // The group of the NEXT step is loaded before this step's look-ups (the 12-byte load used to be issued and waited for in the
        // same step), and inside a step the table reads of the pixels k + 1, k + 2 are issued before pixel k is interpolated
        // (lut_issue / lut_finish, as in the fused first kernel of the Laplace path).
        auto fetch = [&]() __attribute__((always_inline)) {
            // streaming data bypasses the caches' retention (nontemporal): the L cells a CU keeps re-reading stay in its L1
            LcPx4 v;
            const uint32_t* pi = reinterpret_cast<const uint32_t*>(in + (size_t)f * in_sstride + (size_t)y * in_stride + (size_t)xu * 12);
            v.a = __builtin_nontemporal_load(pi); v.b = __builtin_nontemporal_load(pi + 1); v.c = __builtin_nontemporal_load(pi + 2);
            return v;
        };
        LcPx4 vn{};
        if (u < uend) vn = fetch();
        while (u < uend) {
            const LcPx4 v = vn;
            const size_t q = ((size_t)f * h + y) * w + (size_t)xu * 4;
            advance();
            if (u < uend) vn = fetch();
            const uint32_t pb[12] = {v.a & 255, (v.a >> 8) & 255, (v.a >> 16) & 255, v.a >> 24, v.b & 255, (v.b >> 8) & 255,
                                     (v.b >> 16) & 255, v.b >> 24, v.c & 255, (v.c >> 8) & 255, (v.c >> 16) & 255, v.c >> 24};
            int iL[4], ia[4], ib[4];
            LutRefs r[4];
            r[0] = lut_issue(pb[0], pb[1], pb[2], s_ab, lut.Lcells);
            r[1] = lut_issue(pb[3], pb[4], pb[5], s_ab, lut.Lcells);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k + 2 < 4) r[k + 2] = lut_issue(pb[3 * (k + 2)], pb[3 * (k + 2) + 1], pb[3 * (k + 2) + 2], s_ab, lut.Lcells);
                __builtin_amdgcn_sched_barrier(0);
                lut_finish(r[k], iL[k], ia[k], ib[k]);
            }
            if (LFLOAT) {
                float* d = Lfp + q;
#pragma unroll
                for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(lut_L(iL[k]), d + k);
            } else {
                uint32_t* d = reinterpret_cast<uint32_t*>(iLp + q);
                __builtin_nontemporal_store((uint32_t)iL[0] | ((uint32_t)iL[1] << 16), d);
                __builtin_nontemporal_store((uint32_t)iL[2] | ((uint32_t)iL[3] << 16), d + 1);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) __builtin_nontemporal_store((uint32_t)ia[k] | ((uint32_t)ib[k] << 16), iabp + q + k);
        }
    } else {
        for (; u < uend; advance()) {
            const uint8_t* px = in + (size_t)f * in_sstride + (size_t)y * in_stride + (size_t)xu * 3;
            const size_t o = ((size_t)f * h + y) * w;
            int iL, ia, ib;
            lut_lab_int(px[0], px[1], px[2], s_ab, lut.Lcells, iL, ia, ib);
            if (LFLOAT) Lfp[o + xu] = lut_L(iL); else iLp[o + xu] = (uint16_t)iL;
            iabp[o + xu] = (uint32_t)ia | ((uint32_t)ib << 16);
        }
    }
}

// nframes frames (batch frames x streams) of w x h BGR pixels at in_sstride.  Exactly one of iL / Lf is non-null.
void lab_lut_planes(Ctx* c, const uint8_t* d_in, long in_stride, long in_sstride, int w, int h, int nframes, uint16_t* iL, float* Lf,
                    uint32_t* iab, hipStream_t s) {
    const bool vec = w % 4 == 0 && in_stride % 4 == 0 && in_sstride % 4 == 0 && ((uintptr_t)d_in % 4) == 0;
    const long units = (long)(vec ? w / 4 : w) * h * nframes;
    // one workgroup per CU (the table takes 148 KB of the CU's 160 KB); few units: fewer workgroups, >= 1024 units each
    long blocks = (units + LC_THREADS - 1) / LC_THREADS;
    if (blocks > c->num_cus) blocks = c->num_cus;
    if (blocks < 1) blocks = 1;
    long per = (units + blocks - 1) / blocks;
    per = (per + LC_THREADS - 1) / LC_THREADS * LC_THREADS;       // whole steps of the workgroup: neighbouring lanes stay neighbours
    blocks = (units + per - 1) / per;
    if (blocks < 1) blocks = 1;
    auto k = vec ? (Lf ? k_lab_planes<true, true> : k_lab_planes<true, false>) : (Lf ? k_lab_planes<false, true> : k_lab_planes<false, false>);
    LVM_LAUNCH(c, "lab_lut", k, dim3((unsigned)blocks), dim3(LC_THREADS), s, d_in, in_stride, in_sstride, w, h, nframes, c->lab_lut, iL, Lf, iab, (int)per);
}

// ---- lvm_debug_sweep_u8_steps: the step table against the operations it replaces, float by float ---------------------------------
// Thread i of the grid takes the bit patterns first + i, first + i + stride, ...: c = the float with that pattern; reference = OpenCV's
// operations one by one (clip01, x 1024, splineInterpolate, x 255 + 1/255, cvRound + saturate -- the EXACT flavour's code); candidate =
// u8_step(4096 c).  Counts the patterns where the two bytes differ and keeps the smallest such pattern.
__global__ __launch_bounds__(256) void k_sweep_u8_steps(LabCoef lab, unsigned long long first, unsigned long long count, unsigned long long* bad, unsigned long long* first_bad) {
    __shared__ __attribute__((aligned(16))) float s_igt[4096];
    __shared__ uint2 s_steps[kU8StepSlices];
    load_invgamma(s_igt, lab.invgamma);
    load_u8steps(s_steps, lab.u8steps);
    __syncthreads();
    unsigned long long nbad = 0, fb = ~0ull;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint32_t bits = (uint32_t)(first + i);
        const float c = __uint_as_float(bits);
        uint32_t want = 0;
        if (c == c) {                                                       // (NaN: both sides give 0 -- the reference's (int)NaN is not C)
            const float o = spline1024<true>(clip01(c) * 1024.f, s_igt);
            want = __builtin_amdgcn_cvt_pk_u8_f32(o * 255.0f + lab.a255, 0, 0);
        }
        const uint32_t got = u8_step(c * 4096.0f, s_steps);
        if (got != want) { ++nbad; if (first + i < fb) fb = first + i; }
    }
    if (nbad) { atomicAdd(bad, nbad); atomicMin(first_bad, fb); }
}

int sweep_u8_steps(Ctx* c, unsigned long long first, unsigned long long count, unsigned long long* bad, unsigned long long* first_bad, hipStream_t s) {
    unsigned long long* d = nullptr;
    LVM_HIP_TRY(c, hipMalloc((void**)&d, 16));
    const unsigned long long init[2] = {0ull, ~0ull};
    hipError_t e = hipMemcpyAsync(d, init, 16, hipMemcpyHostToDevice, s);
    long blocks = (long)((count + 255) / 256);
    if (blocks > 8L * c->num_cus) blocks = 8L * c->num_cus;
    if (blocks < 1) blocks = 1;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sweep_u8_steps, dim3((unsigned)blocks), dim3(256), 0, s, c->lab, first, count, d, d + 1);
        e = hipGetLastError();
    }
    unsigned long long out[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) { c->err = std::string("sweep_u8_steps: ") + hipGetErrorString(e); return LVM_ERR_HIP; }
    *bad = out[0]; *first_bad = out[1];
    return LVM_OK;
}

// ---- lvm_debug_clock_probe_*: the average shader clock over an interval the HOST chooses ------------------------------------------------
// One lane reads both counters, sleeps and polls a flag in page-locked host memory until the host sets it (or `max_ticks` of the
// 100 MHz counter have passed: the kernel can never hang the device), reads both again.  Launched on the context's auxiliary stream it
// runs beside whatever the main stream executes: one wave slot, a handful of SGPRs.
__global__ void k_clock_probe(const int* stop, unsigned long long* out, unsigned long long max_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = lvm_clock_real(), c0 = lvm_clock_core();
    __atomic_store_n(out + 3, 1ull, __ATOMIC_RELEASE);       // "running": the host waits for it before it enqueues what is to be measured
    unsigned long long r = r0;
    while (r - r0 < max_ticks) {
        if (__atomic_load_n(stop, __ATOMIC_RELAXED) != 0) break;
        for (int i = 0; i < 8; ++i) lvm_sleep();             // ~8 x 127 x 64 clocks between two reads over PCIe
        r = lvm_clock_real();
    }
    const unsigned long long c1 = lvm_clock_core();
    r = lvm_clock_real();
    out[0] = c1 - c0; out[1] = r - r0;
}

int clock_probe_start(Ctx* c, double max_seconds) {
    if (!c->h_probe) {
        LVM_HIP_TRY(c, hipHostMalloc((void**)&c->h_probe, 64, 0));
    }
    if (c->probe_running) { c->err = "clock probe already running"; return LVM_ERR_INVALID; }
    volatile unsigned long long* h = c->h_probe;
    h[0] = 0; h[1] = 0; h[2] = 0; h[3] = 0;                  // [0] cycles, [1] ticks, [2] stop flag, [3] running
    if (max_seconds <= 0.0 || max_seconds > 5.0) max_seconds = 5.0;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, c->aux_stream, reinterpret_cast<const int*>(c->h_probe + 2), c->h_probe,
                       (unsigned long long)(max_seconds * 1e8));
    LVM_HIP_TRY(c, hipGetLastError());
    c->probe_running = true;
    // The lane must be ON the device before the measured work is enqueued: behind a busy queue of full-chip launches it is dispatched
    // only when they drain (measured: started with the work, it covered the last 0.5 ms of a 13.5 ms region).  A few microseconds on an idle device.
    for (int spin = 0; spin < 2000000 && __atomic_load_n(h + 3, __ATOMIC_ACQUIRE) == 0; ++spin) {}
    return LVM_OK;
}

int clock_probe_stop(Ctx* c, double* mhz, double* seconds) {
    if (!c->probe_running) { c->err = "no clock probe running"; return LVM_ERR_INVALID; }
    __atomic_store_n(reinterpret_cast<int*>(c->h_probe + 2), 1, __ATOMIC_RELEASE);
    c->probe_running = false;
    LVM_HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
    const double cyc = (double)c->h_probe[0], tk = (double)c->h_probe[1];
    if (mhz) *mhz = tk > 0.0 ? cyc / tk * 100.0 : 0.0;
    if (seconds) *seconds = tk * 1e-8;
    return LVM_OK;
}

}  // namespace lvm
