// Memory skeleton of the last Laplace kernel (k_lap_final_v4, laplace.hip) with trivial arithmetic: what does its ACCESS PATTERN reach,
// and which change of it moves that?  (round 5, after tools/ubench_stream.hip showed 6.4-6.5 TB/s for the same byte mix as a flat stream)
// Per 4-pixel group and output row: 8 B (uint16 x 4, L plane) + 16 B (dword x 4, (a, b) plane) in, 12 B out; per PAIR of output rows one
// row of the half-resolution motion image cur_1 (3 float planes, taps i0-1 .. i0+2 per lane, i0 = 2 lane + 128 tx).  32 frames of 1080p.
//   CUR   0 = twelve 4-byte loads per row pair (the kernel as it is), 1 = three 8-byte loads + DPP neighbours + masked edge loads,
//         2 = no cur_1 at all (ablation)
//   NT    nontemporal plane loads and output stores
//   PF    0 = plane loads at the top of the step that uses them (as is), 1 = issued one step ahead
//   ALU   dependent fma per pixel and channel standing in for the colour arithmetic (0, 16, 32)
//   grid  persistent (waves take strips round-robin, 512-thread workgroups, G per CU) or one-shot (a wave per strip, raster order)
//   rows  strip height
// Prints GB/s of compulsory bytes (12 B per pixel).   hipcc --offload-arch=gfx950 -O3 tools/ubench_strips.hip -o tools/ubench_strips
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct SArgs { const uint16_t* pL; const uint32_t* pAB; const float* cur1; uint8_t* out; int w, h, w1, h1, strips_x, strips_y, frames, rows, oneshot; };
struct __attribute__((packed, aligned(4))) B12 { uint32_t a, b, c; };
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u3 __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ float dpp_shr1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }

struct Raw { u2 l; u4 ab; };
template <int NT> __device__ __forceinline__ Raw ld_raw(const SArgs& q, size_t poff, int gy, int gx) {
    const size_t o = poff + (size_t)gy * q.w + gx;
    Raw r;
    if (NT) { r.l = __builtin_nontemporal_load(reinterpret_cast<const u2*>(q.pL + o)); r.ab = __builtin_nontemporal_load(reinterpret_cast<const u4*>(q.pAB + o)); }
    else { r.l = *reinterpret_cast<const u2*>(q.pL + o); r.ab = *reinterpret_cast<const u4*>(q.pAB + o); }
    return r;
}
struct Row3 { float x[3], y[3], z[3], w[3]; };

template <int CUR, int NT, int PF, int ALU>
__device__ __forceinline__ void strip(const SArgs& q, int task, int lane) {
    const int per = q.strips_x * q.strips_y;
    const int b = task / per, r = task - b * per, ty = r / q.strips_x, tx = r - ty * q.strips_x;
    const int gx = tx * 256 + 4 * lane, y0 = ty * q.rows;
    const bool active = gx < q.w;
    if (CUR != 1 && !active) return;
    const size_t poff = (size_t)b * q.w * q.h;
    uint8_t* dst = q.out + poff * 3;
    const float* pl = q.cur1 + (size_t)b * 3 * ((size_t)q.w1 * q.h1);
    const int i0 = gx >> 1, w1 = q.w1, h1 = q.h1;
    const int cm1 = i0 > 0 ? i0 - 1 : 1, cp1 = i0 + 1 < w1 ? i0 + 1 : w1 - 1, cp2 = i0 + 2 < w1 ? i0 + 2 : w1 - 1;
    const size_t ps = (size_t)w1 * h1;
    const int ci = i0 < w1 ? i0 : w1 - 2;                    // (idle lanes of the last strip still take part in the DPP exchange)
    auto hrow = [&](int sy) __attribute__((always_inline)) {
        Row3 o;
        sy = sy < 0 ? 1 : (sy >= h1 ? h1 - 1 : sy);
        const float* row = pl + (size_t)sy * w1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* rc = row + c * ps;
            float sm1, s0, s1, s2;
            if (CUR == 0) { sm1 = rc[cm1]; s0 = rc[i0]; s1 = rc[cp1]; s2 = rc[cp2]; }
            else if (CUR == 1) {
                const f2 v = *reinterpret_cast<const f2*>(rc + ci);
                s0 = v.x; s1 = v.y;
                float e = 0.f;
                if (lane == 0 || lane == 63) e = rc[lane == 0 ? cm1 : cp2];
                sm1 = dpp_shr1(s1); s2 = dpp_shl1(s0);
                if (lane == 0) sm1 = e;
                if (lane == 63) s2 = e;
            } else { sm1 = s0 = s1 = s2 = (float)sy; }
            o.x[c] = sm1 + s0 * 6.f + s1; o.y[c] = (s0 + s1) * 4.f; o.z[c] = s0 + s1 * 6.f + s2; o.w[c] = (s1 + s2) * 4.f;
        }
        return o;
    };
    auto emit = [&](const Raw& p, const float (&m)[3][4], int gy) __attribute__((always_inline)) {
        float v[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t l = k < 2 ? p.l.x : p.l.y;
            const float L = (float)((k & 1) ? (l >> 16) : (l & 0xffffu)), a = (float)(p.ab[k] & 0xffffu), bb = (float)(p.ab[k] >> 16);
            float t0 = L + m[0][k], t1 = a + m[1][k], t2 = bb + m[2][k];
#pragma unroll
            for (int i = 0; i < ALU; ++i) { t0 = __builtin_fmaf(t0, 0.999f, t1); t1 = __builtin_fmaf(t1, 0.998f, t2); t2 = __builtin_fmaf(t2, 0.997f, t0); }
            v[3 * k] = t0; v[3 * k + 1] = t1; v[3 * k + 2] = t2;
        }
        u3 o;
        o.x = __builtin_amdgcn_cvt_pk_u8_f32(v[3], 3, __builtin_amdgcn_cvt_pk_u8_f32(v[2], 2, __builtin_amdgcn_cvt_pk_u8_f32(v[1], 1, __builtin_amdgcn_cvt_pk_u8_f32(v[0], 0, 0))));
        o.y = __builtin_amdgcn_cvt_pk_u8_f32(v[7], 3, __builtin_amdgcn_cvt_pk_u8_f32(v[6], 2, __builtin_amdgcn_cvt_pk_u8_f32(v[5], 1, __builtin_amdgcn_cvt_pk_u8_f32(v[4], 0, 0))));
        o.z = __builtin_amdgcn_cvt_pk_u8_f32(v[11], 3, __builtin_amdgcn_cvt_pk_u8_f32(v[10], 2, __builtin_amdgcn_cvt_pk_u8_f32(v[9], 1, __builtin_amdgcn_cvt_pk_u8_f32(v[8], 0, 0))));
        u3* p3 = reinterpret_cast<u3*>(dst + ((size_t)gy * q.w + gx) * 3);
        if (!active) return;
        if (NT) __builtin_nontemporal_store(o, p3); else *p3 = o;
    };
    const int yend = y0 + q.rows < q.h ? y0 + q.rows : q.h;
    int gy = y0, j = y0 >> 1;
    Row3 A = hrow(j - 1), B = hrow(j), C = hrow(j + 1);
    const int gxc = active ? gx : 0;
    Raw ne, no;
    if (PF) { ne = ld_raw<NT>(q, poff, gy, gxc); no = ld_raw<NT>(q, poff, gy + 1 < yend ? gy + 1 : gy, gxc); }
    while (gy < yend) {
        Raw pe, po;
        if (PF) { pe = ne; po = no; }
        else { pe = ld_raw<NT>(q, poff, gy, gxc); po = ld_raw<NT>(q, poff, gy + 1 < yend ? gy + 1 : gy, gxc); }
        float m[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c) { m[c][0] = A.x[c] + B.x[c] * 6.f + C.x[c]; m[c][1] = A.y[c] + B.y[c] * 6.f + C.y[c]; m[c][2] = A.z[c] + B.z[c] * 6.f + C.z[c]; m[c][3] = A.w[c] + B.w[c] * 6.f + C.w[c]; }
        const bool more = gy + 2 < yend;
        if (more) {
            A = hrow(j + 2);
            if (PF) { ne = ld_raw<NT>(q, poff, gy + 2, gxc); no = ld_raw<NT>(q, poff, gy + 3 < yend ? gy + 3 : gy + 2, gxc); }
        }
        emit(pe, m, gy);
        if (gy + 1 < yend) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { m[c][0] = B.x[c] + C.x[c]; m[c][1] = B.y[c] + C.y[c]; m[c][2] = B.z[c] + C.z[c]; m[c][3] = B.w[c] + C.w[c]; }
            emit(po, m, gy + 1);
        }
        gy += 2; ++j;
        const Row3 t = A; A = B; B = C; C = t;          // (the real kernel unrolls three steps instead of moving the window)
    }
}

template <int CUR, int NT, int PF, int ALU>
__global__ __launch_bounds__(512) void k_strips(SArgs q) {
    __shared__ float s_tab[4096];
    if (!q.oneshot) { for (int i = threadIdx.x; i < 4096; i += 512) s_tab[i] = (float)i; __syncthreads(); }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int ntasks = q.strips_x * q.strips_y * q.frames;
    for (int task = blockIdx.x * 8 + wave; task < ntasks; task += gridDim.x * 8) strip<CUR, NT, PF, ALU>(q, task, lane);
    if (q.w < 0) q.out[0] = (uint8_t)s_tab[lane];
}

static SArgs g;
static double g_bytes;
template <int CUR, int NT, int PF, int ALU>
static void run(const char* tag, int rows, int per_cu, int oneshot) {
    SArgs q = g;
    q.rows = rows; q.strips_x = (q.w + 255) / 256; q.strips_y = (q.h + rows - 1) / rows; q.oneshot = oneshot;
    const int ntasks = q.strips_x * q.strips_y * q.frames;
    const int grid = oneshot ? (ntasks + 7) / 8 : 256 * per_cu;
    auto k = k_strips<CUR, NT, PF, ALU>;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, q);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, q);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 512, 0));
    const double us = ms * 1e3 / reps;
    printf("%-34s rows %2d %-10s grid %6d (occ %d WG/CU) %7.1f us  %6.0f GB/s\n", tag, rows, oneshot ? "oneshot" : "persistent", grid, occ, us, g_bytes / (us * 1e-6) / 1e9);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}
#define SWEEP(CUR, NT, PF, ALU) do { const char* t = "cur" #CUR " nt" #NT " pf" #PF " alu" #ALU;                   \
    for (int rows : {4, 8, 16, 32}) { run<CUR, NT, PF, ALU>(t, rows, 2, 0); run<CUR, NT, PF, ALU>(t, rows, 0, 1); }        \
    run<CUR, NT, PF, ALU>(t, 8, 1, 0); run<CUR, NT, PF, ALU>(t, 8, 4, 0); } while (0)

int main() {
    const int w = 1920, h = 1080, F = 32, w1 = 960, h1 = 540;
    const size_t n = (size_t)w * h * F, n1 = (size_t)w1 * h1 * F * 3;
    void *pL, *pAB, *c1, *out;
    CK(hipMalloc(&pL, n * 2 + 4096)); CK(hipMalloc(&pAB, n * 4 + 4096)); CK(hipMalloc(&c1, n1 * 4 + 4096)); CK(hipMalloc(&out, n * 3 + 4096));
    CK(hipMemset(pL, 1, n * 2)); CK(hipMemset(pAB, 2, n * 4)); CK(hipMemset(c1, 0, n1 * 4)); CK(hipMemset(out, 0, n * 3));
    g = SArgs{(const uint16_t*)pL, (const uint32_t*)pAB, (const float*)c1, (uint8_t*)out, w, h, w1, h1, 0, 0, F, 8, 0};
    g_bytes = (double)n * 12;
    printf("# %d frames of %d x %d; GB/s of compulsory bytes (12 B per pixel = %.0f MB per launch)\n", F, w, h, g_bytes / 1e6);
    for (int r = 0; r < 30; ++r) run<0, 0, 0, 0>("(clock ramp)", 8, 2, 0);
    SWEEP(0, 0, 0, 0); SWEEP(0, 0, 0, 32); SWEEP(0, 1, 0, 32); SWEEP(0, 1, 1, 32);
    SWEEP(1, 0, 0, 32); SWEEP(1, 1, 0, 32); SWEEP(1, 1, 1, 32); SWEEP(1, 1, 1, 0); SWEEP(1, 1, 1, 16);
    SWEEP(2, 1, 0, 32); SWEEP(2, 1, 1, 32); SWEEP(2, 1, 1, 0);
    return 0;
}
