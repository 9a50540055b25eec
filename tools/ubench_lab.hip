// Cycles per pixel of the colour arithmetic of the first / last Laplace kernels in isolation (registers + the two LDS
// tables, no global traffic), for candidate formulations.  Built with the library's flags (see tools/_run scripts):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude \
//         -Ilive-video-magnification_amd/csrc tools/ubench_lab.hip -o /tmp/ubench_lab [-fno-slp-vectorize]
#include <cstdio>
#include <vector>
#include "pyramid.h"
using namespace lvm;


// ---- experimental formulations measured here and NOT adopted by the library (see profiles/README.md, round 2) ----
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
// forward conversion with a wave-uniform shortcut past the CIE linear branch
__device__ __forceinline__ void lab_fwd4_shortcut(const float (&Bl)[4], const float (&Gl)[4], const float (&Rl)[4], const float* fw,
                                                  float (&L)[4], float (&a)[4], float (&b)[4]) {
    float X[4], Y[4], Z[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        X[k] = __builtin_fmaf(Bl[k], fw[0], __builtin_fmaf(Gl[k], fw[1], Rl[k] * fw[2]));
        Y[k] = __builtin_fmaf(Bl[k], fw[3], __builtin_fmaf(Gl[k], fw[4], Rl[k] * fw[5]));
        Z[k] = __builtin_fmaf(Bl[k], fw[6], __builtin_fmaf(Gl[k], fw[7], Rl[k] * fw[8]));
    }
    float m = min3f(X[0], Y[0], Z[0]);
#pragma unroll
    for (int k = 1; k < 4; ++k) m = __builtin_fminf(m, min3f(X[k], Y[k], Z[k]));
    if (__builtin_amdgcn_ballot_w64(!(m > 0.008856f)) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float FX = lab_cbrt<false>(X[k]), FY = lab_cbrt<false>(Y[k]), FZ = lab_cbrt<false>(Z[k]);
            L[k] = __builtin_fmaf(116.f, FY, -16.f); a[k] = 500.f * (FX - FY); b[k] = 200.f * (FY - FZ);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) lin_bgr_to_lab<false>(Bl[k], Gl[k], Rl[k], fw, L[k], a[k], b[k]);
    }
}
// inverse conversion: shortcut past the linear branches + analytic inverse gamma (exp2(log2(c) / 2.4)) when every
// linear-light value of the wave is >= 0.01, else the library's spline path
__device__ __forceinline__ void lab_inv4_shortcut(const float (&L)[4], const float (&a)[4], const float (&b)[4], const float* iv,
                                                  const float* igt, float (&o)[12]) {
    const float fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    float fy[4], fx[4], fz[4], c[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        fy[k] = (L[k] + 16.0f) * (1.0f / 116.0f);
        fx[k] = __builtin_fmaf(a[k], 1.0f / 500.0f, fy[k]);
        fz[k] = __builtin_fmaf(b[k], -1.0f / 200.0f, fy[k]);
    }
    float m = min3f(fy[0], fx[0], fz[0]);
#pragma unroll
    for (int k = 1; k < 4; ++k) m = __builtin_fminf(m, min3f(fy[k], fx[k], fz[k]));
    if (__builtin_amdgcn_ballot_w64(!(m > fThresh + 1.0e-4f)) != 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) lab_to_bgr<false>(L[k], a[k], b[k], iv, igt, o[3 * k], o[3 * k + 1], o[3 * k + 2]);
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float y = fy[k] * fy[k] * fy[k], x3 = fx[k] * fx[k] * fx[k], z3 = fz[k] * fz[k] * fz[k];
        c[3 * k] = __builtin_fmaf(iv[0], x3, __builtin_fmaf(iv[1], y, iv[2] * z3));
        c[3 * k + 1] = __builtin_fmaf(iv[3], x3, __builtin_fmaf(iv[4], y, iv[5] * z3));
        c[3 * k + 2] = __builtin_fmaf(iv[6], x3, __builtin_fmaf(iv[7], y, iv[8] * z3));
    }
    float m2 = min3f(c[0], c[1], c[2]);
#pragma unroll
    for (int k = 1; k < 4; ++k) m2 = __builtin_fminf(m2, min3f(c[3 * k], c[3 * k + 1], c[3 * k + 2]));
    if (__builtin_amdgcn_ballot_w64(!(m2 > 10.24f)) == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k)
            o[k] = __builtin_fmaf(__builtin_amdgcn_exp2f(__builtin_fmaf(__builtin_amdgcn_logf(c[k]), 1.0f / 2.4f, -10.0f / 2.4f)), 1.055f, -0.055f);
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) o[k] = spline1024<false>(clip1024_open(c[k]), igt);
    }
}

// ---- explicit pixel-pair arithmetic: every fma-class operation on a <2 x float> (v_pk_fma/mul/add_f32) ----------
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float x) { f2 r; r.x = x; r.y = x; return r; }
__device__ __forceinline__ f2 cbrt2(f2 x) {
    f2 l; l.x = __builtin_amdgcn_logf(x.x); l.y = __builtin_amdgcn_logf(x.y);
    l = l * splat(0.33333334f);
    f2 r; r.x = __builtin_amdgcn_exp2f(l.x); r.y = __builtin_amdgcn_exp2f(l.y);
    return r;
}
__device__ __forceinline__ f2 sel2(f2 x, float thr, f2 a, f2 b) { f2 r; r.x = x.x > thr ? a.x : b.x; r.y = x.y > thr ? a.y : b.y; return r; }
__device__ __forceinline__ void ub_fwd_pair(f2 B, f2 G, f2 R, const float* fw, f2& L, f2& a, f2& b) {
    const f2 X = fma2(B, splat(fw[0]), fma2(G, splat(fw[1]), R * splat(fw[2])));
    const f2 Y = fma2(B, splat(fw[3]), fma2(G, splat(fw[4]), R * splat(fw[5])));
    const f2 Z = fma2(B, splat(fw[6]), fma2(G, splat(fw[7]), R * splat(fw[8])));
    const f2 k = splat(7.787f), c = splat(16.0f / 116.0f);
    const f2 FX = sel2(X, 0.008856f, cbrt2(X), fma2(k, X, c));
    const f2 FY = sel2(Y, 0.008856f, cbrt2(Y), fma2(k, Y, c));
    const f2 FZ = sel2(Z, 0.008856f, cbrt2(Z), fma2(k, Z, c));
    L = fma2(splat(116.f), FY, splat(-16.f));
    a = splat(500.f) * (FX - FY);
    b = splat(200.f) * (FY - FZ);
}
__device__ __forceinline__ f2 spline2(f2 x, const float* tab) {
    const int i0 = (int)x.x, i1 = (int)x.y;
    f2 fr; fr.x = __builtin_amdgcn_fractf(x.x); fr.y = __builtin_amdgcn_fractf(x.y);
    const float4 t0 = *reinterpret_cast<const float4*>(tab + i0 * 4), t1 = *reinterpret_cast<const float4*>(tab + i1 * 4);
    f2 w; w.x = t0.w; w.y = t1.w; f2 z; z.x = t0.z; z.y = t1.z; f2 y; y.x = t0.y; y.y = t1.y; f2 xx; xx.x = t0.x; xx.y = t1.x;
    return fma2(fma2(fma2(w, fr, z), fr, y), fr, xx);
}
__device__ __forceinline__ f2 clip2(f2 v) { f2 r; r.x = clip1024_open(v.x); r.y = clip1024_open(v.y); return r; }
__device__ __forceinline__ void ub_inv_pair(f2 L, f2 a, f2 b, const float* iv, const float* igt, f2& o0, f2& o1, f2& o2) {
    const float lThresh = 0.008856f * 903.3f, fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    const f2 ylin = L * splat(1.0f / 903.3f), fyc = (L + splat(16.0f)) * splat(1.0f / 116.0f);
    const f2 fyl = fma2(splat(7.787f), ylin, splat(16.0f / 116.0f));
    f2 fy, y; const f2 y3 = fyc * fyc * fyc;
    fy.x = L.x <= lThresh ? fyl.x : fyc.x; fy.y = L.y <= lThresh ? fyl.y : fyc.y;
    y.x = L.x <= lThresh ? ylin.x : y3.x; y.y = L.y <= lThresh ? ylin.y : y3.y;
    f2 fx = fma2(a, splat(1.0f / 500.0f), fy), fz = fma2(b, splat(-1.0f / 200.0f), fy);
    const f2 fxl = (fx - splat(16.0f / 116.0f)) * splat(1.0f / 7.787f), fzl = (fz - splat(16.0f / 116.0f)) * splat(1.0f / 7.787f);
    const f2 fx3 = fx * fx * fx, fz3 = fz * fz * fz;
    fx.x = fx.x <= fThresh ? fxl.x : fx3.x; fx.y = fx.y <= fThresh ? fxl.y : fx3.y;
    fz.x = fz.x <= fThresh ? fzl.x : fz3.x; fz.y = fz.y <= fThresh ? fzl.y : fz3.y;
    const f2 c0 = fma2(splat(iv[0]), fx, fma2(splat(iv[1]), y, splat(iv[2]) * fz));
    const f2 c1 = fma2(splat(iv[3]), fx, fma2(splat(iv[4]), y, splat(iv[5]) * fz));
    const f2 c2 = fma2(splat(iv[6]), fx, fma2(splat(iv[7]), y, splat(iv[8]) * fz));
    o0 = spline2(clip2(c0), igt); o1 = spline2(clip2(c1), igt); o2 = spline2(clip2(c2), igt);
}

#define ITERS 512
// VARIANT 0: round-1 per-pixel functions (selects everywhere); 1: 4-pixel functions with the wave-uniform shortcut
// WHAT 0: forward + motion add + inverse + pack (last kernel); 1: forward only (first kernel)
template <int VARIANT, int WHAT>
__global__ __launch_bounds__(512) void k_lab(unsigned* out, LabCoef lab, unsigned seed, float mscale) {
    __shared__ __attribute__((aligned(16))) float s_igt[4096];
    __shared__ float s_gam[256];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) reinterpret_cast<float4*>(s_igt)[i] = reinterpret_cast<const float4*>(lab.invgamma)[i];
    if (threadIdx.x < 256) s_gam[threadIdx.x] = lab.gamma_u8[threadIdx.x];
    __syncthreads();
    unsigned st = seed ^ (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    unsigned acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        Px4 pin;
        // bytes 48..175: every pixel above the CIE / gamma thresholds (the shortcuts fire); the benchmark clip has 3.5 % of its
        // pixels below 26 and a pixel below the thresholds in practically every 256-pixel wave row: there they never fire
        st = st * 1664525u + 1013904223u; pin.a = (st & 0x7f7f7f7fu) + 0x30303030u;
        st = st * 1664525u + 1013904223u; pin.b = (st & 0x7f7f7f7fu) + 0x30303030u;
        st = st * 1664525u + 1013904223u; pin.c = (st & 0x7f7f7f7fu) + 0x30303030u;
        int Bv[4], Gv[4], Rv[4];
        unpack_px4(pin, Bv, Gv, Rv);
        float L[4], A[4], Bb[4];
        if (VARIANT == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) lin_bgr_to_lab<false>(s_gam[Bv[k]], s_gam[Gv[k]], s_gam[Rv[k]], lab.fwd, L[k], A[k], Bb[k]);
        } else if (VARIANT == 2) {
#pragma unroll
            for (int k = 0; k < 4; k += 2) {
                f2 Bp, Gp, Rp, Lp, Ap, Bq;
                Bp.x = s_gam[Bv[k]]; Bp.y = s_gam[Bv[k + 1]]; Gp.x = s_gam[Gv[k]]; Gp.y = s_gam[Gv[k + 1]]; Rp.x = s_gam[Rv[k]]; Rp.y = s_gam[Rv[k + 1]];
                ub_fwd_pair(Bp, Gp, Rp, lab.fwd, Lp, Ap, Bq);
                L[k] = Lp.x; L[k + 1] = Lp.y; A[k] = Ap.x; A[k + 1] = Ap.y; Bb[k] = Bq.x; Bb[k + 1] = Bq.y;
            }
        } else {
            float Bl[4], Gl[4], Rl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { Bl[k] = s_gam[Bv[k]]; Gl[k] = s_gam[Gv[k]]; Rl[k] = s_gam[Rv[k]]; }
            lab_fwd4_shortcut(Bl, Gl, Rl, lab.fwd, L, A, Bb);
        }
        if (WHAT == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc ^= __float_as_uint(L[k]) + __float_as_uint(A[k]) * 3u + __float_as_uint(Bb[k]) * 5u;
            st ^= acc >> 9;
            continue;
        }
        const float m0 = mscale * (float)(st >> 24), m1 = mscale * (float)((st >> 16) & 255), m2 = mscale * (float)((st >> 8) & 255);
        float ov[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) { L[k] = __builtin_fmaf(m0, 0.015625f, L[k]); A[k] = __builtin_fmaf(m1, 0.0015625f, A[k]); Bb[k] = __builtin_fmaf(m2, 0.0015625f, Bb[k]); }
        if (VARIANT == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) lab_to_bgr<false>(L[k], A[k], Bb[k], lab.inv1024, s_igt, ov[3 * k], ov[3 * k + 1], ov[3 * k + 2]);
        } else if (VARIANT == 2) {
#pragma unroll
            for (int k = 0; k < 4; k += 2) {
                f2 Lp, Ap, Bq, o0, o1, o2;
                Lp.x = L[k]; Lp.y = L[k + 1]; Ap.x = A[k]; Ap.y = A[k + 1]; Bq.x = Bb[k]; Bq.y = Bb[k + 1];
                ub_inv_pair(Lp, Ap, Bq, lab.inv1024, s_igt, o0, o1, o2);
                ov[3 * k] = o0.x; ov[3 * k + 1] = o1.x; ov[3 * k + 2] = o2.x; ov[3 * k + 3] = o0.y; ov[3 * k + 4] = o1.y; ov[3 * k + 5] = o2.y;
            }
        } else {
            lab_inv4_shortcut(L, A, Bb, lab.inv1024, s_igt, ov);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) ov[k] = __builtin_fmaf(ov[k], 255.0f, lab.a255);
        const unsigned qa = pack_u8x4(ov[0], ov[1], ov[2], ov[3]), qb = pack_u8x4(ov[4], ov[5], ov[6], ov[7]), qc = pack_u8x4(ov[8], ov[9], ov[10], ov[11]);
        acc ^= qa + qb * 3u + qc * 5u;
        st ^= acc >> 9;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    float g[256], ig[4096];
    LabCoef lab{};
    build_lab_tables(g, ig, lab.fwd, lab.inv);
    for (int i = 0; i < 9; ++i) lab.inv1024[i] = lab.inv[i] * 1024.0f;
    float *dg, *dig; unsigned* dout;
    (void)hipMalloc(&dg, sizeof(g)); (void)hipMalloc(&dig, sizeof(ig)); (void)hipMalloc(&dout, 64);
    (void)hipMemcpy(dg, g, sizeof(g), hipMemcpyHostToDevice); (void)hipMemcpy(dig, ig, sizeof(ig), hipMemcpyHostToDevice);
    lab.gamma_u8 = dg; lab.invgamma = dig; lab.a255 = (float)(1.0 / 255.0f);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 2;                       // 2 x 512 threads per CU = 4 waves per SIMD
    struct { const char* name; void (*k)(unsigned*, LabCoef, unsigned, float); } ks[] = {
        {"last kernel colour math, round-1 functions", k_lab<0, 0>}, {"last kernel colour math, uniform shortcut", k_lab<1, 0>},
        {"first kernel colour math, round-1 functions", k_lab<0, 1>}, {"first kernel colour math, uniform shortcut", k_lab<1, 1>},
        {"last kernel colour math, explicit pixel pairs", k_lab<2, 0>}, {"first kernel colour math, explicit pixel pairs", k_lab<2, 1>}};
    for (auto& e : ks) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(512), 0, 0, dout, lab, 12345u, 0.01f);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        // per SIMD: 4 waves x ITERS x 4 pixels (per lane)
        const double ns_px = (double)best * 1e6 / (4.0 * ITERS * 4);
        printf("%-48s %8.3f ms  %7.3f ns per wave-pixel per SIMD (%.1f cycles at 2.4 GHz) -> a 1080p frame: %.2f us\n", e.name, best, ns_px,
               ns_px * 2.4, ns_px * 1920.0 * 1080.0 / 64.0 / 1024.0 * 1e-3);
    }
    // occupancy sweep (round 2, late): the same arithmetic with W = 1, 2, 4, 8 waves per SIMD (W workgroups of 256 threads per CU).
    // What a kernel that owns its pixels for a whole temporal batch (IIR state in registers) would get from a single stream.
    for (int W : {1, 2, 4, 8}) {
        for (int which = 0; which < 2; ++which) {
            auto k = which ? k_lab<0, 1> : k_lab<0, 0>;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k, dim3(256 * W), dim3(256), 0, 0, dout, lab, 12345u, 0.01f);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double ns_px = (double)best * 1e6 / ((double)W * ITERS * 4);
            printf("W = %d waves per SIMD, %s kernel colour math: %8.3f ms  %7.3f ns per wave-pixel per SIMD -> a 1080p frame: %.2f us\n", W,
                   which ? "first" : "last", best, ns_px, ns_px * 1920.0 * 1080.0 / 64.0 / 1024.0 * 1e-3);
        }
    }
    return 0;
}
