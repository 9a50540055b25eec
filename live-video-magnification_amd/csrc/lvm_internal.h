// lvm_internal.h -- shared host/device declarations of liblvm_hip.so (gfx950 only).
//
// Arithmetic contract (DESIGN.md "Numerics"): float32 storage, every reference Mat-level
// operation rounded on its own, FMA only where the reference's OpenCV inner loops use one
// (filter taps).  The translation units are built with -ffp-contract=off and spell every
// fused multiply-add as __builtin_fmaf, so results do not depend on compiler contraction.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "lvm_hip.h"
#include "lab_lut.h"
#include <lvm_gfx950.h>   // gfx950-only primitives (constant-address-space tables, raw buffer loads / stores, v_dot2 / v_perm / v_mul_u32_u24)

namespace lvm {

constexpr int kMaxLevels = 24;

// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------
// OpenCV borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}
// value select that stays a v_cndmask (the hint keeps the optimiser from turning it back into a
// divergent branch around the evaluation of one operand)
__device__ __forceinline__ float sel(bool c, float a, float b) { return __builtin_unpredictable(c) ? a : b; }

// convertTo(CV_8U): saturate_cast<uchar>(cvRound(v)), cvRound = round-half-even
__device__ __forceinline__ uint8_t sat_u8(float v) {
    if (!(v == v)) return 0;
    float r = rintf(v);
    r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
    return (uint8_t)r;
}

// Four convertTo(CV_8U) results packed into one dword.  v_cvt_pk_u8_f32 converts with round-half-even,
// saturates to [0, 255] and maps NaN to 0 (checked on gfx950 with tools/probe_isa.hip), i.e. it IS
// saturate_cast<uchar>(cvRound(v)), and it writes the byte in place: 4 instructions per dword.
__device__ __forceinline__ uint32_t pack_u8x4(float v0, float v1, float v2, float v3) {
    uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(v0, 0, 0);
    r = __builtin_amdgcn_cvt_pk_u8_f32(v1, 1, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(v2, 2, r);
    return __builtin_amdgcn_cvt_pk_u8_f32(v3, 3, r);
}

// Whole-wave shifts by one lane (DPP wave_shr:1 / wave_shl:1 cross all 64 lanes on gfx950, tools/probe_isa.hip):
// lane i receives lane i - 1's (shr) / lane i + 1's (shl) value; lane 0 / lane 63 receive 0.  bound_ctrl = 1 ("out-of-range source
// lanes read 0"): without it the instruction KEEPS the destination for those lanes, so the compiler has to write the 0 first -- one
// v_mov_b32 in front of every DPP move, 520 of the 4435 vector instructions of k_rz_blur_strips until round 4.
__device__ __forceinline__ float dpp_shr1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }

// ---- XCD-aware workgroup order ---------------------------------------------------------------------------------------
// MI355X has 8 XCDs with private 4 MB L2s and places workgroup b of a launch on XCD b % 8 (observed, not contractual --
// MI355X_MICROARCH.md; used for speed only, results never depend on it).  With tiles handed out in launch order,
// neighbouring tiles -- which share halo rows / columns -- land on different L2s and every shared line is fetched again
// through the fabric.  The remap gives every XCD a CONTIGUOUS range of the linear tile order instead (bijective for any
// workgroup count), so halo re-reads hit the XCD's own L2.  LVM_XCD_SWIZZLE=0 builds without it (A/B measurements).
// Measured per kernel (round 3, profiles/README.md) and kept only where it pays: k_lap_up at level 1 117 -> 105 us per 32
// frames, k_rz_final 400 -> 391; the Riesz blur kernel got SLOWER (730 -> 875 us: eight distant regions of seven planes
// streamed at once) and keeps the launch order, the split / collapse / phase kernels did not move.
#ifndef LVM_XCD_SWIZZLE
#define LVM_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned nb) {
#if LVM_XCD_SWIZZLE
    const unsigned q = nb >> 3, r = nb & 7u, x = b & 7u, i = b >> 3;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + i;
#else
    (void)nb; return b;
#endif
}
struct Bid3 { int x, y, z; };
// the same for a 3-D grid (dispatch order: x fastest)
__device__ __forceinline__ Bid3 xcd_swizzle3() {
    Bid3 r;
#if LVM_XCD_SWIZZLE
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned s = xcd_swizzle(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
    const unsigned t = s / gx;
    r.x = (int)(s - t * gx); r.z = (int)(t / gy); r.y = (int)(t - (unsigned)r.z * gy);
#else
    r.x = blockIdx.x; r.y = blockIdx.y; r.z = blockIdx.z;
#endif
    return r;
}

// Lab conversion tables/coefficients (reference: MagnifyCore.hpp:90,152,219,275 call
// cv::cvtColor COLOR_BGR2Lab / COLOR_Lab2BGR on float [0,1]; OpenCV 4 color_lab.cpp float path).
#ifndef LVM_FAST_FMA
#define LVM_FAST_FMA 1      // default (non-exact) flavour: pyramid tap sums of the first / last kernels as fma chains
#endif
struct LabCoef {
    float fwd[9];             // BGR(linear) -> XYZ/white, row-major, column 0 multiplies B
    float inv[9];             // XYZ -> BGR(linear), row 0 produces B
    float inv1024[9];         // inv * 1024 (exact): the fast flavour gets the spline coordinate straight out of the matrix
    float inv4096[9];         // inv * 4096 (exact): ... and the slice coordinate of the u8 step table (u8_step)
    const uint2* u8steps;     // [kU8StepSlices] { threshold * 4096 | +inf, value at the slice's start } (device; lab_tables.cpp build_u8_steps)
    const float* gamma_u8;    // [256]  sRGB gamma of u8/255 (device)
    const float* invgamma;    // [1024*4] cubic-spline coefficients of the inverse gamma (device)
    float a255;               // float(1.0/255.0f)
};
// Integer Lab planes of the frames of a launch (labconv.hip): frame b's pixel (y, x) at (b * h + y) * w + x.  Null in the
// analytic flavour and for non-Lab input (gray frames, the colour mode).
struct LabPlanes { const uint16_t* iL; const uint32_t* iab; };

// cv::cubeRoot (core/mathfuncs.cpp): exponent split + quartic rational polynomial in float64
template <bool IEEE_DIV = true>
__device__ __forceinline__ float cv_cube_root(float value) {
    int vi = __float_as_int(value);
    int ix = vi & 0x7fffffff;
    unsigned s = (unsigned)vi & 0x80000000u;
    int ex = (ix >> 23) - 127;
    int shx = ex % 3;
    shx -= shx >= 0 ? 3 : 0;
    ex = (ex - shx) / 3;
    double fr = (double)__int_as_float((ix & ((1 << 23) - 1)) | ((shx + 127) << 23));
    const double num = ((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr +
                          119.1654824285581628956914143) * fr + 13.43250139086239872172837314) * fr +
                        0.1636161226585754240958355063);
    const double den = ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr +
                          168.5254414101568283957668343) * fr + 33.9905941350215598754191872) * fr + 1.0);
    if (IEEE_DIV) fr = num / den;
    else {
        // den is in [1, 400]: reciprocal estimate + two Newton steps (relative error ~1e-16 instead of the correctly rounded
        // quotient); the float64 result is rounded to float next, so the two forms differ for ~1e-9 of the arguments (one ulp)
        double r = __builtin_amdgcn_rcp(den);
        r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
        fr = num * r;
    }
    unsigned r = (unsigned)__float_as_int((float)fr);
    r = (r + ((unsigned)ex << 23) + s) & ((((unsigned)vi * 2u) != 0u) ? 0xffffffffu : 0u);
    return __int_as_float((int)r);
}

// Cube root used by the forward Lab conversion.  EXACT = cv::cubeRoot (float64 rational polynomial,
// the oracle's arithmetic); otherwise exp2(log2(x)/3) on the hardware transcendental units
// (v_log_f32 / v_exp_f32, x is never denormal here): relative error ~2e-7, i.e. L moves by < 3e-5 of
// its 0..100 range, at 3 instructions instead of ~60.
template <bool EXACT>
__device__ __forceinline__ float lab_cbrt(float x) {
    if (EXACT) return cv_cube_root<true>(x);
    return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * 0.33333334f);
}
// RGB2Lab_f scalar path on gamma-expanded B,G,R.  The non-EXACT flavour contracts the matrix rows
// into fma chains and evaluates both branches of f(t) (select instead of divergent branches).
template <bool EXACT>
__device__ __forceinline__ void lin_bgr_to_lab(float B, float G, float R, const float* fw, float& L,
                                               float& a, float& b) {
    const float _a = 16.0f / 116.0f;
    float X, Y, Z, FX, FY, FZ;
    if (EXACT) {
        X = B * fw[0] + G * fw[1] + R * fw[2];
        Y = B * fw[3] + G * fw[4] + R * fw[5];
        Z = B * fw[6] + G * fw[7] + R * fw[8];
        FX = X > 0.008856f ? lab_cbrt<true>(X) : (7.787f * X + _a);
        FY = Y > 0.008856f ? lab_cbrt<true>(Y) : (7.787f * Y + _a);
        FZ = Z > 0.008856f ? lab_cbrt<true>(Z) : (7.787f * Z + _a);
        L = Y > 0.008856f ? (116.f * FY - 16.f) : (903.3f * Y);
    } else {
        X = __builtin_fmaf(B, fw[0], __builtin_fmaf(G, fw[1], R * fw[2]));
        Y = __builtin_fmaf(B, fw[3], __builtin_fmaf(G, fw[4], R * fw[5]));
        Z = __builtin_fmaf(B, fw[6], __builtin_fmaf(G, fw[7], R * fw[8]));
        // (X, Y, Z >= 0: the cube root of a value under the threshold is finite and discarded below)
        const float cx = lab_cbrt<false>(X), cy = lab_cbrt<false>(Y), cz = lab_cbrt<false>(Z);
        FX = X > 0.008856f ? cx : __builtin_fmaf(7.787f, X, _a);
        FY = Y > 0.008856f ? cy : __builtin_fmaf(7.787f, Y, _a);
        FZ = Z > 0.008856f ? cz : __builtin_fmaf(7.787f, Z, _a);
        // L = 116 f(Y) - 16 on both sides of the threshold: below it 116 (7.787 Y + 16/116) - 16 = 903.292 Y against the
        // reference's 903.3 Y, < 7e-5 of the 0..100 range (one select and one multiply less per pixel)
        L = __builtin_fmaf(116.f, FY, -16.f);
    }
    a = 500.f * (FX - FY);
    b = 200.f * (FY - FZ);
}
// splineInterpolate (color_lab.cpp), 1024 knots; tab is 16-byte aligned (one 128-bit read per knot)
template <bool EXACT>
__device__ __forceinline__ float spline1024(float x, const float* tab) {
    if (EXACT) {
        int ix = (int)x;
        ix = ix < 0 ? 0 : (ix > 1023 ? 1023 : ix);
        x -= (float)ix;
        const float4 t = *reinterpret_cast<const float4*>(tab + ix * 4);
        return ((t.w * x + t.z) * x + t.y) * x + t.x;
    }
    // x was clamped to [0, 1024) by the caller: knot = trunc(x), argument = fract(x) (one v_fract_f32)
    const int ix = (int)x;
    const float fr = __builtin_amdgcn_fractf(x);
    const float4 t = *reinterpret_cast<const float4*>(tab + ix * 4);
    return __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(t.w, fr, t.z), fr, t.y), fr, t.x);
}
__device__ __forceinline__ float clip01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
// same clamp as one v_med3_f32 (inputs are finite on the fast path)
__device__ __forceinline__ float clip01_fast(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 1.f); }
// clamp to [0, 1): the largest float below 1 instead of 1 moves the inverse-gamma result by < 1e-7 and
// lets spline1024 skip the knot clamp
__device__ __forceinline__ float clip01_open(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 0.99999994f); }
__device__ __forceinline__ float clip1024_open(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 1023.99994f); }   // = clip01_open(v / 1024) * 1024
// The output quantiser u8 = cvRound(255 * invGamma(clip01(c)) + 1/255) as a table hit + a compare (lab_tables.cpp build_u8_steps): c4096 = 4096 c;
// s_steps = the table in LDS.  NaN -> 0 like the spline path (v_med3_f32 returns the minimum when an input is NaN).
__device__ __forceinline__ uint32_t u8_step(float c4096, const uint2* s_steps) {
    const float cc = __builtin_amdgcn_fmed3f(c4096, 0.f, 4095.99976f);
    const uint2 e = s_steps[(int)cc];
    return e.y + (cc >= __uint_as_float(e.x) ? 1u : 0u);
}
// convertTo(CV_8U) for finite input: round-half-even, clamp, convert (3 instructions)
__device__ __forceinline__ uint32_t sat_u8_fast(float v) { return (uint32_t)__builtin_amdgcn_fmed3f(rintf(v), 0.f, 255.f); }
// The default flavour's Lab -> linear BGR: reciprocal multiplies, fma chains, selects.  iv = inv scaled by a power of two (inv1024 /
// inv4096): the scale commutes with every rounding, so the spline coordinate 1024 c / the step-table coordinate 4096 c comes out directly.
__device__ __forceinline__ void lab_to_linear_fast(float li, float ai, float bi, const float* iv, float& c0, float& c1, float& c2) {
    const float lThresh = 0.008856f * 903.3f;
    const float fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    const float ylin = li * (1.0f / 903.3f), fyc = (li + 16.0f) * (1.0f / 116.0f);
    const bool lo = li <= lThresh;
    const float fy = lo ? __builtin_fmaf(7.787f, ylin, 16.0f / 116.0f) : fyc;
    const float y = lo ? ylin : fyc * fyc * fyc;
    float fx = __builtin_fmaf(ai, 1.0f / 500.0f, fy), fz = __builtin_fmaf(bi, -1.0f / 200.0f, fy);
    fx = (fx <= fThresh) ? (fx - 16.0f / 116.0f) * (1.0f / 7.787f) : fx * fx * fx;
    fz = (fz <= fThresh) ? (fz - 16.0f / 116.0f) * (1.0f / 7.787f) : fz * fz * fz;
    c0 = __builtin_fmaf(iv[0], fx, __builtin_fmaf(iv[1], y, iv[2] * fz));
    c1 = __builtin_fmaf(iv[3], fx, __builtin_fmaf(iv[4], y, iv[5] * fz));
    c2 = __builtin_fmaf(iv[6], fx, __builtin_fmaf(iv[7], y, iv[8] * fz));
}
// ... and straight on to the three output bytes (round 6): the quantiser as a step table instead of spline + scale + round + clamp
__device__ __forceinline__ void lab_to_u8(float li, float ai, float bi, const float* iv4096, const uint2* s_steps, uint32_t& u0, uint32_t& u1, uint32_t& u2) {
    float c0, c1, c2;
    lab_to_linear_fast(li, ai, bi, iv4096, c0, c1, c2);
    u0 = u8_step(c0, s_steps); u1 = u8_step(c1, s_steps); u2 = u8_step(c2, s_steps);
}
// Lab2RGBfloat::process + inverse gamma; igt = inverse-gamma spline table (LDS).  EXACT keeps
// OpenCV's divisions by 903.3 / 116 / 500 / 200 / 7.787 and its unfused products; otherwise
// reciprocal multiplies, fma chains and selects.
template <bool EXACT>
__device__ __forceinline__ void lab_to_bgr(float li, float ai, float bi, const float* iv,
                                           const float* igt, float& o0, float& o1, float& o2) {
    const float lThresh = 0.008856f * 903.3f;
    const float fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    float y, fy, fx, fz, c0, c1, c2;
    if (EXACT) {
        if (li <= lThresh) { y = li / 903.3f; fy = 7.787f * y + 16.0f / 116.0f; }
        else { fy = (li + 16.0f) / 116.0f; y = fy * fy * fy; }
        fx = ai / 500.0f + fy; fz = fy - bi / 200.0f;
        fx = (fx <= fThresh) ? (fx - 16.0f / 116.0f) / 7.787f : fx * fx * fx;
        fz = (fz <= fThresh) ? (fz - 16.0f / 116.0f) / 7.787f : fz * fz * fz;
        c0 = iv[0] * fx + iv[1] * y + iv[2] * fz;
        c1 = iv[3] * fx + iv[4] * y + iv[5] * fz;
        c2 = iv[6] * fx + iv[7] * y + iv[8] * fz;
    } else {
        lab_to_linear_fast(li, ai, bi, iv, c0, c1, c2);      // iv = inv1024 here
    }
    if (EXACT) {
        o0 = spline1024<true>(clip01(c0) * 1024.f, igt);
        o1 = spline1024<true>(clip01(c1) * 1024.f, igt);
        o2 = spline1024<true>(clip01(c2) * 1024.f, igt);
    } else {
        o0 = spline1024<false>(clip1024_open(c0), igt);
        o1 = spline1024<false>(clip1024_open(c1), igt);
        o2 = spline1024<false>(clip1024_open(c2), igt);
    }
}
// The default flavour's Lab2RGBfloat for a PAIR of pixels: the same operations as lab_to_bgr<false>, two pixels per v_pk_*_f32
// (gfx950 issues a packed FP32 operation at the rate of a scalar one: the matrix, the cube / linear branches and the scalings are
// half the instructions; the spline look-ups, clamps and selects stay per pixel).  iv = inv1024.
// (lvm_f2 / f2_fma: lvm_gfx950.h -- v_pk_fma_f32)
__device__ __forceinline__ lvm_f2 f2_set(float a, float b) { lvm_f2 v = {a, b}; return v; }
__device__ __forceinline__ lvm_f2 f2_all(float a) { lvm_f2 v = {a, a}; return v; }
__device__ __forceinline__ void lab_to_linear_pair(lvm_f2 li, lvm_f2 ai, lvm_f2 bi, const float* iv, lvm_f2& c0, lvm_f2& c1, lvm_f2& c2) {
    const float lThresh = 0.008856f * 903.3f;
    const float fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    const lvm_f2 ylin = li * f2_all(1.0f / 903.3f), fyc = (li + f2_all(16.0f)) * f2_all(1.0f / 116.0f);
    const lvm_f2 fyl = f2_fma(f2_all(7.787f), ylin, f2_all(16.0f / 116.0f)), yc = fyc * fyc * fyc;
    lvm_f2 fy, y;
#pragma unroll
    for (int k = 0; k < 2; ++k) { const bool lo = li[k] <= lThresh; fy[k] = lo ? fyl[k] : fyc[k]; y[k] = lo ? ylin[k] : yc[k]; }
    lvm_f2 fx = f2_fma(ai, f2_all(1.0f / 500.0f), fy), fz = f2_fma(bi, f2_all(-1.0f / 200.0f), fy);
    const lvm_f2 fxl = (fx - f2_all(16.0f / 116.0f)) * f2_all(1.0f / 7.787f), fxc = fx * fx * fx;
    const lvm_f2 fzl = (fz - f2_all(16.0f / 116.0f)) * f2_all(1.0f / 7.787f), fzc = fz * fz * fz;
#pragma unroll
    for (int k = 0; k < 2; ++k) { fx[k] = (fx[k] <= fThresh) ? fxl[k] : fxc[k]; fz[k] = (fz[k] <= fThresh) ? fzl[k] : fzc[k]; }
    c0 = f2_fma(f2_all(iv[0]), fx, f2_fma(f2_all(iv[1]), y, f2_all(iv[2]) * fz));
    c1 = f2_fma(f2_all(iv[3]), fx, f2_fma(f2_all(iv[4]), y, f2_all(iv[5]) * fz));
    c2 = f2_fma(f2_all(iv[6]), fx, f2_fma(f2_all(iv[7]), y, f2_all(iv[8]) * fz));
}
__device__ __forceinline__ void lab_to_bgr_pair(lvm_f2 li, lvm_f2 ai, lvm_f2 bi, const float* iv, const float* igt, lvm_f2& o0, lvm_f2& o1, lvm_f2& o2) {
    lvm_f2 c0, c1, c2;
    lab_to_linear_pair(li, ai, bi, iv, c0, c1, c2);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        o0[k] = spline1024<false>(clip1024_open(c0[k]), igt);
        o1[k] = spline1024<false>(clip1024_open(c1[k]), igt);
        o2[k] = spline1024<false>(clip1024_open(c2[k]), igt);
    }
}
// ---- flavours of the colour arithmetic (template parameter FL of every kernel that converts) ------------------------
// FL_LUT_FAST  (default): forward = OpenCV 4's interpolated 33^3 table (integer, bit-exact against the oracle), inverse and
//              pyramid taps with reciprocal multiplies / fma chains / selects;
// FL_LUT_EXACT (lvm_debug_exact_lab): the same forward table, every float operation in OpenCV's order (bit-identical to the
//              oracle on the CPU emulation build);
// FL_ANALYTIC  (lvm_debug_lab_analytic): the cube-root form RGB2Lab_f computes when its interpolation is switched off,
//              OpenCV's operation order (the oracle with lvmo_set_lab_lut(0)).
enum { FL_LUT_FAST = 0, FL_LUT_EXACT = 1, FL_ANALYTIC = 2 };
constexpr bool fl_exact(int FL) { return FL != FL_LUT_FAST; }
constexpr bool fl_lut(int FL) { return FL != FL_ANALYTIC; }
// Output kernels: the default flavour quantises through the u8 step table (u8_step) unless the float frame is asked for
// (lvm_debug_keep_float: the spline's float values are the product then); the two debug flavours keep OpenCV's operations one by one.
#ifndef LVM_U8_STEPS
#define LVM_U8_STEPS 1       // 0: spline + scale + round in every flavour (A/B measurements)
#endif
constexpr bool fin_steps(int FL, bool DBG) { return LVM_U8_STEPS && FL == FL_LUT_FAST && !DBG; }
// Lab of pixel (gy, gx) of frame b.  LUT flavours: the integer planes written by labconv.hip (every frame is converted
// once); analytic flavour: the u8 frame + the 256-entry gamma table in LDS (s_gam).
template <int FL>
__device__ __forceinline__ void fetch_lab_px(const uint8_t* __restrict__ frame, long in_stride, const LabPlanes& lp, size_t plane_off, int w,
                                             int gy, int gx, const float* s_gam, const LabCoef& lab, float& L, float& a, float& b) {
    if (fl_lut(FL)) {
        const size_t i = plane_off + (size_t)gy * w + gx;
        lab_from_planes(lp.iL[i], lp.iab[i], L, a, b);
    } else {
        const uint8_t* p = frame + (size_t)gy * in_stride + (size_t)gx * 3;
        lin_bgr_to_lab<true>(s_gam[p[0]], s_gam[p[1]], s_gam[p[2]], lab.fwd, L, a, b);
    }
}
// A group of 4 adjacent pixels as loaded: u8 flavour d[0..2] = 12 bytes of BGR; plane flavour d[0..1] = 4 x iL,
// d[2..5] = 4 x (ia | ib << 16).  (gx % 4 == 0, rows dword aligned.)
struct Raw4 { uint32_t d[6]; };
template <bool PLANES, bool STREAM = false>     // STREAM: the planes are read once by this launch (nontemporal, lvm_gfx950.h)
__device__ __forceinline__ Raw4 load_raw4(const uint8_t* __restrict__ frame, long in_stride, const LabPlanes& lp, size_t plane_off, int w, int gy, unsigned gx) {
    Raw4 r{};
    if (PLANES) {
        const size_t i = plane_off + (size_t)gy * w + gx;
        const uint2 l = STREAM ? ld_stream_u32x2(lp.iL + i) : *reinterpret_cast<const uint2*>(lp.iL + i);
        const uint4 ab = STREAM ? ld_stream_u32x4(lp.iab + i) : *reinterpret_cast<const uint4*>(lp.iab + i);
        r.d[0] = l.x; r.d[1] = l.y; r.d[2] = ab.x; r.d[3] = ab.y; r.d[4] = ab.z; r.d[5] = ab.w;
    } else {
        struct __attribute__((packed, aligned(4))) P3 { uint32_t a, b, c; };
        const P3 v = *reinterpret_cast<const P3*>(frame + (size_t)gy * in_stride + gx * 3u);
        r.d[0] = v.a; r.d[1] = v.b; r.d[2] = v.c;
    }
    return r;
}
// pixel k of a group as B, G, R bytes (u8 flavour)
__device__ __forceinline__ void raw4_bgr(const Raw4& r, int (&B)[4], int (&G)[4], int (&R)[4]) {
    B[0] = r.d[0] & 255; G[0] = (r.d[0] >> 8) & 255; R[0] = (r.d[0] >> 16) & 255;
    B[1] = r.d[0] >> 24; G[1] = r.d[1] & 255; R[1] = (r.d[1] >> 8) & 255;
    B[2] = (r.d[1] >> 16) & 255; G[2] = r.d[1] >> 24; R[2] = r.d[2] & 255;
    B[3] = (r.d[2] >> 8) & 255; G[3] = (r.d[2] >> 16) & 255; R[3] = r.d[2] >> 24;
}
// the 4 pixels of a group as Lab
template <int FL>
__device__ __forceinline__ void raw4_to_lab(const Raw4& r, const float* s_gam, const LabCoef& lab, float (&L)[4], float (&a)[4], float (&b)[4]) {
    if (fl_lut(FL)) {
        lab_from_planes(r.d[0] & 0xffffu, r.d[2], L[0], a[0], b[0]); lab_from_planes(r.d[0] >> 16, r.d[3], L[1], a[1], b[1]);
        lab_from_planes(r.d[1] & 0xffffu, r.d[4], L[2], a[2], b[2]); lab_from_planes(r.d[1] >> 16, r.d[5], L[3], a[3], b[3]);
    } else {
        int B[4], G[4], R[4];
        raw4_bgr(r, B, G, R);
#pragma unroll
        for (int k = 0; k < 4; ++k) lin_bgr_to_lab<true>(s_gam[B[k]], s_gam[G[k]], s_gam[R[k]], lab.fwd, L[k], a[k], b[k]);
    }
}
// the u8 step table into LDS (any workgroup size)
__device__ __forceinline__ void load_u8steps(uint2* s_steps, const uint2* g) {
    const uint4* src = reinterpret_cast<const uint4*>(g);
    uint4* dst = reinterpret_cast<uint4*>(s_steps);
    for (int i = threadIdx.x; i < kU8StepSlices / 2; i += blockDim.x) dst[i] = src[i];
}
// cooperative loads of the two Lab tables into LDS (256 threads)
__device__ __forceinline__ void load_gamma_u8(float* s_gam, const float* g) { s_gam[threadIdx.x] = g[threadIdx.x]; }
__device__ __forceinline__ void load_invgamma(float* s_igt, const float* g) {
    const float4* src = reinterpret_cast<const float4*>(g);
    float4* dst = reinterpret_cast<float4*>(s_igt);
    for (int i = threadIdx.x; i < 1024; i += 256) dst[i] = src[i];
}

// pyrUp horizontal pass for destination column gx from source row `s` whose element for
// source column i sits at s[i - sx0] (OpenCV pyrUp_ border rules, see laplace.hip).
__device__ __forceinline__ float pyrup_h(const float* s, int gx, int sx0, int sw) {
    // all border variants are evaluated on clamped indices and the result selected: no divergent branches
    const int i = gx >> 1, li = i - sx0;
    const bool first = i == 0, last = i == sw - 1;
    const float sm1 = s[first ? li : li - 1], s0 = s[li], s1 = s[last ? li : li + 1];
    const float p6 = s0 * 6.f;
    const float even = sel(first, p6 + s1 * 2.f, sel(last, sm1 + s0 * 7.f, sm1 + p6 + s1));
    const float odd = sel(last, s0 * 8.f, (s0 + s1) * 4.f);
    return sel((gx & 1) == 0, even, odd);
}

// ---------------------------------------------------------------------------------------
// host-side context
// ---------------------------------------------------------------------------------------
struct ProfEvent { int name; hipEvent_t e0, e1; };
struct ProfTotal { std::string name; double ms = 0; long long n = 0; };

struct LevelGeom { int w, h; size_t n; };   // n = w*h

struct Ctx;
int lab_flavour(const Ctx* c);
struct FrameIO {
    const uint8_t* d_in; ptrdiff_t in_stride, in_sstride;
    uint8_t* d_out; ptrdiff_t out_stride, out_sstride;
    int w, h, channels;
};

struct ModeState {
    virtual ~ModeState() {}
};

struct Ctx {
    int device = 0;
    int nstreams = 1;
    hipStream_t own_stream = nullptr;
    std::string err;
    // StructuralTracker (reference: MagnifyCore.hpp:45-80)
    int t_mode = LVM_MODE_NONE, t_levels = -1, t_channels = -1, t_w = 0, t_h = 0;
    uint64_t t_pre = 0;
    // constant tables
    float* d_gamma_u8 = nullptr;
    float* d_invgamma = nullptr;
    uint2* d_u8steps = nullptr;
    // spatial tiling demonstrator (lvm_tile_riesz_*): 1 = stripe context, the frame stops after normalize + amplify; 2 = coarse context, the "frame" is a
    // float plane (tile_plane_in) and the collapse of level 0 goes to tile_plane_out instead of through Lab2BGR
    int tile_mode = 0; const float* tile_plane_in = nullptr; float* tile_plane_out = nullptr;
    // the export's text overlay as per-pixel tables (compose.hip, lvm_export_set_overlay): one device block, per label the offsets of its classes / tables
    uint8_t* d_overlay = nullptr; int overlay_n = 0; int ov_x[4] = {}, ov_y[4] = {}, ov_w[4] = {}, ov_h[4] = {}; size_t ov_cls[4] = {}, ov_fn[4] = {};
    unsigned long long* h_probe = nullptr; bool probe_running = false;      // lvm_debug_clock_probe_* (page-locked: cycles, ticks, stop flag)
    LabCoef lab{};
    // per-mode state (allocated for the tracked geometry)
    ModeState* state = nullptr;
    // host-path staging
    uint8_t *d_in = nullptr, *d_out = nullptr;
    size_t stage_cap = 0;
    // instrumentation
    bool keep_float = false;
    float* d_float = nullptr; size_t float_cap = 0; size_t float_count = 0;
    bool profiling = false;
    std::string prof_only; bool prof_skip = false;   // lvm_profile_only: bracket launches of this report name only
    std::vector<ProfEvent> prof_events;
    std::vector<ProfTotal> prof_totals;
    int pipeline_depth = 0;           // 0 = every call completes its own frame; 1 = outputs lag one call (Laplace)
    hipStream_t aux_stream = nullptr; // second stream of the cross-frame pipeline
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<std::pair<hipStream_t, hipEvent_t>> caller_events;   // one event per distinct caller stream, re-recorded after every enqueue (sync_streams)
    bool caller_overflow = false;     // more caller streams than events: sync_streams synchronises the device
    int max_frames = 0;               // lvm_set_max_frames: temporal-batch buffers are sized for this many frames up front
    bool exact_lab = false;   // debug: OpenCV-order float arithmetic everywhere (bit-faithful to the oracle)
    bool lab_analytic = false;   // debug: analytic forward Lab (OpenCV with its interpolation switched off) instead of the 33^3 table
    uint32_t* d_lab_ab = nullptr; uint4* d_lab_Lcells = nullptr;   // the forward table in its device layouts (lab_lut.h)
    LabLut lab_lut{};
    int num_cus = 256;
    std::vector<int16_t> lab_lut_compact;   // the same table, [r][q][p][3] (lvm_get_lab_lut)
    // preprocess stage (preprocess.hip): area tables of the current geometry, staging of the host chain
    void* pre_tables = nullptr;
    uint8_t *d_pre_in = nullptr, *d_pre_out = nullptr, *d_chain_out = nullptr; size_t pre_in_cap = 0, pre_out_cap = 0, chain_out_cap = 0;
    uint8_t* d_canvas = nullptr; size_t canvas_cap = 0;     // lvm_export_frames: the composed canvases of a batch
    uint8_t* d_pre_tap = nullptr; size_t pre_tap_cap = 0;   // the colour frames in front of GrayscaleProcessor (runChainOnce's `original`)
    // lvm_export_frames' three-stage pipeline: uploads, kernels and downloads of consecutive sub-batches on their own queues
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    std::vector<hipEvent_t> ev_up, ev_done;
    void* mjpeg = nullptr;            // mjpeg.hip: tables, header and scratch of the Motion-JPEG encoder
    void* mjpeg_dec = nullptr;        // mjpeg_decode.hip: per-frame tables and scratch of the decoder
};

inline int lab_flavour(const Ctx* c) { return c->lab_analytic ? FL_ANALYTIC : (c->exact_lab ? FL_LUT_EXACT : FL_LUT_FAST); }
// kernel instantiation of the context's flavour: LVM_FL_PICK(fl, k_name, other template arguments...)
#define LVM_FL_PICK(fl, K, ...) ((fl) == FL_ANALYTIC ? K<__VA_ARGS__, FL_ANALYTIC> : ((fl) == FL_LUT_EXACT ? K<__VA_ARGS__, FL_LUT_EXACT> : K<__VA_ARGS__, FL_LUT_FAST>))
#define LVM_FL_PICK0(fl, K) ((fl) == FL_ANALYTIC ? K<FL_ANALYTIC> : ((fl) == FL_LUT_EXACT ? K<FL_LUT_EXACT> : K<FL_LUT_FAST>))

void sync_streams(Ctx* c);
void mark_enqueued(Ctx* c, hipStream_t s);
void fail_state(Ctx* c, hipStream_t s);
int ensure_float(Ctx* c, size_t count);
void prof_begin(Ctx* c, const char* name, hipStream_t s);
void prof_end(Ctx* c, hipStream_t s);

#define LVM_HIP_TRY(c, expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            (c)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                 \
            return LVM_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)

// Report name of a per-level launch ("lap_up_l1"); only built while profiling (the macro below evaluates its
// name argument inside the profiling branch).
struct LName {
    char s[40];
    LName(const char* base, int level) { std::snprintf(s, sizeof(s), "%s_l%d", base, level); }
    operator const char*() const { return s; }
};

// Launch with optional event bracketing.  `nm` is the kernel's report name; template kernels
// with several arguments are passed through a function-pointer variable.
#define LVM_LAUNCH(c, nm, kern, grid, block, stream, ...)                                 \
    do {                                                                                  \
        if ((c)->profiling) lvm::prof_begin((c), nm, (stream));                           \
        hipLaunchKernelGGL(kern, grid, block, 0, (stream), __VA_ARGS__);                  \
        if ((c)->profiling) lvm::prof_end((c), (stream));                                 \
    } while (0)



// labconv.hip: u8 BGR frames -> integer Lab planes (exactly one of iL / Lf is non-null)
void lab_lut_planes(Ctx* c, const uint8_t* d_in, long in_stride, long in_sstride, int w, int h, int nframes, uint16_t* iL, float* Lf,
                    uint32_t* iab, hipStream_t s);

// preprocess.hip
void preprocess_geometry(const lvm_preprocess_params& pp, int w, int h, int channels, int* rx, int* ry, int* rw, int* rh,
                         int* ow, int* oh, int* och);
int preprocess_device(Ctx* c, const lvm_preprocess_params& pp, const uint8_t* d_in, int w, int h, int channels, ptrdiff_t in_stride,
                      ptrdiff_t in_sstride, uint8_t* d_out, ptrdiff_t out_stride, ptrdiff_t out_sstride, hipStream_t s,
                      uint8_t* d_tap = nullptr, ptrdiff_t tap_stride = 0, ptrdiff_t tap_sstride = 0);
void preprocess_release(Ctx* c);

// compose.hip
int compose_geometry(int split, int ow, int oh, int pw, int ph, int* pane_w, int* pane_h, int* canvas_w, int* canvas_h);
// mjpeg.hip: Motion-JPEG encode of device-resident BGR frames (cv::VideoWriter::write for ExportFormat::AviMjpg, Exporter.cpp:107-117, :259)
size_t mjpeg_bound(int w, int h);
void mjpeg_release(Ctx* c);
void mjpeg_set_restart(Ctx* c, int mcus);
int mjpeg_begin(Ctx* c, int w, int h, int quality, int max_frames_per_call, size_t total_frames, size_t capacity, hipStream_t s);
int mjpeg_encode_device(Ctx* c, const uint8_t* d_bgr, ptrdiff_t stride, ptrdiff_t fstride, int nframes, int frame0, size_t capacity, hipStream_t s);
void mjpeg_decode_release(Ctx* c);
int mjpeg_decode_begin(Ctx* c, const uint8_t* jpegs, const size_t* offsets, int n, int w, int h, hipStream_t s);
int mjpeg_decode_enqueue(Ctx* c, int f0, int nf, uint8_t* d_bgr, ptrdiff_t stride, ptrdiff_t fstride, hipStream_t s);
int mjpeg_decode_finish(Ctx* c, hipStream_t s);
int mjpeg_decode_device(Ctx* c, const uint8_t* jpegs, const size_t* offsets, int n, int w, int h, uint8_t* d_bgr, ptrdiff_t stride, ptrdiff_t fstride, hipStream_t s);
int mjpeg_drain(Ctx* c, uint8_t* out_host, size_t upto_call);
void mjpeg_abort(Ctx* c);   // error paths: waits for the downloads mjpeg_drain has queued
int mjpeg_finish(Ctx* c, size_t total_frames, uint8_t* out_host, size_t* offsets, hipStream_t s);
int overlay_set(Ctx* c, int n, const lvm_overlay_label* labels);
int overlay_device(Ctx* c, uint8_t* d_canvas, int cw, int chh, ptrdiff_t stride, ptrdiff_t fstride, int n_frames, hipStream_t s);
void overlay_release(Ctx* c);
int compose_device(Ctx* c, int split, const uint8_t* d_orig, int ow, int oh, int och, ptrdiff_t ostride, ptrdiff_t osstride,
                   const uint8_t* d_proc, int pw, int ph, int pch, ptrdiff_t pstride, ptrdiff_t psstride, uint8_t* d_canvas,
                   ptrdiff_t cstride, ptrdiff_t csstride, hipStream_t s);

constexpr int kColorBatchMax = 32;    // frames of one colour-mode temporal batch (spare slots of the window ring)

// mode entry points (laplace.hip / riesz.hip / color.hip).  Return LVM_OK or an error;
// *produced follows the reference's passthrough rules.
int laplace_flush(Ctx* c, hipStream_t s);
int laplace_process_frames(Ctx* c, const lvm_params& p, const FrameIO& io, int nt, hipStream_t s);
bool laplace_can_batch(const Ctx* c);
int riesz_process_frames(Ctx* c, const lvm_params& p, const FrameIO& io, int nt, hipStream_t s);
int riesz_tile_residual(Ctx* c, float* d_dst, int* rw, int* rh, hipStream_t s);
int riesz_tile_finish(Ctx* c, const lvm_params& p, const FrameIO& io, const float* d_residual, hipStream_t s);
bool riesz_can_batch(const Ctx* c, const lvm_params& p);
int color_process_frames(Ctx* c, const lvm_params& p, const FrameIO& io, int nt, hipStream_t s);
bool color_can_batch(const Ctx* c, const lvm_params& p, int nt);
int laplace_process(Ctx* c, const lvm_params& p, int levels, const FrameIO& io, hipStream_t s, int* produced);
int riesz_process(Ctx* c, const lvm_params& p, int levels, const FrameIO& io, hipStream_t s, int* produced);
int color_process(Ctx* c, const lvm_params& p, int levels, const FrameIO& io, hipStream_t s, int* produced);

// host tables (lab_tables.cpp)
void build_lab_tables(float gamma_u8[256], float invgamma[4096], float fwd[9], float inv[9]);
int clock_probe_start(Ctx* c, double max_seconds);            // labconv.hip
int clock_probe_stop(Ctx* c, double* mhz, double* seconds);
int sweep_u8_steps(Ctx* c, unsigned long long first, unsigned long long count, unsigned long long* bad, unsigned long long* first_bad, hipStream_t s);   // labconv.hip
bool build_u8_steps(const float invgamma[4096], uint32_t steps[2 * kU8StepSlices]);   // false: the quantiser is not the step function the kernels assume
void build_lab_lut_compact(std::vector<int16_t>& compact);
void lab_lut_device_tables(const int16_t* compact, std::vector<uint32_t>& ab, std::vector<int16_t>& lcells);
bool lab_lut_fine_index_ok();
int upload_lab_lut(Ctx* c);
void butterworth2(double Wn, double a[3], double b[3]);
int max_levels(int w, int h);
int optimal_buffer_size(int fps);

}  // namespace lvm
