// riesz.hip -- Riesz-pyramid phase magnification on gfx950.
//
// Replaces magcore::magnifyRiesz (reference: processing/magnification/MagnifyCore.hpp:209-279)
// with RieszPyramid / RieszPyramidLevel (RieszPyramid.cpp:8-339) and RieszTemporalFilter
// (TemporalFilter.cpp:299-362).  Only the Lab L plane is processed; a,b are recomputed from the
// u8 frame in the last kernel.
//
// HBM layout per context (planar float32, plane = stream): for every band level l = 0..L-2
//   band_l   current high-pass band                       (RieszPyramidLevel::itsLowpass)
//   P_l, R1p_l, R2p_l   prior frame's band and Riesz pair (the only fields of `old` ever read)
//   ph_l[2]  accumulated quaternionic phase (shared by both Butterworth filters)
//   lo_l[4], hi_l[4]    Direct-Form-II registers of the two order-2 Butterworth low-passes
//   amp_l, tc_l, ts_l   sqrt amplitude and (hi - lo) * amp   (inputs of the 13-tap blurs)
//   bandA_l  amplified band (collapse input)
// plus the octaves oct_l (oct_0 = L plane) and the collapse results res_l.
//
// Launch sequence per frame:
//   k_rz_lab               u8 BGR -> L plane
//   k_rz_split  x (L-1)    9x9 high-pass (band) + 9x9 low-pass at even pixels (next octave), one LDS tile
//   k_rz_phase  x (L-1)    Riesz pair (5-tap H/V), quaternion phase difference vs prior, amplitude,
//                          phase accumulation, both IIR filters, prior <- current
//   k_rz_blur_amp x (L-1)  three separable 13-tap Gaussians (amp, c, s) + phase-shift of the band
//   k_rz_collapse x (L-2)  res_l = lp(zero-inject(res_{l+1})) (polyphase) + hp(bandA_l)
//   k_rz_final             level-0 collapse + Lab2BGR(L', a, b) -> u8
// Summation order of every filter equals the oracle's (row-major non-zero taps, fma chain).
#include <cmath>

#include "lvm_internal.h"

namespace lvm {

// RieszPyramid.cpp:146-167
__device__ const float kLp9[81] = {
    -0.0001f, -0.0007f, -0.0023f, -0.0046f, -0.0057f, -0.0046f, -0.0023f, -0.0007f, -0.0001f,
    -0.0007f, -0.0030f, -0.0047f, -0.0025f, -0.0003f, -0.0025f, -0.0047f, -0.0030f, -0.0007f,
    -0.0023f, -0.0047f, 0.0054f, 0.0272f, 0.0387f, 0.0272f, 0.0054f, -0.0047f, -0.0023f,
    -0.0046f, -0.0025f, 0.0272f, 0.0706f, 0.0910f, 0.0706f, 0.0272f, -0.0025f, -0.0046f,
    -0.0057f, -0.0003f, 0.0387f, 0.0910f, 0.1138f, 0.0910f, 0.0387f, -0.0003f, -0.0057f,
    -0.0046f, -0.0025f, 0.0272f, 0.0706f, 0.0910f, 0.0706f, 0.0272f, -0.0025f, -0.0046f,
    -0.0023f, -0.0047f, 0.0054f, 0.0272f, 0.0387f, 0.0272f, 0.0054f, -0.0047f, -0.0023f,
    -0.0007f, -0.0030f, -0.0047f, -0.0025f, -0.0003f, -0.0025f, -0.0047f, -0.0030f, -0.0007f,
    -0.0001f, -0.0007f, -0.0023f, -0.0046f, -0.0057f, -0.0046f, -0.0023f, -0.0007f, -0.0001f};
__device__ const float kHp9[81] = {
    0.0000f, 0.0003f, 0.0011f, 0.0022f, 0.0027f, 0.0022f, 0.0011f, 0.0003f, 0.0000f,
    0.0003f, 0.0020f, 0.0059f, 0.0103f, 0.0123f, 0.0103f, 0.0059f, 0.0020f, 0.0003f,
    0.0011f, 0.0059f, 0.0151f, 0.0249f, 0.0292f, 0.0249f, 0.0151f, 0.0059f, 0.0011f,
    0.0022f, 0.0103f, 0.0249f, 0.0402f, 0.0469f, 0.0402f, 0.0249f, 0.0103f, 0.0022f,
    0.0027f, 0.0123f, 0.0292f, 0.0469f, -0.9455f, 0.0469f, 0.0292f, 0.0123f, 0.0027f,
    0.0022f, 0.0103f, 0.0249f, 0.0402f, 0.0469f, 0.0402f, 0.0249f, 0.0103f, 0.0022f,
    0.0011f, 0.0059f, 0.0151f, 0.0249f, 0.0292f, 0.0249f, 0.0151f, 0.0059f, 0.0011f,
    0.0003f, 0.0020f, 0.0059f, 0.0103f, 0.0123f, 0.0103f, 0.0059f, 0.0020f, 0.0003f,
    0.0000f, 0.0003f, 0.0011f, 0.0022f, 0.0027f, 0.0022f, 0.0011f, 0.0003f, 0.0000f};

// ---- u8 BGR -> L plane (MagnifyCore.hpp:218-222) ---------------------------------------------
__global__ __launch_bounds__(256) void k_rz_lab(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                                int w, int h, float* __restrict__ Lp, LabCoef lab) {
    __shared__ float s_gam[256];
    load_gamma_u8(s_gam, lab.gamma_u8);
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= w) return;
    const uint8_t* p = in + (size_t)b * in_sstride + (size_t)y * in_stride + (size_t)x * 3;
    float L, a, bb;
    // always the exact cube root: L feeds the ill-conditioned acos(q0/|q|) step (DESIGN.md "Numerics")
    lin_bgr_to_lab<true>(s_gam[p[0]], s_gam[p[1]], s_gam[p[2]], lab.fwd, L, a, bb);
    Lp[((size_t)b * h + y) * w + x] = L;
}

// ---- 9x9 split: band = hp9 * oct, next octave = (2 lp9 * oct) at even pixels ------------------
// RieszPyramid.cpp:215-238 (buildPyramid) + subsample (:254-278).  Tile 32x16, halo 4.
constexpr int ST_W = 32, ST_H = 16, SH = 4;
constexpr int SS_W = ST_W + 2 * SH, SS_H = ST_H + 2 * SH;

__device__ __forceinline__ float conv9(const float (&s)[SS_H][SS_W + 1], int lx, int ly, const float* k, float kscale) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float kv = k[i * 9 + j] * kscale;   // x2 is exact
            if (kv != 0.f) acc = __builtin_fmaf(kv, s[ly + i][lx + j], acc);
        }
    return acc;
}

__global__ __launch_bounds__(256) void k_rz_split(const float* __restrict__ oct, int w, int h,
                                                  float* __restrict__ band, float* __restrict__ next, int nw, int nh) {
    __shared__ float s[SS_H][SS_W + 1];
    const int x0 = blockIdx.x * ST_W, y0 = blockIdx.y * ST_H;
    const float* src = oct + (size_t)blockIdx.z * w * h;
    for (int i = threadIdx.x; i < SS_H * SS_W; i += 256) {
        const int ly = i / SS_W, lx = i - ly * SS_W;
        s[ly][lx] = src[(size_t)reflect101(y0 - SH + ly, h) * w + reflect101(x0 - SH + lx, w)];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ST_H * ST_W; i += 256) {
        const int y = i / ST_W, x = i - y * ST_W;
        const int gx = x0 + x, gy = y0 + y;
        if (gx < w && gy < h) band[((size_t)blockIdx.z * h + gy) * w + gx] = conv9(s, x, y, kHp9, 1.0f);   // :227
    }
    for (int i = threadIdx.x; i < (ST_H / 2) * (ST_W / 2); i += 256) {
        const int y = (i / (ST_W / 2)) * 2, x = (i % (ST_W / 2)) * 2;
        const int gx = x0 + x, gy = y0 + y;
        if (gx < w && gy < h)
            next[((size_t)blockIdx.z * nh + gy / 2) * nw + gx / 2] = conv9(s, x, y, kLp9, 2.0f);            // :232-234
    }
}

// ---- phase difference + amplitude + temporal filters -----------------------------------------
// RieszPyramidLevel::build (:66-78), computePhaseDifferenceAndAmplitude (:81-111),
// RieszTemporalFilter::IIRTemporalFilter (TemporalFilter.cpp:340-351), *old = *cur (:267).
struct PhaseArgs {
    const float* band;                 // current band
    float *P, *R1p, *R2p;              // prior (read), then overwritten with current
    float *phc, *phs;                  // accumulated phase
    float *lo0c, *lo0s, *lo1c, *lo1s;  // low-cutoff filter registers
    float *hi0c, *hi0s, *hi1c, *hi1s;  // high-cutoff filter registers
    float *amp, *tc, *ts;              // outputs
    int w, h;
    double la1, la2, lb0, lb1, lb2, ha1, ha2, hb0, hb1, hb2;
    int mode;                          // 0 = normal, 1 = seed with zero Riesz pair (init), 2 = seed with actual pair
};
constexpr int PT_W = 32, PT_H = 8;

// arcCos (RieszPyramid.cpp:8-23): out-of-range input returns -1 / +1, not pi / 0
__device__ __forceinline__ float arc_cos(float x) {
    if (x < -1.0f) return -1.0f;
    if (x > 1.0f) return 1.0f;
    return acosf(x);
}
__device__ __forceinline__ float mul_sd(float x, double s) { return (float)((double)x * s); }

__global__ __launch_bounds__(256) void k_rz_phase(PhaseArgs a) {
    __shared__ float s[PT_H + 4][PT_W + 4 + 1];
    const int x0 = blockIdx.x * PT_W, y0 = blockIdx.y * PT_H;
    const size_t pl = (size_t)blockIdx.z * a.w * a.h;
    for (int i = threadIdx.x; i < (PT_H + 4) * (PT_W + 4); i += 256) {
        const int ly = i / (PT_W + 4), lx = i - ly * (PT_W + 4);
        s[ly][lx] = a.band[pl + (size_t)reflect101(y0 - 2 + ly, a.h) * a.w + reflect101(x0 - 2 + lx, a.w)];
    }
    __syncthreads();
    const int x = threadIdx.x % PT_W, y = threadIdx.x / PT_W;
    const int gx = x0 + x, gy = y0 + y;
    if (gx >= a.w || gy >= a.h) return;
    const size_t idx = pl + (size_t)gy * a.w + gx;
    const float p = s[y + 2][x + 2];
    // filter2D with [-0.2 -0.48 0 0.48 0.2] (1x5) and its transpose: non-zero taps, fma chain
    float r1 = __builtin_fmaf(-0.2f, s[y + 2][x], 0.f);
    r1 = __builtin_fmaf(-0.48f, s[y + 2][x + 1], r1);
    r1 = __builtin_fmaf(0.48f, s[y + 2][x + 3], r1);
    r1 = __builtin_fmaf(0.2f, s[y + 2][x + 4], r1);
    float r2 = __builtin_fmaf(-0.2f, s[y][x + 2], 0.f);
    r2 = __builtin_fmaf(-0.48f, s[y + 1][x + 2], r2);
    r2 = __builtin_fmaf(0.48f, s[y + 3][x + 2], r2);
    r2 = __builtin_fmaf(0.2f, s[y + 4][x + 2], r2);
    if (a.mode != 0) {   // seed: prior <- current (Riesz pair zeroed by RieszPyramid::init), filters cleared
        a.P[idx] = p;
        a.R1p[idx] = a.mode == 1 ? 0.f : r1;
        a.R2p[idx] = a.mode == 1 ? 0.f : r2;
        a.phc[idx] = 0.f; a.phs[idx] = 0.f;
        a.lo0c[idx] = 0.f; a.lo0s[idx] = 0.f; a.lo1c[idx] = 0.f; a.lo1s[idx] = 0.f;
        a.hi0c[idx] = 0.f; a.hi0s[idx] = 0.f; a.hi1c[idx] = 0.f; a.hi1s[idx] = 0.f;
        return;
    }
    const float Pp = a.P[idx], R1 = a.R1p[idx], R2 = a.R2p[idx];
    const float q0 = (p * Pp + r1 * R1) + r2 * R2;                     // :82-84
    const float np = p * (-1.f);
    const float q1 = R1 * np + r1 * Pp;                                // :86
    const float q2 = R2 * np + r2 * Pp;
    const float xy = q1 * q1 + q2 * q2;                                // :89
    const float ampq = sqrtf(q0 * q0 + xy);                            // :91
    const float phi = arc_cos(q0 / ampq);                              // :93-97
    const float sxy = sqrtf(xy);                                       // :99-100
    float dc = (q1 / sxy) * phi, ds = (q2 / sxy) * phi;                // :102-104
    if (dc != dc) dc = 0.f;                                            // :105-106
    if (ds != ds) ds = 0.f;
    const float am = sqrtf(ampq);                                      // :108
    // IIRTemporalFilter for the low and the high cutoff (TemporalFilter.cpp:343-350); both keep
    // their own copy of the accumulated phase in the reference, the copies are always equal.
    const float phc = a.phc[idx] + dc, phs = a.phs[idx] + ds;
    a.phc[idx] = phc; a.phs[idx] = phs;
    const float ylc = mul_sd(phc, a.lb0) + a.lo0c[idx];
    const float yls = mul_sd(phs, a.lb0) + a.lo0s[idx];
    a.lo0c[idx] = (mul_sd(phc, a.lb1) + a.lo1c[idx]) - mul_sd(ylc, a.la1);
    a.lo0s[idx] = (mul_sd(phs, a.lb1) + a.lo1s[idx]) - mul_sd(yls, a.la1);
    a.lo1c[idx] = mul_sd(phc, a.lb2) - mul_sd(ylc, a.la2);
    a.lo1s[idx] = mul_sd(phs, a.lb2) - mul_sd(yls, a.la2);
    const float yhc = mul_sd(phc, a.hb0) + a.hi0c[idx];
    const float yhs = mul_sd(phs, a.hb0) + a.hi0s[idx];
    a.hi0c[idx] = (mul_sd(phc, a.hb1) + a.hi1c[idx]) - mul_sd(yhc, a.ha1);
    a.hi0s[idx] = (mul_sd(phs, a.hb1) + a.hi1s[idx]) - mul_sd(yhs, a.ha1);
    a.hi1c[idx] = mul_sd(phc, a.hb2) - mul_sd(yhc, a.ha2);
    a.hi1s[idx] = mul_sd(phs, a.hb2) - mul_sd(yhs, a.ha2);
    a.amp[idx] = am;
    a.tc[idx] = (yhc - ylc) * am;                                      // RieszPyramid.cpp:118-120
    a.ts[idx] = (yhs - yls) * am;
    a.P[idx] = p; a.R1p[idx] = r1; a.R2p[idx] = r2;                    // MagnifyCore.hpp:267
}

// ---- 3 x separable Gaussian-13 + amplify ------------------------------------------------------
// GaussianBlur(13x13, sigma 3) of amp (RieszPyramid.cpp:110), sepFilter2D of c, s (:121-124),
// then RieszPyramidLevel::amplify (:129-144).  Tile 32x32, halo 6.
constexpr int BT = 32, BH = 6, BS = BT + 2 * BH;
struct BlurArgs {
    const float *amp, *tc, *ts, *band, *R1, *R2;
    float* bandA;
    int w, h;
    float g[13];
    float alpha, thr;
};

__global__ __launch_bounds__(256) void k_rz_blur_amp(BlurArgs a) {
    __shared__ float s[3][BS][BS + 1];
    __shared__ float hr[3][BS][BT + 1];
    const int x0 = blockIdx.x * BT, y0 = blockIdx.y * BT;
    const size_t pl = (size_t)blockIdx.z * a.w * a.h;
    for (int i = threadIdx.x; i < BS * BS; i += 256) {
        const int ly = i / BS, lx = i - ly * BS;
        const size_t si = pl + (size_t)reflect101(y0 - BH + ly, a.h) * a.w + reflect101(x0 - BH + lx, a.w);
        s[0][ly][lx] = a.amp[si]; s[1][ly][lx] = a.tc[si]; s[2][ly][lx] = a.ts[si];
    }
    __syncthreads();
    // RowFilter: acc = k0*S0; acc = fma(kj, Sj, acc), left to right
    for (int i = threadIdx.x; i < 3 * BS * BT; i += 256) {
        const int f = i / (BS * BT), r = i - f * (BS * BT);
        const int ly = r / BT, x = r - ly * BT;
        float acc = a.g[0] * s[f][ly][x];
#pragma unroll
        for (int j = 1; j < 13; ++j) acc = __builtin_fmaf(a.g[j], s[f][ly][x + j], acc);
        hr[f][ly][x] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BT * BT; i += 256) {
        const int y = i / BT, x = i - y * BT;
        const int gx = x0 + x, gy = y0 + y;
        if (gx >= a.w || gy >= a.h) continue;
        float v[3];
#pragma unroll
        for (int f = 0; f < 3; ++f) {   // SymmColumnFilter: centre, then fma(kj, S[+j] + S[-j])
            float acc = a.g[6] * hr[f][y + BH][x];
#pragma unroll
            for (int j = 1; j <= 6; ++j) acc = __builtin_fmaf(a.g[6 + j], hr[f][y + BH + j][x] + hr[f][y + BH - j][x], acc);
            v[f] = acc;
        }
        const size_t idx = pl + (size_t)gy * a.w + gx;
        const float c = v[1] / v[0], sn = v[2] / v[0];                 // :125-126
        const float magV = sqrtf(c * c + sn * sn);                     // :133-134
        float magV2 = magV * a.alpha;                                  // :135
        magV2 = magV2 > a.thr ? a.thr : magV2;                         // :136 THRESH_TRUNC
        const float cp = cosf(magV2), sp = sinf(magV2);                // :138
        float pair = (a.R1[idx] * c + a.R2[idx] * sn) / magV;          // :139-140
        if (pair != pair) pair = 0.f;                                  // :141
        a.bandA[idx] = a.band[idx] * cp - pair * sp;                   // :143
    }
}

// ---- collapse (RieszPyramid.cpp:304-325) ------------------------------------------------------
// res_l = filter2D(zero-injected nearest-upsample of res_{l+1}, 2 lp9) + filter2D(bandA_l, hp9).
// The zero-injected image is non-zero only at even (x,y) (REFLECT_101 keeps parity), so only taps
// with j == x and i == y (mod 2) are visited -- in the same row-major order as the full sum.
__device__ __forceinline__ void collapse_stage(float (&sb)[SS_H][SS_W + 1], float (&su)[SS_H][SS_W + 1],
                                               const float* __restrict__ bandA, const float* __restrict__ resn,
                                               int w, int h, int nw, int nh, int x0, int y0) {
    for (int i = threadIdx.x; i < SS_H * SS_W; i += 256) {
        const int ly = i / SS_W, lx = i - ly * SS_W;
        const int yr = reflect101(y0 - SH + ly, h), xr = reflect101(x0 - SH + lx, w);
        sb[ly][lx] = bandA[(size_t)yr * w + xr];
        float u = 0.f;
        if (((xr | yr) & 1) == 0) {   // injectZerosEven (:280-302) of resize(INTER_NEAREST) (:314)
            const int sx = xr / 2 < nw ? xr / 2 : nw - 1, sy = yr / 2 < nh ? yr / 2 : nh - 1;
            u = resn[(size_t)sy * nw + sx];
        }
        su[ly][lx] = u;
    }
}
__device__ __forceinline__ float collapse_px(const float (&sb)[SS_H][SS_W + 1], const float (&su)[SS_H][SS_W + 1],
                                             int x, int y, int gx, int gy) {
    float lp = 0.f;
    const int i0 = gy & 1, j0 = gx & 1;   // (gy + i - 4) even <=> i == gy (mod 2)
    for (int i = i0; i < 9; i += 2)
        for (int j = j0; j < 9; j += 2) lp = __builtin_fmaf(kLp9[i * 9 + j] * 2.0f, su[y + i][x + j], lp);
    const float hp = conv9(sb, x, y, kHp9, 1.0f);
    return lp + hp;                                                     // :322
}

__global__ __launch_bounds__(256) void k_rz_collapse(const float* __restrict__ bandA, const float* __restrict__ resn,
                                                     float* __restrict__ res, int w, int h, int nw, int nh) {
    __shared__ float sb[SS_H][SS_W + 1], su[SS_H][SS_W + 1];
    const int x0 = blockIdx.x * ST_W, y0 = blockIdx.y * ST_H;
    const size_t pl = (size_t)blockIdx.z * w * h, pn = (size_t)blockIdx.z * nw * nh;
    collapse_stage(sb, su, bandA + pl, resn + pn, w, h, nw, nh, x0, y0);
    __syncthreads();
    for (int i = threadIdx.x; i < ST_H * ST_W; i += 256) {
        const int y = i / ST_W, x = i - y * ST_W;
        const int gx = x0 + x, gy = y0 + y;
        if (gx < w && gy < h) res[pl + (size_t)gy * w + gx] = collapse_px(sb, su, x, y, gx, gy);
    }
}

// level-0 collapse (or plain L plane when there are no bands) + Lab2BGR + u8 (MagnifyCore.hpp:272-277)
template <bool BANDS, bool EXACT>
__global__ __launch_bounds__(256) void k_rz_final(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                                  uint8_t* __restrict__ out, long out_stride, long out_sstride, int w, int h,
                                                  const float* __restrict__ bandA, const float* __restrict__ resn, int nw,
                                                  int nh, LabCoef lab, int tiles_x, int tiles_y, int nstreams,
                                                  float* __restrict__ dbg) {
    __shared__ __attribute__((aligned(16))) float s_igt[4096];
    __shared__ float s_gam[256];
    __shared__ float sb[SS_H][SS_W + 1], su[SS_H][SS_W + 1];
    load_invgamma(s_igt, lab.invgamma);
    load_gamma_u8(s_gam, lab.gamma_u8);
    __syncthreads();
    const int ntiles = tiles_x * tiles_y * nstreams;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (tiles_x * tiles_y);
        const int r = t - b * (tiles_x * tiles_y);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int x0 = tx * ST_W, y0 = ty * ST_H;
        if (BANDS) {
            collapse_stage(sb, su, bandA + (size_t)b * w * h, resn + (size_t)b * nw * nh, w, h, nw, nh, x0, y0);
            __syncthreads();
        }
        const uint8_t* src = in + (size_t)b * in_sstride;
        uint8_t* dst = out + (size_t)b * out_sstride;
        for (int i = threadIdx.x; i < ST_H * ST_W; i += 256) {
            const int y = i / ST_W, x = i - y * ST_W;
            const int gx = x0 + x, gy = y0 + y;
            if (gx >= w || gy >= h) continue;
            const uint8_t* p = src + (size_t)gy * in_stride + (size_t)gx * 3;
            uint8_t* q = dst + (size_t)gy * out_stride + (size_t)gx * 3;
            float L, a, bb;
            // without bands L itself is the output luminance: keep it exact then
            lin_bgr_to_lab<EXACT || !BANDS>(s_gam[p[0]], s_gam[p[1]], s_gam[p[2]], lab.fwd, L, a, bb);
            if (BANDS) L = collapse_px(sb, su, x, y, gx, gy);
            float o0, o1, o2;
            lab_to_bgr<EXACT>(L, a, bb, lab.inv, s_igt, o0, o1, o2);
            if (dbg && b == 0) { float* d = dbg + ((size_t)gy * w + gx) * 3; d[0] = o0; d[1] = o1; d[2] = o2; }
            q[0] = sat_u8(o0 * 255.0f + lab.a255);
            q[1] = sat_u8(o1 * 255.0f + lab.a255);
            q[2] = sat_u8(o2 * 255.0f + lab.a255);
        }
        if (BANDS) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct RieszState : ModeState {
    int levels = 0;
    LevelGeom g[kMaxLevels + 1];
    float* arena = nullptr;
    float* oct[kMaxLevels + 1] = {};
    float* res[kMaxLevels + 1] = {};
    float* f[kMaxLevels + 1][19] = {};   // per band level: band,P,R1p,R2p,phc,phs,lo0c,lo0s,lo1c,lo1s,hi0c,hi0s,hi1c,hi1s,amp,tc,ts,bandA
    bool inited = false;
    double lo_freq = 0, hi_freq = 0, fps = 0;
    double la[3] = {}, lb[3] = {}, ha[3] = {}, hb[3] = {};
    bool steady(const lvm_params& p) const override {
        return inited && lo_freq == p.coLow && hi_freq == p.coHigh && !std::isnan(la[0]) && !std::isnan(ha[0]);
    }
    ~RieszState() override { if (arena) (void)hipFree(arena); }
};
enum { F_BAND, F_P, F_R1, F_R2, F_PHC, F_PHS, F_LO0C, F_LO0S, F_LO1C, F_LO1S, F_HI0C, F_HI0S, F_HI1C, F_HI1S, F_AMP, F_TC, F_TS, F_BANDA, F_COUNT };

static int riesz_alloc(Ctx* c, RieszState* st, int w, int h, int levels) {
    st->levels = levels;
    const int NS = c->nstreams;
    st->g[0] = {w, h, (size_t)w * h};
    for (int l = 1; l < levels; ++l) {
        const int lw = st->g[l - 1].w / 2 + (st->g[l - 1].w % 2), lh = st->g[l - 1].h / 2 + (st->g[l - 1].h % 2);
        st->g[l] = {lw, lh, (size_t)lw * lh};
    }
    auto pad = [](size_t n) { return (n + 63) & ~(size_t)63; };
    size_t total = 0;
    for (int l = 0; l < levels; ++l) total += 2 * pad(st->g[l].n * NS);                 // oct, res
    for (int l = 0; l < levels - 1; ++l) total += (size_t)F_COUNT * pad(st->g[l].n * NS);
    if (hipMalloc((void**)&st->arena, (total ? total : 64) * sizeof(float)) != hipSuccess) {
        st->arena = nullptr; c->err = "riesz: hipMalloc failed"; return LVM_ERR_OOM;
    }
    float* p = st->arena;
    for (int l = 0; l < levels; ++l) { st->oct[l] = p; p += pad(st->g[l].n * NS); st->res[l] = p; p += pad(st->g[l].n * NS); }
    for (int l = 0; l < levels - 1; ++l)
        for (int k = 0; k < F_COUNT; ++k) { st->f[l][k] = p; p += pad(st->g[l].n * NS); }
    return LVM_OK;
}

static void riesz_coeffs(double frq, double fps, double a[3], double b[3]) {   // TemporalFilter.cpp:324-327
    const double Wn = fps == 0.0 ? 0.0 : frq / (fps / 2.0);
    butterworth2(Wn, a, b);
}

int riesz_process(Ctx* c, const lvm_params& p, int levels, const FrameIO& io, hipStream_t s, int* produced) {
    *produced = 0;
    if (io.channels < 3) return LVM_OK;                                          // MagnifyCore.hpp:212
    RieszState* st = static_cast<RieszState*>(c->state);
    if (!st) {
        st = new RieszState();
        c->state = st;
        const int rc = riesz_alloc(c, st, io.w, io.h, levels);
        if (rc != LVM_OK) return rc;
    }
    const int NS = c->nstreams, w = io.w, h = io.h;
    const dim3 blk(256);
    const int nb = levels - 1;   // number of band levels

    // L plane + pyramid of the current frame (needed by every path below)
    {
        const dim3 grid((w + 255) / 256, h, NS);
        LVM_LAUNCH(c, "rz_lab", k_rz_lab, grid, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, w, h, st->oct[0], c->lab);
    }
    for (int l = 0; l < nb; ++l) {
        const LevelGeom &a = st->g[l], &b = st->g[l + 1];
        const dim3 grid((a.w + ST_W - 1) / ST_W, (a.h + ST_H - 1) / ST_H, NS);
        LVM_LAUNCH(c, "rz_split", k_rz_split, grid, blk, s, (const float*)st->oct[l], a.w, a.h, st->f[l][F_BAND], st->oct[l + 1], b.w, b.h);
    }
    auto launch_phase = [&](int mode) {
        for (int l = 0; l < nb; ++l) {
            PhaseArgs a;
            float** f = st->f[l];
            a.band = f[F_BAND]; a.P = f[F_P]; a.R1p = f[F_R1]; a.R2p = f[F_R2]; a.phc = f[F_PHC]; a.phs = f[F_PHS];
            a.lo0c = f[F_LO0C]; a.lo0s = f[F_LO0S]; a.lo1c = f[F_LO1C]; a.lo1s = f[F_LO1S];
            a.hi0c = f[F_HI0C]; a.hi0s = f[F_HI0S]; a.hi1c = f[F_HI1C]; a.hi1s = f[F_HI1S];
            a.amp = f[F_AMP]; a.tc = f[F_TC]; a.ts = f[F_TS];
            a.w = st->g[l].w; a.h = st->g[l].h;
            a.la1 = st->la[1]; a.la2 = st->la[2]; a.lb0 = st->lb[0]; a.lb1 = st->lb[1]; a.lb2 = st->lb[2];
            a.ha1 = st->ha[1]; a.ha2 = st->ha[2]; a.hb0 = st->hb[0]; a.hb1 = st->hb[1]; a.hb2 = st->hb[2];
            a.mode = mode;
            const dim3 grid((a.w + PT_W - 1) / PT_W, (a.h + PT_H - 1) / PT_H, NS);
            LVM_LAUNCH(c, mode ? "rz_seed" : "rz_phase", k_rz_phase, grid, blk, s, a);
        }
    };
    // first frame ever, or degenerate coefficients: init and pass the frame through (:226-240)
    if (!st->inited || std::isnan(st->la[0]) || std::isnan(st->ha[0])) {
        st->lo_freq = p.coLow; st->hi_freq = p.coHigh; st->fps = p.framerate;
        riesz_coeffs(st->lo_freq, st->fps, st->la, st->lb);
        riesz_coeffs(st->hi_freq, st->fps, st->ha, st->hb);
        launch_phase(1);
        st->inited = true;
        LVM_HIP_TRY(c, hipGetLastError());
        return LVM_OK;
    }
    // cutoff changed: new coefficients, both filters cleared, prior rebuilt from this frame (:243-254)
    bool reseed = false;
    if (st->lo_freq != p.coLow) { st->lo_freq = p.coLow; riesz_coeffs(st->lo_freq, st->fps, st->la, st->lb); reseed = true; }
    if (st->hi_freq != p.coHigh) { st->hi_freq = p.coHigh; riesz_coeffs(st->hi_freq, st->fps, st->ha, st->hb); reseed = true; }
    if (reseed) launch_phase(2);
    launch_phase(0);                                                             // :256-267
    // amplify (:269)
    float gk[13];
    {
        double t[13], sum = 0;   // getGaussianKernel(13, 3, CV_32F)
        for (int i = 0; i < 13; ++i) { const double x = i - 6.0; t[i] = std::exp(-0.5 / 9.0 * x * x); sum += t[i]; }
        sum = 1.0 / sum;
        for (int i = 0; i < 13; ++i) gk[i] = (float)(t[i] * sum);
    }
    const double PI_PERCENT = 3.1415926535897932384626433832795 / 100.0;
    for (int l = nb - 1; l >= 0; --l) {
        BlurArgs a;
        float** f = st->f[l];
        a.amp = f[F_AMP]; a.tc = f[F_TC]; a.ts = f[F_TS]; a.band = f[F_BAND]; a.R1 = f[F_R1]; a.R2 = f[F_R2]; a.bandA = f[F_BANDA];
        a.w = st->g[l].w; a.h = st->g[l].h;
        for (int i = 0; i < 13; ++i) a.g[i] = gk[i];
        a.alpha = (float)p.amplification; a.thr = (float)(p.coWavelength * PI_PERCENT);
        const dim3 grid((a.w + BT - 1) / BT, (a.h + BT - 1) / BT, NS);
        LVM_LAUNCH(c, "rz_blur_amp", k_rz_blur_amp, grid, blk, s, a);
    }
    // collapse (:270): res_{L-1} = residual octave
    const float* resn = st->oct[levels - 1];
    for (int l = nb - 1; l >= 1; --l) {
        const LevelGeom &a = st->g[l], &b = st->g[l + 1];
        const dim3 grid((a.w + ST_W - 1) / ST_W, (a.h + ST_H - 1) / ST_H, NS);
        LVM_LAUNCH(c, "rz_collapse", k_rz_collapse, grid, blk, s, (const float*)st->f[l][F_BANDA], resn, st->res[l], a.w, a.h, b.w, b.h);
        resn = st->res[l];
    }
    {
        const int tx = (w + ST_W - 1) / ST_W, ty = (h + ST_H - 1) / ST_H;
        const int ntiles = tx * ty * NS;
        const dim3 grid(ntiles < 2048 ? ntiles : 2048);
        float* dbg = c->keep_float ? c->d_float : nullptr;
        auto kfb = c->exact_lab ? k_rz_final<true, true> : k_rz_final<true, false>;
        auto kfn = c->exact_lab ? k_rz_final<false, true> : k_rz_final<false, false>;
        if (nb >= 1)
            LVM_LAUNCH(c, "rz_final", kfb, grid, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.d_out,
                       (long)io.out_stride, (long)io.out_sstride, w, h, (const float*)st->f[0][F_BANDA], resn, st->g[1].w, st->g[1].h,
                       c->lab, tx, ty, NS, dbg);
        else
            LVM_LAUNCH(c, "rz_final", kfn, grid, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.d_out,
                       (long)io.out_stride, (long)io.out_sstride, w, h, (const float*)nullptr, (const float*)nullptr, 0, 0,
                       c->lab, tx, ty, NS, dbg);
    }
    LVM_HIP_TRY(c, hipGetLastError());
    *produced = 1;
    return LVM_OK;
}

}  // namespace lvm
