// riesz.hip -- Riesz-pyramid phase magnification on gfx950.
//
// Replaces magcore::magnifyRiesz (reference: processing/magnification/MagnifyCore.hpp:209-279)
// with RieszPyramid / RieszPyramidLevel (RieszPyramid.cpp:8-339) and RieszTemporalFilter
// (TemporalFilter.cpp:299-362).  Only the Lab L plane is processed; the frame goes through OpenCV's forward Lab table
// exactly once (labconv.hip): L becomes the float plane oct_0 of the pyramid, (ia, ib) one dword per pixel that the
// last kernel reads back for the inverse conversion (lvm_debug_lab_analytic: the cube-root L plane of k_rz_lab4 instead).
//
// HBM layout per context (planar float32, plane = stream): for every band level l = 0..L-2
//   band_l   current high-pass band                       (RieszPyramidLevel::itsLowpass)
//   P_l, R1p_l, R2p_l   prior frame's band and Riesz pair (the only fields of `old` ever read)
//   ph_l[2]  accumulated quaternionic phase (shared by both Butterworth filters)
//   lo_l[4], hi_l[4]    Direct-Form-II registers of the two order-2 Butterworth low-passes
//   amp_l, tc_l, ts_l   sqrt amplitude and (hi - lo) * amp   (inputs of the 13-tap blurs)
//   bandA_l  amplified band (collapse input)
// plus the octaves oct_l (oct_0 = L plane), the collapse results res_l and the (ia, ib) dword plane of the frame.
//
// Launch sequence (one frame, or a temporal batch of T frames of every stream):
//   k_lab_planes (labconv.hip)       u8 BGR -> forward table -> float L plane + (ia, ib) plane
//   k_rz_split_rows | k_rz_split2 | k_rz_split x (L-1)   9x9 high-pass (band) + 9x9 low-pass at even pixels (next octave):
//                                    wave strips with DPP halo (large levels) | one LDS tile
//   k_rz_phase4 | k_rz_phase (one launch per variant, all band levels)   Riesz pair (5-tap H/V), quaternion phase difference
//                                    vs prior, amplitude, phase accumulation, both IIR filters, prior <- current; in a batch
//                                    the frame loop runs inside the kernel with the 13 state values in registers
//   k_rz_blur_strips | k_rz_blur_amp4 | k_rz_blur_amp   three separable 13-tap Gaussians (amp, c, s) + phase-shift of the
//                                    band (strips: register window + DPP halo, Riesz pair recomputed from the band)
//   k_rz_collapse x (L-2)            res_l = lp(zero-inject(res_{l+1})) (polyphase) + hp(bandA_l)
//   k_rz_final                       level-0 collapse + Lab2BGR(L', a, b) -> u8, (a, b) from the dword plane
// Summation order of every filter equals the oracle's (row-major non-zero taps, fma chain).
#include <cmath>
#include <type_traits>

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
// L of one pixel, ANALYTIC flavour only (lvm_debug_lab_analytic): the Y row of the matrix and cv::cubeRoot's float64
// rational polynomial.  The default flavour takes L from OpenCV's interpolated table (labconv.hip writes the L plane).
template <bool EXACT>
__device__ __forceinline__ float rz_lum(float B, float G, float R, const float* fw) {
    const float Y = B * fw[3] + G * fw[4] + R * fw[5];
    const bool hi = Y > 0.008856f;
    const float FY = hi ? cv_cube_root<EXACT>(Y) : (7.787f * Y + 16.0f / 116.0f);
    return hi ? (116.f * FY - 16.f) : (903.3f * Y);
}
__global__ __launch_bounds__(256) void k_rz_lab(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                                int w, int h, float* __restrict__ Lp, LabCoef lab) {
    __shared__ float s_gam[256];
    load_gamma_u8(s_gam, lab.gamma_u8);
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= w) return;
    const uint8_t* p = in + (size_t)b * in_sstride + (size_t)y * in_stride + (size_t)x * 3;
    Lp[((size_t)b * h + y) * w + x] = rz_lum<true>(s_gam[p[0]], s_gam[p[1]], s_gam[p[2]], lab.fwd);
}
// Vectorised variant (4-pixel groups dword aligned): one 12-byte load and one 16-byte store per lane, and
// a workgroup walks over kLabIters x 1024 pixels so the gamma table is loaded once per 4096 pixels instead
// of once per 256.  Same arithmetic.
constexpr int kLabIters = 4;
struct __attribute__((packed, aligned(4))) RzIn4 { uint32_t a, b, c; };
__global__ __launch_bounds__(256) void k_rz_lab4(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                                 int w, int h, float* __restrict__ Lp, LabCoef lab) {
    __shared__ float s_gam[256];
    load_gamma_u8(s_gam, lab.gamma_u8);
    __syncthreads();
    const int gpr = w >> 2, ngroups = gpr * h, b = blockIdx.y;
#pragma unroll 1
    for (int it = 0; it < kLabIters; ++it) {
        const int g = (blockIdx.x * kLabIters + it) * 256 + threadIdx.x;
        if (g >= ngroups) return;
        const int y = g / gpr, x = (g - y * gpr) * 4;
        const RzIn4 v = *reinterpret_cast<const RzIn4*>(in + (size_t)b * in_sstride + (size_t)y * in_stride + (size_t)x * 3);
        const uint32_t pb[12] = {v.a & 255, (v.a >> 8) & 255, (v.a >> 16) & 255, v.a >> 24, v.b & 255, (v.b >> 8) & 255,
                                 (v.b >> 16) & 255, v.b >> 24, v.c & 255, (v.c >> 8) & 255, (v.c >> 16) & 255, v.c >> 24};
        float L[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) L[k] = rz_lum<true>(s_gam[pb[3 * k]], s_gam[pb[3 * k + 1]], s_gam[pb[3 * k + 2]], lab.fwd);
        *reinterpret_cast<float4*>(Lp + ((size_t)b * h + y) * w + x) = make_float4(L[0], L[1], L[2], L[3]);
    }
}

// ---- 9x9 split: band = hp9 * oct, next octave = (2 lp9 * oct) at even pixels ------------------
// RieszPyramid.cpp:215-238 (buildPyramid) + subsample (:254-278).
constexpr int SH = 4;                      // halo of the 9x9 kernels

// Register-blocked variants: tile 64x16 (+ halo 4), row pitch 72 floats so that the 12 (16) floats a
// thread needs per kernel row are three (four) aligned 128-bit LDS reads.  Each output keeps its own
// accumulator and receives its 81 taps in row-major order (filter2D's order).
constexpr int CW = 64, CH = 16, CSW = CW + 2 * SH, CSH = CH + 2 * SH;
// Row pitch of the staged tiles.  A wave's 128-bit reads come from lanes 0-15 in one tile row and lanes 16-31, 32-47,
// 48-63 two rows further down each; ds_read_b128 serves 16 lanes at a time drawn from two such 16-lane rows
// (MI355X_MICROARCH.md, LDS), so it is conflict-free iff rows two apart start on the same bank: 2 * pitch = 0 (mod 64
// dwords).  72 (= tile + halo) put 55-66 % of the LDS cycles of these kernels into bank conflicts (profiles/).
#ifndef LVM_RZ_PITCH
#define LVM_RZ_PITCH 96
#endif
constexpr int CSP = LVM_RZ_PITCH;
static_assert(CSP >= CSW && CSP % 4 == 0, "pitch");

// 4 adjacent outputs (lx..lx+3, ly), lx % 4 == 0.  Fully unrolled: the 81 coefficients become
// immediates and zero taps vanish (measured faster than a rolled row loop with scalar coefficient
// loads, despite the higher register count).
__device__ __forceinline__ void conv9x4(const float (&s)[CSH][CSP], int lx, int ly, const float* k, float kscale, float (&o)[4]) {
    o[0] = o[1] = o[2] = o[3] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(&s[ly + i][lx]);
        const float4 b = *reinterpret_cast<const float4*>(&s[ly + i][lx + 4]);
        const float4 c = *reinterpret_cast<const float4*>(&s[ly + i][lx + 8]);
        const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float kv = k[i * 9 + j] * kscale;   // x2 is exact
            if (kv != 0.f) {
                o[0] = __builtin_fmaf(kv, v[j], o[0]); o[1] = __builtin_fmaf(kv, v[j + 1], o[1]);
                o[2] = __builtin_fmaf(kv, v[j + 2], o[2]); o[3] = __builtin_fmaf(kv, v[j + 3], o[3]);
            }
        }
    }
}
__device__ __forceinline__ void stage_reflect(float (&s)[CSH][CSP], const float* __restrict__ src, int w, int h, int x0, int y0) {
    for (int i = threadIdx.x; i < CSH * CSW; i += 256) {
        const int ly = i / CSW, lx = i - ly * CSW;
        s[ly][lx] = src[(size_t)reflect101(y0 - SH + ly, h) * w + reflect101(x0 - SH + lx, w)];
    }
}

__global__ __launch_bounds__(256) void k_rz_split(const float* __restrict__ oct, int w, int h,
                                                     float* __restrict__ band, float* __restrict__ next, int nw, int nh) {
    __shared__ __attribute__((aligned(16))) float s[CSH][CSP];
    const Bid3 bq{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int x0 = bq.x * CW, y0 = bq.y * CH;
    stage_reflect(s, oct + (size_t)bq.z * w * h, w, h, x0, y0);
    __syncthreads();
    {   // high-pass band at every pixel: 4 per thread
        const int y = threadIdx.x >> 4, x = (threadIdx.x & 15) * 4;
        const int gx = x0 + x, gy = y0 + y;
        if (gx < w && gy < h) {
            float o[4];
            conv9x4(s, x, y, kHp9, 1.0f, o);                                          // :227
            float* d = band + ((size_t)bq.z * h + gy) * w + gx;
            if ((w & 3) == 0) *reinterpret_cast<float4*>(d) = make_float4(o[0], o[1], o[2], o[3]);   // gx % 4 == 0, planes 256-B aligned
            else {
#pragma unroll
                for (int m = 0; m < 4; ++m) if (gx + m < w) d[m] = o[m];
            }
        }
    }
    {   // 2 x low-pass at even pixels only: 32 x 8 per tile, ONE per thread so that all four waves share the
        // work (the 9 floats of a kernel row start at an even column: five aligned 64-bit LDS reads)
        const int v = threadIdx.x >> 5, q = threadIdx.x & 31;
        const int y = 2 * v, x = 2 * q;
        const int gx = x0 + x, gy = y0 + y;
        if (gx < w && gy < h) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const float2 a = *reinterpret_cast<const float2*>(&s[y + i][x]);
                const float2 b = *reinterpret_cast<const float2*>(&s[y + i][x + 2]);
                const float2 c = *reinterpret_cast<const float2*>(&s[y + i][x + 4]);
                const float2 d = *reinterpret_cast<const float2*>(&s[y + i][x + 6]);
                const float e = s[y + i][x + 8];
                const float t[9] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y, e};
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const float kv = kLp9[i * 9 + j] * 2.0f;                           // x2 is exact
                    if (kv != 0.f) acc = __builtin_fmaf(kv, t[j], acc);               // :232-234, row-major taps
                }
            }
            next[((size_t)bq.z * nh + gy / 2) * nw + gx / 2] = acc;
        }
    }
}

// Variant with twice the tile height (64 x 32) and 4 x 2 outputs per thread: k_rz_split keeps the LDS pipe ~70 %
// busy (SQ counters, profiles/), this one reads 10 instead of 18 kernel rows per 8 high-pass outputs and 11
// instead of 18 per 2 low-pass outputs (110 B of LDS traffic per pixel instead of 195).  Same accumulation
// order per output.  Used for planes whose width is a multiple of 4.
constexpr int C2H = 32, C2SH = C2H + 2 * SH;
__global__ __launch_bounds__(256) void k_rz_split2(const float* __restrict__ oct, int w, int h,
                                                      float* __restrict__ band, float* __restrict__ next, int nw, int nh) {
    __shared__ __attribute__((aligned(16))) float s[C2SH][CSP];
    const Bid3 bq{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int x0 = bq.x * CW, y0 = bq.y * C2H;
    const float* src = oct + (size_t)bq.z * w * h;
    const bool interior = x0 - SH >= 0 && x0 + CW + SH <= w && y0 - SH >= 0 && y0 + C2H + SH <= h;
    if (interior) {
        for (int i = threadIdx.x; i < C2SH * (CSW / 4); i += 256) {
            const int ly = i / (CSW / 4), g = i - ly * (CSW / 4);
            *reinterpret_cast<float4*>(&s[ly][4 * g]) = *reinterpret_cast<const float4*>(src + (size_t)(y0 - SH + ly) * w + (x0 - SH + 4 * g));
        }
    } else {
        for (int i = threadIdx.x; i < C2SH * CSW; i += 256) {
            const int ly = i / CSW, lx = i - ly * CSW;
            s[ly][lx] = src[(size_t)reflect101(y0 - SH + ly, h) * w + reflect101(x0 - SH + lx, w)];
        }
    }
    __syncthreads();
    {   // high-pass band: 4 columns x 2 rows per thread, kernel rows 0..9 of the window feed both output rows
        const int yq = threadIdx.x >> 4, x = (threadIdx.x & 15) * 4, y = 2 * yq;
        const int gx = x0 + x, gy = y0 + y;
        float o0[4] = {0.f, 0.f, 0.f, 0.f}, o1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const float4 a = *reinterpret_cast<const float4*>(&s[y + r][x]);
            const float4 b = *reinterpret_cast<const float4*>(&s[y + r][x + 4]);
            const float4 c = *reinterpret_cast<const float4*>(&s[y + r][x + 8]);
            const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
            if (r < 9) {                           // kernel row r of output row 0
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const float kv = kHp9[r * 9 + j];
                    if (kv != 0.f) {
                        o0[0] = __builtin_fmaf(kv, v[j], o0[0]); o0[1] = __builtin_fmaf(kv, v[j + 1], o0[1]);
                        o0[2] = __builtin_fmaf(kv, v[j + 2], o0[2]); o0[3] = __builtin_fmaf(kv, v[j + 3], o0[3]);
                    }
                }
            }
            if (r >= 1) {                          // kernel row r-1 of output row 1
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const float kv = kHp9[(r - 1) * 9 + j];
                    if (kv != 0.f) {
                        o1[0] = __builtin_fmaf(kv, v[j], o1[0]); o1[1] = __builtin_fmaf(kv, v[j + 1], o1[1]);
                        o1[2] = __builtin_fmaf(kv, v[j + 2], o1[2]); o1[3] = __builtin_fmaf(kv, v[j + 3], o1[3]);
                    }
                }
            }
        }
        if (gx < w && gy < h) {                    // w % 4 == 0: whole groups are inside
            float* d = band + ((size_t)bq.z * h + gy) * w + gx;
            *reinterpret_cast<float4*>(d) = make_float4(o0[0], o0[1], o0[2], o0[3]);                  // RieszPyramid.cpp:227
            if (gy + 1 < h) *reinterpret_cast<float4*>(d + w) = make_float4(o1[0], o1[1], o1[2], o1[3]);
        }
    }
    {   // 2 x low-pass at even pixels: 32 x 16 per tile, two vertically adjacent ones (rows y, y + 2) per thread
        const int v2 = threadIdx.x >> 5, q = threadIdx.x & 31;
        const int y = 4 * v2, x = 2 * q;
        const int gx = x0 + x, gy = y0 + y;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < 11; ++r) {
            const float2 a = *reinterpret_cast<const float2*>(&s[y + r][x]);
            const float2 b = *reinterpret_cast<const float2*>(&s[y + r][x + 2]);
            const float2 c = *reinterpret_cast<const float2*>(&s[y + r][x + 4]);
            const float2 d = *reinterpret_cast<const float2*>(&s[y + r][x + 6]);
            const float e = s[y + r][x + 8];
            const float tt[9] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y, e};
            if (r < 9) {
#pragma unroll
                for (int j = 0; j < 9; ++j) { const float kv = kLp9[r * 9 + j] * 2.0f; if (kv != 0.f) a0 = __builtin_fmaf(kv, tt[j], a0); }   // :232-234
            }
            if (r >= 2) {
#pragma unroll
                for (int j = 0; j < 9; ++j) { const float kv = kLp9[(r - 2) * 9 + j] * 2.0f; if (kv != 0.f) a1 = __builtin_fmaf(kv, tt[j], a1); }
            }
        }
        if (gx < w && gy < h) {
            float* d = next + ((size_t)bq.z * nh + gy / 2) * nw + gx / 2;
            d[0] = a0;
            if (gy + 2 < h) d[nw] = a1;
        }
    }
}

// ---- the same split as WAVE STRIPS: no LDS, no barrier ------------------------------------------------------------
// The tiled kernels above keep the LDS pipe 50-70 % busy (every fma operand is an LDS read) at 30-45 % VALU and two to
// four waves per SIMD: neither pipe saturates.  Here a wave owns a strip of 248 columns x `rows` output rows of one
// plane and walks down the INPUT rows: lane i holds the four columns 4g .. 4g+3 (g = 62 tx - 1 + i) of the current
// input row -- one 16-byte load -- and gets the four columns left and right of them from its neighbours (eight DPP
// wave shifts; lanes 0 and 63 only feed).  An input row y contributes kernel row i to output row y + 4 - i, so nine
// partial output rows live in registers (9 x 4 accumulators for the high-pass, 9 x 2 for the low-pass at even
// pixels); the row that just received kernel row 8 is complete, is stored and its slot cleared for the row nine
// further down.  Every output still receives its 81 taps in row-major order (kernel rows arrive top to bottom, taps
// left to right within a row): the same fma chain as conv9x4, bit for bit.  The loop is unrolled over 18 steps (nine slot
// phases x the two row parities) so that every accumulator index and the rows that carry a low-pass output are compile-time
// constants; no control flow between the steps of a group of nine.
// REFLECT_101: rows by reflecting the row index of the load; columns at the image edge from the lane's own and its
// inner neighbour's values (columns -4..-1 are columns 4, 3, 2, 1; columns w..w+3 are w-2, w-3, w-4, w-5).
constexpr int SR_THREADS = 256, SR_OWN = 62;
// The high-pass accumulators of a slot are two register pairs -- outputs (0, 1) and (2, 3) -- and a tap is ONE v_pk_fma_f32 per
// pair: tap j of outputs (m, m + 1) multiplies (v[j + m], v[j + m + 1]), which is the pair V[(j + m) / 2] of the row for an even
// j + m and W[(j + m - 1) / 2] -- the same row shifted by one column -- for an odd one (VGPR pairs are even-aligned on gfx950, so
// the shifted copy costs five register pairs; 154 packed + 81 scalar instead of 389 scalar fmas per lane and input row).
// Each half of a packed fma is an IEEE fma: the chain per output is the one of conv9x4, bit for bit.
template <int PHASE>
__device__ __forceinline__ void hp_row_taps(const lvm_f2 (&V)[6], const lvm_f2 (&W)[5], lvm_f2 (&hp)[9][2]) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int slot = (PHASE + 9 - i) % 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float kv = kHp9[i * 9 + j];
            if (kv != 0.f) {
                const lvm_f2 k2 = f2_all(kv);
                hp[slot][0] = f2_fma(k2, (j & 1) ? W[(j - 1) / 2] : V[j / 2], hp[slot][0]);
                hp[slot][1] = f2_fma(k2, (j & 1) ? W[(j + 1) / 2] : V[j / 2 + 1], hp[slot][1]);
            }
        }
    }
}
template <int PHASE, bool even_row>
__device__ __forceinline__ void split_row_taps(const lvm_f2 (&V)[6], const lvm_f2 (&W)[5], lvm_f2 (&hp)[9][2], float (&lp)[9][2]) {
    hp_row_taps<PHASE>(V, W, hp);
    // low-pass only at even output rows: kernel rows of this input row's parity
    auto v = [&](int j) __attribute__((always_inline)) { return V[j >> 1][j & 1]; };
    if (even_row) {
#pragma unroll
        for (int i = 0; i < 9; i += 2) {
            const int slot = (PHASE + 9 - i) % 9;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const float kv = kLp9[i * 9 + j] * 2.0f;
                if (kv != 0.f) { lp[slot][0] = __builtin_fmaf(kv, v(j), lp[slot][0]); lp[slot][1] = __builtin_fmaf(kv, v(j + 2), lp[slot][1]); }
            }
        }
    } else {
#pragma unroll
        for (int i = 1; i < 9; i += 2) {
            const int slot = (PHASE + 9 - i) % 9;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const float kv = kLp9[i * 9 + j] * 2.0f;
                if (kv != 0.f) { lp[slot][0] = __builtin_fmaf(kv, v(j), lp[slot][0]); lp[slot][1] = __builtin_fmaf(kv, v(j + 2), lp[slot][1]); }
            }
        }
    }
}
// columns c-4 .. c+7 of one row from the lane's own four and its neighbours' (first / last: the lane owns the image's
// first / last column group), as the pairs V[k] = (column c-4+2k, c-3+2k) and W[k] = (c-3+2k, c-2+2k)
__device__ __forceinline__ void row12(const float4 v, bool first, bool last, lvm_f2 (&V)[6], lvm_f2 (&W)[5]) {
    float4 L, R;
    L.x = dpp_shr1(v.x); L.y = dpp_shr1(v.y); L.z = dpp_shr1(v.z); L.w = dpp_shr1(v.w);
    R.x = dpp_shl1(v.x); R.y = dpp_shl1(v.y); R.z = dpp_shl1(v.z); R.w = dpp_shl1(v.w);
    const float4 Lf = make_float4(R.x, v.w, v.z, v.y), Rl = make_float4(v.z, v.y, v.x, L.w);
    L.x = sel(first, Lf.x, L.x); L.y = sel(first, Lf.y, L.y); L.z = sel(first, Lf.z, L.z); L.w = sel(first, Lf.w, L.w);
    R.x = sel(last, Rl.x, R.x); R.y = sel(last, Rl.y, R.y); R.z = sel(last, Rl.z, R.z); R.w = sel(last, Rl.w, R.w);
    V[0] = f2_set(L.x, L.y); V[1] = f2_set(L.z, L.w); V[2] = f2_set(v.x, v.y); V[3] = f2_set(v.z, v.w); V[4] = f2_set(R.x, R.y); V[5] = f2_set(R.z, R.w);
    W[0] = f2_set(L.y, L.z); W[1] = f2_set(L.w, v.x); W[2] = f2_set(v.y, v.z); W[3] = f2_set(v.w, R.x); W[4] = f2_set(R.y, R.z);
}
__global__ __launch_bounds__(SR_THREADS, 4) void k_rz_split_rows(const float* __restrict__ oct, int w, int h, float* __restrict__ band,
                                                                  float* __restrict__ next, int nw, int nh, int strips_x, int strips_y,
                                                                  int ntasks, int rows) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int task = blockIdx.x * (SR_THREADS / 64) + wave;
    if (task >= ntasks) return;
    const int z = task / (strips_x * strips_y);
    const int r = task - z * (strips_x * strips_y);
    const int ty = r / strips_x, tx = r - ty * strips_x;
    const int G = w >> 2;                                           // w % 4 == 0, G >= 2
    const int g = tx * SR_OWN - 1 + lane;
    const int gl = g < 0 ? 0 : (g > G - 1 ? G - 1 : g);
    const bool owner = lane >= 1 && lane <= SR_OWN && g >= 0 && g < G;
    const bool first = g == 0, last = g == G - 1;
    // raw buffer addressing: one resource per plane (wave-uniform base), the lane's column group as a 32-bit byte offset, the row as
    // a scalar byte offset.  Stores are UNCONDITIONAL instructions: a lane that owns nothing (and every lane while the rows above
    // the strip complete) stores at an offset outside the resource, which the hardware drops.  (With the stores inside a branch the
    // compiler sinks the whole fma chain of the completed row into that branch and keeps nine rows of operands alive for it.)
    const uint32_t pbytes = (uint32_t)w * (uint32_t)h * 4u, nbytes = (uint32_t)nw * (uint32_t)nh * 4u;
    const BufRsrc rs = buf_rsrc(oct + (size_t)z * w * h, pbytes);
    const BufRsrc rb = buf_rsrc(band + (size_t)z * w * h, pbytes);
    const BufRsrc rn = buf_rsrc(next + (size_t)z * nw * nh, nbytes);
    constexpr uint32_t kDrop = 0x80000000u;                         // + any row offset (< 2^31) stays outside every plane (< 2^31 bytes)
    const uint32_t lsrc = 16u * (uint32_t)gl, lband = owner ? 16u * (uint32_t)gl : kDrop, lnext = owner ? 8u * (uint32_t)gl : kDrop;
    const int y0 = ty * rows;                                       // rows is even: the strip starts on an even row
    const int yend = y0 + rows < h ? y0 + rows : h;
    const int nq = yend - y0 + 8;                                   // input rows y0 - 4 .. yend + 3
    lvm_f2 hp[9][2];
    float lp[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k) { hp[k][0] = hp[k][1] = f2_all(0.f); lp[k][0] = lp[k][1] = 0.f; }
    // REFLECT_101 row index of the NEXT load, advanced by selects: no control flow between the steps (a branch there -- reflect101's
    // loop -- splits the steps into basic blocks, and the compiler then sinks every fma chain down to the block of its store, keeping
    // the operand rows of nine steps alive: 1 KB of scratch per lane in the ISA)
    int ry = reflect101(y0 - 4, h), rdir = reflect101(y0 - 3, h) - ry;    // rdir = +1 / -1 (0: a one-row plane)
    auto advance_row = [&]() __attribute__((always_inline)) {       // ry <- the reflected index of the row after it
        const int t = ry + rdir;
        const bool lo = t < 0, hi = t > h - 1;
        ry = lo ? (h > 1 ? 1 : 0) : (hi ? (h > 1 ? h - 2 : 0) : t);
        rdir = lo ? 1 : (hi ? -1 : rdir);
    };
    float4 nxt = buf_ld_f32x4(rs, lsrc, (uint32_t)ry * (uint32_t)w * 4u);
    int q = 0;
#define LVM_SPLIT_STEP(P18)                                                                                        \
    {                                                                                                               \
        if ((P18) % 9 == 0 && q >= nq) break;                       /* two exits per 18 steps: up to 8 extra rows (loads reflected, stores dropped) */ \
        constexpr int P = (P18) % 9;                                /* q = P18 (mod 18): slot phase and row parity are compile-time */ \
        constexpr bool EVEN = ((P18) & 1) == 0;                     /* input row y0 - 4 + q and output row y0 - 8 + q, y0 even */ \
        const float4 cur = nxt;                                                                                     \
        advance_row();                                              /* row y0 - 3 + q, reflected */                 \
        nxt = buf_ld_f32x4(rs, lsrc, (uint32_t)ry * (uint32_t)w * 4u);                                              \
        lvm_issue_fence();                                          /* the load is ISSUED here, a whole step ahead of its use */ \
        lvm_f2 V[6], W[5];                                                                                          \
        row12(cur, first, last, V, W);                                                                              \
        split_row_taps<P, EVEN>(V, W, hp, lp);                                                                      \
        constexpr int E = (P + 1) % 9;                              /* output row o = y0 - 8 + q is complete */     \
        const bool inside = q >= 8 && q < nq;                       /* (rows above and below the strip: dropped) */ \
        const uint32_t o = inside ? (uint32_t)(y0 - 8 + q) : 0u;                                                    \
        buf_st_f32x4(hp[E][0][0], hp[E][0][1], hp[E][1][0], hp[E][1][1], rb, inside ? lband : kDrop, o * (uint32_t)w * 4u);   /* RieszPyramid.cpp:227 */ \
        if (EVEN) buf_st_f32x2(lp[E][0], lp[E][1], rn, inside ? lnext : kDrop, (o >> 1) * (uint32_t)nw * 4u);   /* :232-234, subsample :254-278 */ \
        hp[E][0] = hp[E][1] = f2_all(0.f); lp[E][0] = lp[E][1] = 0.f;                                               \
        ++q;                                                                                                        \
        _Pragma("unroll") for (int d = 1; d < 9; ++d) {             /* one row at a time: every partial sum exists HERE (lvm_pin) */ \
            lvm_pin(hp[(E + d) % 9][0], hp[(E + d) % 9][1]);                                                        \
            if ((((P18) + d) & 1) == 0) lvm_pin(lp[(E + d) % 9][0], lp[(E + d) % 9][1]);   /* slots of even output rows */ \
        }                                                                                                           \
    }
    for (;;) {
        LVM_SPLIT_STEP(0) LVM_SPLIT_STEP(1) LVM_SPLIT_STEP(2) LVM_SPLIT_STEP(3) LVM_SPLIT_STEP(4) LVM_SPLIT_STEP(5)
        LVM_SPLIT_STEP(6) LVM_SPLIT_STEP(7) LVM_SPLIT_STEP(8) LVM_SPLIT_STEP(9) LVM_SPLIT_STEP(10) LVM_SPLIT_STEP(11)
        LVM_SPLIT_STEP(12) LVM_SPLIT_STEP(13) LVM_SPLIT_STEP(14) LVM_SPLIT_STEP(15) LVM_SPLIT_STEP(16) LVM_SPLIT_STEP(17)
    }
#undef LVM_SPLIT_STEP
}

// ---- phase difference + amplitude + temporal filters -----------------------------------------
// RieszPyramidLevel::build (:66-78), computePhaseDifferenceAndAmplitude (:81-111),
// RieszTemporalFilter::IIRTemporalFilter (TemporalFilter.cpp:340-351), *old = *cur (:267).
constexpr int kMaxBands = 12;
struct PhaseLv {                       // one band level
    const float* band;                 // current band
    float *P, *R1p, *R2p;              // prior (read), then overwritten with current
    float *phc, *phs;                  // accumulated phase
    float *lo0c, *lo0s, *lo1c, *lo1s;  // low-cutoff filter registers
    float *hi0c, *hi0s, *hi1c, *hi1s;  // high-cutoff filter registers
    float *amp, *tc, *ts;              // per-frame outputs
    float *R1c, *R2c;                  // per-frame Riesz pair of the current band (read by the amplify step);
                                       // aliases R1p / R2p in per-frame mode
    int w, h, tx, ty, block0;          // geometry, tiles per stream, first workgroup of this level
    long fs;                           // frame stride (floats) of band / amp / tc / ts / R1c / R2c
};
// All band levels in ONE launch: the levels are independent in this stage, and the small ones are
// pure launch latency on their own.
struct SdF { float hi, lo; };           // a float64 coefficient as the sum of two floats
struct PhaseArgs {
    PhaseLv lv[kMaxBands];
    int nlv;
    double la1, la2, lb0, lb1, lb2, ha1, ha2, hb0, hb1, hb2;
    SdF fla1, fla2, flb0, flb1, flb2, fha1, fha2, fhb0, fhb1, fhb2;      // the same coefficients as float pairs (default flavour)
    int mode;                          // 0 = normal, 1 = seed with zero Riesz pair (init), 2 = seed with actual pair
    int nt;                            // frames handled by this launch, in temporal order
};
constexpr int PT_W = 32, PT_H = 8;

// arcCos (RieszPyramid.cpp:8-23): out-of-range input returns -1 / +1, not pi / 0
__device__ __forceinline__ float arc_cos(float x) {
    if (x < -1.0f) return -1.0f;
    if (x > 1.0f) return 1.0f;
    return acosf(x);
}
__device__ __forceinline__ float mul_sd(float x, double s) { return (float)((double)x * s); }
// The same product for the default flavour without the float64 unit: s = hi + lo (two floats, exact to 2^-48), x * hi is
// exact inside the fma, so fma(x, hi, x * lo) is the float nearest to x * s except in double-rounding ties (~2^-24 of
// the cases, one ulp).  Three conversions / f64 operations become two f32 ones, 20 times per pixel and frame.
__device__ __forceinline__ float mul_sd(float x, SdF s) { return __builtin_fmaf(x, s.hi, x * s.lo); }

// With nt > 1 the workgroup walks over nt consecutive frames: the 13 state values of a pixel (prior
// band + Riesz pair, accumulated phase, 8 filter registers) stay in registers and move through HBM
// once per launch instead of once per frame; the band tile of frame t+1 is prefetched while frame t
// is processed.
template <bool EXACT>
__global__ __launch_bounds__(256) void k_rz_phase(PhaseArgs aa) {
    __shared__ float s[PT_H + 4][PT_W + 4 + 1];
    int lvl = 0;
    const int bid = (int)blockIdx.x;       // (the XCD-aware order of lvm_internal.h was measured here: blur 730 -> 875 us per 32 frames, phase unchanged)
    while (lvl + 1 < aa.nlv && bid >= aa.lv[lvl + 1].block0) ++lvl;
    const PhaseLv& a = aa.lv[lvl];
    const int tq = bid - a.block0;
    const int bs = tq / (a.tx * a.ty), tr = tq - bs * (a.tx * a.ty);
    const int x0 = (tr % a.tx) * PT_W, y0 = (tr / a.tx) * PT_H;
    const size_t pl = (size_t)bs * a.w * a.h;
    constexpr int NSE = ((PT_H + 4) * (PT_W + 4) + 255) / 256;
    size_t soff[NSE]; int sdst[NSE]; float pre[NSE];
#pragma unroll
    for (int k = 0; k < NSE; ++k) {
        const int i = threadIdx.x + k * 256;
        sdst[k] = -1; soff[k] = pl;
        if (i < (PT_H + 4) * (PT_W + 4)) {
            const int ly = i / (PT_W + 4), lx = i - ly * (PT_W + 4);
            soff[k] = pl + (size_t)reflect101(y0 - 2 + ly, a.h) * a.w + reflect101(x0 - 2 + lx, a.w);
            sdst[k] = ly * (PT_W + 4 + 1) + lx;
        }
        pre[k] = a.band[soff[k]];
    }
    const int x = threadIdx.x % PT_W, y = threadIdx.x / PT_W;
    const int gx = x0 + x, gy = y0 + y;
    const bool ok = gx < a.w && gy < a.h;
    const size_t idx = pl + (size_t)(ok ? gy : 0) * a.w + (ok ? gx : 0);
    float Pp = 0.f, R1 = 0.f, R2 = 0.f, phc = 0.f, phs = 0.f;
    float lo0c = 0.f, lo0s = 0.f, lo1c = 0.f, lo1s = 0.f, hi0c = 0.f, hi0s = 0.f, hi1c = 0.f, hi1s = 0.f;
    if (aa.mode == 0 && ok) {
        Pp = a.P[idx]; R1 = a.R1p[idx]; R2 = a.R2p[idx]; phc = a.phc[idx]; phs = a.phs[idx];
        lo0c = a.lo0c[idx]; lo0s = a.lo0s[idx]; lo1c = a.lo1c[idx]; lo1s = a.lo1s[idx];
        hi0c = a.hi0c[idx]; hi0s = a.hi0s[idx]; hi1c = a.hi1c[idx]; hi1s = a.hi1s[idx];
    }
    for (int t = 0; t < aa.nt; ++t) {
        if (t > 0) __syncthreads();
#pragma unroll
        for (int k = 0; k < NSE; ++k) if (sdst[k] >= 0) (&s[0][0])[sdst[k]] = pre[k];
        __syncthreads();
        if (t + 1 < aa.nt) {
#pragma unroll
            for (int k = 0; k < NSE; ++k) pre[k] = a.band[(size_t)(t + 1) * a.fs + soff[k]];
        }
        if (!ok) continue;
        const size_t fidx = (size_t)t * a.fs + idx;
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
        if (aa.mode != 0) {   // seed: prior <- current (Riesz pair zeroed by RieszPyramid::init), filters cleared
            Pp = p; R1 = aa.mode == 1 ? 0.f : r1; R2 = aa.mode == 1 ? 0.f : r2;
            continue;
        }
        const float q0 = (p * Pp + r1 * R1) + r2 * R2;                     // :82-84
        const float np = p * (-1.f);
        const float q1 = R1 * np + r1 * Pp;                                // :86
        const float q2 = R2 * np + r2 * Pp;
        const float xy = q1 * q1 + q2 * q2;                                // :89
        const float ampq = sqrtf(q0 * q0 + xy);                            // :91
        const float phi = arc_cos(q0 / ampq);                              // :93-97
        // The acos argument above keeps IEEE sqrt and division in every flavour (it is the ill-conditioned step);
        // the direction q/|q| and the amplitude are well-conditioned: the default flavour uses v_rsq / v_sqrt there.
        float dc, ds;
        if (EXACT) {
            const float sxy = sqrtf(xy);                                   // :99-100
            dc = (q1 / sxy) * phi; ds = (q2 / sxy) * phi;                  // :102-104
        } else {
            // v_rsq_f32 flushes a denormal argument to 0 (-> +inf) where sqrt + divide stay finite: rescale by 2^64
            // below 2^-100 (xy == 0 still gives inf -> 0 * inf = NaN -> patched to 0 below, like the reference's 0 / 0)
            const bool tiny = xy < 0x1p-100f;
            float rs = __builtin_amdgcn_rsqf(tiny ? xy * 0x1p64f : xy);
            rs = tiny ? rs * 0x1p32f : rs;
            dc = (q1 * rs) * phi; ds = (q2 * rs) * phi;
        }
        if (dc != dc) dc = 0.f;                                            // :105-106
        if (ds != ds) ds = 0.f;
        const float am = EXACT ? sqrtf(ampq) : __builtin_amdgcn_sqrtf(ampq);   // :108
        // IIRTemporalFilter for the low and the high cutoff (TemporalFilter.cpp:343-350); both keep
        // their own copy of the accumulated phase in the reference, the copies are always equal.
        phc = phc + dc; phs = phs + ds;
#define MSD(x, c) (EXACT ? mul_sd((x), aa.c) : mul_sd((x), aa.f##c))
        const float ylc = MSD(phc, lb0) + lo0c;
        const float yls = MSD(phs, lb0) + lo0s;
        lo0c = (MSD(phc, lb1) + lo1c) - MSD(ylc, la1);
        lo0s = (MSD(phs, lb1) + lo1s) - MSD(yls, la1);
        lo1c = MSD(phc, lb2) - MSD(ylc, la2);
        lo1s = MSD(phs, lb2) - MSD(yls, la2);
        const float yhc = MSD(phc, hb0) + hi0c;
        const float yhs = MSD(phs, hb0) + hi0s;
        hi0c = (MSD(phc, hb1) + hi1c) - MSD(yhc, ha1);
        hi0s = (MSD(phs, hb1) + hi1s) - MSD(yhs, ha1);
        hi1c = MSD(phc, hb2) - MSD(yhc, ha2);
        hi1s = MSD(phs, hb2) - MSD(yhs, ha2);
#undef MSD
        a.amp[fidx] = am;
        a.tc[fidx] = (yhc - ylc) * am;                                     // RieszPyramid.cpp:118-120
        a.ts[fidx] = (yhs - yls) * am;
        if (a.R1c != a.R1p) { a.R1c[fidx] = r1; a.R2c[fidx] = r2; }        // batched frames keep their own pair
        Pp = p; R1 = r1; R2 = r2;                                          // MagnifyCore.hpp:267
    }
    if (!ok) return;
    a.P[idx] = Pp; a.R1p[idx] = R1; a.R2p[idx] = R2;
    a.phc[idx] = phc; a.phs[idx] = phs;
    a.lo0c[idx] = lo0c; a.lo0s[idx] = lo0s; a.lo1c[idx] = lo1c; a.lo1s[idx] = lo1s;
    a.hi0c[idx] = hi0c; a.hi0s[idx] = hi0s; a.hi1c[idx] = hi1c; a.hi1s[idx] = hi1s;
}

// The same step with FOUR horizontally adjacent pixels per thread (levels whose width is a multiple of 4): tile 64 x 16, the
// band tile staged and read as 128-bit vectors (halo 2 rows above / below, columns x0 - 4 .. x0 + 67 so that a thread's own
// four values sit on a 16-byte boundary), the 13 state planes and the 5 per-frame outputs moved as one 16-byte access per
// thread and plane instead of four 4-byte ones, and a third less halo traffic than the 32 x 8 tile (1.33 x instead of
// 1.69 x the band).  Same per-pixel arithmetic, same order: bit-identical to k_rz_phase.
constexpr int P4_W = 64, P4_H = 16, P4_SW = P4_W + 8, P4_SH = P4_H + 4;
template <bool EXACT>
__global__ __launch_bounds__(256) void k_rz_phase4(PhaseArgs aa) {
    __shared__ __attribute__((aligned(16))) float s[P4_SH][P4_SW];
    int lvl = 0;
    const int bid = (int)blockIdx.x;       // (the XCD-aware order of lvm_internal.h was measured here: blur 730 -> 875 us per 32 frames, phase unchanged)
    while (lvl + 1 < aa.nlv && bid >= aa.lv[lvl + 1].block0) ++lvl;
    const PhaseLv& a = aa.lv[lvl];
    const int tq = bid - a.block0;
    const int bs = tq / (a.tx * a.ty), tr = tq - bs * (a.tx * a.ty);
    const int x0 = (tr % a.tx) * P4_W, y0 = (tr / a.tx) * P4_H;
    const size_t pl = (size_t)bs * a.w * a.h;
    // staging: P4_SH rows x 18 column groups of 4; group g covers image columns x0 - 4 + 4 g .. + 3.  Groups inside the image
    // are one aligned 16-byte load (row reflected); the groups hanging over the left / right edge are gathered per element.
    constexpr int NG = P4_SH * (P4_SW / 4), NSE = (NG + 255) / 256;
    int sdst[NSE]; size_t sbase[NSE]; int scol[NSE]; bool svec[NSE];
    float4 pre[NSE];
    auto gather = [&](const float* band, int k) __attribute__((always_inline)) {
        if (svec[k]) return *reinterpret_cast<const float4*>(band + sbase[k] + scol[k]);
        float4 v;
        v.x = band[sbase[k] + reflect101(scol[k], a.w)]; v.y = band[sbase[k] + reflect101(scol[k] + 1, a.w)];
        v.z = band[sbase[k] + reflect101(scol[k] + 2, a.w)]; v.w = band[sbase[k] + reflect101(scol[k] + 3, a.w)];
        return v;
    };
#pragma unroll
    for (int k = 0; k < NSE; ++k) {
        const int i = threadIdx.x + k * 256;
        sdst[k] = -1; sbase[k] = pl; scol[k] = 0; svec[k] = true;
        if (i < NG) {
            const int ly = i / (P4_SW / 4), g = i - ly * (P4_SW / 4);
            sbase[k] = pl + (size_t)reflect101(y0 - 2 + ly, a.h) * a.w;
            scol[k] = x0 - 4 + 4 * g;
            svec[k] = scol[k] >= 0 && scol[k] + 3 < a.w;
            sdst[k] = ly * P4_SW + 4 * g;
        }
        pre[k] = gather(a.band, k);
    }
    const int x = 4 * (threadIdx.x & 15), y = threadIdx.x >> 4;
    const int gx = x0 + x, gy = y0 + y;
    const bool ok = gx < a.w && gy < a.h;                      // (w % 4 == 0: the four pixels are inside or outside together)
    const size_t idx = pl + (size_t)(ok ? gy : 0) * a.w + (ok ? gx : 0);
    float Pp[4] = {}, R1[4] = {}, R2[4] = {}, phc[4] = {}, phs[4] = {};
    float lo0c[4] = {}, lo0s[4] = {}, lo1c[4] = {}, lo1s[4] = {}, hi0c[4] = {}, hi0s[4] = {}, hi1c[4] = {}, hi1s[4] = {};
    auto ld4 = [&](const float* p, float (&o)[4]) __attribute__((always_inline)) {
        const float4 v = ld_stream_f32x4(p + idx); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;      // (state: read once per launch)
    };
    auto st4 = [&](float* p, size_t at, const float (&o)[4]) __attribute__((always_inline)) {      // state / per-frame outputs: next read by a later launch
        st_stream_f32x4(p + at, o[0], o[1], o[2], o[3]);
    };
    if (aa.mode == 0 && ok) {
        ld4(a.P, Pp); ld4(a.R1p, R1); ld4(a.R2p, R2); ld4(a.phc, phc); ld4(a.phs, phs);
        ld4(a.lo0c, lo0c); ld4(a.lo0s, lo0s); ld4(a.lo1c, lo1c); ld4(a.lo1s, lo1s);
        ld4(a.hi0c, hi0c); ld4(a.hi0s, hi0s); ld4(a.hi1c, hi1c); ld4(a.hi1s, hi1s);
    }
    for (int t = 0; t < aa.nt; ++t) {
        if (t > 0) __syncthreads();
#pragma unroll
        for (int k = 0; k < NSE; ++k) if (sdst[k] >= 0) *reinterpret_cast<float4*>(&s[0][0] + sdst[k]) = pre[k];
        __syncthreads();
        if (t + 1 < aa.nt) {
#pragma unroll
            for (int k = 0; k < NSE; ++k) pre[k] = gather(a.band + (size_t)(t + 1) * a.fs, k);
        }
        if (!ok) continue;
        const size_t fidx = (size_t)t * a.fs + idx;
        // local column of image column x0 + c is c + 4: the thread's pixels are columns x + 4 .. x + 7 of the tile
        float hrow[12], vcol[5][4];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(&s[y + 2][x + 4 * q]);
            hrow[4 * q] = v.x; hrow[4 * q + 1] = v.y; hrow[4 * q + 2] = v.z; hrow[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            if (r == 2) { vcol[2][0] = hrow[4]; vcol[2][1] = hrow[5]; vcol[2][2] = hrow[6]; vcol[2][3] = hrow[7]; continue; }
            const float4 v = *reinterpret_cast<const float4*>(&s[y + r][x + 4]);
            vcol[r][0] = v.x; vcol[r][1] = v.y; vcol[r][2] = v.z; vcol[r][3] = v.w;
        }
        float amv[4], tcv[4], tsv[4], r1v[4], r2v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float p = hrow[4 + k];
            float r1 = __builtin_fmaf(-0.2f, hrow[2 + k], 0.f);
            r1 = __builtin_fmaf(-0.48f, hrow[3 + k], r1);
            r1 = __builtin_fmaf(0.48f, hrow[5 + k], r1);
            r1 = __builtin_fmaf(0.2f, hrow[6 + k], r1);
            float r2 = __builtin_fmaf(-0.2f, vcol[0][k], 0.f);
            r2 = __builtin_fmaf(-0.48f, vcol[1][k], r2);
            r2 = __builtin_fmaf(0.48f, vcol[3][k], r2);
            r2 = __builtin_fmaf(0.2f, vcol[4][k], r2);
            r1v[k] = r1; r2v[k] = r2;
            if (aa.mode != 0) {
                Pp[k] = p; R1[k] = aa.mode == 1 ? 0.f : r1; R2[k] = aa.mode == 1 ? 0.f : r2;
                continue;
            }
            const float q0 = (p * Pp[k] + r1 * R1[k]) + r2 * R2[k];
            const float np = p * (-1.f);
            const float q1 = R1[k] * np + r1 * Pp[k];
            const float q2 = R2[k] * np + r2 * Pp[k];
            const float xy = q1 * q1 + q2 * q2;
            const float ampq = sqrtf(q0 * q0 + xy);
            const float phi = arc_cos(q0 / ampq);
            float dc, ds;
            if (EXACT) {
                const float sxy = sqrtf(xy);
                dc = (q1 / sxy) * phi; ds = (q2 / sxy) * phi;
            } else {
                const bool tiny = xy < 0x1p-100f;
                float rs = __builtin_amdgcn_rsqf(tiny ? xy * 0x1p64f : xy);
                rs = tiny ? rs * 0x1p32f : rs;
                dc = (q1 * rs) * phi; ds = (q2 * rs) * phi;
            }
            if (dc != dc) dc = 0.f;
            if (ds != ds) ds = 0.f;
            const float am = EXACT ? sqrtf(ampq) : __builtin_amdgcn_sqrtf(ampq);
            phc[k] = phc[k] + dc; phs[k] = phs[k] + ds;
#define MSD(x, c) (EXACT ? mul_sd((x), aa.c) : mul_sd((x), aa.f##c))
            const float ylc = MSD(phc[k], lb0) + lo0c[k];
            const float yls = MSD(phs[k], lb0) + lo0s[k];
            lo0c[k] = (MSD(phc[k], lb1) + lo1c[k]) - MSD(ylc, la1);
            lo0s[k] = (MSD(phs[k], lb1) + lo1s[k]) - MSD(yls, la1);
            lo1c[k] = MSD(phc[k], lb2) - MSD(ylc, la2);
            lo1s[k] = MSD(phs[k], lb2) - MSD(yls, la2);
            const float yhc = MSD(phc[k], hb0) + hi0c[k];
            const float yhs = MSD(phs[k], hb0) + hi0s[k];
            hi0c[k] = (MSD(phc[k], hb1) + hi1c[k]) - MSD(yhc, ha1);
            hi0s[k] = (MSD(phs[k], hb1) + hi1s[k]) - MSD(yhs, ha1);
            hi1c[k] = MSD(phc[k], hb2) - MSD(yhc, ha2);
            hi1s[k] = MSD(phs[k], hb2) - MSD(yhs, ha2);
#undef MSD
            amv[k] = am; tcv[k] = (yhc - ylc) * am; tsv[k] = (yhs - yls) * am;
            Pp[k] = p; R1[k] = r1; R2[k] = r2;
        }
        if (aa.mode != 0) continue;
        st4(a.amp, fidx, amv); st4(a.tc, fidx, tcv); st4(a.ts, fidx, tsv);
        if (a.R1c != a.R1p) { st4(a.R1c, fidx, r1v); st4(a.R2c, fidx, r2v); }
    }
    if (!ok) return;
    st4(a.P, idx, Pp); st4(a.R1p, idx, R1); st4(a.R2p, idx, R2); st4(a.phc, idx, phc); st4(a.phs, idx, phs);
    st4(a.lo0c, idx, lo0c); st4(a.lo0s, idx, lo0s); st4(a.lo1c, idx, lo1c); st4(a.lo1s, idx, lo1s);
    st4(a.hi0c, idx, hi0c); st4(a.hi0s, idx, hi0s); st4(a.hi1c, idx, hi1c); st4(a.hi1s, idx, hi1s);
}

// ---- 3 x separable Gaussian-13 + amplify ------------------------------------------------------
// GaussianBlur(13x13, sigma 3) of amp (RieszPyramid.cpp:110), sepFilter2D of c, s (:121-124),
// then RieszPyramidLevel::amplify (:129-144).  Tile 32x32, halo 6.
constexpr int BT = 32, BTH = 16, BH = 6, BS = BT + 2 * BH, BSH = BTH + 2 * BH;
struct BlurLv { const float *amp, *tc, *ts, *band, *R1, *R2; float* bandA; int w, h, tx, ty, block0; };
struct BlurArgs {                      // all band levels in one launch (independent in this stage)
    BlurLv lv[kMaxBands];
    int nlv;
    float g[13];
    float alpha, thr;
    int exact;                         // 1: libm-order amplify (cosf/sinf, IEEE divisions); 0: hardware sin/cos and reciprocals
};

// RieszPyramidLevel::amplify for one pixel (RieszPyramid.cpp:125-143) given the three blurred values.
// EXACT: the reference's operations (IEEE divisions, sqrtf, cosf/sinf) -- what the oracle computes.
// Otherwise: reciprocals and the hardware sine/cosine (v_sin_f32 / v_cos_f32 take revolutions; the angle is
// in [0, pi], absolute error ~1e-6): the phase-shifted band moves by ~1e-6 of its amplitude, and nothing
// downstream feeds back into the temporal state.  The precise sinf/cosf pair alone is ~200 instructions
// per pixel, more than the three 13-tap Gaussians together.
template <bool EXACT>
__device__ __forceinline__ float rz_amplify(float v0, float v1, float v2, float r1, float r2, float band, float alpha, float thr) {
    if (EXACT) {
        const float c = v1 / v0, sn = v2 / v0;                         // :125-126
        const float magV = sqrtf(c * c + sn * sn);                     // :133-134
        float magV2 = magV * alpha;                                     // :135
        magV2 = magV2 < thr ? magV2 : thr;                               // :136 THRESH_TRUNC = v_min(src, thr): NaN -> thr (minps)
        const float cp = cosf(magV2), sp = sinf(magV2);                // :138
        float pair = (r1 * c + r2 * sn) / magV;                        // :139-140
        if (pair != pair) pair = 0.f;                                  // :141
        return band * cp - pair * sp;                                  // :143
    }
    const float iv = __builtin_amdgcn_rcpf(v0);
    const float c = v1 * iv, sn = v2 * iv;
    const float m2 = c * c + sn * sn;
    const float magV = __builtin_amdgcn_sqrtf(m2);
    float magV2 = magV * alpha;
    magV2 = magV2 < thr ? magV2 : thr;                               // NaN -> thr, like the exact flavour
    const float rev = magV2 * 0.15915494309189535f;
    const float cp = __builtin_amdgcn_cosf(rev), sp = __builtin_amdgcn_sinf(rev);
    float pair = (r1 * c + r2 * sn) * __builtin_amdgcn_rcpf(magV);
    if (pair != pair) pair = 0.f;
    return band * cp - pair * sp;
}

template <bool EXACT>
__global__ __launch_bounds__(256) void k_rz_blur_amp(BlurArgs aa) {
    __shared__ float s[3][BSH][BS + 1];
    __shared__ float hr[3][BSH][BT + 1];
    int lvl = 0;
    const int bid = (int)blockIdx.x;       // (the XCD-aware order of lvm_internal.h was measured here: blur 730 -> 875 us per 32 frames, phase unchanged)
    while (lvl + 1 < aa.nlv && bid >= aa.lv[lvl + 1].block0) ++lvl;
    const BlurLv& a = aa.lv[lvl];
    const int t = bid - a.block0;
    const int bs = t / (a.tx * a.ty), tr = t - bs * (a.tx * a.ty);
    const int x0 = (tr % a.tx) * BT, y0 = (tr / a.tx) * BTH;
    const size_t pl = (size_t)bs * a.w * a.h;
    for (int i = threadIdx.x; i < BSH * BS; i += 256) {
        const int ly = i / BS, lx = i - ly * BS;
        const size_t si = pl + (size_t)reflect101(y0 - BH + ly, a.h) * a.w + reflect101(x0 - BH + lx, a.w);
        s[0][ly][lx] = a.amp[si]; s[1][ly][lx] = a.tc[si]; s[2][ly][lx] = a.ts[si];
    }
    __syncthreads();
    // RowFilter: acc = k0*S0; acc = fma(kj, Sj, acc), left to right
    for (int i = threadIdx.x; i < 3 * BSH * BT; i += 256) {
        const int f = i / (BSH * BT), r = i - f * (BSH * BT);
        const int ly = r / BT, x = r - ly * BT;
        float acc = aa.g[0] * s[f][ly][x];
#pragma unroll
        for (int j = 1; j < 13; ++j) acc = __builtin_fmaf(aa.g[j], s[f][ly][x + j], acc);
        hr[f][ly][x] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BTH * BT; i += 256) {
        const int y = i / BT, x = i - y * BT;
        const int gx = x0 + x, gy = y0 + y;
        if (gx >= a.w || gy >= a.h) continue;
        float v[3];
#pragma unroll
        for (int f = 0; f < 3; ++f) {   // SymmColumnFilter: centre, then fma(kj, S[+j] + S[-j])
            float acc = aa.g[6] * hr[f][y + BH][x];
#pragma unroll
            for (int j = 1; j <= 6; ++j) acc = __builtin_fmaf(aa.g[6 + j], hr[f][y + BH + j][x] + hr[f][y + BH - j][x], acc);
            v[f] = acc;
        }
        const size_t idx = pl + (size_t)gy * a.w + gx;
        a.bandA[idx] = rz_amplify<EXACT>(v[0], v[1], v[2], a.R1[idx], a.R2[idx], a.band[idx], aa.alpha, aa.thr);
    }
}

// Register-blocked variant for levels whose width is a multiple of 4 (the large ones).  Tile 64 x 32,
// staged with an 8-column / 6-row halo so that interior tiles load 128-bit vectors.  The three planes
// (amp, c, s) go through the SAME LDS buffers one after the other (25 KB per workgroup instead of 76 KB:
// six workgroups per CU hide the staging latency), their column-pass results wait in registers for the
// amplify step.  Row pass: a work item is four adjacent outputs of one staged row, their source values
// come from five aligned 128-bit LDS reads (20 bytes per output instead of 52).  Column pass + amplify: a
// thread owns a 4 x 2 pixel block, 14 128-bit reads per plane.  Every output keeps its own accumulator
// and receives its taps in the order of the scalar kernel, so both kernels give identical bits.
#ifndef LVM_BLUR_H
#define LVM_BLUR_H 32              // tile height of k_rz_blur_amp4 (32: 256 threads; 64: 512 threads, 19 % instead of 37 % halo rows)
#endif
constexpr int B2W = 64, B2H = LVM_BLUR_H, B2T = 8 * B2H, B2HX = 8, B2HY = 6, B2SW = B2W + 2 * B2HX, B2SH = B2H + 2 * B2HY;
#ifndef LVM_BLUR_WAVES
#define LVM_BLUR_WAVES 0           // minimum waves per SIMD asked of the register allocator (0: none; 86 VGPRs = 5 waves)
#endif
#if LVM_BLUR_WAVES
#define LVM_BLUR_BOUNDS __launch_bounds__(B2T, LVM_BLUR_WAVES)
#else
#define LVM_BLUR_BOUNDS __launch_bounds__(B2T)
#endif
template <bool EXACT>
__global__ LVM_BLUR_BOUNDS void k_rz_blur_amp4(BlurArgs aa) {
    __shared__ __attribute__((aligned(16))) float s[B2SH][B2SW];
    __shared__ __attribute__((aligned(16))) float hr[B2SH][B2W];
    int lvl = 0;
    const int bid = (int)blockIdx.x;       // (the XCD-aware order of lvm_internal.h was measured here: blur 730 -> 875 us per 32 frames, phase unchanged)
    while (lvl + 1 < aa.nlv && bid >= aa.lv[lvl + 1].block0) ++lvl;
    const BlurLv& a = aa.lv[lvl];
    const int t = bid - a.block0;
    const int bs = t / (a.tx * a.ty), tr = t - bs * (a.tx * a.ty);
    const int x0 = (tr % a.tx) * B2W, y0 = (tr / a.tx) * B2H;
    const size_t pl = (size_t)bs * a.w * a.h;
    const bool interior = x0 - B2HX >= 0 && x0 + B2W + B2HX <= a.w && y0 - B2HY >= 0 && y0 + B2H + B2HY <= a.h;
    const int gq = threadIdx.x & 15, yq = threadIdx.x >> 4;           // column pass: 16 column groups x 16 row pairs
    const int x = 4 * gq, y = 2 * yq;
    float bl[3][2][4];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const float* src = (f == 0 ? a.amp : (f == 1 ? a.tc : a.ts)) + pl;
        if (f > 0) __syncthreads();                                   // the row pass of plane f-1 has finished reading s
        if (interior) {
#pragma unroll 1
            for (int i = threadIdx.x; i < B2SH * (B2SW / 4); i += B2T) {
                const int ly = i / (B2SW / 4), g = i - ly * (B2SW / 4);
                *reinterpret_cast<float4*>(&s[ly][4 * g]) =
                    *reinterpret_cast<const float4*>(src + (size_t)(y0 - B2HY + ly) * a.w + (x0 - B2HX + 4 * g));
            }
        } else {
#pragma unroll 1
            for (int i = threadIdx.x; i < B2SH * B2SW; i += B2T) {
                const int ly = i / B2SW, lx = i - ly * B2SW;
                s[ly][lx] = src[(size_t)reflect101(y0 - B2HY + ly, a.h) * a.w + reflect101(x0 - B2HX + lx, a.w)];
            }
        }
        __syncthreads();                                              // ... and the column pass of plane f-1 reading hr
        // RowFilter: acc = k0*S0; acc = fma(kj, Sj, acc), left to right.  Output column x reads staged columns
        // x + 2 .. x + 14 (the staging halo is 8, the filter radius 6): elements 2 .. 17 of five aligned vectors.
#pragma unroll 1
        for (int i = threadIdx.x; i < B2SH * (B2W / 4); i += B2T) {
            const int ly = i / (B2W / 4), g = i - ly * (B2W / 4);
            float v[20];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const float4 p4 = *reinterpret_cast<const float4*>(&s[ly][4 * g + 4 * q]);
                v[4 * q] = p4.x; v[4 * q + 1] = p4.y; v[4 * q + 2] = p4.z; v[4 * q + 3] = p4.w;
            }
            float o[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float acc = aa.g[0] * v[m + 2];
#pragma unroll
                for (int j = 1; j < 13; ++j) acc = __builtin_fmaf(aa.g[j], v[m + 2 + j], acc);
                o[m] = acc;
            }
            *reinterpret_cast<float4*>(&hr[ly][4 * g]) = make_float4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
        // SymmColumnFilter: centre, then fma(kj, S[+j] + S[-j]); hr rows y .. y+13 serve both output rows.
        // Rows are fetched as the two accumulations reach them (row y+6+j of the first output is row
        // y+7+(j-1) of the second), so only a few are live at a time.
        float4 up0 = *reinterpret_cast<const float4*>(&hr[y + 7][x]);   // centre of output row 1, first "up" row of output row 0
        float4 dn1 = *reinterpret_cast<const float4*>(&hr[y + 6][x]);   // centre of output row 0, first "down" row of output row 1
        float acc0[4], acc1[4];
        {
            const float* c0 = &dn1.x; const float* c1 = &up0.x;
#pragma unroll
            for (int m = 0; m < 4; ++m) { acc0[m] = aa.g[6] * c0[m]; acc1[m] = aa.g[6] * c1[m]; }
        }
#pragma unroll
        for (int j = 1; j <= 6; ++j) {
            const float4 dn0 = *reinterpret_cast<const float4*>(&hr[y + 6 - j][x]);
            const float4 up1 = *reinterpret_cast<const float4*>(&hr[y + 7 + j][x]);
            const float* u0 = &up0.x; const float* d0 = &dn0.x; const float* u1 = &up1.x; const float* d1 = &dn1.x;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc0[m] = __builtin_fmaf(aa.g[6 + j], u0[m] + d0[m], acc0[m]);
                acc1[m] = __builtin_fmaf(aa.g[6 + j], u1[m] + d1[m], acc1[m]);
            }
            up0 = up1; dn1 = dn0;          // row y+7+j is the next "up" row of output 0, row y+6-j the next "down" row of output 1
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) { bl[f][0][m] = acc0[m]; bl[f][1][m] = acc1[m]; }
    }
    const int gx = x0 + x, gy = y0 + y;
    if (gx >= a.w || gy >= a.h) return;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        if (gy + d >= a.h) break;
        const size_t idx = pl + (size_t)(gy + d) * a.w + gx;
        const float4 r1 = *reinterpret_cast<const float4*>(a.R1 + idx), r2 = *reinterpret_cast<const float4*>(a.R2 + idx),
                     bd = *reinterpret_cast<const float4*>(a.band + idx);
        const float* R1 = &r1.x; const float* R2 = &r2.x; const float* Bd = &bd.x;
        float o[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            o[m] = rz_amplify<EXACT>(bl[0][d][m], bl[1][d][m], bl[2][d][m], R1[m], R2[m], Bd[m], aa.alpha, aa.thr);
        *reinterpret_cast<float4*>(a.bandA + idx) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- collapse (RieszPyramid.cpp:304-325) ------------------------------------------------------
// res_l = filter2D(zero-injected nearest-upsample of res_{l+1}, 2 lp9) + filter2D(bandA_l, hp9).
// The zero-injected image is non-zero only at even (x,y) (REFLECT_101 keeps parity), so only taps
// with j == x and i == y (mod 2) are visited -- in the same row-major order as the full sum.
// COMPACT (planes with even width and height: REFLECT_101 keeps the parity of every coordinate there, so the odd rows and
// columns of the zero-injected tile are all zeros): the tile holds only its even rows / columns, i.e. 12 x 36 coarse values
// instead of 24 x 72 with three quarters of them zero -- half the LDS bytes read by the polyphase low-pass (three 64-bit
// reads per kernel row instead of three 128-bit ones), a fifth of the staging stores, 7 KB less LDS per workgroup.
constexpr int CCH = CSH / 2, CCP = 40;                    // compact tile: rows, columns, row pitch (floats)
template <bool COMPACT> struct CollapseTile { float v[COMPACT ? CCH : CSH][COMPACT ? CCP : CSP]; };
template <bool COMPACT>
__device__ __forceinline__ void collapse_stage(float (&sb)[CSH][CSP], CollapseTile<COMPACT>& su,
                                               const float* __restrict__ bandA, const float* __restrict__ resn,
                                               int w, int h, int nw, int nh, int x0, int y0) {
    // Interior tiles of planes whose width is a multiple of 4 need no reflection: the band tile is 432 aligned
    // 128-bit loads, and the zero-injected tile is its 12 even rows, each 36 consecutive coarse values spread as
    // (v, 0, v', 0), plus 12 rows of zeros -- about a tenth of the instructions of the per-element path below,
    // which was ~40 % of this kernel's instruction count.
    const bool interior = (w & 3) == 0 && x0 - SH >= 0 && x0 + CW + SH <= w && y0 - SH >= 0 && y0 + CH + SH <= h;
    if (interior) {
        for (int i = threadIdx.x; i < CSH * (CSW / 4); i += 256) {
            const int ly = i / (CSW / 4), g = i - ly * (CSW / 4);
            *reinterpret_cast<float4*>(&sb[ly][4 * g]) =
                *reinterpret_cast<const float4*>(bandA + (size_t)(y0 - SH + ly) * w + (x0 - SH + 4 * g));
        }
        // tile origin (x0 - 4, y0 - 4) is even-even, so local parity = image parity
        const int cx0 = (x0 - SH) >> 1, cy0 = (y0 - SH) >> 1;
        for (int i = threadIdx.x; i < (CSH / 2) * (CSW / 4); i += 256) {
            const int r = i / (CSW / 4), g = i - r * (CSW / 4);
            const float2 v = *reinterpret_cast<const float2*>(resn + (size_t)(cy0 + r) * nw + (cx0 + 2 * g));   // 8-byte aligned: cx0, nw even
            if (COMPACT) *reinterpret_cast<float2*>(&su.v[r][2 * g]) = v;
            else {
                *reinterpret_cast<float4*>(&su.v[2 * r][4 * g]) = make_float4(v.x, 0.f, v.y, 0.f);
                *reinterpret_cast<float4*>(&su.v[2 * r + 1][4 * g]) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < CSH * CSW; i += 256) {
        const int ly = i / CSW, lx = i - ly * CSW;
        const int yr = reflect101(y0 - SH + ly, h), xr = reflect101(x0 - SH + lx, w);
        sb[ly][lx] = bandA[(size_t)yr * w + xr];
        float u = 0.f;
        const bool even = ((xr | yr) & 1) == 0;
        if (even) {   // injectZerosEven (:280-302) of resize(INTER_NEAREST) (:314)
            const int sx = xr / 2 < nw ? xr / 2 : nw - 1, sy = yr / 2 < nh ? yr / 2 : nh - 1;
            u = resn[(size_t)sy * nw + sx];
        }
        if (!COMPACT) su.v[ly][lx] = u;
        else if (((lx | ly) & 1) == 0) su.v[ly >> 1][lx >> 1] = u;   // (even plane sizes: local parity = image parity, `even` holds)
    }
}
// 4 adjacent outputs (lx..lx+3, ly), lx % 4 == 0; gx0 = image column of lx (even, tiles start at
// multiples of 64), gy = image row.  Polyphase low-pass of the zero-injected image + high-pass.
// polyphase low-pass of the zero-injected tile for the 4 outputs; ODD = parity of the image row (the kernel rows
// i == gy (mod 2) are the only ones that meet non-zero samples: 0,2,4,6,8 or 1,3,5,7)
template <bool ODD, bool COMPACT>
__device__ __forceinline__ void collapse_lp4(const CollapseTile<COMPACT>& su, int lx, int ly, float (&lp)[4]) {
    lp[0] = lp[1] = lp[2] = lp[3] = 0.f;
#pragma unroll
    for (int ii = 0; ii < (ODD ? 4 : 5); ++ii) {
        const int i = ODD ? 2 * ii + 1 : 2 * ii;
        float v[12];
        if (COMPACT) {               // ly + i is even (kernel rows of the output row's parity), lx a multiple of 4
            const float2 a = *reinterpret_cast<const float2*>(&su.v[(ly + i) >> 1][lx >> 1]);
            const float2 b = *reinterpret_cast<const float2*>(&su.v[(ly + i) >> 1][(lx >> 1) + 2]);
            const float2 c = *reinterpret_cast<const float2*>(&su.v[(ly + i) >> 1][(lx >> 1) + 4]);
            v[0] = a.x; v[2] = a.y; v[4] = b.x; v[6] = b.y; v[8] = c.x; v[10] = c.y;
            v[1] = v[3] = v[5] = v[7] = v[9] = v[11] = 0.f;   // never read below
        } else {
            const float4 a = *reinterpret_cast<const float4*>(&su.v[ly + i][lx]);
            const float4 b = *reinterpret_cast<const float4*>(&su.v[ly + i][lx + 4]);
            const float4 c = *reinterpret_cast<const float4*>(&su.v[ly + i][lx + 8]);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {           // output m (column parity m & 1) uses taps j == m (mod 2)
            const float kv = kLp9[i * 9 + j] * 2.0f;
            if ((j & 1) == 0) { lp[0] = __builtin_fmaf(kv, v[j], lp[0]); lp[2] = __builtin_fmaf(kv, v[j + 2], lp[2]); }
            else { lp[1] = __builtin_fmaf(kv, v[j + 1], lp[1]); lp[3] = __builtin_fmaf(kv, v[j + 3], lp[3]); }
        }
    }
}
// The row parity is uniform across a wave (collapse_row() below maps waves 0-1 to the even rows of the tile
// and waves 2-3 to the odd ones), so the parity test is a scalar branch and every tap weight an immediate.
__device__ __forceinline__ int collapse_row() { return ((threadIdx.x >> 4) & 7) * 2 + (threadIdx.x >> 7); }
template <bool COMPACT>
__device__ __forceinline__ void collapse_px4(const float (&sb)[CSH][CSP], const CollapseTile<COMPACT>& su,
                                             int lx, int ly, int gy, float (&o)[4]) {
    float lp[4];
    if (__builtin_amdgcn_readfirstlane(gy & 1)) collapse_lp4<true, COMPACT>(su, lx, ly, lp);
    else collapse_lp4<false, COMPACT>(su, lx, ly, lp);
    float hp[4];
    conv9x4(sb, lx, ly, kHp9, 1.0f, hp);
#pragma unroll
    for (int m = 0; m < 4; ++m) o[m] = lp[m] + hp[m];                                 // :322
}

// (a 64 x 32 / 4 x 2 variant of this kernel, like k_rz_split2, needs 300 registers and measured 48 us against 29 us)
template <bool COMPACT>
__global__ __launch_bounds__(256) void k_rz_collapse(const float* __restrict__ bandA, const float* __restrict__ resn,
                                                     float* __restrict__ res, int w, int h, int nw, int nh) {
    __shared__ __attribute__((aligned(16))) float sb[CSH][CSP];
    __shared__ __attribute__((aligned(16))) CollapseTile<COMPACT> su;
    const Bid3 bq{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};      // (XCD-aware order measured: level 1 73 -> 77 us, the others equal)
    const int x0 = bq.x * CW, y0 = bq.y * CH;
    const size_t pl = (size_t)bq.z * w * h, pn = (size_t)bq.z * nw * nh;
    collapse_stage(sb, su, bandA + pl, resn + pn, w, h, nw, nh, x0, y0);
    __syncthreads();
    const int y = collapse_row(), x = (threadIdx.x & 15) * 4;
    const int gx = x0 + x, gy = y0 + y;
    if (gx < w && gy < h) {
        float o[4];
        collapse_px4(sb, su, x, y, gy, o);
        float* d = res + pl + (size_t)gy * w + gx;
        if ((w & 3) == 0) *reinterpret_cast<float4*>(d) = make_float4(o[0], o[1], o[2], o[3]);
        else {
#pragma unroll
            for (int m = 0; m < 4; ++m) if (gx + m < w) d[m] = o[m];
        }
    }
}

// ---- the same stage as wave strips: no LDS tile, no barrier (round 3) ---------------------------------------------------
// The tiled kernels stage (64 + 16) x (32 + 12) values per 64 x 32 outputs of each of the three blurred planes: measured
// (calibrated FETCH_SIZE, profiles/) the stage moved 1.68 x its compulsory bytes at the HBM / Infinity-Cache ceiling.  Here a
// wave owns a strip of 116 columns x `rows` rows of one plane-frame; lane i holds the two adjacent columns
// c0 = 116 tx - 6 + 2 i, c0 + 1 of the current input row of amp / c / s (REFLECT_101 applied by the loads) and gets the six
// columns left and right of them from its neighbours with whole-wave DPP shifts (lanes 3 .. 60 produce outputs, the three
// lanes at either end only feed them).  The row-filtered values of the last 13 input rows of the three planes live in
// registers (3 x 13 x 2); once 13 rows are in, every new input row completes one output row: column filter, amplify,
// one 8-byte store.  Horizontal re-reads shrink to 128 / 116, vertical ones to (rows + 12) / rows.  Every output receives
// its taps in the order of the tiled kernels (RowFilter left to right, SymmColumnFilter centre then +-j): identical bits.
__device__ __attribute__((noinline)) float rz_amplify_exact_call(float v0, float v1, float v2, float r1, float r2, float band, float alpha, float thr) {
    return rz_amplify<true>(v0, v1, v2, r1, r2, band, alpha, thr);
}
constexpr int BSW = 116, BS_THREADS = 256;
struct BlurStripLv { const float *amp, *tc, *ts, *band, *R1, *R2; float* bandA; int w, h, sx, sy, rows, task0; };
struct BlurStripArgs { BlurStripLv lv[kMaxBands]; int nlv, ntasks; float g[13]; float alpha, thr; };
template <bool EXACT>
__global__ __launch_bounds__(BS_THREADS) void k_rz_blur_strips(BlurStripArgs aa) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // (launch order; the XCD-aware order of lvm_internal.h was measured here too: 550 -> 719-734 us per 32 frames)
    const int task = blockIdx.x * (BS_THREADS / 64) + wave;
    if (task >= aa.ntasks) return;
    int lvl = 0;
    while (lvl + 1 < aa.nlv && task >= aa.lv[lvl + 1].task0) ++lvl;
    const BlurStripLv& a = aa.lv[lvl];
    const int tq = task - a.task0;
    const int bs = tq / (a.sx * a.sy), tr = tq - bs * (a.sx * a.sy);
    const int ty = tr / a.sx, tx = tr - ty * a.sx;
    const int w = a.w, h = a.h;
    const int c0 = tx * BSW - 6 + 2 * lane;
    const unsigned o0 = 4u * (unsigned)reflect101(c0, w), o1 = 4u * (unsigned)reflect101(c0 + 1, w);     // byte offsets inside a row
    const bool owner = lane >= 3 && lane <= 60 && c0 < w;              // (w even, c0 even: c0 + 1 < w too)
    const size_t pl = (size_t)bs * w * h;
    const char* pamp = reinterpret_cast<const char*>(a.amp + pl);
    const char* ptc = reinterpret_cast<const char*>(a.tc + pl);
    const char* pts = reinterpret_cast<const char*>(a.ts + pl);
    const char* pband = reinterpret_cast<const char*>(a.band + pl);
    const int y0 = ty * a.rows, yend = y0 + a.rows < h ? y0 + a.rows : h;
    const int nin = yend - y0 + 12;                                    // input rows y0 - 6 .. yend + 5
    lvm_f2 H[3][13];                                                   // (column c0, column c0 + 1) per slot: every filter step below is ONE packed operation
    // the lane's raw values of one input row (two columns of amp / c / s); fetched one row AHEAD of their use, so that a
    // wave has a row of loads in flight while it filters the previous one
    // ... plus the two columns of band row y0 - 10 + i: the amplify step's Riesz pair (5-tap horizontal / vertical filters of
    // the band, RieszPyramid.cpp:71; what k_rz_phase computes and used to store per frame) is RECOMPUTED here from a 5-row
    // window of the band -- 8 bytes per pixel less to write in the phase kernel and 8 less to read here, both of which run
    // at the bandwidth ceiling.  Same fma chains: identical bits.
    struct Raw6 { float v[8]; };
    auto fetch = [&](int i) __attribute__((always_inline)) {
        const size_t ro = (size_t)reflect101(y0 - 6 + i, h) * w * sizeof(float);
        Raw6 r;
        // (plain loads: the nontemporal form measured 505-510 -> 532 us per 32 frames here, round 5 -- strips re-read their neighbours' halo rows / columns)
        r.v[0] = *reinterpret_cast<const float*>(pamp + ro + o0); r.v[1] = *reinterpret_cast<const float*>(pamp + ro + o1);
        r.v[2] = *reinterpret_cast<const float*>(ptc + ro + o0); r.v[3] = *reinterpret_cast<const float*>(ptc + ro + o1);
        r.v[4] = *reinterpret_cast<const float*>(pts + ro + o0); r.v[5] = *reinterpret_cast<const float*>(pts + ro + o1);
        r.v[6] = r.v[7] = 0.f;
        if (i >= 8) {                                                  // band rows y0 - 2 .. yend + 1 (wave-uniform)
            const size_t rb = (size_t)reflect101(y0 - 10 + i, h) * w * sizeof(float);
            r.v[6] = *reinterpret_cast<const float*>(pband + rb + o0); r.v[7] = *reinterpret_cast<const float*>(pband + rb + o1);
        }
        return r;
    };
    lvm_f2 bw[5] = {};                                                 // band rows y - 2 .. y + 2 of the output row being completed
    // Row-filtered pair of one plane from the lane's two values.  Round 4: the kernel is bound by vector issue (2.9e8 wave instructions
    // per 32 frames, profiles/r04), and a lane's TWO columns are a natural float pair: out = sum_j g[j] * (S[j], S[j + 1]) is 13 packed
    // fma (v_pk_fma_f32, full rate on gfx950) instead of 26 scalar ones.  The pairs with even j are the lanes' own pairs as they arrive
    // over DPP, those with odd j are re-paired neighbours (v_pk_mov_b32).  Each element still receives its taps left to right: same bits.
    auto hpass = [&](const float v0, const float v1) __attribute__((always_inline)) {
        // A[k] = columns (c0 - 6 + 2 k, c0 - 5 + 2 k): the pairs of the lanes 3 - k to the left ... 3 to the right, written by the DPP moves
        // straight into register pairs; B[k] = (A[k].hi, A[k + 1].lo) = the pairs that start at an odd column
        lvm_f2 A[7];
        A[3] = f2_set(v0, v1);
        A[2][0] = dpp_shr1(A[3][0]); A[2][1] = dpp_shr1(A[3][1]); A[1][0] = dpp_shr1(A[2][0]); A[1][1] = dpp_shr1(A[2][1]);
        A[0][0] = dpp_shr1(A[1][0]); A[0][1] = dpp_shr1(A[1][1]);
        A[4][0] = dpp_shl1(A[3][0]); A[4][1] = dpp_shl1(A[3][1]); A[5][0] = dpp_shl1(A[4][0]); A[5][1] = dpp_shl1(A[4][1]);
        A[6][0] = dpp_shl1(A[5][0]); A[6][1] = dpp_shl1(A[5][1]);
        lvm_f2 acc = f2_all(aa.g[0]) * A[0];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            acc = f2_fma(f2_all(aa.g[2 * k + 1]), f2_set(A[k][1], A[k + 1][0]), acc);
            acc = f2_fma(f2_all(aa.g[2 * k + 2]), A[k + 1], acc);
        }
        return acc;
    };
    Raw6 nxt = fetch(0);
    for (int i0 = 0; i0 < nin; i0 += 13) {
#pragma unroll
        for (int ph = 0; ph < 13; ++ph) {
            const int i = i0 + ph;
            if (i >= nin) continue;                                    // (wave-uniform; `continue` keeps the phase loop unrollable)
            const Raw6 cur = nxt;
            if (i + 1 < nin) nxt = fetch(i + 1);
            const int y = y0 + i - 12;                                 // window: row y - 6 + k sits in slot (ph + 1 + k) % 13
            // band window: rows y - 2 .. y + 2 (the row fetched for this step is y + 2)
#pragma unroll
            for (int k = 0; k < 4; ++k) bw[k] = bw[k + 1];
            bw[4] = f2_set(cur.v[6], cur.v[7]);
            H[0][ph] = hpass(cur.v[0], cur.v[1]); H[1][ph] = hpass(cur.v[2], cur.v[3]); H[2][ph] = hpass(cur.v[4], cur.v[5]);
            if (i >= 12) {
                lvm_f2 v[3];
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    lvm_f2 acc = f2_all(aa.g[6]) * H[f][(ph + 7) % 13];
#pragma unroll
                    for (int j = 1; j <= 6; ++j) acc = f2_fma(f2_all(aa.g[6 + j]), H[f][(ph + 7 + j) % 13] + H[f][(ph + 7 - j + 13) % 13], acc);
                    v[f] = acc;
                }
                // Riesz pair of the band at the lane's two columns: [-0.2 -0.48 0 0.48 0.2] along the row (neighbour columns from
                // the adjacent lanes) and along the column (the window); every lane takes part in the DPP exchange
                lvm_f2 bd, q1, q2;
                {
                    const float c0v = bw[2][0], c1v = bw[2][1];
                    const float lm2 = dpp_shr1(c0v), lm1 = dpp_shr1(c1v), rp2 = dpp_shl1(c0v), rp3 = dpp_shl1(c1v);   // columns c0-2, c0-1, c0+2, c0+3
                    bd = bw[2];
                    q1 = f2_fma(f2_all(0.2f), f2_set(rp2, rp3), f2_fma(f2_all(0.48f), f2_set(c1v, rp2), f2_fma(f2_all(-0.48f), f2_set(lm1, c0v), f2_fma(f2_all(-0.2f), f2_set(lm2, lm1), f2_all(0.f)))));
                    q2 = f2_fma(f2_all(0.2f), bw[4], f2_fma(f2_all(0.48f), bw[3], f2_fma(f2_all(-0.48f), bw[1], f2_fma(f2_all(-0.2f), bw[0], f2_all(0.f)))));
                }
                if (owner) {
                    const size_t idx = pl + (size_t)y * w + c0;
                    float2 o;
                    if (EXACT) {      // the libm sine / cosine of the exact flavour is large: out of line, or the 13-phase loop does not unroll
                        o.x = rz_amplify_exact_call(v[0][0], v[1][0], v[2][0], q1[0], q2[0], bd[0], aa.alpha, aa.thr);
                        o.y = rz_amplify_exact_call(v[0][1], v[1][1], v[2][1], q1[1], q2[1], bd[1], aa.alpha, aa.thr);
                    } else {
                        o.x = rz_amplify<false>(v[0][0], v[1][0], v[2][0], q1[0], q2[0], bd[0], aa.alpha, aa.thr);
                        o.y = rz_amplify<false>(v[0][1], v[1][1], v[2][1], q1[1], q2[1], bd[1], aa.alpha, aa.thr);
                    }
                    *reinterpret_cast<float2*>(a.bandA + idx) = o;
                }
            }
        }
    }
}

// level-0 collapse (or plain L plane when there are no bands) + Lab2BGR + u8 (MagnifyCore.hpp:272-277).
// 4 pixels per thread; VEC = the frame's 4-pixel groups are dword aligned (12-byte loads/stores).
struct __attribute__((packed, aligned(4))) RzPx4 { uint32_t a, b, c; };
template <bool BANDS, int FL, bool VEC, bool COMPACT, bool DBG>   // DBG: also store the float frame (compile time: no per-pixel branch otherwise)
__global__ __launch_bounds__(256) void k_rz_final(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                                  uint8_t* __restrict__ out, long out_stride, long out_sstride, int w, int h,
                                                  const float* __restrict__ bandA, const float* __restrict__ resn, int nw,
                                                  int nh, LabCoef lab, int tiles_x, int tiles_y, int nstreams,
                                                  float* __restrict__ dbg, const float* __restrict__ Lplane, const uint32_t* __restrict__ iab) {
    constexpr bool EXACT = fl_exact(FL);
    __shared__ __attribute__((aligned(16))) float s_igt[4096];
    __shared__ float s_gam[fl_lut(FL) ? 1 : 256];
    __shared__ __attribute__((aligned(16))) float sb[CSH][CSP];
    __shared__ __attribute__((aligned(16))) CollapseTile<COMPACT> su;
    load_invgamma(s_igt, lab.invgamma);
    if (!fl_lut(FL)) load_gamma_u8(s_gam, lab.gamma_u8);
    __syncthreads();
    const int ntiles = tiles_x * tiles_y * nstreams;
    const int y = collapse_row(), x = (threadIdx.x & 15) * 4;
    // every workgroup takes a contiguous run of tiles, every XCD a contiguous run of workgroups (xcd_swizzle)
    const int tper = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t0 = (int)xcd_swizzle(blockIdx.x, gridDim.x) * tper, t1 = t0 + tper < ntiles ? t0 + tper : ntiles;
    for (int t = t0; t < t1; ++t) {
        const int b = t / (tiles_x * tiles_y);
        const int r = t - b * (tiles_x * tiles_y);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int x0 = tx * CW, y0 = ty * CH;
        const int gx = x0 + x, gy = y0 + y;
        const bool ok = gx < w && gy < h;
        // phase A: collapse (register-heavy 9x9 taps); its 4 results go through LDS so that the register
        // allocation of phase B (colour math) does not add to it
        if (BANDS) {
            collapse_stage(sb, su, bandA + (size_t)b * w * h, resn + (size_t)b * nw * nh, w, h, nw, nh, x0, y0);
            __syncthreads();
            float Lc[4] = {0.f, 0.f, 0.f, 0.f};
            if (ok) collapse_px4(sb, su, x, y, gy, Lc);
            __syncthreads();                                   // all reads of sb/su done: reuse sb rows 0..15 for the results
            *reinterpret_cast<float4*>(&sb[y][x]) = make_float4(Lc[0], Lc[1], Lc[2], Lc[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ok) {
            // Lab(in): analytic flavour from the u8 frame; LUT flavours from the planes the conversion kernel wrote
            float Lin[4] = {0.f, 0.f, 0.f, 0.f}, ain[4], bin[4];
            if (fl_lut(FL)) {
                const size_t i = ((size_t)b * h + gy) * w + gx;
                uint32_t q[4];
                if (VEC) { const uint4 v = *reinterpret_cast<const uint4*>(iab + i); q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
                else {
#pragma unroll
                    for (int m = 0; m < 4; ++m) q[m] = (gx + m < w) ? iab[i + m] : 0u;
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    ain[m] = lut_ab((int)(q[m] & 0xffffu)); bin[m] = lut_ab((int)(q[m] >> 16));
                    if (!BANDS) Lin[m] = (gx + m < w) ? Lplane[i + m] : 0.f;
                }
            } else {
                const uint8_t* p = in + (size_t)b * in_sstride + (size_t)gy * in_stride + (size_t)gx * 3;
                uint32_t pb[12];
                if (VEC) {
                    const RzPx4 v = *reinterpret_cast<const RzPx4*>(p);
                    pb[0] = v.a & 255; pb[1] = (v.a >> 8) & 255; pb[2] = (v.a >> 16) & 255; pb[3] = v.a >> 24;
                    pb[4] = v.b & 255; pb[5] = (v.b >> 8) & 255; pb[6] = (v.b >> 16) & 255; pb[7] = v.b >> 24;
                    pb[8] = v.c & 255; pb[9] = (v.c >> 8) & 255; pb[10] = (v.c >> 16) & 255; pb[11] = v.c >> 24;
                } else {
#pragma unroll
                    for (int m = 0; m < 12; ++m) pb[m] = (gx + m / 3 < w) ? p[m] : 0;
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) lin_bgr_to_lab<true>(s_gam[pb[3 * m]], s_gam[pb[3 * m + 1]], s_gam[pb[3 * m + 2]], lab.fwd, Lin[m], ain[m], bin[m]);
            }
            float4 Lq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BANDS) Lq = *reinterpret_cast<const float4*>(&sb[y][x]);   // written by this very thread
            const float Lc[4] = {Lq.x, Lq.y, Lq.z, Lq.w};
            uint32_t ob[12];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float L = BANDS ? Lc[m] : Lin[m], a = ain[m], bb = bin[m];
                float o0, o1, o2;
                lab_to_bgr<EXACT>(L, a, bb, EXACT ? lab.inv : lab.inv1024, s_igt, o0, o1, o2);
                if (DBG && dbg && b == 0 && gx + m < w) { float* d = dbg + ((size_t)gy * w + gx + m) * 3; d[0] = o0; d[1] = o1; d[2] = o2; }
                ob[3 * m] = sat_u8(o0 * 255.0f + lab.a255);
                ob[3 * m + 1] = sat_u8(o1 * 255.0f + lab.a255);
                ob[3 * m + 2] = sat_u8(o2 * 255.0f + lab.a255);
            }
            uint8_t* q = out + (size_t)b * out_sstride + (size_t)gy * out_stride + (size_t)gx * 3;
            if (VEC) {
                RzPx4 qo;
                qo.a = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
                qo.b = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
                qo.c = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
                *reinterpret_cast<RzPx4*>(q) = qo;
            } else {
#pragma unroll
                for (int m = 0; m < 12; ++m) if (gx + m / 3 < w) q[m] = (uint8_t)ob[m];
            }
        }
        if (BANDS) __syncthreads();
    }
}

// ---- collapse and output as WAVE STRIPS (planes with even width and height, width a multiple of 4) ---------------------------
// The tiled kernels above read every fma operand from LDS (k_rz_final: 2.1 TB/s, neither pipe saturated).  Here a wave walks down
// the rows of bandA_l exactly as k_rz_split_rows walks down the octave: lane i holds columns 4g .. 4g+3, the 9x9 high-pass is the
// same nine-slot scatter (packed pairs (0, 1), (2, 3)).  The zero-injected image only has samples at even rows and columns, so at
// every EVEN input row the lane also takes the coarse row res_{l+1}[y / 2] -- its own two values 2g, 2g+1 plus two from each
// neighbour -- and scatters kernel row i of 2 lp9 into slot y + 4 - i: output m (column parity m & 1) meets the taps j = m (mod 2),
// i.e. the coarse columns 2g-2 .. 2g+2 (m = 0), 2g-1 .. 2g+2 (1), 2g-1 .. 2g+3 (2), 2g .. 2g+3 (3).  Outputs (0, 2) and (1, 3)
// share their weights and read adjacent coarse values, so the low-pass accumulators are the pairs (0, 2), (1, 3).  Every output
// receives the taps of collapse_px4 in the same order: the visited taps of 2 lp9 row-major, the taps of hp9 row-major, lp + hp.
// REFLECT_101 of the zero-injected image: rows through the reflected row index (even planes: parity is kept); columns -4, -2 are
// the coarse columns 2, 1 and columns w, w + 2 the coarse columns nw - 1, nw - 2.
// FINAL: the completed row of L' goes through Lab2BGR with (a, b) of the frame's integer plane and leaves as u8 (MagnifyCore.hpp:272-277).
constexpr int CS_THREADS = 256;
struct CollapseStripArgs {
    const float* bandA; const float* resn; float* res;          // [z][h][w], [z][nh][nw], [z][h][w] (res: !FINAL)
    int w, h, nw, nh, strips_x, strips_y, ntasks, rows;
    const uint32_t* iab; uint8_t* out; long out_stride, out_sstride; float* dbg; LabCoef lab;   // FINAL
};
// coarse columns 2g-2 .. 2g+3 as the pairs A[k] = (c[2k], c[2k+1]), B[k] = (c[2k+1], c[2k+2])
__device__ __forceinline__ void coarse6(const float2 v, bool first, bool last, lvm_f2 (&A)[3], lvm_f2 (&B)[2]) {
    const float Lx = dpp_shr1(v.x), Ly = dpp_shr1(v.y), Rx = dpp_shl1(v.x), Ry = dpp_shl1(v.y);
    const float c0 = sel(first, Rx, Lx), c1 = sel(first, v.y, Ly), c4 = sel(last, v.y, Rx), c5 = sel(last, v.x, Ry);
    A[0] = f2_set(c0, c1); A[1] = f2_set(v.x, v.y); A[2] = f2_set(c4, c5);
    B[0] = f2_set(c1, v.x); B[1] = f2_set(v.y, c4);
}
template <int PHASE>
__device__ __forceinline__ void collapse_lp_taps(const lvm_f2 (&A)[3], const lvm_f2 (&B)[2], lvm_f2 (&lp)[9][2]) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int slot = (PHASE + 9 - i) % 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const lvm_f2 k2 = f2_all(kLp9[i * 9 + j] * 2.0f);
            const int t = (j & 1) ? (j + 1) / 2 : j / 2;                 // first coarse column of the pair
            const lvm_f2 c = (t & 1) ? B[(t - 1) / 2] : A[t / 2];
            if ((j & 1) == 0) lp[slot][0] = f2_fma(k2, c, lp[slot][0]);   // outputs 0, 2
            else lp[slot][1] = f2_fma(k2, c, lp[slot][1]);                // outputs 1, 3
        }
    }
}
// one completed row of L' (four pixels of a lane) -> Lab2BGR with the frame's (a, b) -> u8 (and the float frame when DBG)
template <int FL, bool DBG>
__device__ __forceinline__ void collapse_emit(float L0, float L1, float L2, float L3, const float4 iabq, const LabCoef& lab, const float* s_igt,
                                              const BufRsrc& ro, uint32_t voff, uint32_t soff, const BufRsrc& rd, uint32_t dvoff, uint32_t dsoff) {
    constexpr bool EXACT = fl_exact(FL);
    const float Lc[4] = {L0, L1, L2, L3};
    const uint32_t q[4] = {__float_as_uint(iabq.x), __float_as_uint(iabq.y), __float_as_uint(iabq.z), __float_as_uint(iabq.w)};
    if (fin_steps(FL, DBG)) {        // s_igt = the u8 step table: linear BGR -> byte by a table hit and a compare (lvm_internal.h u8_step)
        const uint2* steps = reinterpret_cast<const uint2*>(s_igt);
        uint32_t u[4][3];
#pragma unroll
        for (int m = 0; m < 4; m += 2) {
            lvm_f2 c0, c1, c2;
            lab_to_linear_pair(f2_set(Lc[m], Lc[m + 1]), f2_set(lut_ab((int)(q[m] & 0xffffu)), lut_ab((int)(q[m + 1] & 0xffffu))),
                               f2_set(lut_ab((int)(q[m] >> 16)), lut_ab((int)(q[m + 1] >> 16))), lab.inv4096, c0, c1, c2);
#pragma unroll
            for (int k = 0; k < 2; ++k) { u[m + k][0] = u8_step(c0[k], steps); u[m + k][1] = u8_step(c1[k], steps); u[m + k][2] = u8_step(c2[k], steps); }
        }
        B96 v;
        v.a = lvm_pack_b4(u[0][0], u[0][1], u[0][2], u[1][0]);
        v.b = lvm_pack_b4(u[1][1], u[1][2], u[2][0], u[2][1]);
        v.c = lvm_pack_b4(u[2][2], u[3][0], u[3][1], u[3][2]);
        buf_sts_b96(v, ro, voff, soff);
        return;
    }
    float o[4][3];
    if (EXACT) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
            lab_to_bgr<true>(Lc[m], lut_ab((int)(q[m] & 0xffffu)), lut_ab((int)(q[m] >> 16)), lab.inv, s_igt, o[m][0], o[m][1], o[m][2]);
    } else {
#pragma unroll
        for (int m = 0; m < 4; m += 2) {
            lvm_f2 o0, o1, o2;
            lab_to_bgr_pair(f2_set(Lc[m], Lc[m + 1]), f2_set(lut_ab((int)(q[m] & 0xffffu)), lut_ab((int)(q[m + 1] & 0xffffu))),
                            f2_set(lut_ab((int)(q[m] >> 16)), lut_ab((int)(q[m + 1] >> 16))), lab.inv1024, s_igt, o0, o1, o2);
            o[m][0] = o0[0]; o[m][1] = o1[0]; o[m][2] = o2[0]; o[m + 1][0] = o0[1]; o[m + 1][1] = o1[1]; o[m + 1][2] = o2[1];
        }
    }
    if (DBG) {
        buf_st_f32x4(o[0][0], o[0][1], o[0][2], o[1][0], rd, dvoff, dsoff);
        buf_st_f32x4(o[1][1], o[1][2], o[2][0], o[2][1], rd, dvoff, dsoff + 16u);
        buf_st_f32x4(o[2][2], o[3][0], o[3][1], o[3][2], rd, dvoff, dsoff + 32u);
    }
    float t[4][3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < 3; ++k) t[m][k] = o[m][k] * 255.0f + lab.a255;
    B96 v;
    v.a = pack_u8x4(t[0][0], t[0][1], t[0][2], t[1][0]);
    v.b = pack_u8x4(t[1][1], t[1][2], t[2][0], t[2][1]);
    v.c = pack_u8x4(t[2][2], t[3][0], t[3][1], t[3][2]);
    buf_sts_b96(v, ro, voff, soff);                 // (the output frame is not read again on the device: streaming store)
}
template <bool FINAL, int FL, bool DBG>
__global__ __launch_bounds__(CS_THREADS, 3) void k_rz_collapse_strips(CollapseStripArgs a) {
    constexpr bool EXACT = fl_exact(FL);
    __shared__ __attribute__((aligned(16))) float s_igt[FINAL ? (fin_steps(FL, DBG) ? 2 * kU8StepSlices : 4096) : 4];      // spline | u8 step table
    if (FINAL) {
        if (fin_steps(FL, DBG)) load_u8steps(reinterpret_cast<uint2*>(s_igt), a.lab.u8steps); else load_invgamma(s_igt, a.lab.invgamma);
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int task = blockIdx.x * (CS_THREADS / 64) + wave;
    if (task >= a.ntasks) return;
    const int w = a.w, h = a.h, nw = a.nw, nh = a.nh;
    const int z = task / (a.strips_x * a.strips_y);
    const int r = task - z * (a.strips_x * a.strips_y);
    const int ty = r / a.strips_x, tx = r - ty * a.strips_x;
    const int G = w >> 2;                                           // w % 4 == 0, G >= 2
    const int g = tx * SR_OWN - 1 + lane;
    const int gl = g < 0 ? 0 : (g > G - 1 ? G - 1 : g);
    const bool owner = lane >= 1 && lane <= SR_OWN && g >= 0 && g < G;
    const bool first = g == 0, last = g == G - 1;
    constexpr uint32_t kDrop = 0x80000000u;
    const uint32_t pbytes = (uint32_t)w * (uint32_t)h * 4u;
    const BufRsrc rb = buf_rsrc(a.bandA + (size_t)z * w * h, pbytes);
    const BufRsrc rc = buf_rsrc(a.resn + (size_t)z * nw * nh, (uint32_t)nw * (uint32_t)nh * 4u);
    const BufRsrc ro = FINAL ? buf_rsrc(a.out + (size_t)z * a.out_sstride, (uint32_t)h * (uint32_t)a.out_stride)
                             : buf_rsrc(a.res + (size_t)z * w * h, pbytes);
    const BufRsrc ri = buf_rsrc(FINAL ? (const void*)(a.iab + (size_t)z * w * h) : (const void*)a.bandA, FINAL ? pbytes : 0u);
    const BufRsrc rd = buf_rsrc(a.dbg, (DBG && FINAL && a.dbg && z == 0) ? pbytes * 3u : 0u);   // float frame of stream 0
    const uint32_t lband = 16u * (uint32_t)gl, lcoarse = 8u * (uint32_t)gl;
    const uint32_t lout = owner ? (FINAL ? 12u : 16u) * (uint32_t)gl : kDrop, ldbg = owner ? 48u * (uint32_t)gl : kDrop;
    const uint32_t ostride = FINAL ? (uint32_t)a.out_stride : (uint32_t)w * 4u;
    const int y0 = ty * a.rows;                                     // rows is even
    const int yend = y0 + a.rows < h ? y0 + a.rows : h;
    const int nq = yend - y0 + 8;                                   // input rows y0 - 4 .. yend + 3
    lvm_f2 hp[9][2], lp[9][2];
#pragma unroll
    for (int k = 0; k < 9; ++k) { hp[k][0] = hp[k][1] = f2_all(0.f); lp[k][0] = lp[k][1] = f2_all(0.f); }
    int ry = reflect101(y0 - 4, h), rdir = reflect101(y0 - 3, h) - ry;
    auto advance_row = [&]() __attribute__((always_inline)) {
        const int t = ry + rdir;
        const bool lo = t < 0, hi = t > h - 1;
        ry = lo ? (h > 1 ? 1 : 0) : (hi ? (h > 1 ? h - 2 : 0) : t);
        rdir = lo ? 1 : (hi ? -1 : rdir);
    };
    // in flight per wave: the band row of the next step, the coarse row of the next even step, the (a, b) row of the next output row
    float4 nxt = buf_ld_f32x4(rb, lband, (uint32_t)ry * (uint32_t)w * 4u);
    float2 nxtc;
    { const lvm_f2 c = buf_ld_f32x2(rc, lcoarse, (uint32_t)(ry >> 1) * (uint32_t)nw * 4u); nxtc = make_float2(c[0], c[1]); }
    float4 nxti = make_float4(0.f, 0.f, 0.f, 0.f);
    int q = 0;
#define LVM_COLLAPSE_STEP(P18)                                                                                     \
    {                                                                                                               \
        if ((P18) % 9 == 0 && q >= nq) break;                                                                       \
        constexpr int P = (P18) % 9;                                                                                \
        constexpr bool EVEN = ((P18) & 1) == 0;                     /* input row y0 - 4 + q and output row y0 - 8 + q, y0 even */ \
        const float4 cur = nxt;                                                                                     \
        const float2 curc = nxtc;                                                                                   \
        const float4 curi = nxti;                                                                                   \
        advance_row();                                              /* row y0 - 3 + q, reflected */                 \
        nxt = buf_ld_f32x4(rb, lband, (uint32_t)ry * (uint32_t)w * 4u);                                             \
        if (!EVEN) { const lvm_f2 c = buf_ld_f32x2(rc, lcoarse, (uint32_t)(ry >> 1) * (uint32_t)nw * 4u); nxtc = make_float2(c[0], c[1]); }   /* that row is even */ \
        if (FINAL) {                                                /* (a, b) of output row y0 - 7 + q: complete one step from now */ \
            const int orow = y0 - 7 + q;                                                                            \
            nxti = buf_lds_f32x4(ri, lband, (uint32_t)(orow < 0 ? 0 : (orow > h - 1 ? h - 1 : orow)) * (uint32_t)w * 4u);   \
        }                                                                                                           \
        lvm_issue_fence();                                                                                          \
        lvm_f2 V[6], W[5];                                                                                          \
        row12(cur, first, last, V, W);                                                                              \
        hp_row_taps<P>(V, W, hp);                                                                                   \
        if (EVEN) {                                                                                                 \
            lvm_f2 A[3], B[2];                                                                                      \
            coarse6(curc, first, last, A, B);                                                                       \
            collapse_lp_taps<P>(A, B, lp);                                                                          \
        }                                                                                                           \
        constexpr int E = (P + 1) % 9;                              /* output row o = y0 - 8 + q is complete */     \
        const bool inside = q >= 8 && q < nq;                                                                       \
        const uint32_t o = inside ? (uint32_t)(y0 - 8 + q) : 0u;                                                    \
        const float L0 = lp[E][0][0] + hp[E][0][0], L1 = lp[E][1][0] + hp[E][0][1];   /* RieszPyramid.cpp:322 */    \
        const float L2 = lp[E][0][1] + hp[E][1][0], L3 = lp[E][1][1] + hp[E][1][1];                                 \
        if (!FINAL) buf_st_f32x4(L0, L1, L2, L3, ro, inside ? lout : kDrop, o * ostride);                           \
        else collapse_emit<FL, DBG>(L0, L1, L2, L3, curi, a.lab, s_igt, ro, inside ? lout : kDrop, o * ostride, rd, inside ? ldbg : kDrop, o * (uint32_t)w * 12u); \
        hp[E][0] = hp[E][1] = f2_all(0.f); lp[E][0] = lp[E][1] = f2_all(0.f);                                      \
        ++q;                                                                                                        \
        _Pragma("unroll") for (int d = 1; d < 9; ++d) {                                                             \
            lvm_pin(hp[(E + d) % 9][0], hp[(E + d) % 9][1]);                                                        \
            lvm_pin(lp[(E + d) % 9][0], lp[(E + d) % 9][1]);                                                        \
        }                                                                                                           \
    }
    for (;;) {
        LVM_COLLAPSE_STEP(0) LVM_COLLAPSE_STEP(1) LVM_COLLAPSE_STEP(2) LVM_COLLAPSE_STEP(3) LVM_COLLAPSE_STEP(4) LVM_COLLAPSE_STEP(5)
        LVM_COLLAPSE_STEP(6) LVM_COLLAPSE_STEP(7) LVM_COLLAPSE_STEP(8) LVM_COLLAPSE_STEP(9) LVM_COLLAPSE_STEP(10) LVM_COLLAPSE_STEP(11)
        LVM_COLLAPSE_STEP(12) LVM_COLLAPSE_STEP(13) LVM_COLLAPSE_STEP(14) LVM_COLLAPSE_STEP(15) LVM_COLLAPSE_STEP(16) LVM_COLLAPSE_STEP(17)
    }
#undef LVM_COLLAPSE_STEP
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
constexpr int F_ALL_N = 21;
struct RieszState : ModeState {
    int levels = 0;
    LevelGeom g[kMaxLevels + 1];
    float* arena = nullptr;
    float* oct[kMaxLevels + 1] = {};
    float* res[kMaxLevels + 1] = {};
    float* f[kMaxLevels + 1][F_ALL_N] = {};
    // temporal batching: per-frame fields (band, amp, tc, ts, bandA, R1c, R2c), octaves and collapse results of tcap frames
    uint32_t* iab = nullptr; uint32_t* iab_t = nullptr;      // (ia | ib << 16) planes of the input frames (labconv.hip)
    int tcap = 0; float* tarena = nullptr;
    float* ft[kMaxLevels + 1][F_ALL_N] = {}; float* oct_t[kMaxLevels + 1] = {}; float* res_t[kMaxLevels + 1] = {};   // per band level: band,P,R1p,R2p,phc,phs,lo0c,lo0s,lo1c,lo1s,hi0c,hi0s,hi1c,hi1s,amp,tc,ts,bandA
    bool inited = false;
    bool split_rows = true;          // LDS-free wave-strip split (LVM_RZ_SPLIT_ROWS=0: the tiled kernels) ...
    long split_rows_min = 10000000;  // ... for launches of at least this many plane-pixels (LVM_RZ_SPLIT_ROWS_MIN); below, the tiled kernels measure equal or faster
    bool phase4 = true;              // 4-pixels-per-thread phase kernel on levels whose width is a multiple of 4 (LVM_RZ_PHASE4=0: scalar kernel)
    int phase4_min_frames = 2;       // ... from this many frames x streams per launch (LVM_RZ_PHASE4_MIN_FRAMES): ONE frame of one stream has too few 4-pixel threads to
                                     // hide the state loads -- the one-pixel kernel runs it in 58 us against 64 (round 5)
    int fin_groups = 0;              // workgroups of the persistent last kernel (LVM_RZ_FIN_GROUPS; 0 = a sixth of the tiles, at least 2048)
    int split_strip = 0;             // rows per strip of k_rz_split_rows (LVM_RZ_SPLIT_STRIP; 0 = chosen per launch)
    bool collapse_strips = true;     // collapse / output as wave strips (LVM_RZ_COLLAPSE_STRIPS=0: the tiled kernels) ...
    long collapse_strips_min = 10000000;   // ... for launches of at least this many plane-pixels (LVM_RZ_COLLAPSE_STRIPS_MIN)
    int collapse_strip = 0;          // rows per strip of k_rz_collapse_strips (LVM_RZ_COLLAPSE_STRIP; 0 = chosen per launch)
    bool compact = true;             // compact zero-injected tile in the collapse kernels (LVM_RZ_COMPACT=0: the full 24 x 72 tile)
    bool split2 = true;              // 64 x 32 tiles with 4 x 2 outputs per thread in the 9x9 split (LVM_RZ_SPLIT2=0: k_rz_split)
    long split2_min = 600000;        // ... on launches of at least this many plane-pixels (LVM_RZ_SPLIT2_MIN): a per-frame call's levels >= 1 are a few workgroups whose time is
                                     // the length of one thread's instruction stream -- one output per thread: 11 -> 8-9 us per launch (round 5)
    bool blur4 = true;               // register-blocked Gaussian/amplify kernel on the large levels (LVM_RZ_BLUR4=0: scalar kernel everywhere)
    bool blur_strips = true;         // LDS-free wave-strip Gaussian/amplify kernel (LVM_RZ_BLUR_STRIPS=0: the tiled kernels) ...
    long blur_strips_min = 2000;     // ... for levels of at least this many plane-pixels per launch (LVM_RZ_BLUR_STRIPS_MIN).  Round 5: 2^19 -> 2000 -- per-frame calls used to
                                     // send level 1 (518 400 px) to the tiled kernel and the small levels to a third launch: 242 -> 205 us per 1080p frame with every even-width level in the ONE strip launch
    int blur_strip_rows = 64;        // rows per strip (LVM_RZ_BLUR_STRIP_ROWS), halved until a level has 4096 strips
    double lo_freq = 0, hi_freq = 0, fps = 0;
    double la[3] = {}, lb[3] = {}, ha[3] = {}, hb[3] = {};
    bool steady(const lvm_params& p) const {
        return inited && lo_freq == p.coLow && hi_freq == p.coHigh && !std::isnan(la[0]) && !std::isnan(ha[0]);
    }
    ~RieszState() override { if (arena) (void)hipFree(arena); if (tarena) (void)hipFree(tarena); }
};
enum { F_BAND, F_P, F_R1, F_R2, F_PHC, F_PHS, F_LO0C, F_LO0S, F_LO1C, F_LO1S, F_HI0C, F_HI0S, F_HI1C, F_HI1S, F_AMP, F_TC, F_TS, F_BANDA, F_COUNT,
       F_R1C = F_COUNT, F_R2C, F_ALL };   // F_R1C / F_R2C: per-frame Riesz pair (aliases F_R1 / F_R2 in per-frame mode)

static int riesz_reserve_frames(Ctx* c, RieszState* st, int nt, hipStream_t s);
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
    total += pad(st->g[0].n * NS);                                                      // iab
    if (hipMalloc((void**)&st->arena, (total ? total : 64) * sizeof(float)) != hipSuccess) {
        st->arena = nullptr; c->err = "riesz: hipMalloc failed"; return LVM_ERR_OOM;
    }
    float* p = st->arena;
    for (int l = 0; l < levels; ++l) { st->oct[l] = p; p += pad(st->g[l].n * NS); st->res[l] = p; p += pad(st->g[l].n * NS); }
    for (int l = 0; l < levels - 1; ++l)
        for (int k = 0; k < F_COUNT; ++k) { st->f[l][k] = p; p += pad(st->g[l].n * NS); }
    for (int l = 0; l < levels - 1; ++l) { st->f[l][F_R1C] = st->f[l][F_R1]; st->f[l][F_R2C] = st->f[l][F_R2]; }
    st->iab = reinterpret_cast<uint32_t*>(p); p += pad(st->g[0].n * NS);
    return LVM_OK;
}

static void riesz_coeffs(double frq, double fps, double a[3], double b[3]) {   // TemporalFilter.cpp:324-327
    const double Wn = fps == 0.0 ? 0.0 : frq / (fps / 2.0);
    butterworth2(Wn, a, b);
}

// One buffer set = where the per-frame arrays of nt frames live ([frame][stream] planes).
struct RzBufs { float** oct; float** res; float* (*pf)[F_ALL_N]; int nt; uint32_t* iab; };
struct RieszState;
static bool rz_level_uses_strips(const RieszState* st, int l, int NZ);
// Rows per strip of a 9x9 strip kernel (k_rz_split_rows, k_rz_collapse_strips).  A strip of r rows runs r + 8 steps rounded up to groups
// of nine, the kernels are bound by their fma issue and a launch hands every SIMD ceil(strips / 1024) waves (workgroups of four waves,
// one per SIMD): cost = steps x waves per SIMD, with at least four waves per SIMD assumed (fewer do not hide the loads).
static int strip_rows_by_work(int sx, int h, int NZ) {
    long best = -1;
    int rows = 8;
    for (int r = 8; r <= 256; r += 2) {
        const long tasks = (long)sx * ((h + r - 1) / r) * NZ;
        const long steps = (r + 8 + 8) / 9 * 9;
        const long per_simd = (tasks + 1023) / 1024;
        const long cost = steps * (per_simd < 4 ? 4 : per_simd);
        if (best < 0 || cost < best) { best = cost; rows = r; }
    }
    return rows;
}

// pyramid of the nt frames: L plane + 9x9 split chain
static void rz_build(Ctx* c, RieszState* st, const FrameIO& io, const RzBufs& B, hipStream_t s) {
    const int NZ = c->nstreams * B.nt, w = io.w, h = io.h, nb = st->levels - 1;
    const dim3 blk(256);
    if (c->tile_mode == 2) {
        // spatial tiling, the gathered coarse levels (lvm_tile_riesz_planes): the "frame" IS an octave of another context's pyramid
        (void)hipMemcpyAsync(B.oct[0], c->tile_plane_in, (size_t)w * h * sizeof(float), hipMemcpyDeviceToDevice, s);
    } else if (fl_lut(lab_flavour(c))) {
        // OpenCV's forward table, once per frame: the float L plane for the pyramid and (ia, ib) for the output kernel
        lab_lut_planes(c, io.d_in, (long)io.in_stride, (long)io.in_sstride, w, h, NZ, nullptr, B.oct[0], B.iab, s);
    } else if (w % 4 == 0 && io.in_stride % 4 == 0 && io.in_sstride % 4 == 0 && ((uintptr_t)io.d_in % 4) == 0) {
        const long groups = (long)(w / 4) * h;
        LVM_LAUNCH(c, "rz_lab", k_rz_lab4, dim3((unsigned)((groups + 256 * kLabIters - 1) / (256 * kLabIters)), NZ), blk, s, io.d_in,
                   (long)io.in_stride, (long)io.in_sstride, w, h, B.oct[0], c->lab);
    } else {
        LVM_LAUNCH(c, "rz_lab", k_rz_lab, dim3((w + 255) / 256, h, NZ), blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, w, h, B.oct[0], c->lab);
    }
    for (int l = 0; l < nb; ++l) {
        const LevelGeom &a = st->g[l], &b = st->g[l + 1];
        if (st->split_rows && a.w % 4 == 0 && a.w >= 8 && (long)a.n * NZ >= st->split_rows_min && (long)a.n * 4 < (1L << 31)) {   // (32-bit byte offsets inside a plane)
            // wave strips (no LDS), rows per strip by strip_rows_by_work: 32 frames of 1080p run 68-row strips -- 4096 strips of 81
            // steps, four waves on every SIMD (100 VGPRs).  LVM_RZ_SPLIT_STRIP overrides the choice.
            const int sx = (a.w / 4 + SR_OWN - 1) / SR_OWN;
            int rows = st->split_strip;
            if (rows <= 0) rows = strip_rows_by_work(sx, a.h, NZ);
            const int sy = (a.h + rows - 1) / rows;
            const long ntasks = (long)sx * sy * NZ;
            LVM_LAUNCH(c, LName("rz_split", l), k_rz_split_rows, dim3((unsigned)((ntasks + SR_THREADS / 64 - 1) / (SR_THREADS / 64))), dim3(SR_THREADS), s,
                       (const float*)B.oct[l], a.w, a.h, B.pf[l][F_BAND], B.oct[l + 1], b.w, b.h, sx, sy, (int)ntasks, rows);
            continue;
        }
        if (st->split2 && a.w % 4 == 0 && (long)a.n * NZ >= st->split2_min) {
            const dim3 grid2((a.w + CW - 1) / CW, (a.h + C2H - 1) / C2H, NZ);
            LVM_LAUNCH(c, LName("rz_split", l), k_rz_split2, grid2, blk, s, (const float*)B.oct[l], a.w, a.h, B.pf[l][F_BAND], B.oct[l + 1], b.w, b.h);
            continue;
        }
        const dim3 grid((a.w + CW - 1) / CW, (a.h + CH - 1) / CH, NZ);
        LVM_LAUNCH(c, LName("rz_split", l), k_rz_split, grid, blk, s, (const float*)B.oct[l], a.w, a.h, B.pf[l][F_BAND], B.oct[l + 1], b.w, b.h);
    }
}

static void rz_phase(Ctx* c, RieszState* st, const RzBufs& B, int mode, hipStream_t s) {
    const int NS = c->nstreams, nb = st->levels - 1;
    if (nb < 1) return;
    PhaseArgs a;
    a.nlv = nb; a.mode = mode; a.nt = B.nt;
    a.la1 = st->la[1]; a.la2 = st->la[2]; a.lb0 = st->lb[0]; a.lb1 = st->lb[1]; a.lb2 = st->lb[2];
    a.ha1 = st->ha[1]; a.ha2 = st->ha[2]; a.hb0 = st->hb[0]; a.hb1 = st->hb[1]; a.hb2 = st->hb[2];
    auto split = [](double v) { SdF r; r.hi = (float)v; r.lo = (float)(v - (double)r.hi); return r; };
    a.fla1 = split(a.la1); a.fla2 = split(a.la2); a.flb0 = split(a.lb0); a.flb1 = split(a.lb1); a.flb2 = split(a.lb2);
    a.fha1 = split(a.ha1); a.fha2 = split(a.ha2); a.fhb0 = split(a.hb0); a.fhb1 = split(a.hb1); a.fhb2 = split(a.hb2);
    // levels whose width is a multiple of 4 go to the 4-pixels-per-thread kernel (64 x 16 tiles), the others to the scalar one
    // (32 x 8 tiles): two launches, each with its own level table
    PhaseArgs a4 = a;
    int n1 = 0, n4 = 0, blocks = 0, blocks4 = 0;
    for (int l = 0; l < nb; ++l) {
        const bool vec = st->phase4 && st->g[l].w % 4 == 0 && st->g[l].w >= 8 && NS * B.nt >= st->phase4_min_frames;
        PhaseLv& v = vec ? a4.lv[n4++] : a.lv[n1++];
        float** f = st->f[l];          // state planes
        float** q = B.pf[l];           // per-frame planes
        v.band = q[F_BAND]; v.P = f[F_P]; v.R1p = f[F_R1]; v.R2p = f[F_R2]; v.phc = f[F_PHC]; v.phs = f[F_PHS];
        v.lo0c = f[F_LO0C]; v.lo0s = f[F_LO0S]; v.lo1c = f[F_LO1C]; v.lo1s = f[F_LO1S];
        v.hi0c = f[F_HI0C]; v.hi0s = f[F_HI0S]; v.hi1c = f[F_HI1C]; v.hi1s = f[F_HI1S];
        v.amp = q[F_AMP]; v.tc = q[F_TC]; v.ts = q[F_TS]; v.R1c = q[F_R1C]; v.R2c = q[F_R2C];
        // the strip form of the amplify stage recomputes the Riesz pair from the band: no per-frame copy for its levels
        if (rz_level_uses_strips(st, l, NS * B.nt)) { v.R1c = v.R1p; v.R2c = v.R2p; }
        v.w = st->g[l].w; v.h = st->g[l].h;
        v.tx = (v.w + (vec ? P4_W : PT_W) - 1) / (vec ? P4_W : PT_W); v.ty = (v.h + (vec ? P4_H : PT_H) - 1) / (vec ? P4_H : PT_H);
        v.block0 = vec ? blocks4 : blocks;
        v.fs = (long)NS * (long)st->g[l].n;
        (vec ? blocks4 : blocks) += v.tx * v.ty * NS;
    }
    a.nlv = n1; a4.nlv = n4;
    if (n4) LVM_LAUNCH(c, mode ? "rz_seed" : "rz_phase", (lab_flavour(c) != FL_LUT_FAST) ? k_rz_phase4<true> : k_rz_phase4<false>, dim3(blocks4), dim3(256), s, a4);
    if (n1) LVM_LAUNCH(c, mode ? "rz_seed_small" : "rz_phase_small", (lab_flavour(c) != FL_LUT_FAST) ? k_rz_phase<true> : k_rz_phase<false>, dim3(blocks), dim3(256), s, a);
}

static bool rz_level_uses_strips(const RieszState* st, int l, int NZ) {
    return st->blur_strips && st->g[l].w % 2 == 0 && (long)st->g[l].n * NZ >= st->blur_strips_min;
}

// amplify + collapse + output of the nt frames (RieszPyramid.cpp:248-252, 304-325; MagnifyCore.hpp:269-277)
// rz_finish = rz_amplify (normalize + amplify, RieszPyramid.cpp:114-144) + rz_collapse_out (collapsePyramid :304-325 + the epilogue).  The two
// halves are separate entry points for the spatial-tiling demonstrator (lvm_tile_riesz_*): a stripe context stops after the first, the
// collapse then starts from a residual that another context computed.
static void rz_collapse_out(Ctx* c, RieszState* st, const lvm_params& p, const FrameIO& io, const RzBufs& B, hipStream_t s, const float* residual, float* plane_out);
static void rz_amplify(Ctx* c, RieszState* st, const lvm_params& p, const FrameIO& io, const RzBufs& B, hipStream_t s);
static void rz_finish(Ctx* c, RieszState* st, const lvm_params& p, const FrameIO& io, const RzBufs& B, hipStream_t s) {
    rz_amplify(c, st, p, io, B, s);
    if (c->tile_mode == 1) return;                                   // stripe context, stage 1: lvm_tile_riesz_stage2 collapses
    rz_collapse_out(c, st, p, io, B, s, B.oct[st->levels - 1], c->tile_mode == 2 ? c->tile_plane_out : nullptr);
}
static void rz_amplify(Ctx* c, RieszState* st, const lvm_params& p, const FrameIO& io, const RzBufs& B, hipStream_t s) {
    const int NZ = c->nstreams * B.nt, levels = st->levels, nb = levels - 1;
    const dim3 blk(256);
    if (nb >= 1) {
        BlurArgs a;
        a.nlv = nb;
        double t[13], sum = 0;   // getGaussianKernel(13, 3, CV_32F)
        for (int i = 0; i < 13; ++i) { const double x = i - 6.0; t[i] = std::exp(-0.5 / 9.0 * x * x); sum += t[i]; }
        sum = 1.0 / sum;
        for (int i = 0; i < 13; ++i) a.g[i] = (float)(t[i] * sum);
        const double PI_PERCENT = 3.1415926535897932384626433832795 / 100.0;
        a.alpha = (float)p.amplification; a.thr = (float)(p.coWavelength * PI_PERCENT);
        // levels whose width is a multiple of 4 (and not tiny) take the register-blocked kernel, the rest the scalar one
        BlurArgs a4 = a;
        BlurStripArgs as;
        for (int i = 0; i < 13; ++i) as.g[i] = a.g[i];
        as.alpha = a.alpha; as.thr = a.thr; as.nlv = 0; as.ntasks = 0;
        int blocks = 0, blocks4 = 0, n1 = 0, n4 = 0;
        for (int l = 0; l < nb; ++l) {
            // wave strips (no LDS) for the large levels with an even width: rows per strip chosen so that the launch keeps
            // the resident waves busy (a strip of r rows walks r + 12)
            if (rz_level_uses_strips(st, l, NZ)) {
                BlurStripLv& v = as.lv[as.nlv++];
                float** q = B.pf[l];
                v.amp = q[F_AMP]; v.tc = q[F_TC]; v.ts = q[F_TS]; v.band = q[F_BAND]; v.R1 = q[F_R1C]; v.R2 = q[F_R2C]; v.bandA = q[F_BANDA];
                v.w = st->g[l].w; v.h = st->g[l].h;
                v.sx = (v.w + BSW - 1) / BSW;
                int rows = st->blur_strip_rows;
                while (rows > 16 && (long)v.sx * ((v.h + rows - 1) / rows) * NZ < 4096) rows >>= 1;
                v.rows = rows; v.sy = (v.h + rows - 1) / rows;
                v.task0 = as.ntasks; as.ntasks += v.sx * v.sy * NZ;
                continue;
            }
            const bool big = st->blur4 && st->g[l].w % 4 == 0 && st->g[l].w >= 128 && st->g[l].h >= 64;
            BlurLv& v = big ? a4.lv[n4++] : a.lv[n1++];
            float** q = B.pf[l];
            v.amp = q[F_AMP]; v.tc = q[F_TC]; v.ts = q[F_TS]; v.band = q[F_BAND]; v.R1 = q[F_R1C]; v.R2 = q[F_R2C]; v.bandA = q[F_BANDA];
            v.w = st->g[l].w; v.h = st->g[l].h;
            if (big) { v.tx = (v.w + B2W - 1) / B2W; v.ty = (v.h + B2H - 1) / B2H; v.block0 = blocks4; blocks4 += v.tx * v.ty * NZ; }
            else { v.tx = (v.w + BT - 1) / BT; v.ty = (v.h + BTH - 1) / BTH; v.block0 = blocks; blocks += v.tx * v.ty * NZ; }
        }
        a.nlv = n1; a4.nlv = n4;
        if (as.nlv) LVM_LAUNCH(c, "rz_blur_amp", (lab_flavour(c) != FL_LUT_FAST) ? k_rz_blur_strips<true> : k_rz_blur_strips<false>,
                               dim3((unsigned)((as.ntasks + BS_THREADS / 64 - 1) / (BS_THREADS / 64))), dim3(BS_THREADS), s, as);
        if (n4) LVM_LAUNCH(c, as.nlv ? "rz_blur_amp_tiles" : "rz_blur_amp", (lab_flavour(c) != FL_LUT_FAST) ? k_rz_blur_amp4<true> : k_rz_blur_amp4<false>, dim3(blocks4), dim3(B2T), s, a4);
        if (n1) LVM_LAUNCH(c, "rz_blur_amp_small", (lab_flavour(c) != FL_LUT_FAST) ? k_rz_blur_amp<true> : k_rz_blur_amp<false>, dim3(blocks), blk, s, a);
    }
}
// residual = res_{L-1} (the context's own residual octave, or -- tiling -- the collapsed coarse levels of another context);
// plane_out != null: level 0 is collapsed into this float plane instead of passing through Lab2BGR (tiling, the coarse context)
static void rz_collapse_out(Ctx* c, RieszState* st, const lvm_params& p, const FrameIO& io, const RzBufs& B, hipStream_t s, const float* residual, float* plane_out) {
    const int NZ = c->nstreams * B.nt, w = io.w, h = io.h, levels = st->levels, nb = levels - 1;
    const dim3 blk(256);
    const float* resn = residual;
    // wave strips (k_rz_collapse_strips) for large launches on planes with even sizes and a width that is a multiple of 4
    auto strips_ok = [&](const LevelGeom& a, const LevelGeom& b) {
        return st->collapse_strips && a.w % 4 == 0 && a.w >= 8 && a.h % 2 == 0 && a.h >= 2 && b.w == a.w / 2 && b.h == a.h / 2 &&
               (long)a.n * NZ >= st->collapse_strips_min && (long)a.n * 4 < (1L << 31);      // (32-bit byte offsets inside a plane)
    };
    auto strips_geom = [&](CollapseStripArgs& ca, const LevelGeom& a, const LevelGeom& b) {
        ca.w = a.w; ca.h = a.h; ca.nw = b.w; ca.nh = b.h;
        ca.strips_x = (a.w / 4 + SR_OWN - 1) / SR_OWN;
        ca.rows = st->collapse_strip > 0 ? st->collapse_strip : strip_rows_by_work(ca.strips_x, a.h, NZ);
        ca.strips_y = (a.h + ca.rows - 1) / ca.rows;
        ca.ntasks = ca.strips_x * ca.strips_y * NZ;
    };
    for (int l = nb - 1; l >= (plane_out ? 0 : 1); --l) {
        const LevelGeom &a = st->g[l], &b = st->g[l + 1];
        float* res_l = (l == 0) ? plane_out : B.res[l];
        if (strips_ok(a, b)) {
            CollapseStripArgs ca{};
            ca.bandA = B.pf[l][F_BANDA]; ca.resn = resn; ca.res = res_l;
            strips_geom(ca, a, b);
            LVM_LAUNCH(c, LName("rz_collapse", l), (k_rz_collapse_strips<false, FL_LUT_FAST, false>),
                       dim3((unsigned)((ca.ntasks + CS_THREADS / 64 - 1) / (CS_THREADS / 64))), dim3(CS_THREADS), s, ca);
            resn = res_l;
            continue;
        }
        const dim3 grid((a.w + CW - 1) / CW, (a.h + CH - 1) / CH, NZ);
        const bool compact = st->compact && a.w % 2 == 0 && a.h % 2 == 0;
        LVM_LAUNCH(c, LName("rz_collapse", l), compact ? k_rz_collapse<true> : k_rz_collapse<false>, grid, blk, s, (const float*)B.pf[l][F_BANDA], resn,
                   res_l, a.w, a.h, b.w, b.h);
        resn = res_l;
    }
    if (plane_out) return;
    const int tx = (w + CW - 1) / CW, ty = (h + CH - 1) / CH;
    const int ntiles = tx * ty * NZ;
    // persistent workgroups (the two Lab tables, 17 KB, are loaded once per workgroup), but many more of them than fit on the
    // chip at once: the hardware dispatcher then balances the tail.  1080p, 32 frames (65280 tiles): 2048 workgroups 463-474 us,
    // 5120: 441-446, 10240: 427-433, 20480: 441, one tile per workgroup 487.
    const int gcap = st->fin_groups > 0 ? st->fin_groups : (ntiles / 6 > 2048 ? ntiles / 6 : 2048);
    const dim3 grid(ntiles < gcap ? ntiles : gcap);
    float* dbg = c->keep_float ? c->d_float : nullptr;
    const bool vec = w % 4 == 0 && io.in_stride % 4 == 0 && io.in_sstride % 4 == 0 && io.out_stride % 4 == 0 &&
                     io.out_sstride % 4 == 0 && ((uintptr_t)io.d_in % 4) == 0 && ((uintptr_t)io.d_out % 4) == 0;
    const int fl = lab_flavour(c);
    const bool compact = st->compact && w % 2 == 0 && h % 2 == 0;
    // (BANDS, FL, VEC, COMPACT, DBG) -> instantiation
    auto pick = [&](auto bands) {
        constexpr bool BN = decltype(bands)::value;
        auto p3 = [&](auto e, auto v, auto cp) {
            constexpr int E = decltype(e)::value; constexpr bool V = decltype(v)::value, CP = decltype(cp)::value;
            return dbg ? k_rz_final<BN, E, V, CP, true> : k_rz_final<BN, E, V, CP, false>;
        };
        auto p2 = [&](auto e, auto v) { return (compact && BN) ? p3(e, v, std::true_type{}) : p3(e, v, std::false_type{}); };
        auto p1 = [&](auto e) { return vec ? p2(e, std::true_type{}) : p2(e, std::false_type{}); };
        return fl == FL_ANALYTIC ? p1(std::integral_constant<int, FL_ANALYTIC>{})
                                 : (fl == FL_LUT_EXACT ? p1(std::integral_constant<int, FL_LUT_EXACT>{}) : p1(std::integral_constant<int, FL_LUT_FAST>{}));
    };
    auto kfb = pick(std::true_type{});
    auto kfn = pick(std::false_type{});
    const bool out_ok = io.out_stride % 4 == 0 && io.out_sstride % 4 == 0 && ((uintptr_t)io.d_out % 4) == 0 && (long)io.out_stride * h < (1L << 31);
    if (nb >= 1 && fl != FL_ANALYTIC && out_ok && strips_ok(st->g[0], st->g[1])) {
        CollapseStripArgs ca{};
        ca.bandA = B.pf[0][F_BANDA]; ca.resn = resn; ca.res = nullptr;
        ca.iab = B.iab; ca.out = (uint8_t*)io.d_out; ca.out_stride = (long)io.out_stride; ca.out_sstride = (long)io.out_sstride; ca.dbg = dbg; ca.lab = c->lab;
        strips_geom(ca, st->g[0], st->g[1]);
        auto kf = fl == FL_LUT_EXACT ? (dbg ? k_rz_collapse_strips<true, FL_LUT_EXACT, true> : k_rz_collapse_strips<true, FL_LUT_EXACT, false>)
                                     : (dbg ? k_rz_collapse_strips<true, FL_LUT_FAST, true> : k_rz_collapse_strips<true, FL_LUT_FAST, false>);
        LVM_LAUNCH(c, "rz_final", kf, dim3((unsigned)((ca.ntasks + CS_THREADS / 64 - 1) / (CS_THREADS / 64))), dim3(CS_THREADS), s, ca);
    } else if (nb >= 1)
        LVM_LAUNCH(c, "rz_final", kfb, grid, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.d_out,
                   (long)io.out_stride, (long)io.out_sstride, w, h, (const float*)B.pf[0][F_BANDA], resn, st->g[1].w, st->g[1].h,
                   c->lab, tx, ty, NZ, dbg, (const float*)B.oct[0], (const uint32_t*)B.iab);
    else
        LVM_LAUNCH(c, "rz_final", kfn, grid, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.d_out,
                   (long)io.out_stride, (long)io.out_sstride, w, h, (const float*)nullptr, (const float*)nullptr, 0, 0,
                   c->lab, tx, ty, NZ, dbg, (const float*)B.oct[0], (const uint32_t*)B.iab);
}

int riesz_process(Ctx* c, const lvm_params& p, int levels, const FrameIO& io, hipStream_t s, int* produced) {
    *produced = 0;
    if (io.channels < 3) return LVM_OK;                                          // MagnifyCore.hpp:212
    RieszState* st = static_cast<RieszState*>(c->state);
    if (!st) {
        st = new RieszState();
        if (const char* e = std::getenv("LVM_RZ_BLUR4")) st->blur4 = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_RZ_BLUR_STRIPS")) st->blur_strips = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_RZ_BLUR_STRIPS_MIN")) st->blur_strips_min = std::atol(e);
        if (const char* e = std::getenv("LVM_RZ_BLUR_STRIP_ROWS")) st->blur_strip_rows = std::atoi(e);
        if (const char* e = std::getenv("LVM_RZ_SPLIT2")) st->split2 = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_RZ_SPLIT2_MIN")) st->split2_min = std::atol(e);
        if (const char* e = std::getenv("LVM_RZ_PHASE4_MIN_FRAMES")) st->phase4_min_frames = std::atoi(e);
        if (const char* e = std::getenv("LVM_RZ_COMPACT")) st->compact = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_RZ_PHASE4")) st->phase4 = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_RZ_FIN_GROUPS")) st->fin_groups = std::atoi(e);
        if (const char* e = std::getenv("LVM_RZ_SPLIT_STRIP")) { const int v = std::atoi(e); if (v >= 2 && v % 2 == 0) st->split_strip = v; }
        if (const char* e = std::getenv("LVM_RZ_SPLIT_ROWS")) st->split_rows = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_RZ_COLLAPSE_STRIPS")) st->collapse_strips = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_RZ_COLLAPSE_STRIPS_MIN")) st->collapse_strips_min = std::atol(e);
        if (const char* e = std::getenv("LVM_RZ_COLLAPSE_STRIP")) { const int v = std::atoi(e); if (v >= 2 && v % 2 == 0) st->collapse_strip = v; }
        if (const char* e = std::getenv("LVM_RZ_SPLIT_ROWS_MIN")) st->split_rows_min = std::atol(e);
        c->state = st;
        int rc = riesz_alloc(c, st, io.w, io.h, levels);
        if (rc == LVM_OK && c->max_frames > 1) rc = riesz_reserve_frames(c, st, c->max_frames, s);
        if (rc != LVM_OK) return rc;
    }
    const RzBufs B{st->oct, st->res, st->f, 1, st->iab};
    rz_build(c, st, io, B, s);               // L plane + pyramid of the current frame (needed by every path below)
    // first frame ever, or degenerate coefficients: init and pass the frame through (:226-240)
    if (!st->inited || std::isnan(st->la[0]) || std::isnan(st->ha[0])) {
        st->lo_freq = p.coLow; st->hi_freq = p.coHigh; st->fps = p.framerate;
        riesz_coeffs(st->lo_freq, st->fps, st->la, st->lb);
        riesz_coeffs(st->hi_freq, st->fps, st->ha, st->hb);
        rz_phase(c, st, B, 1, s);
        st->inited = true;
        LVM_HIP_TRY(c, hipGetLastError());
        return LVM_OK;
    }
    // cutoff changed: new coefficients, both filters cleared, prior rebuilt from this frame (:243-254)
    bool reseed = false;
    if (st->lo_freq != p.coLow) { st->lo_freq = p.coLow; riesz_coeffs(st->lo_freq, st->fps, st->la, st->lb); reseed = true; }
    if (st->hi_freq != p.coHigh) { st->hi_freq = p.coHigh; riesz_coeffs(st->hi_freq, st->fps, st->ha, st->hb); reseed = true; }
    if (reseed) rz_phase(c, st, B, 2, s);
    rz_phase(c, st, B, 0, s);                                                    // :256-267
    rz_finish(c, st, p, io, B, s);                                               // :269-277
    LVM_HIP_TRY(c, hipGetLastError());
    *produced = 1;
    return LVM_OK;
}

// ---- spatial tiling of one stream (lvm_tile_riesz_*, a correctness demonstrator: SURVEY.md 8e) -------------------------------------------
// the residual octave of the context's pyramid (octave_{levels-1}): what a stripe context hands to the coarse context
int riesz_tile_residual(Ctx* c, float* d_dst, int* rw, int* rh, hipStream_t s) {
    RieszState* st = dynamic_cast<RieszState*>(c->state);
    if (!st) { c->err = "tiling: no Riesz state"; return LVM_ERR_INVALID; }
    const LevelGeom& g = st->g[st->levels - 1];
    if (rw) *rw = g.w;
    if (rh) *rh = g.h;
    if (d_dst) LVM_HIP_TRY(c, hipMemcpyAsync(d_dst, st->oct[st->levels - 1], g.n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return LVM_OK;
}
// stage 2 of a stripe context: collapse from `d_residual` (the coarse context's result, rows of this stripe) + Lab2BGR + u8
int riesz_tile_finish(Ctx* c, const lvm_params& p, const FrameIO& io, const float* d_residual, hipStream_t s) {
    RieszState* st = dynamic_cast<RieszState*>(c->state);
    if (!st || !st->inited || io.w != st->g[0].w || io.h != st->g[0].h) { c->err = "tiling: stage 2 without a matching stage 1"; return LVM_ERR_INVALID; }
    const RzBufs B{st->oct, st->res, st->f, 1, st->iab};
    rz_collapse_out(c, st, p, io, B, s, d_residual, nullptr);
    LVM_HIP_TRY(c, hipGetLastError());
    return LVM_OK;
}

// Buffers of a temporal batch; sized for max(nt, lvm_set_max_frames hint) so steady-state calls never allocate.
static int riesz_reserve_frames(Ctx* c, RieszState* st, int nt, hipStream_t s) {
    if (nt < c->max_frames) nt = c->max_frames;
    if (nt <= st->tcap) return LVM_OK;
    const int levels = st->levels, NS = c->nstreams;
    LVM_HIP_TRY(c, hipStreamSynchronize(s));
    sync_streams(c);
    if (st->tarena) (void)hipFree(st->tarena);
    st->tarena = nullptr; st->tcap = 0;
    auto pad = [](size_t n) { return (n + 63) & ~(size_t)63; };
    static const int kPer[] = {F_BAND, F_AMP, F_TC, F_TS, F_BANDA, F_R1C, F_R2C};
    size_t total = 64;
    for (int l = 0; l < levels; ++l) total += 2 * pad(st->g[l].n * NS * nt);
    // Levels whose amplify stage runs as strips for EVERY batch size of this context (rz_level_uses_strips already at one
    // frame per launch: level 0 of a 1080p stream) never touch a per-frame copy of the Riesz pair -- rz_phase points those
    // slots at the state planes and the strip kernel recomputes the pair -- so none is allocated (0.5 GB per 32 frames of
    // 1080p at level 0 alone).
    auto needs_pair = [&](int l) { return !rz_level_uses_strips(st, l, NS); };
    for (int l = 0; l < levels - 1; ++l) total += (needs_pair(l) ? 7 : 5) * pad(st->g[l].n * NS * nt);
    total += pad(st->g[0].n * NS * nt);
    if (hipMalloc((void**)&st->tarena, total * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); st->tarena = nullptr; c->err = "riesz: hipMalloc (frames) failed"; return LVM_ERR_OOM; }
    float* q = st->tarena;
    for (int l = 0; l < levels; ++l) { st->oct_t[l] = q; q += pad(st->g[l].n * NS * nt); st->res_t[l] = q; q += pad(st->g[l].n * NS * nt); }
    for (int l = 0; l < levels - 1; ++l) {
        for (int k = 0; k < F_ALL_N; ++k) st->ft[l][k] = st->f[l][k];          // state planes are shared
        for (int k : kPer) {
            if ((k == F_R1C || k == F_R2C) && !needs_pair(l)) { st->ft[l][k] = st->f[l][k == F_R1C ? F_R1 : F_R2]; continue; }   // as riesz_alloc does for per-frame calls
            st->ft[l][k] = q; q += pad(st->g[l].n * NS * nt);
        }
    }
    st->iab_t = reinterpret_cast<uint32_t*>(q); q += pad(st->g[0].n * NS * nt);
    st->tcap = nt;
    return LVM_OK;
}

// Temporal batch (see laplace_process_frames): nt consecutive frames, steady state only.
int riesz_process_frames(Ctx* c, const lvm_params& p, const FrameIO& io, int nt, hipStream_t s) {
    RieszState* st = static_cast<RieszState*>(c->state);
    if (nt > st->tcap) { const int rc = riesz_reserve_frames(c, st, nt, s); if (rc != LVM_OK) return rc; }
    const RzBufs B{st->oct_t, st->res_t, st->ft, nt, st->iab_t};
    rz_build(c, st, io, B, s);
    rz_phase(c, st, B, 0, s);
    rz_finish(c, st, p, io, B, s);
    LVM_HIP_TRY(c, hipGetLastError());
    return LVM_OK;
}

bool riesz_can_batch(const Ctx* c, const lvm_params& p) {
    const RieszState* st = dynamic_cast<const RieszState*>(c->state);
    return st && st->steady(p);
}

}  // namespace lvm
