// pyramid.h -- OpenCV-faithful pyrDown / pyrUp building blocks shared by laplace.hip and color.hip.
// Everything here is a template or inline device function so each translation unit carries its own
// copy (the library is built without relocatable device code).
#pragma once
#include "lvm_internal.h"

namespace lvm {

// ------------------------------------------------------------------------------------------
// pyrDown: horizontal row[x] = s[2x]*6 + (s[2x-1]+s[2x+1])*4 + s[2x-2] + s[2x+2],
//          vertical   dst   = (r2*6 + (r1+r3)*4 + r0 + r4) * (1/256), BORDER_REFLECT_101.
// Output tile DT_W x DT_H per workgroup, source tile (2*DT_W+3) x (2*DT_H+3) staged in LDS.
// ------------------------------------------------------------------------------------------
constexpr int DT_W = 32, DT_H = 16;
constexpr int DS_W = 2 * DT_W + 3, DS_H = 2 * DT_H + 3;

template <int C>
__device__ __forceinline__ void pyrdown_tile(float (&s_src)[C][DS_H][DS_W], float (&s_row)[C][DS_H][DT_W],
                                             float* __restrict__ dst, int dw, int dh, size_t dplane,
                                             int ox0, int oy0) {
    const int tid = threadIdx.x;
    for (int i = tid; i < C * DS_H * DT_W; i += 256) {
        const int c = i / (DS_H * DT_W);
        const int r = i - c * (DS_H * DT_W);
        const int ly = r / DT_W, x = r - ly * DT_W;
        const float* s = &s_src[c][ly][2 * x];
        s_row[c][ly][x] = s[2] * 6.f + (s[1] + s[3]) * 4.f + s[0] + s[4];
    }
    __syncthreads();
    for (int i = tid; i < C * DT_H * DT_W; i += 256) {
        const int c = i / (DT_H * DT_W);
        const int r = i - c * (DT_H * DT_W);
        const int y = r / DT_W, x = r - y * DT_W;
        const int gx = ox0 + x, gy = oy0 + y;
        if (gx < dw && gy < dh) {
            const float r0 = s_row[c][2 * y][x], r1 = s_row[c][2 * y + 1][x], r2 = s_row[c][2 * y + 2][x],
                        r3 = s_row[c][2 * y + 3][x], r4 = s_row[c][2 * y + 4][x];
            dst[c * dplane + (size_t)gy * dw + gx] = (r2 * 6.f + (r1 + r3) * 4.f + r0 + r4) * (1.f / 256.f);
        }
    }
}

// u8 frame -> float (Lab for C == 3, x/255 for C == 1; SCALE255 = false keeps [0,255] for the
// colour mode, MagnifyCore.hpp:169) -> pyrDown -> level-1 planes.
template <int C, bool LAB, int FL>
__global__ __launch_bounds__(256) void k_down0(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                               int w, int h, float* __restrict__ G1, int w1, int h1,
                                               LabCoef lab, float scale, LabPlanes lp) {
    __shared__ float s_src[C][DS_H][DS_W];
    __shared__ float s_row[C][DS_H][DT_W];
    __shared__ float s_gam[LAB && !fl_lut(FL) ? 256 : 1];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    if (LAB && !fl_lut(FL)) { load_gamma_u8(s_gam, lab.gamma_u8); __syncthreads(); }
    const int ox0 = blockIdx.x * DT_W, oy0 = blockIdx.y * DT_H;
    const uint8_t* src = in + (size_t)b * in_sstride;
    for (int i = tid; i < DS_H * DS_W; i += 256) {
        const int ly = i / DS_W, lx = i - ly * DS_W;
        const int gy = reflect101(2 * oy0 - 2 + ly, h), gx = reflect101(2 * ox0 - 2 + lx, w);
        const uint8_t* p = src + (size_t)gy * in_stride + (size_t)gx * C;
        if (LAB) {
            float L, a, bb;
            fetch_lab_px<FL>(src, in_stride, lp, (size_t)b * w * h, w, gy, gx, s_gam, lab, L, a, bb);
            s_src[0][ly][lx] = L; s_src[C > 1 ? 1 : 0][ly][lx] = a; s_src[C > 2 ? 2 : 0][ly][lx] = bb;
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) s_src[c][ly][lx] = (float)p[c] * scale;
        }
    }
    __syncthreads();
    const size_t plane = (size_t)w1 * h1;
    pyrdown_tile<C>(s_src, s_row, G1 + (size_t)b * C * plane, w1, h1, plane, ox0, oy0);
}

// float plane -> next level; blockIdx.z = plane
template <int TU>
__global__ __launch_bounds__(256) void k_pyr_down(const float* __restrict__ src, int w, int h,
                                                  float* __restrict__ dst, int dw, int dh) {
    __shared__ float s_src[1][DS_H][DS_W];
    __shared__ float s_row[1][DS_H][DT_W];
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * DT_W, oy0 = blockIdx.y * DT_H;
    const float* sp = src + (size_t)blockIdx.z * w * h;
    for (int i = tid; i < DS_H * DS_W; i += 256) {
        const int ly = i / DS_W, lx = i - ly * DS_W;
        const int gy = reflect101(2 * oy0 - 2 + ly, h), gx = reflect101(2 * ox0 - 2 + lx, w);
        s_src[0][ly][lx] = sp[(size_t)gy * w + gx];
    }
    __syncthreads();
    pyrdown_tile<1>(s_src, s_row, dst + (size_t)blockIdx.z * dw * dh, dw, dh, (size_t)dw * dh, ox0, oy0);
}

// ------------------------------------------------------------------------------------------
// pyrUp (OpenCV pyrUp_): per source row: even[2i] = s[i-1] + s[i]*6 + s[i+1], odd[2i+1] =
// (s[i]+s[i+1])*4; i == 0: even = s0*6 + s1*2; i == sw-1: even = s[sw-2] + s[sw-1]*7,
// odd = s[sw-1]*8.  Vertical on rows j-1, j, j+1 with row -1 -> 1 and row sh -> sh-1:
// dst[2j] = (r0 + r1*6 + r2)/64, dst[2j+1] = ((r1 + r2)*4)/64.  dsize is 2n or 2n-1.
// Output tile UT_W x UT_H, source tile (UT_W/2+2) x (UT_H/2+2).
// ------------------------------------------------------------------------------------------
#ifndef LVM_UT_W
#define LVM_UT_W 64
#define LVM_UT_H 16
#endif
constexpr int UT_W = LVM_UT_W, UT_H = LVM_UT_H;     // (128 x 8 measured too: see profiles/README.md)
static_assert(UT_W * UT_H == 1024 && UT_W % 2 == 0 && UT_H % 2 == 0, "four pixels per thread of a 256-thread workgroup");
constexpr int US_W = UT_W / 2 + 2, US_H = UT_H / 2 + 2;

// stage the source tile of one plane: rows sy0..sy0+US_H-1 (vertical border map), columns
// sx0..sx0+US_W-1 clamped into the plane (border columns are handled by pyrup_h's formulas)
__device__ __forceinline__ void pyrup_stage(float (&s)[US_H][US_W + 1], const float* __restrict__ src,
                                            int sw, int sh, int sx0, int sy0) {
    for (int i = threadIdx.x; i < US_H * US_W; i += 256) {
        const int ly = i / US_W, lx = i - ly * US_W;
        int gy = sy0 + ly;
        gy = gy < 0 ? 1 : (gy >= sh ? sh - 1 : gy);
        int gx = sx0 + lx;
        gx = gx < 0 ? 0 : (gx >= sw ? sw - 1 : gx);
        s[ly][lx] = src[(size_t)gy * sw + gx];
    }
}
// horizontal pass of the staged tile into hrow[US_H][UT_W]
__device__ __forceinline__ void pyrup_hpass(float (&hrow)[US_H][UT_W + 1], const float (&s)[US_H][US_W + 1],
                                            int x0, int sx0, int sw, int dw) {
    for (int i = threadIdx.x; i < US_H * UT_W; i += 256) {
        const int ly = i / UT_W, x = i - ly * UT_W;
        const int gx = x0 + x;
        hrow[ly][x] = (gx < dw) ? pyrup_h(&s[ly][0], gx, sx0, sw) : 0.f;
    }
}
// vertical pass for destination row gy; lj = local row of source row gy>>1
__device__ __forceinline__ float pyrup_v(const float (&hrow)[US_H][UT_W + 1], int x, int gy, int sy0) {
    const int lj = (gy >> 1) - sy0;
    const float r0 = hrow[lj - 1][x], r1 = hrow[lj][x], r2 = hrow[lj + 1][x];
    return sel((gy & 1) == 0, (r0 + r1 * 6.f + r2) * (1.f / 64.f), ((r1 + r2) * 4.f) * (1.f / 64.f));
}

// ---- vectorised u8 -> planes -> pyrDown (frames whose pixel groups of 4 are dword aligned) ----
struct __attribute__((packed, aligned(4))) Px4 { uint32_t a, b, c; };   // 4 BGR pixels

__device__ __forceinline__ void unpack_px4(const Px4 v, int (&B)[4], int (&G)[4], int (&R)[4]) {
    B[0] = v.a & 255; G[0] = (v.a >> 8) & 255; R[0] = (v.a >> 16) & 255;
    B[1] = v.a >> 24; G[1] = v.b & 255; R[1] = (v.b >> 8) & 255;
    B[2] = (v.b >> 16) & 255; G[2] = v.b >> 24; R[2] = v.c & 255;
    B[3] = (v.c >> 8) & 255; G[3] = (v.c >> 16) & 255; R[3] = v.c >> 24;
}

// u8 BGR -> Lab -> pyrDown -> G_1.  Output tile 32x16; the Lab source tile (35 rows x 72 columns
// starting at source column 64*tx - 4) is staged planar in LDS with a +2 column offset so that the
// 8 floats a pair of adjacent outputs needs are two aligned 128-bit reads; the vertical pass
// slides a 5-row register window (no intermediate LDS image).
constexpr int D0_ROWS = 2 * DT_H + 3, D0_GROUPS = 18, D0_PITCH = 76;
template <bool LAB, int FL>
__global__ __launch_bounds__(256) void k_down0_v4(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                                  int w, int h, float* __restrict__ G1, int w1, int h1, LabCoef lab, LabPlanes lp) {
    constexpr bool PLANES = LAB && fl_lut(FL);
    __shared__ __attribute__((aligned(16))) float s_src[3][D0_ROWS][D0_PITCH];
    __shared__ float s_gam[LAB && !fl_lut(FL) ? 256 : 1];
    if (LAB && !fl_lut(FL)) { load_gamma_u8(s_gam, lab.gamma_u8); __syncthreads(); }
    const int b = blockIdx.z;
    const int ox0 = blockIdx.x * DT_W, oy0 = blockIdx.y * DT_H;
    const int sx0 = 2 * ox0 - 4, sy0 = 2 * oy0 - 2;
    const uint8_t* src = in + (size_t)b * in_sstride;
    // all global loads of this thread are issued before the first use (3 pixel groups per thread)
    constexpr int NG = (D0_ROWS * D0_GROUPS + 255) / 256;
    Raw4 pv[NG];
    int prow[NG], pgy[NG], pgx[NG];
    const size_t poff = (size_t)b * w * h;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int i = threadIdx.x + k * 256;
        const int r = i / D0_GROUPS, g = i - r * D0_GROUPS;
        prow[k] = i < D0_ROWS * D0_GROUPS ? r : -1;
        pgy[k] = reflect101(sy0 + (r < D0_ROWS ? r : 0), h);
        pgx[k] = sx0 + 4 * g;
        pv[k] = Raw4{};
        if (prow[k] >= 0 && pgx[k] >= 0 && pgx[k] + 3 < w) pv[k] = load_raw4<PLANES>(src, in_stride, lp, poff, w, pgy[k], (unsigned)pgx[k]);
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        if (prow[k] < 0) continue;
        const int r = prow[k], gy = pgy[k], gx0 = pgx[k], g = (gx0 - sx0) >> 2;
        float L[4], A[4], Bb[4];
        if (gx0 >= 0 && gx0 + 3 < w) {
            if (LAB) raw4_to_lab<FL>(pv[k], s_gam, lab, L, A, Bb);
            else {
                int Bv[4], Gv[4], Rv[4];
                raw4_bgr(pv[k], Bv, Gv, Rv);
#pragma unroll
                for (int q = 0; q < 4; ++q) { L[q] = (float)Bv[q] * 1.0f; A[q] = (float)Gv[q] * 1.0f; Bb[q] = (float)Rv[q] * 1.0f; }   // colour mode: unscaled planes
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rx = reflect101(gx0 + q, w);
                if (LAB) fetch_lab_px<FL>(src, in_stride, lp, poff, w, gy, rx, s_gam, lab, L[q], A[q], Bb[q]);
                else {
                    const uint8_t* p = src + (size_t)gy * in_stride + (size_t)rx * 3;
                    L[q] = (float)p[0] * 1.0f; A[q] = (float)p[1] * 1.0f; Bb[q] = (float)p[2] * 1.0f;
                }
            }
        }
        float* d0 = &s_src[0][r][4 * g + 2];
        float* d1 = &s_src[1][r][4 * g + 2];
        float* d2 = &s_src[2][r][4 * g + 2];
        *reinterpret_cast<float2*>(d0) = make_float2(L[0], L[1]); *reinterpret_cast<float2*>(d0 + 2) = make_float2(L[2], L[3]);
        *reinterpret_cast<float2*>(d1) = make_float2(A[0], A[1]); *reinterpret_cast<float2*>(d1 + 2) = make_float2(A[2], A[3]);
        *reinterpret_cast<float2*>(d2) = make_float2(Bb[0], Bb[1]); *reinterpret_cast<float2*>(d2 + 2) = make_float2(Bb[2], Bb[3]);
    }
    __syncthreads();
    if (threadIdx.x < 192) {
        const int pair = threadIdx.x & 15, seg = (threadIdx.x >> 4) & 3, ch = threadIdx.x >> 6;
        const int x = 2 * pair;                       // tile-local output column (and x + 1)
        float w0a = 0, w1a = 0, w2a = 0, w3a = 0, w0b = 0, w1b = 0, w2b = 0, w3b = 0;
        float* dst = G1 + ((size_t)b * 3 + ch) * ((size_t)w1 * h1);
#pragma unroll
        for (int rr = 0; rr < 11; ++rr) {
            const float* row = &s_src[ch][8 * seg + rr][4 * pair + 4];
            const float4 u = *reinterpret_cast<const float4*>(row);
            const float4 v = *reinterpret_cast<const float4*>(row + 4);
            // taps of output x: u.x u.y u.z u.w v.x ; of output x+1: u.z u.w v.x v.y v.z
            const float ha = u.z * 6.f + (u.y + u.w) * 4.f + u.x + v.x;
            const float hb = v.x * 6.f + (u.w + v.y) * 4.f + u.z + v.z;
            if (rr >= 4 && (rr & 1) == 0) {
                const int gy = oy0 + 4 * seg + (rr - 4) / 2, gx = ox0 + x;
                const float oa = (w2a * 6.f + (w1a + w3a) * 4.f + w0a + ha) * (1.f / 256.f);
                const float ob = (w2b * 6.f + (w1b + w3b) * 4.f + w0b + hb) * (1.f / 256.f);
                if (gy < h1) {
                    if ((w1 & 1) == 0 && gx + 1 < w1) *reinterpret_cast<float2*>(&dst[(size_t)gy * w1 + gx]) = make_float2(oa, ob);
                    else {
                        if (gx < w1) dst[(size_t)gy * w1 + gx] = oa;
                        if (gx + 1 < w1) dst[(size_t)gy * w1 + gx + 1] = ob;
                    }
                }
            }
            w0a = w1a; w1a = w2a; w2a = w3a; w3a = ha;
            w0b = w1b; w1b = w2b; w2b = w3b; w3b = hb;
        }
    }
}

// ---- pyrUp of float planes without LDS (dsize = 2n exactly, dw % 4 == 0) ----------------------------------
// The tiled k_pyr_up spends ~48 instructions per output pixel (92 % VALU-busy in the colour up chain).  Here a
// lane owns a block of 4 columns x 2 rows (output rows 2j, 2j+1): three source rows x four dword loads, the
// horizontal pass with the border variants selected per lane, both vertical formulas, two 16-byte stores.
template <int TU>
__global__ __launch_bounds__(256) void k_pyr_up_rows(const float* __restrict__ src, int sw, int sh,
                                                     float* __restrict__ dst, int ngroups) {
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    const int dw = 2 * sw, gw = dw >> 2;                 // groups of 4 output columns per row
    const int j = gi / gw, gx = (gi - j * gw) * 4;
    const float* sp = src + (size_t)blockIdx.y * sw * sh;
    float* dp = dst + (size_t)blockIdx.y * (size_t)dw * (2 * sh);
    const int i0 = gx >> 1;
    const int cm1 = i0 > 0 ? i0 - 1 : 0, cp1 = i0 + 1 < sw ? i0 + 1 : sw - 1, cp2 = i0 + 2 < sw ? i0 + 2 : sw - 1;
    const bool f0 = i0 == 0, l0 = i0 == sw - 1, l1 = i0 + 1 == sw - 1;
    float h[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        int sy = j - 1 + q; sy = sy < 0 ? 1 : (sy >= sh ? sh - 1 : sy);       // vertical border map: row -1 -> 1, row sh -> sh - 1
        const float* r = sp + (size_t)sy * sw;
        const float sm1 = r[cm1], s0 = r[i0], s1 = r[cp1], s2 = r[cp2];
        h[q][0] = sel(f0, s0 * 6.f + s1 * 2.f, sel(l0, sm1 + s0 * 7.f, sm1 + s0 * 6.f + s1));
        h[q][1] = sel(l0, s0 * 8.f, (s0 + s1) * 4.f);
        h[q][2] = sel(l1, s0 + s1 * 7.f, s0 + s1 * 6.f + s2);
        h[q][3] = sel(l1, s1 * 8.f, (s1 + s2) * 4.f);
    }
    float e[4], o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e[k] = (h[0][k] + h[1][k] * 6.f + h[2][k]) * (1.f / 64.f);
        o[k] = ((h[1][k] + h[2][k]) * 4.f) * (1.f / 64.f);
    }
    float* d0 = dp + (size_t)(2 * j) * dw + gx;
    *reinterpret_cast<float4*>(d0) = make_float4(e[0], e[1], e[2], e[3]);
    *reinterpret_cast<float4*>(d0 + dw) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- u8 BGR -> (Lab) -> pyrDown -> G_1 as wave strips ----------------------------------------------
// Barrier-free form of k_down0_v4 (same preconditions, same arithmetic).  A wave owns a strip of 124 output
// columns x `rows` output rows of one frame.  Lane i holds the source pixel group 4g .. 4g+3, g = 62 tx - 1 + i,
// of the current source row: one 12-byte load, the colour conversion of its 4 pixels, and -- instead of an LDS
// tile -- two DPP wave shifts per channel hand it the two pixels left of the group and the pixel right of it
// (wave_shr:1 / wave_shl:1 cross all 64 lanes on gfx950).  Lanes 1 .. 62 then own the two outputs 2g, 2g+1;
// lanes 0 and 63 only feed their neighbours.  Groups outside the image are the REFLECT_101 mirror of the
// edge group, obtained by loading that group and swapping its pixels.  The horizontal results slide down
// the strip in a 5-row register window.  Only the gamma table lives in LDS.
constexpr int D0R_THREADS = 256, D0R_OUT = 124;

// Strip height for k_down0_rows: a strip of r output rows converts 2r + 3 source rows, and the launch takes as
// long as the busiest SIMD (1024 of them on MI355X) has strips -- minimise ceil(strips / 1024) * (2r + 3).
inline int down0_rows_choice(int w1, int h1, long frames, long min_tasks, long* tasks_out) {
    const long sx = (w1 + D0R_OUT - 1) / D0R_OUT;
    int best = 8; long best_cost = -1, best_tasks = 0;
    for (int r = 6; r <= 32; ++r) {
        const long tasks = sx * ((h1 + r - 1) / r) * frames;
        if (tasks < min_tasks) continue;                        // keep at least ~4 waves per SIMD in flight
        const long cost = ((tasks + 1023) / 1024) * (2 * r + 3);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; best_tasks = tasks; }
    }
    *tasks_out = best_tasks;                                    // 0: no strip height gives enough strips
    return best;
}
template <bool LAB, int FL>
__global__ __launch_bounds__(D0R_THREADS) void k_down0_rows(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                                            int w, int h, float* __restrict__ G1, int w1, int h1, LabCoef lab,
                                                            int strips_x, int strips_y, int ntasks, int rows, LabPlanes lp) {
    constexpr bool EXACT = fl_exact(FL);
    constexpr bool PLANES = LAB && fl_lut(FL);
    __shared__ float s_gam[LAB && !fl_lut(FL) ? 256 : 1];
    if (LAB && !fl_lut(FL)) { load_gamma_u8(s_gam, lab.gamma_u8); __syncthreads(); }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int task = blockIdx.x * (D0R_THREADS / 64) + wave;
    if (task >= ntasks) return;
    const int b = task / (strips_x * strips_y);
    const int r = task - b * (strips_x * strips_y);
    const int ty = r / strips_x, tx = r - ty * strips_x;
    const int ngroups = w >> 2;                                    // w % 4 == 0
    const int g = tx * (D0R_OUT / 2) - 1 + lane;                    // source group of this lane
    const bool left_mirror = g < 0, right_mirror = g >= ngroups;    // REFLECT_101 mirrors of the edge groups
    const int gl = left_mirror ? 0 : (right_mirror ? ngroups - 1 : g);
    const uint8_t* src = in + (size_t)b * in_sstride;
    const size_t poff = (size_t)b * w * h;
    const int ox = 2 * g, oy0 = ty * rows;
    const bool owner = lane >= 1 && lane <= 62 && g >= 0 && ox < w1;    // (ox even, w1 even: ox + 1 < w1 too)
    // colour planes of the lane's 4 pixels for source row sy, and the two horizontal pyrDown results per plane
    auto fetch = [&](int sy) __attribute__((always_inline)) {
        return load_raw4<PLANES>(src, in_stride, lp, poff, w, reflect101(sy, h), 4u * (unsigned)gl);
    };
    auto hrow = [&](const Raw4 pv, float (&ha)[3], float (&hb)[3]) __attribute__((always_inline)) {
        float P[3][4];
        if (LAB) raw4_to_lab<FL>(pv, s_gam, lab, P[0], P[1], P[2]);
        else {
            int Bv[4], Gv[4], Rv[4];
            raw4_bgr(pv, Bv, Gv, Rv);
#pragma unroll
            for (int q = 0; q < 4; ++q) { P[0][q] = (float)Bv[q] * 1.0f; P[1][q] = (float)Gv[q] * 1.0f; P[2][q] = (float)Rv[q] * 1.0f; }   // colour mode: unscaled planes
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // what the neighbours see of this lane: (p2, p3) towards the right neighbour, p0 towards the left one.
            // Mirrored groups: pixels -2, -1 are pixels 2, 1 of group 0; pixel w is pixel w-2 = pixel 2 of the last group.
            const float give2 = P[c][2], give3 = left_mirror ? P[c][1] : P[c][3], give0 = right_mirror ? P[c][2] : P[c][0];
            const float L2 = dpp_shr1(give2), L3 = dpp_shr1(give3), R0 = dpp_shl1(give0);
            if (!EXACT && LVM_FAST_FMA) {   // default flavour: the same sum as two fmas (two roundings less)
                ha[c] = __builtin_fmaf(P[c][0], 6.f, __builtin_fmaf(L3 + P[c][1], 4.f, L2 + P[c][2]));
                hb[c] = __builtin_fmaf(P[c][2], 6.f, __builtin_fmaf(P[c][1] + P[c][3], 4.f, P[c][0] + R0));
            } else {
                ha[c] = P[c][0] * 6.f + (L3 + P[c][1]) * 4.f + L2 + P[c][2];
                hb[c] = P[c][2] * 6.f + (P[c][1] + P[c][3]) * 4.f + P[c][0] + R0;
            }
        }
    };
    const int yend = oy0 + rows < h1 ? oy0 + rows : h1;
    float a0[3], a1[3], a2[3], a3[3], a4[3], b0[3], b1[3], b2[3], b3[3], b4[3];
    hrow(fetch(2 * oy0 - 2), a0, b0); hrow(fetch(2 * oy0 - 1), a1, b1); hrow(fetch(2 * oy0), a2, b2);
    const size_t plane = (size_t)w1 * h1;
    float* dst = G1 + (size_t)b * 3 * plane;
    // the two source rows of the NEXT output row are fetched before the current ones are converted: a wave otherwise waits
    // for its 12-byte loads once per output row with nothing of its own to do: 159-164 -> 147-151 us per 32 frames
    Raw4 n3 = fetch(2 * oy0 + 1), n4 = fetch(2 * oy0 + 2);
    for (int oy = oy0; oy < yend; ++oy) {
        const Raw4 c3 = n3, c4 = n4;
        if (oy + 1 < yend) { n3 = fetch(2 * oy + 3); n4 = fetch(2 * oy + 4); }
        hrow(c3, a3, b3); hrow(c4, a4, b4);
        if (owner) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float va, vb;
                if (!EXACT && LVM_FAST_FMA) {
                    va = __builtin_fmaf(a2[c], 6.f, __builtin_fmaf(a1[c] + a3[c], 4.f, a0[c] + a4[c])) * (1.f / 256.f);
                    vb = __builtin_fmaf(b2[c], 6.f, __builtin_fmaf(b1[c] + b3[c], 4.f, b0[c] + b4[c])) * (1.f / 256.f);
                } else {
                    va = (a2[c] * 6.f + (a1[c] + a3[c]) * 4.f + a0[c] + a4[c]) * (1.f / 256.f);
                    vb = (b2[c] * 6.f + (b1[c] + b3[c]) * 4.f + b0[c] + b4[c]) * (1.f / 256.f);
                }
                *reinterpret_cast<float2*>(dst + c * plane + (size_t)oy * w1 + ox) = make_float2(va, vb);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { a0[c] = a2[c]; a1[c] = a3[c]; a2[c] = a4[c]; b0[c] = b2[c]; b1[c] = b3[c]; b2[c] = b4[c]; }
    }
}

// ---- u8 BGR -> OpenCV's forward Lab table -> pyrDown -> G_1, AND the integer Lab planes, in one pass ----------------
// The conversion kernel (labconv.hip) and k_down0_rows fused for large launches (temporal batches / many streams): the
// strip walk, DPP halo exchange and 5-row window of k_down0_rows, with every lane converting its 4-pixel group through
// the table (lut_lab_int: (a, b) nodes in LDS, L cells from L2) instead of reading planes back, and storing the integer
// planes the output kernel needs for the pixels it OWNS (lanes 1 .. 62, source rows 2 oy0 .. 2 yend - 1: every pixel of
// the frame exactly once).  Saves the 6 bytes per pixel k_down0_rows would read and one launch; the table keeps one
// persistent 1024-thread workgroup per CU, whose 16 waves take strips round-robin.
#ifndef LVM_D0L_THREADS
#define LVM_D0L_THREADS 1024
#endif
#ifndef LVM_D0L_DEPTH
#define LVM_D0L_DEPTH 3          // table look-ups in flight per lane: the reads of the pixels k + 1, k + 2 are issued before pixel k is interpolated
                                 // (1 = one pixel at a time, the compiler's own overlap).  Measured per 32 frames of 1080p: 262 / 253 / 250 us at depth 1 / 2 / 3
#endif
constexpr int D0L_THREADS = LVM_D0L_THREADS;
// strip height for `waves` resident waves (k_down0_rows: one wave per SIMD slot of its 256-thread workgroups)
inline int down0_lut_rows_choice(int w1, int h1, long frames, long waves, long* tasks_out) {
    const long sx = (w1 + D0R_OUT - 1) / D0R_OUT;
    int best = 8; long best_cost = -1, best_tasks = 0;
    for (int r = 6; r <= 48; ++r) {
        const long tasks = sx * ((h1 + r - 1) / r) * frames;
        if (tasks * 5 < waves * 3) continue;                    // at least 60 % of the resident waves get a strip
        const long cost = ((tasks + waves - 1) / waves) * (2 * r + 3);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; best_tasks = tasks; }
    }
    *tasks_out = best_tasks;
    return best;
}
struct D0LArgs {
    const uint8_t* in; long in_stride, in_sstride; int w, h;
    float* G1; int w1, h1;
    LabLut lut;
    int strips_x, strips_y, ntasks, rows;
    uint16_t* iLp; uint32_t* iabp;
};
// one wave strip (task) of the fused conversion + first pyramid kernel; s_ab = the (a, b) node table in LDS
template <int FL>
__device__ __forceinline__ void down0_lut_strip(const D0LArgs& q, int task, int lane, const uint32_t* s_ab) {
    constexpr bool EXACT = fl_exact(FL);
    const uint8_t* __restrict__ in = q.in; const long in_stride = q.in_stride, in_sstride = q.in_sstride;
    const int w = q.w, h = q.h, w1 = q.w1, h1 = q.h1, strips_x = q.strips_x, strips_y = q.strips_y, rows = q.rows;
    float* __restrict__ G1 = q.G1; const LabLut lut = q.lut;
    uint16_t* __restrict__ iLp = q.iLp; uint32_t* __restrict__ iabp = q.iabp;
    const int ngroups = w >> 2;                                    // w % 4 == 0
    const size_t plane = (size_t)w1 * h1;
    const int b = task / (strips_x * strips_y);
    const int r = task - b * (strips_x * strips_y);
    const int ty = r / strips_x, tx = r - ty * strips_x;
    const int g = tx * (D0R_OUT / 2) - 1 + lane;                    // source group of this lane
    const bool left_mirror = g < 0, right_mirror = g >= ngroups;    // REFLECT_101 mirrors of the edge groups
    const int gl = left_mirror ? 0 : (right_mirror ? ngroups - 1 : g);
    const uint8_t* src = in + (size_t)b * in_sstride;
    const int ox = 2 * g, oy0 = ty * rows;
    const bool owner = lane >= 1 && lane <= 62 && g >= 0 && ox < w1;
    const int yend = oy0 + rows < h1 ? oy0 + rows : h1;
    const int own_lo = 2 * oy0, own_hi = 2 * yend < h ? 2 * yend : h;    // source rows whose planes this strip stores
    uint16_t* pL = iLp + (size_t)b * w * h + 4u * (unsigned)gl;
    uint32_t* pab = iabp + (size_t)b * w * h + 4u * (unsigned)gl;
    struct __attribute__((packed, aligned(4))) P3 { uint32_t a, b, c; };
    auto fetch = [&](int sy) __attribute__((always_inline)) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(src + (size_t)reflect101(sy, h) * in_stride + 12u * (unsigned)gl);
        P3 v; v.a = __builtin_nontemporal_load(q); v.b = __builtin_nontemporal_load(q + 1); v.c = __builtin_nontemporal_load(q + 2);
        return v;
    };
    // source row sy: conversion of the lane's 4 pixels, plane stores, the two horizontal pyrDown results per channel
    auto hrow = [&](const P3 v, int sy, float (&ha)[3], float (&hb)[3]) __attribute__((always_inline)) {
        const uint32_t pb[12] = {v.a & 255, (v.a >> 8) & 255, (v.a >> 16) & 255, v.a >> 24, v.b & 255, (v.b >> 8) & 255,
                                 (v.b >> 16) & 255, v.b >> 24, v.c & 255, (v.c >> 8) & 255, (v.c >> 16) & 255, v.c >> 24};
        int iL[4], ia[4], ib[4];
#if LVM_D0L_DEPTH >= 2
        {   // the look-ups of pixel k + 1 (k + 1, k + 2 with depth 3) are issued before pixel k is interpolated
            LutRefs r[4];
#pragma unroll
            for (int k = 0; k < LVM_D0L_DEPTH - 1; ++k) r[k] = lut_issue(pb[3 * k], pb[3 * k + 1], pb[3 * k + 2], s_ab, lut.Lcells);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k + LVM_D0L_DEPTH - 1 < 4) { const int n = k + LVM_D0L_DEPTH - 1; r[n] = lut_issue(pb[3 * n], pb[3 * n + 1], pb[3 * n + 2], s_ab, lut.Lcells); }
                __builtin_amdgcn_sched_barrier(0);
                lut_finish(r[k], iL[k], ia[k], ib[k]);
            }
        }
#else
#pragma unroll
        for (int k = 0; k < 4; ++k) lut_lab_int(pb[3 * k], pb[3 * k + 1], pb[3 * k + 2], s_ab, lut.Lcells, iL[k], ia[k], ib[k]);
#endif
        if (owner && sy >= own_lo && sy < own_hi) {
            const size_t o = (size_t)sy * w;
            uint32_t* dL = reinterpret_cast<uint32_t*>(pL + o);
            // (nontemporal: plain stores measured 229 -> 232 us for this kernel and 49 -> 63 us for the pyrDown that follows -- the planes
            //  evict G_1 from the caches; profiles/README.md round 4)
            __builtin_nontemporal_store((uint32_t)iL[0] | ((uint32_t)iL[1] << 16), dL);
            __builtin_nontemporal_store((uint32_t)iL[2] | ((uint32_t)iL[3] << 16), dL + 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) __builtin_nontemporal_store((uint32_t)ia[k] | ((uint32_t)ib[k] << 16), pab + o + k);
        }
        float P[3][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { P[0][k] = lut_L(iL[k]); P[1][k] = lut_ab(ia[k]); P[2][k] = lut_ab(ib[k]); }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float give2 = P[c][2], give3 = left_mirror ? P[c][1] : P[c][3], give0 = right_mirror ? P[c][2] : P[c][0];
            const float L2 = dpp_shr1(give2), L3 = dpp_shr1(give3), R0 = dpp_shl1(give0);
            if (!EXACT && LVM_FAST_FMA) {
                ha[c] = __builtin_fmaf(P[c][0], 6.f, __builtin_fmaf(L3 + P[c][1], 4.f, L2 + P[c][2]));
                hb[c] = __builtin_fmaf(P[c][2], 6.f, __builtin_fmaf(P[c][1] + P[c][3], 4.f, P[c][0] + R0));
            } else {
                ha[c] = P[c][0] * 6.f + (L3 + P[c][1]) * 4.f + L2 + P[c][2];
                hb[c] = P[c][2] * 6.f + (P[c][1] + P[c][3]) * 4.f + P[c][0] + R0;
            }
        }
    };
    float a0[3], a1[3], a2[3], a3[3], a4[3], b0[3], b1[3], b2[3], b3[3], b4[3];
    hrow(fetch(2 * oy0 - 2), 2 * oy0 - 2, a0, b0); hrow(fetch(2 * oy0 - 1), 2 * oy0 - 1, a1, b1); hrow(fetch(2 * oy0), 2 * oy0, a2, b2);
    float* dst = G1 + (size_t)b * 3 * plane;
    P3 n3 = fetch(2 * oy0 + 1), n4 = fetch(2 * oy0 + 2);
    for (int oy = oy0; oy < yend; ++oy) {
        const P3 c3 = n3, c4 = n4;
        if (oy + 1 < yend) { n3 = fetch(2 * oy + 3); n4 = fetch(2 * oy + 4); }
        hrow(c3, 2 * oy + 1, a3, b3); hrow(c4, 2 * oy + 2, a4, b4);
        if (owner) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float va, vb;
                if (!EXACT && LVM_FAST_FMA) {
                    va = __builtin_fmaf(a2[c], 6.f, __builtin_fmaf(a1[c] + a3[c], 4.f, a0[c] + a4[c])) * (1.f / 256.f);
                    vb = __builtin_fmaf(b2[c], 6.f, __builtin_fmaf(b1[c] + b3[c], 4.f, b0[c] + b4[c])) * (1.f / 256.f);
                } else {
                    va = (a2[c] * 6.f + (a1[c] + a3[c]) * 4.f + a0[c] + a4[c]) * (1.f / 256.f);
                    vb = (b2[c] * 6.f + (b1[c] + b3[c]) * 4.f + b0[c] + b4[c]) * (1.f / 256.f);
                }
                *reinterpret_cast<float2*>(dst + c * plane + (size_t)oy * w1 + ox) = make_float2(va, vb);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { a0[c] = a2[c]; a1[c] = a3[c]; a2[c] = a4[c]; b0[c] = b2[c]; b1[c] = b3[c]; b2[c] = b4[c]; }
    }
}
template <int FL>
__global__ __launch_bounds__(D0L_THREADS) void k_down0_lut_rows(D0LArgs q) {
    __shared__ uint32_t s_ab[kLabAbWords];
    for (int i = threadIdx.x; i < kLabAbWords; i += D0L_THREADS) s_ab[i] = q.lut.ab[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    for (int task = blockIdx.x * (D0L_THREADS / 64) + wave; task < q.ntasks; task += gridDim.x * (D0L_THREADS / 64))
        down0_lut_strip<FL>(q, task, lane, s_ab);
}


// ---- u8 BGR -> pyrDown -> pyrDown -> G_2 in one pass (colour mode) ------------------------------------------------------------
// Colour magnification only keeps the SMALLEST level of its Gaussian pyramid (the temporal window); G_1 is written by
// k_down0_rows and read once by the next pyrDown: 12.4 MB per 1080p frame of traffic for an image nobody else reads.  This kernel
// keeps k_down0_rows' strip walk for the first level (lane = one 4-pixel source group, DPP halo, 5-row register window) and runs the
// second pyrDown on the level-1 pixels while they are in registers: a lane's two level-1 pixels (columns 2g, 2g + 1) plus three
// neighbours by DPP give level-2 column g, whose horizontal sums slide down a second 5-row window.
// Exactness: the planes are unscaled u8 values, so every level-1 value is k / 256 with k < 2^16 and every level-2 value k / 65536
// with k < 2^24 -- all partial sums are exactly representable floats, and ANY summation order (fma included) gives the reference's
// bits.  (From level 3 on the sums round and the order is pyrdown_tile's again: those levels stay with the generic kernels.)
// Borders: BORDER_REFLECT_101 of the SOURCE comes from the mirrored row / group loads as in k_down0_rows; REFLECT_101 of LEVEL 1 is
// a property of level-1 values -- columns -2, -1 are columns 2, 1 and column w1 is column w1 - 2 (per-lane selects in the border
// lanes), rows -2, -1 are rows 2, 1 (the top strip starts at level-1 row 0 and mirrors its window) and rows >= h1 are copies of
// window entries.  Requires w % 8 == 0 (w1 even, one level-2 column per source group) and buffer-addressable frames.
constexpr int D01_THREADS = 256, D01_OUT = 60;      // level-2 columns per wave (lanes 2 .. 61)
// strip height: a strip of r level-2 rows converts 4 r + 9 source rows; as down0_rows_choice
inline int down01_rows_choice(int w2, int h2, long frames, long min_tasks, long* tasks_out) {
    const long sx = (w2 + D01_OUT - 1) / D01_OUT;
    int best = 8; long best_cost = -1, best_tasks = 0;
    for (int r = 4; r <= 34; ++r) {
        const long tasks = sx * ((h2 + r - 1) / r) * frames;
        if (tasks < min_tasks) continue;
        const long cost = ((tasks + 1023) / 1024) * (4 * r + 9);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = r; best_tasks = tasks; }
    }
    *tasks_out = best_tasks;
    return best;
}
struct D01Args {
    const uint8_t* in; long in_stride, in_sstride; int w, h;
    float* G2; int w2, h2, h1;
    int strips_x, strips_y, ntasks, rows;
};
template <int TU>
__global__ __launch_bounds__(D01_THREADS) void k_down01_rows(D01Args q) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int task = blockIdx.x * (D01_THREADS / 64) + wave;
    if (task >= q.ntasks) return;
    const int w = q.w, h = q.h, w2 = q.w2, h2 = q.h2, h1 = q.h1, rows = q.rows;
    const int b = task / (q.strips_x * q.strips_y);
    const int r = task - b * (q.strips_x * q.strips_y);
    const int ty = r / q.strips_x, tx = r - ty * q.strips_x;
    const int ngroups = w >> 2;                                    // == w2
    const int g = tx * D01_OUT - 2 + lane;                          // source group = level-2 column of this lane
    const bool left_mirror = g < 0, right_mirror = g >= ngroups;    // source REFLECT_101 mirrors of the edge groups (only g = -1, ngroups are used)
    const int gl = left_mirror ? 0 : (right_mirror ? ngroups - 1 : g);
    const bool first2 = g == 0, last2 = g == w2 - 1;               // level-1 REFLECT_101 lanes
    const bool owner = lane >= 2 && lane <= 61 && g >= 0 && g < w2;
    const unsigned in_stride = (unsigned)q.in_stride;
    const BufRsrc rin = buf_rsrc(q.in + (size_t)b * q.in_sstride, in_stride * (unsigned)(h - 1) + (unsigned)w * 3u);
    const unsigned voff = 12u * (unsigned)gl;
    auto fetch = [&](int sy) __attribute__((always_inline)) { return buf_ld_b96(rin, voff, (unsigned)reflect101(sy, h) * in_stride); };
    // horizontal level-1 sums of one source row at the lane's two level-1 columns (k_down0_rows' exchange)
    auto hrow = [&](const B96 v, float (&ha)[3], float (&hb)[3]) __attribute__((always_inline)) {
        const uint32_t pb[12] = {v.a & 255, (v.a >> 8) & 255, (v.a >> 16) & 255, v.a >> 24, v.b & 255, (v.b >> 8) & 255,
                                 (v.b >> 16) & 255, v.b >> 24, v.c & 255, (v.c >> 8) & 255, (v.c >> 16) & 255, v.c >> 24};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p0 = (float)pb[c], p1 = (float)pb[3 + c], p2 = (float)pb[6 + c], p3 = (float)pb[9 + c];
            const float give2 = p2, give3 = left_mirror ? p1 : p3, give0 = right_mirror ? p2 : p0;
            const float L2 = dpp_shr1(give2), L3 = dpp_shr1(give3), R0 = dpp_shl1(give0);
            ha[c] = __builtin_fmaf(p0, 6.f, __builtin_fmaf(L3 + p1, 4.f, L2 + p2));
            hb[c] = __builtin_fmaf(p2, 6.f, __builtin_fmaf(p1 + p3, 4.f, p0 + R0));
        }
    };
    const int Y0 = ty * rows, Yend = Y0 + rows < h2 ? Y0 + rows : h2;
    const bool top = Y0 == 0;
    int rv = top ? 0 : 2 * Y0 - 2;                                 // next level-1 row to make
    float a0[3], a1[3], a2[3], a3[3], a4[3], b0[3], b1[3], b2[3], b3[3], b4[3];
    hrow(fetch(2 * rv - 2), a0, b0); hrow(fetch(2 * rv - 1), a1, b1); hrow(fetch(2 * rv), a2, b2);
    // two alternating sets of prefetched source rows: the rows of level-1 row rv (set A), rv + 1 (set B), and so on -- a set is
    // re-loaded right after its use for the level-1 row two steps later, so no register in flight is ever copied
    B96 nA0 = fetch(2 * rv + 1), nA1 = fetch(2 * rv + 2), nB0 = fetch(2 * rv + 3), nB1 = fetch(2 * rv + 4);
    // level-1 row rv from the source window -> its level-2 horizontal sums at column g (one value per channel)
    auto l1 = [&](B96& n0, B96& n1, float (&H2)[3]) __attribute__((always_inline)) {
        hrow(n0, a3, b3); hrow(n1, a4, b4);
        n0 = fetch(2 * rv + 5); n1 = fetch(2 * rv + 6);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float va = __builtin_fmaf(a2[c], 6.f, __builtin_fmaf(a1[c] + a3[c], 4.f, a0[c] + a4[c])) * (1.f / 256.f);
            const float vb = __builtin_fmaf(b2[c], 6.f, __builtin_fmaf(b1[c] + b3[c], 4.f, b0[c] + b4[c])) * (1.f / 256.f);
            float Lm2 = dpp_shr1(va), Lm1 = dpp_shr1(vb), Rp = dpp_shl1(va);
            Lm2 = sel(first2, Rp, Lm2); Lm1 = sel(first2, vb, Lm1);   // level-1 columns -2, -1 = columns 2, 1
            Rp = sel(last2, va, Rp);                                  // level-1 column w1 = column w1 - 2
            H2[c] = __builtin_fmaf(va, 6.f, __builtin_fmaf(Lm1 + vb, 4.f, Lm2 + Rp));
            a0[c] = a2[c]; a1[c] = a3[c]; a2[c] = a4[c]; b0[c] = b2[c]; b1[c] = b3[c]; b2[c] = b4[c];
        }
        ++rv;
    };
    float w0[3] = {0.f, 0.f, 0.f}, w1[3] = {0.f, 0.f, 0.f}, w2r[3], w3[3], w4[3];
    // an odd number of rows before the loop (1 for the top strip, 3 otherwise): the loop continues with set B, then A
    if (top) l1(nA0, nA1, w2r);
    else { l1(nA0, nA1, w0); l1(nB0, nB1, w1); l1(nA0, nA1, w2r); }
    const size_t plane = (size_t)w2 * h2;
    float* dst = q.G2 + (size_t)b * 3 * plane + g;
    for (int Y = Y0; Y < Yend; ++Y) {
        // window rows 2Y - 2 .. 2Y + 2 in w0, w1, w2r, w3, w4; rows past the last level-1 row are REFLECT_101 copies of window entries
        // (row h1 + k = row h1 - 2 - k), but their source rows are still loaded and converted (static schedule; the results are dropped)
        float t3[3], t4[3];
        const int r3 = rv;
        l1(nB0, nB1, t3);
        const int r4 = rv;
        l1(nA0, nA1, t4);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float e3 = t3[c], e4 = t4[c];
            if (r3 >= h1) { const int k = 2 * h1 - 2 - r3 - (2 * Y - 2); e3 = k == 0 ? w0[c] : (k == 1 ? w1[c] : w2r[c]); }
            if (r4 >= h1) { const int k = 2 * h1 - 2 - r4 - (2 * Y - 2); e4 = k == 0 ? w0[c] : (k == 1 ? w1[c] : (k == 2 ? w2r[c] : e3)); }
            w3[c] = e3; w4[c] = e4;
        }
        if (top && Y == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { w0[c] = w4[c]; w1[c] = w3[c]; }
        }
        if (owner) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                dst[c * plane + (size_t)Y * w2] = __builtin_fmaf(w2r[c], 6.f, __builtin_fmaf(w1[c] + w3[c], 4.f, w0[c] + w4[c])) * (1.f / 256.f);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { w0[c] = w2r[c]; w1[c] = w3[c]; w2r[c] = w4[c]; }
    }
}

// ---- pyrDown of large float planes: wave strips, no LDS ---------------------------------------------
// Every wave owns a strip of 128 output columns x `rows` output rows of one plane; a lane produces two
// adjacent outputs per row.  The 7 source values a lane needs from a source row (columns 2x-2 .. 2x+4)
// come straight from global memory as an aligned 8-byte + 16-byte + 4-byte load (the overlap between
// neighbouring lanes is served by the vector L1), the horizontal results slide down the strip in a
// 5-row register window, and there is no barrier and no index division anywhere.  Requires w % 4 == 0
// (16-byte aligned column groups); arithmetic and operation order are those of pyrdown_tile.
constexpr int PD_THREADS = 256;
template <int TU>
__global__ __launch_bounds__(PD_THREADS) void k_pyr_down_rows(const float* __restrict__ src, int w, int h,
                                                              float* __restrict__ dst, int dw, int dh,
                                                              int strips_x, int strips_y, int ntasks, int rows) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int task = blockIdx.x * (PD_THREADS / 64) + wave;
    if (task >= ntasks) return;
    const int pl = task / (strips_x * strips_y);
    const int r = task - pl * (strips_x * strips_y);
    const int ty = r / strips_x, tx = r - ty * strips_x;
    const int ox = tx * 128 + 2 * lane, oy0 = ty * rows;
    if (ox >= dw) return;
    const float* sp = src + (size_t)pl * ((size_t)w * h);
    float* dp = dst + (size_t)pl * ((size_t)dw * dh);
    const int c = 2 * ox;                                  // centre column of the first output
    const bool interior = c >= 2 && c + 4 < w;
    // byte offsets of the taps (edge lanes: REFLECT_101 per tap)
    unsigned off[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) off[k] = 4u * (unsigned)reflect101(c - 2 + k, w);
    const unsigned offc = 4u * (unsigned)c;
    auto hpair = [&](int sy, float& ha, float& hb) __attribute__((always_inline)) {
        const char* row = reinterpret_cast<const char*>(sp + (size_t)reflect101(sy, h) * w);
        float t[7];
        if (interior) {
            const float2 a = *reinterpret_cast<const float2*>(row + offc - 8);
            const float4 m = *reinterpret_cast<const float4*>(row + offc);
            t[0] = a.x; t[1] = a.y; t[2] = m.x; t[3] = m.y; t[4] = m.z; t[5] = m.w;
            t[6] = *reinterpret_cast<const float*>(row + offc + 16);
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) t[k] = *reinterpret_cast<const float*>(row + off[k]);
        }
        ha = t[2] * 6.f + (t[1] + t[3]) * 4.f + t[0] + t[4];
        hb = t[4] * 6.f + (t[3] + t[5]) * 4.f + t[2] + t[6];
    };
    const int yend = oy0 + rows < dh ? oy0 + rows : dh;
    float a0, a1, a2, a3, a4, b0, b1, b2, b3, b4;
    hpair(2 * oy0 - 2, a0, b0); hpair(2 * oy0 - 1, a1, b1); hpair(2 * oy0, a2, b2);
    const bool pair_store = (dw & 1) == 0;
    for (int oy = oy0; oy < yend; ++oy) {
        hpair(2 * oy + 1, a3, b3); hpair(2 * oy + 2, a4, b4);
        const float va = (a2 * 6.f + (a1 + a3) * 4.f + a0 + a4) * (1.f / 256.f);
        const float vb = (b2 * 6.f + (b1 + b3) * 4.f + b0 + b4) * (1.f / 256.f);
        float* q = dp + (size_t)oy * dw + ox;
        if (pair_store) *reinterpret_cast<float2*>(q) = make_float2(va, vb);
        else { q[0] = va; if (ox + 1 < dw) q[1] = vb; }
        a0 = a2; a1 = a3; a2 = a4; b0 = b2; b1 = b3; b2 = b4;
    }
}

// ---- three pyramid levels in one launch ------------------------------------------------------
// G_l -> G_{l+1}, G_{l+2}, G_{l+3} (NL = 2 or 3 output levels).  A workgroup owns an 8x8 tile of the
// coarsest output (16x16 / 32x32 of the finer ones) and recomputes the halos it needs: the G_l
// region (<= 85x85) is staged in LDS once, every intermediate level lives only in LDS.  Halo
// positions outside an image are never computed: readers apply REFLECT_101 at that level and the
// mirrored position is always inside the stored (clipped) region.  Replaces NL launches that are
// each launch/latency-bound (6-7 us on MI355X) by one.
constexpr int ML_T = 8;                       // owned tile of the coarsest level
template <int NL>
__global__ __launch_bounds__(256) void k_pyr_down_multi(const float* __restrict__ src, int w0, int h0,
                                                        float* __restrict__ d1, int w1, int h1,
                                                        float* __restrict__ d2, int w2, int h2,
                                                        float* __restrict__ d3, int w3, int h3) {
    constexpr int T3 = ML_T, T2 = (NL == 3) ? 2 * ML_T : ML_T, T1 = 2 * T2;   // owned tile sizes per level
    constexpr int R2 = (NL == 3) ? 2 * T3 + 3 : T2;          // stored extent of level l+2
    constexpr int R1 = 2 * R2 + 3;                           // stored extent of level l+1
    constexpr int R0 = 2 * R1 + 3;                           // staged extent of level l
    __shared__ float s0[R0][R0 + 1];
    __shared__ float ht[R0][R1 + 1];                         // horizontal temporaries (reused per level)
    __shared__ float s1[R1][R1 + 1];
    __shared__ float s2[R2][R2 + 1];
    const int tid = threadIdx.x;
    const size_t pl = blockIdx.z;
    const float* sp = src + pl * ((size_t)w0 * h0);
    // origins (may be negative / beyond the image; stored regions are clipped to the image)
    const int o2x = (NL == 3) ? 2 * (blockIdx.x * T3) - 2 : blockIdx.x * T2, o2y = (NL == 3) ? 2 * (blockIdx.y * T3) - 2 : blockIdx.y * T2;
    const int o1x = 2 * o2x - 2, o1y = 2 * o2y - 2;
    const int o0x = 2 * o1x - 2, o0y = 2 * o1y - 2;
    // stage level l (REFLECT_101 of the source image)
    for (int i = tid; i < R0 * R0; i += 256) {
        const int ly = i / R0, lx = i - ly * R0;
        s0[ly][lx] = sp[(size_t)reflect101(o0y + ly, h0) * w0 + reflect101(o0x + lx, w0)];
    }
    __syncthreads();
    // level l+1: local (ly, lx) <-> image (o1y + ly, o1x + lx); source local = 2*l + {0..4}
    for (int i = tid; i < R0 * R1; i += 256) {
        const int ly = i / R1, x = i - ly * R1;
        const float* s = &s0[ly][2 * x];
        ht[ly][x] = s[2] * 6.f + (s[1] + s[3]) * 4.f + s[0] + s[4];
    }
    __syncthreads();
    for (int i = tid; i < R1 * R1; i += 256) {
        const int y = i / R1, x = i - y * R1;
        const float v = (ht[2 * y + 2][x] * 6.f + (ht[2 * y + 1][x] + ht[2 * y + 3][x]) * 4.f + ht[2 * y][x] + ht[2 * y + 4][x]) * (1.f / 256.f);
        s1[y][x] = v;
        const int gx = o1x + x, gy = o1y + y;
        const int ox = gx - (int)blockIdx.x * T1, oy = gy - (int)blockIdx.y * T1;       // owned part
        if (ox >= 0 && ox < T1 && oy >= 0 && oy < T1 && gx < w1 && gy < h1) d1[pl * ((size_t)w1 * h1) + (size_t)gy * w1 + gx] = v;
    }
    __syncthreads();
    // level l+2 from s1 with REFLECT_101 at level l+1 (mirrored coordinates stay inside the stored region)
    for (int i = tid; i < R1 * R2; i += 256) {
        const int ly = i / R2, x = i - ly * R2;
        const int gx = o2x + x;
        float v = 0.f;
        const int gy1 = o1y + ly;
        if (gx >= 0 && gx < w2 && gy1 >= 0 && gy1 < h1) {
            const float* s = &s1[ly][0];
            const int c0 = reflect101(2 * gx - 2, w1) - o1x, c1 = reflect101(2 * gx - 1, w1) - o1x, c2 = 2 * gx - o1x,
                      c3 = reflect101(2 * gx + 1, w1) - o1x, c4 = reflect101(2 * gx + 2, w1) - o1x;
            v = s[c2] * 6.f + (s[c1] + s[c3]) * 4.f + s[c0] + s[c4];
        }
        ht[ly][x] = v;
    }
    __syncthreads();
    for (int i = tid; i < R2 * R2; i += 256) {
        const int y = i / R2, x = i - y * R2;
        const int gx = o2x + x, gy = o2y + y;
        float v = 0.f;
        if (gx >= 0 && gx < w2 && gy >= 0 && gy < h2) {
            const int r0 = reflect101(2 * gy - 2, h1) - o1y, r1 = reflect101(2 * gy - 1, h1) - o1y, r2 = 2 * gy - o1y,
                      r3 = reflect101(2 * gy + 1, h1) - o1y, r4 = reflect101(2 * gy + 2, h1) - o1y;
            v = (ht[r2][x] * 6.f + (ht[r1][x] + ht[r3][x]) * 4.f + ht[r0][x] + ht[r4][x]) * (1.f / 256.f);
            const int ox = gx - (int)blockIdx.x * T2, oy = gy - (int)blockIdx.y * T2;
            if (ox >= 0 && ox < T2 && oy >= 0 && oy < T2) d2[pl * ((size_t)w2 * h2) + (size_t)gy * w2 + gx] = v;
        }
        s2[y][x] = v;
    }
    if (NL == 3) {
        __syncthreads();
        for (int i = tid; i < R2 * T3; i += 256) {
            const int ly = i / T3, x = i - ly * T3;
            const int gx = blockIdx.x * T3 + x, gy2 = o2y + ly;
            float v = 0.f;
            if (gx < w3 && gy2 >= 0 && gy2 < h2) {
                const float* s = &s2[ly][0];
                const int c0 = reflect101(2 * gx - 2, w2) - o2x, c1 = reflect101(2 * gx - 1, w2) - o2x, c2 = 2 * gx - o2x,
                          c3 = reflect101(2 * gx + 1, w2) - o2x, c4 = reflect101(2 * gx + 2, w2) - o2x;
                v = s[c2] * 6.f + (s[c1] + s[c3]) * 4.f + s[c0] + s[c4];
            }
            ht[ly][x] = v;
        }
        __syncthreads();
        for (int i = tid; i < T3 * T3; i += 256) {
            const int y = i / T3, x = i - y * T3;
            const int gx = blockIdx.x * T3 + x, gy = blockIdx.y * T3 + y;
            if (gx < w3 && gy < h3) {
                const int r0 = reflect101(2 * gy - 2, h2) - o2y, r1 = reflect101(2 * gy - 1, h2) - o2y, r2 = 2 * gy - o2y,
                          r3 = reflect101(2 * gy + 1, h2) - o2y, r4 = reflect101(2 * gy + 2, h2) - o2y;
                d3[pl * ((size_t)w3 * h3) + (size_t)gy * w3 + gx] =
                    (ht[r2][x] * 6.f + (ht[r1][x] + ht[r3][x]) * 4.f + ht[r0][x] + ht[r4][x]) * (1.f / 256.f);
            }
        }
    }
}

// generic pyrUp of float planes (dsize = dw x dh, 2n or 2n-1); blockIdx.z = plane
template <int TU>
__global__ __launch_bounds__(256) void k_pyr_up(const float* __restrict__ src, int sw, int sh,
                                                float* __restrict__ dst, int dw, int dh) {
    __shared__ float s_g[US_H][US_W + 1];
    __shared__ float h_g[US_H][UT_W + 1];
    const int x0 = blockIdx.x * UT_W, y0 = blockIdx.y * UT_H;
    const int sx0 = x0 / 2 - 1, sy0 = y0 / 2 - 1;
    pyrup_stage(s_g, src + (size_t)blockIdx.z * sw * sh, sw, sh, sx0, sy0);
    __syncthreads();
    pyrup_hpass(h_g, s_g, x0, sx0, sw, dw);
    __syncthreads();
    for (int i = threadIdx.x; i < UT_H * UT_W; i += 256) {
        const int y = i / UT_W, x = i - y * UT_W;
        const int gx = x0 + x, gy = y0 + y;
        if (gx < dw && gy < dh) dst[(size_t)blockIdx.z * dw * dh + (size_t)gy * dw + gx] = pyrup_v(h_g, x, gy, sy0);
    }
}

}  // namespace lvm
