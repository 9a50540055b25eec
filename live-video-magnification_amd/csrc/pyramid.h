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
template <int C, bool LAB, bool EXACT>
__global__ __launch_bounds__(256) void k_down0(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                               int w, int h, float* __restrict__ G1, int w1, int h1,
                                               LabCoef lab, float scale) {
    __shared__ float s_src[C][DS_H][DS_W];
    __shared__ float s_row[C][DS_H][DT_W];
    __shared__ float s_gam[LAB ? 256 : 1];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    if (LAB) { load_gamma_u8(s_gam, lab.gamma_u8); __syncthreads(); }
    const int ox0 = blockIdx.x * DT_W, oy0 = blockIdx.y * DT_H;
    const uint8_t* src = in + (size_t)b * in_sstride;
    for (int i = tid; i < DS_H * DS_W; i += 256) {
        const int ly = i / DS_W, lx = i - ly * DS_W;
        const int gy = reflect101(2 * oy0 - 2 + ly, h), gx = reflect101(2 * ox0 - 2 + lx, w);
        const uint8_t* p = src + (size_t)gy * in_stride + (size_t)gx * C;
        if (LAB) {
            float L, a, bb;
            lin_bgr_to_lab<EXACT>(s_gam[p[0]], s_gam[LAB ? p[1] : 0], s_gam[LAB ? p[2] : 0], lab.fwd, L, a, bb);
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
constexpr int UT_W = 64, UT_H = 16;
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
    if ((gy & 1) == 0) return (hrow[lj - 1][x] + hrow[lj][x] * 6.f + hrow[lj + 1][x]) * (1.f / 64.f);
    return ((hrow[lj][x] + hrow[lj + 1][x]) * 4.f) * (1.f / 64.f);
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
