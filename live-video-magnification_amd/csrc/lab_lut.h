// lab_lut.h -- OpenCV 4's DEFAULT forward float BGR -> Lab: the trilinear-interpolated 33^3 int16 table.
//
// Reference: MagnifyCore.hpp:90 and :219 call cv::cvtColor(COLOR_BGR2Lab) on a CV_32FC3 image in [0, 1].  In OpenCV 4
// (imgproc/src/color_lab.cpp) that call runs RGB2Labfloat with useInterpolation = true (sRGB, default coefficients and
// white point): every channel is rounded to 1/16384 (LAB_BASE), the cell of a 33 x 33 x 33 int16 table is found from the
// top 5 bits, the next 4 bits are the trilinear weights (trilinearInterpolate: sum of 8 corners x weight products,
// CV_DESCALE by 12 bits), and the integers become floats as L = iL * 100 / 16384, a = ia * 256 / 16384 - 128 (b alike).
// The analytic cube-root form is what OpenCV computes only when that interpolation is disabled.
//
// This stage feeds u8 frames scaled by float(1/255) (convertTo, MagnifyCore.hpp:89,218), so a channel takes 256 values and
// its rounded 14-bit value c = cvRound(float(u) * a255 * 16384) satisfies  c >> 5 == (514 u + 4) >> 8  for every u
// (checked exhaustively by tests/test_lab_lut.py and by lab_tables.cpp at start-up): cell index and weight are two bit
// fields of one v_mad_u32_u24, no table look-up and no float operation.  Everything after that is integer arithmetic,
// so the result is BIT-EXACT against the oracle's restatement (oracle/lvm_oracle.c bgr2lab_lut_px).
//
// Table layout in HBM / L2 (built by lab_tables.cpp, 575 KB, resident in every XCD's 4 MB L2): one 16-byte NODE per grid
// point (p = R index fastest, q = G, r = B slowest) holding the B-direction pairs of the three channels,
//     { L[r], L[r+1], a[r], a[r+1], b[r], b[r+1], 0, 0 }   (int16 each; r + 1 clamped to 32),
// so that one v_dot2_i32_i16 with the packed weights (w * (16 - z), w * z) folds a pair.  A pixel reads the 4 nodes
// (p + dx, q + dy, r): two 32-byte runs.  OpenCV's own layout (8 replicated corners per CELL, 3 x 16 bytes per pixel,
// 1.7 MB) was measured too (tools/ubench_lut.hip, profiles/README.md).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lvm {

constexpr int kLabLutDim = 33;
constexpr int kLabLutNodes = kLabLutDim * kLabLutDim * kLabLutDim;
// nodes p + 1 / q + 1 of an edge cell (weight 0) index past the cube: the allocation is padded by 35 nodes
constexpr int kLabLutNodesPadded = kLabLutNodes + 35;

#ifndef LVM_EMU_NO_DOT2
typedef short lut_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int lut_dot2(uint32_t pair, uint32_t wts, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(lut_s2, pair), __builtin_bit_cast(lut_s2, wts), acc, false);
}
#else           // tests/emu (g++): the same sum spelled out
__device__ __forceinline__ int lut_dot2(uint32_t pair, uint32_t wts, int acc) {
    return acc + (int)(int16_t)(pair & 0xffff) * (int)(int16_t)(wts & 0xffff) + (int)(int16_t)(pair >> 16) * (int)(int16_t)(wts >> 16);
}
#endif

// fine grid coordinate of a u8 channel value: bits 4.. = cell, bits 0..3 = weight of the upper neighbour
__device__ __forceinline__ uint32_t lut_fine(uint32_t u) { return (u * 514u + 4u) >> 8; }

// integer Lab of one pixel: iL in [0, 16384], ia, ib = (a + 128) / 256 * 16384
__device__ __forceinline__ void lut_lab_int(uint32_t B, uint32_t G, uint32_t R, const uint4* __restrict__ nodes, int& iL, int& ia, int& ib) {
    const uint32_t fr = lut_fine(R), fg = lut_fine(G), fb = lut_fine(B);
    const uint32_t x = fr & 15u, y = fg & 15u, z = fb & 15u;
    const uint32_t n = (fr >> 4) + 33u * (fg >> 4) + 1089u * (fb >> 4);
    const uint4 n00 = nodes[n], n10 = nodes[n + 1], n01 = nodes[n + 33], n11 = nodes[n + 34];
    const uint32_t wz = (16u - z) | (z << 16);                 // (16 - z, z) as an int16 pair
    const uint32_t x0 = 16u - x, y0 = 16u - y;
    // w(dx, dy) * (16 - z | z): each half <= 4096, no carry between the halves
    const uint32_t w00 = (x0 * y0) * wz, w10 = (x * y0) * wz, w01 = (x0 * y) * wz, w11 = (x * y) * wz;
    const int rnd = 1 << 11;                                  // CV_DESCALE(v, 12) = (v + 2048) >> 12
    iL = lut_dot2(n11.x, w11, lut_dot2(n01.x, w01, lut_dot2(n10.x, w10, lut_dot2(n00.x, w00, rnd)))) >> 12;
    ia = lut_dot2(n11.y, w11, lut_dot2(n01.y, w01, lut_dot2(n10.y, w10, lut_dot2(n00.y, w00, rnd)))) >> 12;
    ib = lut_dot2(n11.z, w11, lut_dot2(n01.z, w01, lut_dot2(n10.z, w10, lut_dot2(n00.z, w00, rnd)))) >> 12;
}
// the float values RGB2Labfloat stores (each product is exact in float32: a power-of-two scale, 25 * iL < 2^24)
__device__ __forceinline__ float lut_L(int iL) { return (float)iL * (100.0f / 16384.0f); }
__device__ __forceinline__ float lut_ab(int i) { return (float)i * (1.0f / 64.0f) - 128.0f; }

// the L channel alone (Riesz L plane): the first dword of each node
__device__ __forceinline__ float lut_lab_L(uint32_t B, uint32_t G, uint32_t R, const uint4* __restrict__ nodes) {
    const uint32_t fr = lut_fine(R), fg = lut_fine(G), fb = lut_fine(B);
    const uint32_t x = fr & 15u, y = fg & 15u, z = fb & 15u;
    const uint32_t n = (fr >> 4) + 33u * (fg >> 4) + 1089u * (fb >> 4);
    const uint32_t l00 = nodes[n].x, l10 = nodes[n + 1].x, l01 = nodes[n + 33].x, l11 = nodes[n + 34].x;
    const uint32_t wz = (16u - z) | (z << 16), x0 = 16u - x, y0 = 16u - y;
    const int iL = lut_dot2(l11, (x * y) * wz, lut_dot2(l01, (x0 * y) * wz, lut_dot2(l10, (x * y0) * wz, lut_dot2(l00, (x0 * y0) * wz, 1 << 11)))) >> 12;
    return lut_L(iL);
}

__device__ __forceinline__ void lut_lab(uint32_t B, uint32_t G, uint32_t R, const uint4* __restrict__ nodes, float& L, float& a, float& b) {
    int iL, ia, ib;
    lut_lab_int(B, G, R, nodes, iL, ia, ib);
    L = lut_L(iL); a = lut_ab(ia); b = lut_ab(ib);
}

}  // namespace lvm
