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
// (checked exhaustively by tests/test_lab_lut.py and by lab_tables.cpp whenever a context is created): cell index and
// weight are two bit fields of one v_mad_u32_u24, no table look-up and no float operation.  Everything after that is
// integer arithmetic, so the result is BIT-EXACT against the oracle's restatement (oracle/lvm_oracle.c bgr2lab_lut_px).
//
// Where the table lives on MI355X (measured: tools/ubench_lut.hip, tools/probe_gather.hip, profiles/README.md round 3).
// A pixel needs 8 corners x 3 channels = 48 bytes of table.  Gathered from L2 (OpenCV's own layout: 8 replicated corners per
// cell, 1.7 MB) the conversion costs 10 us per 1080p frame -- a random L2-resident gather is ~140 cycles per wave
// instruction on a CU, whatever its width -- against 3 us for the analytic form.  The 160 KB LDS cannot hold three
// channels (215 KB as int16) but it does hold TWO as one dword per grid node, (a | b << 16), 144 KB, one 1024-thread
// workgroup per CU: 8 random ds_read_b32 per pixel; L comes from an L2-resident table of CELLS (8 corners x int16 = ONE
// 16-byte gather per pixel, 575 KB).  LDS pipe, texture-address pipe and VALU then carry about a third of the work each:
// 5.5 us per 1080p frame.  v_perm_b32 turns two B-neighbour node dwords into the (a[r], a[r+1]) / (b[r], b[r+1]) pairs that
// v_dot2_i32_i16 folds with the packed weights (w (16 - z), w z).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <lvm_gfx950.h>

namespace lvm {

constexpr int kU8StepSlices = 4096;       // slices of the output quantiser's step table (lab_tables.cpp build_u8_steps)
constexpr int kLabLutDim = 33;
constexpr int kLabLutNodes = kLabLutDim * kLabLutDim * kLabLutDim;
// node index n = p + 33 q + 1089 r (p, q, r = R, G, B grid index).  Neighbours n + 1, n + 33, n + 34 of an edge cell carry
// weight 0 and may index past the cube: the node table is padded by 34 entries (the B neighbour is clamped instead).
// Device layouts (round 4: two instruction diets of the conversion, ~94 -> ~89 vector instructions per pixel):
//  * every entry is stored as 2 v + 1 (uint16): sum w (2 v + 1) = 2 sum w v + 4096 (the eight weights always add up to 16^3), so
//    CV_DESCALE(sum w v, 12) = (sum w v + 2048) >> 12 = (sum w (2 v + 1)) >> 13 -- the rounding constant rides in the table and the
//    accumulator chains start from the inline constant 0 (three v_mov per pixel less); unsigned dot products (2 v + 1 <= 32769);
//  * the B neighbour of a node is ALWAYS n + 1089: for u = 255 (cell 32) its weight is 0 and the node table is padded by one more
//    B plane of zeros instead of clamping the index (compare + select + add -> add);
//  * (measured and NOT taken: the L cell indexed by its origin node n -- the index the (a, b) look-up needs anyway, ~10 instructions per
//    pixel less than the 2 x 2 x 2-blocked cell number -- made the fused first kernel SLOWER, 247 -> 297 us per 32 frames, and the
//    conversion kernel 239 -> 270: the kernels are bound by the L gather's L1 misses, not by vector issue; LVM_LUT_LCELL_BLOCKED=0 builds it)
#ifndef LVM_LUT_LCELL_BLOCKED
#define LVM_LUT_LCELL_BLOCKED 1
#endif
constexpr int kLabAbWords = kLabLutNodes + 1089 + 34 + 1;
constexpr int kLabLCells = LVM_LUT_LCELL_BLOCKED ? 17 * 17 * 17 * 8 : kLabLutNodes;       // cells in 2 x 2 x 2 blocks (lcell_index) | by origin node

// lut_dot2 / lut_lo2 / lut_hi2 / lut_mul24 (v_dot2_u32_u16, v_perm_b32, v_mul_u32_u24): lvm_gfx950.h

// fine grid coordinate of a u8 channel value: bits 4.. = cell, bits 0..3 = weight of the upper neighbour
__device__ __forceinline__ uint32_t lut_fine(uint32_t u) { return (u * 514u + 4u) >> 8; }

// device tables of a context (lab_tables.cpp builds them from the compact table)
struct LabLut {
    const uint32_t* ab;       // [kLabAbWords]  (2 a + 1) | (2 b + 1) << 16 per node (copied into LDS by the conversion kernel)
    const uint4* Lcells;      // [kLabLCells]   the 8 L corners of a cell as 2 L + 1 (uint16 index 4 dp + 2 dq + dr), by origin node
};

// integer Lab of one pixel: iL in [0, 16384], ia, ib = (a + 128) / 256 * 16384.  s_ab = the node table in LDS.
// Two halves, so that a caller can have the look-ups of the NEXT pixel in flight while this one is interpolated: lut_issue computes
// the addresses and weights and issues the nine reads (one 16-byte gather from L2, four ds_read2_b32), lut_finish interpolates.
struct LutRefs { uint4 cL; uint32_t d00, d10, d01, d11, e00, e10, e01, e11, w00, w10, w01, w11; };
__device__ __forceinline__ LutRefs lut_issue(uint32_t B, uint32_t G, uint32_t R, const uint32_t* s_ab, const uint4* __restrict__ Lcells) {
    LutRefs q;
    const uint32_t fr = lut_fine(R), fg = lut_fine(G), fb = lut_fine(B);
    const uint32_t x = fr & 15u, y = fg & 15u, z = fb & 15u, tb = fb >> 4;
    const uint32_t n = (fr >> 4) + 33u * (fg >> 4) + 1089u * tb;
    const uint32_t n1 = n + 1089u;                               // B neighbour (tb == 32 only for u = 255, where z == 0: the padding plane)
    // the L cell: one 16-byte gather at a 32-bit byte offset from the uniform base
#if LVM_LUT_LCELL_BLOCKED
    // cells in 2 x 2 x 2 blocks (one 128-byte line each): neighbouring colours share lines, the gathers hit the CU's L1 more often
    const uint32_t nc = (((fr >> 5) + 17u * (fg >> 5) + 289u * (fb >> 5)) << 3) | ((fr >> 4) & 1u) | ((fg >> 3) & 2u) | ((fb >> 2) & 4u);
#else
    const uint32_t nc = n;
#endif
    q.cL = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Lcells) + (nc << 4));
    q.d00 = s_ab[n]; q.d10 = s_ab[n + 1]; q.d01 = s_ab[n + 33]; q.d11 = s_ab[n + 34];
    q.e00 = s_ab[n1]; q.e10 = s_ab[n1 + 1]; q.e01 = s_ab[n1 + 33]; q.e11 = s_ab[n1 + 34];
    const uint32_t wz = (16u - z) | (z << 16);                 // (16 - z, z) as an int16 pair
    const uint32_t x0 = 16u - x, y0 = 16u - y;
    // w(dp, dq) * (16 - z | z): each half <= 4096, no carry between the halves
    q.w00 = lut_mul24(lut_mul24(x0, y0), wz); q.w10 = lut_mul24(lut_mul24(x, y0), wz); q.w01 = lut_mul24(lut_mul24(x0, y), wz);
    q.w11 = lut_mul24(lut_mul24(x, y), wz);
    return q;
}
__device__ __forceinline__ void lut_finish(const LutRefs& q, int& iL, int& ia, int& ib) {
    // entries are 2 v + 1: (sum w (2 v + 1)) >> 13 == CV_DESCALE(sum w v, 12)
    ia = (int)(lut_dot2(lut_lo2(q.d11, q.e11), q.w11, lut_dot2(lut_lo2(q.d01, q.e01), q.w01, lut_dot2(lut_lo2(q.d10, q.e10), q.w10, lut_dot2(lut_lo2(q.d00, q.e00), q.w00, 0u)))) >> 13);
    ib = (int)(lut_dot2(lut_hi2(q.d11, q.e11), q.w11, lut_dot2(lut_hi2(q.d01, q.e01), q.w01, lut_dot2(lut_hi2(q.d10, q.e10), q.w10, lut_dot2(lut_hi2(q.d00, q.e00), q.w00, 0u)))) >> 13);
    // cell dwords: x = (dp 0, dq 0), y = (0, 1), z = (1, 0), w = (1, 1), each the (dr 0, dr 1) pair
    iL = (int)(lut_dot2(q.cL.w, q.w11, lut_dot2(q.cL.y, q.w01, lut_dot2(q.cL.z, q.w10, lut_dot2(q.cL.x, q.w00, 0u)))) >> 13);
}
__device__ __forceinline__ void lut_lab_int(uint32_t B, uint32_t G, uint32_t R, const uint32_t* s_ab, const uint4* __restrict__ Lcells,
                                            int& iL, int& ia, int& ib) {
    const LutRefs q = lut_issue(B, G, R, s_ab, Lcells);
    lut_finish(q, iL, ia, ib);
}
// the float values RGB2Labfloat stores (each product is exact in float32: a power-of-two scale, 25 * iL < 2^24)
__device__ __forceinline__ float lut_L(int iL) { return (float)iL * (100.0f / 16384.0f); }
__device__ __forceinline__ float lut_ab(int i) { return (float)i * (1.0f / 64.0f) - 128.0f; }

// ---- integer Lab planes: what the conversion kernel (labconv.hip) hands to the kernels that used to convert ------------
// iL: uint16 per pixel; iab: ia | ib << 16 per pixel; one plane pair per frame, frame stride = w * h elements.
__device__ __forceinline__ void lab_from_planes(uint32_t iL, uint32_t iab, float& L, float& a, float& b) {
    L = lut_L((int)iL); a = lut_ab((int)(iab & 0xffffu)); b = lut_ab((int)(iab >> 16));
}

}  // namespace lvm
