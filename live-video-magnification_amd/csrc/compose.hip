// compose.hip -- the export hand-off on the device.
//
// Replaces the pane composition of Exporter::compose (reference: export/Exporter.cpp:22-34 toBgr, :53-88 compose,
// export/ExportTypes.hpp:11 SplitMode): both frames widened to BGR (gray -> b = g = r), cropped to their common EVEN
// size, and written next to each other (LeftRight), above each other (TopBottom) or alone (None) into one BGR canvas.
// Inputs are the magnifier's device output and the device copy of the original, so an export that encodes on the
// device -- or downloads ONE canvas instead of two frames -- needs no second PCIe crossing of the original.
// Byte work, bit-exact against the oracle.  The text overlay (cv::putText with anti-aliased Hershey strokes,
// Exporter.cpp:36-50) is not restated: a caller that wants labels draws them on the downloaded canvas as before.
#include "lvm_internal.h"

namespace lvm {

struct ComposeArgs {
    const uint8_t* src[2]; long stride[2], sstride[2]; int cn[2];   // [0] original, [1] processed
    uint8_t* dst; long dst_stride, dst_sstride;
    int w, h;                 // pane size (common even size)
    int npanes, dx1, dy1;     // pane 1 (processed) origin on the canvas; with one pane only src[1] is used
};

// One thread per group of 4 canvas pixels of one pane row (12 output bytes as three dwords when the canvas row
// allows it, bytes otherwise); blockIdx.z = stream * npanes + pane.
template <bool VEC>
__global__ __launch_bounds__(256) void k_compose(ComposeArgs a) {
    const int gx = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (gx >= a.w || y >= a.h) return;
    const int pane = a.npanes == 2 ? (int)(blockIdx.z & 1) : 1, stream = a.npanes == 2 ? (int)(blockIdx.z >> 1) : (int)blockIdx.z;
    const uint8_t* p = a.src[pane] + (size_t)stream * a.sstride[pane] + (size_t)y * a.stride[pane];
    const int ox = pane ? a.dx1 : 0, oy = pane ? a.dy1 : 0;
    uint8_t* q = a.dst + (size_t)stream * a.dst_sstride + (size_t)(oy + y) * a.dst_stride + (size_t)(ox + gx) * 3;
    const int n = a.w - gx < 4 ? a.w - gx : 4;
    uint8_t v[12];
    if (a.cn[pane] == 3) {
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = k < 3 * n ? p[(size_t)gx * 3 + k] : 0;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint8_t g = k < n ? p[gx + k] : 0; v[3 * k] = g; v[3 * k + 1] = g; v[3 * k + 2] = g; }   // COLOR_GRAY2BGR
    }
    if (VEC && n == 4) {
        uint32_t* q4 = reinterpret_cast<uint32_t*>(q);
        q4[0] = v[0] | (v[1] << 8) | (v[2] << 16) | ((uint32_t)v[3] << 24);
        q4[1] = v[4] | (v[5] << 8) | (v[6] << 16) | ((uint32_t)v[7] << 24);
        q4[2] = v[8] | (v[9] << 8) | (v[10] << 16) | ((uint32_t)v[11] << 24);
    } else {
        for (int k = 0; k < 3 * n; ++k) q[k] = v[k];
    }
}

// Exporter.cpp:55-66: canvas size and pane size for the two frame sizes
int compose_geometry(int split, int ow, int oh, int pw, int ph, int* pane_w, int* pane_h, int* canvas_w, int* canvas_h) {
    int w, h;
    if (split == LVM_SPLIT_NONE) { w = pw & ~1; h = ph & ~1; }                         // :56
    else { w = (ow < pw ? ow : pw) & ~1; h = (oh < ph ? oh : ph) & ~1; }               // :63-64
    if (w <= 0 || h <= 0) { *pane_w = *pane_h = *canvas_w = *canvas_h = 0; return 0; }   // :57, :65 (empty Mat)
    *pane_w = w; *pane_h = h;
    *canvas_w = split == LVM_SPLIT_LEFT_RIGHT ? 2 * w : w;                             // :71
    *canvas_h = split == LVM_SPLIT_TOP_BOTTOM ? 2 * h : h;                             // :79
    return 1;
}

int compose_device(Ctx* c, int split, const uint8_t* d_orig, int ow, int oh, int och, ptrdiff_t ostride, ptrdiff_t osstride,
                   const uint8_t* d_proc, int pw, int ph, int pch, ptrdiff_t pstride, ptrdiff_t psstride, uint8_t* d_canvas,
                   ptrdiff_t cstride, ptrdiff_t csstride, hipStream_t s) {
    if (split < LVM_SPLIT_NONE || split > LVM_SPLIT_TOP_BOTTOM) { c->err = "invalid split mode"; return LVM_ERR_INVALID; }
    if (!d_proc || !d_canvas || (pch != 1 && pch != 3)) { c->err = "bad processed frame"; return LVM_ERR_INVALID; }
    if (split != LVM_SPLIT_NONE && !d_orig) { d_orig = d_proc; ow = pw; oh = ph; och = pch; ostride = pstride; osstride = psstride; }   // :62 fall back
    if (split != LVM_SPLIT_NONE && och != 1 && och != 3) { c->err = "bad original frame"; return LVM_ERR_INVALID; }
    int w, h, cw, chh;
    if (!compose_geometry(split, ow, oh, pw, ph, &w, &h, &cw, &chh)) return LVM_OK;    // empty canvas: nothing to write
    if (cstride < (ptrdiff_t)cw * 3) { c->err = "canvas stride too small"; return LVM_ERR_INVALID; }
    // the kernel indexes the frames with the caller's strides: rows must hold their pixels, streams must not overlap
    if (pstride < (ptrdiff_t)pw * pch || (split != LVM_SPLIT_NONE && ostride < (ptrdiff_t)ow * och)) { c->err = "frame stride too small"; return LVM_ERR_INVALID; }
    if (c->nstreams > 1 && (csstride < (ptrdiff_t)chh * cstride || psstride < (ptrdiff_t)ph * pstride ||
                            (split != LVM_SPLIT_NONE && osstride < (ptrdiff_t)oh * ostride))) { c->err = "stream stride too small"; return LVM_ERR_INVALID; }
    ComposeArgs a;
    a.src[0] = d_orig; a.stride[0] = ostride; a.sstride[0] = osstride; a.cn[0] = och;
    a.src[1] = d_proc; a.stride[1] = pstride; a.sstride[1] = psstride; a.cn[1] = pch;
    a.dst = d_canvas; a.dst_stride = cstride; a.dst_sstride = csstride;
    a.w = w; a.h = h;
    a.npanes = split == LVM_SPLIT_NONE ? 1 : 2;
    a.dx1 = split == LVM_SPLIT_LEFT_RIGHT ? w : 0;
    a.dy1 = split == LVM_SPLIT_TOP_BOTTOM ? h : 0;
    if (split == LVM_SPLIT_NONE) { a.dx1 = 0; a.dy1 = 0; }
    // dword stores need every pane row segment dword-aligned: canvas base, strides and the pane-1 column offset (3 w bytes)
    const bool vec = ((uintptr_t)d_canvas % 4) == 0 && cstride % 4 == 0 && csstride % 4 == 0 && (a.dx1 * 3) % 4 == 0;
    const dim3 grid(((w + 3) / 4 + 63) / 64, (h + 3) / 4, (unsigned)(c->nstreams * a.npanes)), blk(256);
    if (vec) LVM_LAUNCH(c, "compose", k_compose<true>, grid, blk, s, a);
    else LVM_LAUNCH(c, "compose", k_compose<false>, grid, blk, s, a);
    LVM_HIP_TRY(c, hipGetLastError());
    return LVM_OK;
}

}  // namespace lvm
