// compose.hip -- the export hand-off on the device.
//
// Replaces the pane composition of Exporter::compose (reference: export/Exporter.cpp:22-34 toBgr, :53-88 compose,
// export/ExportTypes.hpp:11 SplitMode): both frames widened to BGR (gray -> b = g = r), cropped to their common EVEN
// size, and written next to each other (LeftRight), above each other (TopBottom) or alone (None) into one BGR canvas.
// Inputs are the magnifier's device output and the device copy of the original, so an export that encodes on the
// device -- or downloads ONE canvas instead of two frames -- needs no second PCIe crossing of the original.
// Byte work, bit-exact against the oracle.
//
// The text overlay (drawLabel, Exporter.cpp:36-50: addWeighted(roi, 0.35, black, 0.65) + cv::putText(LINE_AA, white)) is NOT restated
// -- OpenCV's anti-aliased Hershey strokes are not something to reproduce from memory -- and need not be: both steps read-modify-write
// single pixels, so what a label does to a canvas pixel is a FUNCTION OF THAT PIXEL'S BYTE, the same for B, G and R (white on black),
// fixed for the whole export (the label depends on the canvas size only).  Round 6: the reference-side shim renders the label once per
// export with the reference's own calls onto 256 constant canvases, which yields that function for every pixel of the label's rectangle
// exactly, whatever the OpenCV build does; pixels with the same function share a class.  k_overlay_labels applies the tables to the
// composed canvases where they lie: exact by construction, and an export with `textOverlay` stays on the device (and on the MJPEG path).
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

// ---- the text overlay as per-pixel tables (lvm_export_set_overlay) ----------------------------------------------------------------
constexpr int kMaxOverlayLabels = 4;
struct OverlayArgs {
    int n; int x[kMaxOverlayLabels], y[kMaxOverlayLabels], w[kMaxOverlayLabels], h[kMaxOverlayLabels];
    int first[kMaxOverlayLabels + 1];                   // prefix sums of w * h: thread index -> label
    const uint16_t* cls[kMaxOverlayLabels]; const uint8_t* fn[kMaxOverlayLabels];
    uint8_t* canvas; long stride, fstride;
};
// one thread per label pixel, blockIdx.y = frame: the three bytes of the pixel through the pixel's table
__global__ __launch_bounds__(256) void k_overlay_labels(OverlayArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.first[a.n]) return;
    int l = 0;
    while (l + 1 < a.n && i >= a.first[l + 1]) ++l;
    const int k = i - a.first[l], ly = k / a.w[l], lx = k - ly * a.w[l];
    const uint8_t* f = a.fn[l] + (size_t)a.cls[l][k] * 256;
    uint8_t* q = a.canvas + (size_t)blockIdx.y * a.fstride + (size_t)(a.y[l] + ly) * a.stride + (size_t)(a.x[l] + lx) * 3;
    q[0] = f[q[0]]; q[1] = f[q[1]]; q[2] = f[q[2]];
}

void overlay_release(Ctx* c) {
    if (c->d_overlay) (void)hipFree(c->d_overlay);
    c->d_overlay = nullptr; c->overlay_n = 0;
}

int overlay_set(Ctx* c, int n, const lvm_overlay_label* labels) {
    sync_streams(c);                                    // (a running export may still read the old tables)
    (void)hipStreamSynchronize(c->own_stream);
    overlay_release(c);
    if (n == 0) return LVM_OK;
    if (n < 0 || n > kMaxOverlayLabels || !labels) { c->err = "overlay: 0..4 labels"; return LVM_ERR_INVALID; }
    size_t total = 0;
    for (int l = 0; l < n; ++l) {
        const lvm_overlay_label& L = labels[l];
        if (L.w < 1 || L.h < 1 || L.x < 0 || L.y < 0 || L.n_classes < 1 || L.n_classes > 65536 || !L.cls || !L.fn || (long)L.w * L.h > (1L << 24)) { c->err = "overlay: bad label"; return LVM_ERR_INVALID; }
        for (long k = 0; k < (long)L.w * L.h; ++k) if (L.cls[k] >= L.n_classes) { c->err = "overlay: class index outside the label's tables"; return LVM_ERR_INVALID; }
        total += (((size_t)L.w * L.h * 2 + 255) & ~(size_t)255) + (size_t)L.n_classes * 256;
    }
    std::vector<uint8_t> host(total);
    LVM_HIP_TRY(c, hipMalloc((void**)&c->d_overlay, total));
    size_t off = 0;
    for (int l = 0; l < n; ++l) {
        const lvm_overlay_label& L = labels[l];
        c->ov_x[l] = L.x; c->ov_y[l] = L.y; c->ov_w[l] = L.w; c->ov_h[l] = L.h;
        c->ov_cls[l] = off; std::memcpy(host.data() + off, L.cls, (size_t)L.w * L.h * 2); off += ((size_t)L.w * L.h * 2 + 255) & ~(size_t)255;
        c->ov_fn[l] = off; std::memcpy(host.data() + off, L.fn, (size_t)L.n_classes * 256); off += (size_t)L.n_classes * 256;
    }
    if (hipMemcpy(c->d_overlay, host.data(), total, hipMemcpyHostToDevice) != hipSuccess) { overlay_release(c); c->err = "overlay: upload failed"; return LVM_ERR_HIP; }
    c->overlay_n = n;
    return LVM_OK;
}

// the labels onto n_frames canvases of cw x chh (nothing to do without labels)
int overlay_device(Ctx* c, uint8_t* d_canvas, int cw, int chh, ptrdiff_t stride, ptrdiff_t fstride, int n_frames, hipStream_t s) {
    if (c->overlay_n == 0 || n_frames < 1) return LVM_OK;
    if (!d_canvas || stride < (ptrdiff_t)cw * 3 || (n_frames > 1 && fstride < (ptrdiff_t)chh * stride)) { c->err = "overlay: bad canvas arguments"; return LVM_ERR_INVALID; }
    OverlayArgs a{};
    a.n = c->overlay_n; a.first[0] = 0;
    for (int l = 0; l < a.n; ++l) {
        // drawLabel clips its rectangle to the canvas BEFORE it draws (Exporter.cpp:44): the tables describe the clipped rectangle of ONE
        // canvas size -- a label that does not fit this canvas was rendered for another geometry
        if (c->ov_x[l] + c->ov_w[l] > cw || c->ov_y[l] + c->ov_h[l] > chh) { c->err = "overlay: a label lies outside the canvas (rendered for another canvas size?)"; return LVM_ERR_INVALID; }
        a.x[l] = c->ov_x[l]; a.y[l] = c->ov_y[l]; a.w[l] = c->ov_w[l]; a.h[l] = c->ov_h[l];
        a.first[l + 1] = a.first[l] + a.w[l] * a.h[l];
        a.cls[l] = reinterpret_cast<const uint16_t*>(c->d_overlay + c->ov_cls[l]); a.fn[l] = c->d_overlay + c->ov_fn[l];
    }
    a.canvas = d_canvas; a.stride = (long)stride; a.fstride = (long)fstride;
    LVM_LAUNCH(c, "overlay", k_overlay_labels, dim3((unsigned)((a.first[a.n] + 255) / 256), (unsigned)n_frames), dim3(256), s, a);
    LVM_HIP_TRY(c, hipGetLastError());
    return LVM_OK;
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
