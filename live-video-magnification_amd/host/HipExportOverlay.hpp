// HipExportOverlay.hpp -- the export's text overlay as per-pixel tables, read off the reference's OWN drawing code
// (the reference side of lvm_export_set_overlay; round 6).
//
// Reference: drawLabel (src/export/Exporter.cpp:36-50) darkens the rectangle behind a caption with
// cv::addWeighted(roi, 0.35, black, 0.65) and writes the caption with cv::putText(FONT_HERSHEY_SIMPLEX, white, LINE_AA); compose calls it
// twice per canvas (:74-77 / :82-85).  Both steps read-modify-write single pixels with the same arithmetic for B, G and R, and the label
// depends on the canvas size only.  So instead of restating OpenCV's anti-aliased Hershey strokes, this header lets the reference draw:
//
//     overlay_tables(cw, ch, draw)   calls `draw` -- the reference's own label code -- on 256 canvases of constant value v = 0..255 and
//                                    reads off, for every pixel the drawing touches, the function v -> new byte.  Exact for whatever the
//                                    linked OpenCV does; pixels with equal functions share a class; touched pixels are grouped into at most
//                                    four rectangles (one per caption).
//
// The tables go to lvm_export_set_overlay once per canvas geometry (ExportRunner::set_canvas_drawer does that), the device applies them to
// every composed canvas, and an export with `textOverlay` keeps the device -- and the Motion-JPEG -- path.  No OpenCV in this header: the
// drawer is a callback on a raw BGR canvas (the livim binding in HipExportRunner.hpp wraps a cv::Mat view around it).
//
// What `draw` must be for this to be exact: a function of each pixel's own previous value (no blur, no reads of neighbours), equal for the
// three channels.  overlay_tables CHECKS both on the canvases it draws (throws lvm::Error otherwise): the same drawing applied to a
// two-valued checkerboard canvas must equal the tables applied to it.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "lvm.hpp"

namespace lvm {

struct OverlayLabel {
    int x = 0, y = 0, w = 0, h = 0;
    std::vector<std::uint16_t> cls;          // [h][w]
    std::vector<std::uint8_t> fn;            // [n_classes][256]
    int n_classes() const { return (int)(fn.size() / 256); }
};
// draws the export's labels onto a BGR canvas of cw x ch (row stride in bytes), exactly as the reference's compose does for `overlay == true`
using CanvasDrawer = std::function<void(std::uint8_t* canvas, int cw, int ch, std::ptrdiff_t stride)>;

inline std::vector<OverlayLabel> overlay_tables(int cw, int ch, const CanvasDrawer& draw) {
    if (cw < 1 || ch < 1 || !draw) throw Error(LVM_ERR_INVALID, "overlay_tables: bad arguments");
    const std::size_t row = (std::size_t)cw * 3;
    std::vector<std::uint8_t> canvas(row * ch);
    // pass 1: which pixels does the drawing touch?  (a pixel may keep SOME values -- white text on a white canvas -- so several are probed)
    std::vector<std::uint8_t> touched((std::size_t)cw * ch, 0);
    for (int v : {0, 37, 128, 200, 255}) {
        std::memset(canvas.data(), v, canvas.size());
        draw(canvas.data(), cw, ch, (std::ptrdiff_t)row);
        for (int y = 0; y < ch; ++y)
            for (int x = 0; x < cw; ++x) {
                const std::uint8_t* p = canvas.data() + (std::size_t)y * row + (std::size_t)x * 3;
                if (p[0] != v || p[1] != v || p[2] != v) touched[(std::size_t)y * cw + x] = 1;
            }
    }
    // rectangles: runs of touched columns (gaps under 8 columns bridged), each with its own row range, then the same split by rows
    // when one run still holds two captions above each other (TopBottom)
    struct R { int x0, x1, y0, y1; };
    std::vector<R> rects;
    auto rows_of = [&](int x0, int x1, int ya, int yb, std::vector<std::pair<int, int>>& runs) {
        runs.clear();
        int start = -1, last = -100;
        for (int y = ya; y <= yb; ++y) {
            bool any = false;
            for (int x = x0; x <= x1 && !any; ++x) any = touched[(std::size_t)y * cw + x] != 0;
            if (any) { if (start < 0) start = y; else if (y - last > 8) { runs.push_back({start, last}); start = y; } last = y; }
        }
        if (start >= 0) runs.push_back({start, last});
    };
    {
        std::vector<std::uint8_t> col(cw, 0);
        for (int y = 0; y < ch; ++y) for (int x = 0; x < cw; ++x) if (touched[(std::size_t)y * cw + x]) col[x] = 1;
        int start = -1, last = -100;
        std::vector<std::pair<int, int>> cruns, rruns;
        for (int x = 0; x < cw; ++x) if (col[x]) { if (start < 0) start = x; else if (x - last > 8) { cruns.push_back({start, last}); start = x; } last = x; }
        if (start >= 0) cruns.push_back({start, last});
        for (auto& c : cruns) { rows_of(c.first, c.second, 0, ch - 1, rruns); for (auto& r : rruns) rects.push_back(R{c.first, c.second, r.first, r.second}); }
    }
    if (rects.empty()) return {};
    if (rects.size() > 4) throw Error(LVM_ERR_INVALID, "overlay_tables: the drawing touches more than four separate regions");
    // pass 2: the function v -> byte of every pixel of the rectangles
    std::vector<std::vector<std::uint8_t>> f(rects.size());          // [rect][pixel][256]
    for (std::size_t r = 0; r < rects.size(); ++r) f[r].resize((std::size_t)(rects[r].x1 - rects[r].x0 + 1) * (rects[r].y1 - rects[r].y0 + 1) * 256);
    for (int v = 0; v < 256; ++v) {
        std::memset(canvas.data(), v, canvas.size());
        draw(canvas.data(), cw, ch, (std::ptrdiff_t)row);
        for (std::size_t r = 0; r < rects.size(); ++r) {
            const int w = rects[r].x1 - rects[r].x0 + 1;
            for (int y = rects[r].y0; y <= rects[r].y1; ++y)
                for (int x = rects[r].x0; x <= rects[r].x1; ++x) {
                    const std::uint8_t* p = canvas.data() + (std::size_t)y * row + (std::size_t)x * 3;
                    if (p[0] != p[1] || p[1] != p[2]) throw Error(LVM_ERR_INVALID, "overlay_tables: the drawing treats B, G and R differently");
                    f[r][((std::size_t)(y - rects[r].y0) * w + (x - rects[r].x0)) * 256 + v] = p[0];
                }
        }
    }
    std::vector<OverlayLabel> out(rects.size());
    for (std::size_t r = 0; r < rects.size(); ++r) {
        OverlayLabel& L = out[r];
        L.x = rects[r].x0; L.y = rects[r].y0; L.w = rects[r].x1 - rects[r].x0 + 1; L.h = rects[r].y1 - rects[r].y0 + 1;
        L.cls.resize((std::size_t)L.w * L.h);
        std::map<std::string, int> ids;
        for (std::size_t k = 0; k < L.cls.size(); ++k) {
            const std::string key(reinterpret_cast<const char*>(&f[r][k * 256]), 256);
            auto it = ids.find(key);
            if (it == ids.end()) {
                if (ids.size() >= 65536) throw Error(LVM_ERR_INVALID, "overlay_tables: more than 65536 distinct pixel functions in one label");
                it = ids.emplace(key, (int)ids.size()).first;
                L.fn.insert(L.fn.end(), key.begin(), key.end());
            }
            L.cls[k] = (std::uint16_t)it->second;
        }
    }
    // the premise, checked: drawn on a canvas that is NOT constant, the result is still what the tables say (a drawing that read its
    // neighbours -- a blur, an alpha-blended image -- would differ here)
    for (int y = 0; y < ch; ++y)
        for (int x = 0; x < cw; ++x) {
            const std::uint8_t v = (std::uint8_t)(((x * 7 + y * 13) & 1) ? 231 - (x * 5 + y * 3) % 97 : 18 + (x * 3 + y * 11) % 89);
            std::uint8_t* p = canvas.data() + (std::size_t)y * row + (std::size_t)x * 3;
            p[0] = v; p[1] = (std::uint8_t)(v ^ 0x55); p[2] = (std::uint8_t)(255 - v);
        }
    std::vector<std::uint8_t> want = canvas;
    draw(canvas.data(), cw, ch, (std::ptrdiff_t)row);
    for (const OverlayLabel& L : out)
        for (int y = 0; y < L.h; ++y)
            for (int x = 0; x < L.w; ++x) {
                const std::uint8_t* fn = L.fn.data() + (std::size_t)L.cls[(std::size_t)y * L.w + x] * 256;
                std::uint8_t* p = want.data() + (std::size_t)(L.y + y) * row + (std::size_t)(L.x + x) * 3;
                p[0] = fn[p[0]]; p[1] = fn[p[1]]; p[2] = fn[p[2]];
            }
    if (want != canvas) throw Error(LVM_ERR_INVALID, "overlay_tables: the drawing is not a per-pixel function of the canvas (it reads other pixels, or touches pixels the constant canvases did not show)");
    return out;
}

// hands the tables to a context (lvm_export_set_overlay); an empty vector switches the overlay off
inline void set_overlay(lvm_ctx* ctx, const std::vector<OverlayLabel>& labels) {
    std::vector<lvm_overlay_label> c(labels.size());
    for (std::size_t i = 0; i < labels.size(); ++i)
        c[i] = lvm_overlay_label{labels[i].x, labels[i].y, labels[i].w, labels[i].h, labels[i].n_classes(), labels[i].cls.data(), labels[i].fn.data()};
    const int rc = lvm_export_set_overlay(ctx, (int)c.size(), c.empty() ? nullptr : c.data());
    if (rc != LVM_OK) throw Error(rc, std::string("lvm: ") + lvm_last_error(ctx));
}

}  // namespace lvm
