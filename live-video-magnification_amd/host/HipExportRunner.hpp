// HipExportRunner.hpp -- the reference's offline export loop over ONE device context, batch by batch
// (the reference side of lvm_export_frames / lvm_compose_device).
//
// Replaces the frame loop of Exporter::run (src/export/Exporter.cpp:216-259).  Everything around that loop -- opening the
// source, the writer fallback list, progress, abort, finalising / removing a partial file (:179-215, :261-289) -- is the
// reference's and stays where it is:
//
//   reference (one frame per iteration)                       here (up to `batch` frames per iteration)
//   ---------------------------------------------------------  ---------------------------------------------------------
//   while (!abort_) {                                (:216)   while (!abort()) {
//     if (!source->next(raw)) break;                 (:217)     pull frames until the batch is full or the source ends;
//     if (raw.empty()) continue;                     (:218)       empty frames are skipped; each kept frame is COPIED into a
//                                                                 pinned slot (next() decodes into `raw` in place, :228-230)
//     in = Frame{seq++, ptsUs = seq * 1e6 / captureFps, ...}    the same metadata per frame (seq, ptsUs), kept with its slot
//     cur = runChainOnce(chain, in, cfg, original)   (:233)     ONE lvm_export_frames call: Preprocess + Grayscale + the
//     canvas = compose(original, cur, split, overlay) (:242)      magnifier as a temporal batch + compose() on the device;
//                                                                 only ROI rows go up, only canvases come down
//     if (preview_) preview_->publish({cur, original})(:235-240) preview: the canvas of the LAST frame of the batch (latest wins)
//     open the writer on the first canvas            (:245-258)   the same, through the traits (first canvas of the run)
//     writer.write(canvas); framesDone_++            (:259-260)   per canvas, in order.  The text overlay (:36-50): with
//   }                                                             set_canvas_drawer() the reference's own label code is turned into
//                                                                 per-pixel tables once per geometry (HipExportOverlay.hpp) and applied
//                                                                 on the device -- also on the Motion-JPEG path; without it the traits
//                                                                 draw on the host canvas before it is written, as before round 6
//
// Frames stay strictly in order (the temporal filters are stateful, Exporter.hpp:18-20); a batch is n consecutive frames, so the
// results are those of the frame-by-frame loop (tests/test_export.py::test_export_runner_*: canvases byte-equal to runChainOnce + compose of
// the CPU oracle).  An abort is honoured between batches and between the writes of a batch ("stops at the next frame
// boundary", Exporter.hpp:33).  Template over a traits type so that it compiles and runs without OpenCV; `livim::HipExportLoop` at
// the end binds it to cv::Mat / IExportFrameSource / cv::VideoWriter inside the reference tree.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "lvm.hpp"
#include "HipExportOverlay.hpp"

namespace lvm {

// What a traits type provides:
//   struct View { const uint8_t* data; int w, h, channels; std::ptrdiff_t stride; bool empty; };
//   using Source = ...;  bool next(Source&, View& raw);       raw stays valid until the next call (decoded in place)
//   using Sink = ...;    bool write(Sink&, std::uint64_t seq, std::int64_t pts_us, std::uint8_t* canvas, int cw, int ch, std::ptrdiff_t stride);
//                        draws the overlay if any, opens the writer on the first canvas, writes; false = cannot write (stop)
//   static bool aborted(const Sink&);
//   (run_mjpeg only)     bool write_jpeg(Sink&, std::uint64_t seq, std::int64_t pts_us, const std::uint8_t* jpeg, std::size_t bytes, int cw, int ch);
//                        the canvas as ONE complete JPEG frame, encoded on the device: what lvm::MjpegAviWriter::write takes
template <class T>
class ExportRunner {
public:
    ExportRunner(int device, int batch) : mag_(device, 1), batch_(batch < 1 ? 1 : batch) {
        if (lvm_set_max_frames(mag_.handle(), batch_) != LVM_OK) throw Error(LVM_ERR_INVALID, "lvm_set_max_frames failed");
    }
    ~ExportRunner() { release(); }
    ExportRunner(const ExportRunner&) = delete;
    ExportRunner& operator=(const ExportRunner&) = delete;

    // The loop.  Returns the number of canvases written.  Throws lvm::Error on a library failure (the reference's catch at
    // Exporter.cpp:283-288 turns any std::exception into the Error phase).
    std::uint64_t run(typename T::Source& src, typename T::Sink& sink, const lvm_preprocess_params& pre, const MagnificationParams& mag, int split,
                      double capture_fps) {
        return run_impl<false>(src, sink, pre, mag, split, capture_fps, 0);
    }
    // The same loop for ExportFormat::AviMjpg (Exporter.cpp:107-117) without the text overlay: the canvases are JPEG-encoded on the device
    // (lvm_export_frames_mjpeg, libjpeg's quality scale 1..100), only the compressed frames come down, the sink gets them through
    // T::write_jpeg -- cv::VideoWriter's software codec is out of the loop.  A frame larger than its raw canvas (noise at quality 100)
    // makes the call fail with lvm::Error: the slots are sized for the canvases.
    std::uint64_t run_mjpeg(typename T::Source& src, typename T::Sink& sink, const lvm_preprocess_params& pre, const MagnificationParams& mag, int split,
                            double capture_fps, int quality) {
        return run_impl<true>(src, sink, pre, mag, split, capture_fps, quality);
    }
    int batch() const { return batch_; }
    // request.textOverlay (ExportTypes.hpp:22): `draw` = the reference's label code for one canvas (what compose does for overlay == true,
    // Exporter.cpp:74-77 / :82-85).  It is run on constant canvases whenever the canvas geometry is (re)established and never on a frame:
    // the device applies the resulting tables (lvm_export_set_overlay).  An empty function switches the overlay off.
    void set_canvas_drawer(CanvasDrawer draw) { drawer_ = std::move(draw); overlay_dirty_ = true; }

private:
    template <bool MJPEG>
    std::uint64_t run_impl(typename T::Source& src, typename T::Sink& sink, const lvm_preprocess_params& pre, const MagnificationParams& mag, int split,
                           double capture_fps, int quality) {
        const lvm_params c = to_c(mag, 0);
        const double interval_us = 1'000'000.0 / (capture_fps > 0.0 ? capture_fps : 30.0);     // Exporter.cpp:196-199, :214
        std::uint64_t seq = 0, written = 0;
        bool more = true;
        while (more && !T::aborted(sink)) {
            int n = 0;
            if (has_carry_) {                                                                   // the frame that changed the geometry starts this batch
                const typename T::View v{carry_.data(), cw0_, ch0_, cc0_, (std::ptrdiff_t)cw0_ * cc0_, false};
                prepare(v, pre, split);
                store(n, v); seqs_[(std::size_t)n++] = seq++;
                has_carry_ = false;
            }
            while (n < batch_) {
                typename T::View raw{};
                if (!T::next(src, raw)) { more = false; break; }                                // :217
                if (raw.empty || !raw.data) continue;                                           // :218
                if (canvas_empty(raw, pre, split)) { ++seq; continue; }                         // :243 `if (canvas.empty()) continue;` (the frame is consumed, nothing is written --
                                                                                                //  but it HAS taken its sequence number, :224 precedes :243: the pts of later frames count it;
                                                                                                //  such a frame cannot pass the magnifier either: both of its sizes are < 2)
                if (!slots_ready(raw)) {
                    if (n == 0) prepare(raw, pre, split);
                    else { keep(raw); break; }                                                  // flush what we have with the old geometry first
                }
                store(n, raw); seqs_[(std::size_t)n++] = seq++;
            }
            if (n > 0) written += flush<MJPEG>(n, sink, pre, c, split, interval_us, quality);
        }
        return written;
    }
    bool slots_ready(const typename T::View& v) const { return in_ && !overlay_dirty_ && v.w == w_ && v.h == h_ && v.channels == ch_; }
    static bool canvas_empty(const typename T::View& v, const lvm_preprocess_params& pre, int split) {
        int cw = 0, ch = 0;
        return lvm_export_geometry(&pre, split, v.w, v.h, v.channels, &cw, &ch) != LVM_OK || cw <= 0 || ch <= 0;
    }
    void keep(const typename T::View& v) {
        cw0_ = v.w; ch0_ = v.h; cc0_ = v.channels;
        carry_.resize((std::size_t)v.w * v.h * v.channels);
        for (int y = 0; y < v.h; ++y) std::memcpy(carry_.data() + (std::size_t)y * v.w * v.channels, v.data + (std::ptrdiff_t)y * v.stride, (std::size_t)v.w * v.channels);
        has_carry_ = true;
    }
    void release() {
        if (in_) lvm_host_free(in_);
        if (out_) lvm_host_free(out_);
        in_ = out_ = nullptr;
    }
    // pinned slots for `batch` input frames and canvases of this geometry (what core/FramePool.cpp:29-36 would hand out)
    void prepare(const typename T::View& v, const lvm_preprocess_params& pre, int split) {
        release();
        w_ = v.w; h_ = v.h; ch_ = v.channels;
        if (lvm_export_geometry(&pre, split, w_, h_, ch_, &cw_, &chh_) != LVM_OK || cw_ <= 0 || chh_ <= 0)
            throw Error(LVM_ERR_INVALID, "export: empty canvas for this geometry");
        frame_bytes_ = (std::size_t)w_ * h_ * ch_; canvas_bytes_ = (std::size_t)cw_ * chh_ * 3;
        if (drawer_) set_overlay(mag_.handle(), overlay_tables(cw_, chh_, drawer_));        // the labels of THIS canvas size
        else if (overlay_dirty_) set_overlay(mag_.handle(), {});
        overlay_dirty_ = false;
        void* a = nullptr; void* b = nullptr;
        if (lvm_host_alloc(frame_bytes_ * batch_, &a) != LVM_OK || lvm_host_alloc(canvas_bytes_ * batch_, &b) != LVM_OK) {
            if (a) lvm_host_free(a);
            throw Error(LVM_ERR_OOM, "export: lvm_host_alloc failed");
        }
        in_ = static_cast<std::uint8_t*>(a); out_ = static_cast<std::uint8_t*>(b);
        seqs_.assign((std::size_t)batch_, 0);
    }
    void store(int k, const typename T::View& v) {
        std::uint8_t* d = in_ + (std::size_t)k * frame_bytes_;
        const std::size_t row = (std::size_t)w_ * ch_;
        for (int y = 0; y < h_; ++y) std::memcpy(d + (std::size_t)y * row, v.data + (std::ptrdiff_t)y * v.stride, row);
    }
    template <bool MJPEG>
    std::uint64_t flush(int n, typename T::Sink& sink, const lvm_preprocess_params& pre, const lvm_params& c, int split, double interval_us, int quality) {
        std::vector<const std::uint8_t*> fin((std::size_t)n);
        std::vector<std::uint8_t*> can((std::size_t)n);
        std::vector<int> produced((std::size_t)n, 0);
        std::vector<std::size_t> offs((std::size_t)n + 1, 0);
        for (int k = 0; k < n; ++k) { fin[(std::size_t)k] = in_ + (std::size_t)k * frame_bytes_; can[(std::size_t)k] = out_ + (std::size_t)k * canvas_bytes_; }
        int rc;
        if constexpr (MJPEG)             // the canvas slots receive the JPEG frames back to back
            rc = lvm_export_frames_mjpeg(mag_.handle(), &pre, &c, split, n, fin.data(), w_, h_, ch_, (std::ptrdiff_t)w_ * ch_, quality, out_,
                                         canvas_bytes_ * (std::size_t)batch_, offs.data(), produced.data());
        else
            rc = lvm_export_frames(mag_.handle(), &pre, &c, split, n, fin.data(), w_, h_, ch_, (std::ptrdiff_t)w_ * ch_, can.data(),
                                   (std::ptrdiff_t)cw_ * 3, produced.data());
        if (rc != LVM_OK) throw Error(rc, std::string("lvm: ") + lvm_last_error(mag_.handle()));
        std::uint64_t written = 0;
        for (int k = 0; k < n; ++k) {
            if (T::aborted(sink)) break;                                                         // "stops at the next frame boundary"
            const std::int64_t pts = static_cast<std::int64_t>(static_cast<double>(seqs_[(std::size_t)k]) * interval_us);   // :224
            if constexpr (MJPEG) {
                if (!T::write_jpeg(sink, seqs_[(std::size_t)k], pts, out_ + offs[(std::size_t)k], offs[(std::size_t)k + 1] - offs[(std::size_t)k], cw_, chh_)) return written;
            } else {
                if (!T::write(sink, seqs_[(std::size_t)k], pts, can[(std::size_t)k], cw_, chh_, (std::ptrdiff_t)cw_ * 3)) return written;
            }
            ++written;                                                                           // framesDone_ (:260)
        }
        return written;
    }

    Magnifier mag_;
    int batch_;
    int w_ = 0, h_ = 0, ch_ = 0, cw_ = 0, chh_ = 0;
    std::size_t frame_bytes_ = 0, canvas_bytes_ = 0;
    std::uint8_t *in_ = nullptr, *out_ = nullptr;
    std::vector<std::uint64_t> seqs_;
    std::vector<std::uint8_t> carry_; bool has_carry_ = false; int cw0_ = 0, ch0_ = 0, cc0_ = 0;
    CanvasDrawer drawer_; bool overlay_dirty_ = false;
};

}  // namespace lvm

// ---- binding to the reference's types (compiles inside the reference tree only) -------------------------------
#if defined(LVM_WITH_LIVIM_HEADERS)
#include <atomic>
#include <functional>
#include <opencv2/core.hpp>

#include "export/ExportTypes.hpp"
#include "export/IExportFrameSource.hpp"
#include "processing/IProcessor.hpp"

namespace livim {

// Sink = what Exporter::run does with a canvas (Exporter.cpp:242-260): overlay, lazy writer open, write, progress.  The
// reference passes its own lambdas, so openWriter's fallback list and the cv::putText overlay stay in Exporter.cpp.
struct LivimExportSink {
    std::function<bool(cv::Mat& canvas)> write_canvas;         // overlay + openWriter-on-first + writer.write; false = failed
    // HipExportLoop::run_mjpeg (ExportFormat::AviMjpg without the text overlay): the canvas arrives as a finished JPEG frame -- open an
    // lvm::MjpegAviWriter (HipMjpegWriter.hpp) of cw x ch on the first one instead of the cv::VideoWriter, write(jpeg, bytes)
    std::function<bool(const std::uint8_t* jpeg, std::size_t bytes, int cw, int ch)> write_jpeg;
    const std::atomic<bool>* abort = nullptr;                  // Exporter::abort_
    std::atomic<int>* frames_done = nullptr;                   // Exporter::framesDone_
};
struct LivimExportTraits {
    struct View { const std::uint8_t* data; int w, h, channels; std::ptrdiff_t stride; bool empty; };
    struct Source { IExportFrameSource* src; cv::Mat raw; };
    static bool next(Source& s, View& v) {
        if (!s.src->next(s.raw)) return false;
        v = View{s.raw.data, s.raw.cols, s.raw.rows, s.raw.channels(), static_cast<std::ptrdiff_t>(s.raw.step), s.raw.empty()};
        return true;
    }
    using Sink = LivimExportSink;
    static bool write(Sink& k, std::uint64_t, std::int64_t, std::uint8_t* canvas, int cw, int ch, std::ptrdiff_t stride) {
        cv::Mat m(ch, cw, CV_8UC3, canvas, static_cast<size_t>(stride));      // a view of the pinned canvas slot
        if (!k.write_canvas(m)) return false;
        if (k.frames_done) k.frames_done->fetch_add(1, std::memory_order_relaxed);
        return true;
    }
    static bool write_jpeg(Sink& k, std::uint64_t, std::int64_t, const std::uint8_t* jpeg, std::size_t bytes, int cw, int ch) {
        if (!k.write_jpeg || !k.write_jpeg(jpeg, bytes, cw, ch)) return false;
        if (k.frames_done) k.frames_done->fetch_add(1, std::memory_order_relaxed);
        return true;
    }
    static bool aborted(const Sink& k) { return k.abort && k.abort->load(std::memory_order_acquire); }
};
using HipExportLoop = lvm::ExportRunner<LivimExportTraits>;

// request.textOverlay: the reference's label code (drawLabel twice, as compose does it; INTEGRATION.md section 5 moves those lines into
// `drawLabels(cv::Mat&, SplitMode)`) as the runner's canvas drawer -- a cv::Mat view around the raw canvas, nothing else
inline lvm::CanvasDrawer export_canvas_drawer(std::function<void(cv::Mat& canvas)> draw_labels) {
    return [draw_labels](std::uint8_t* canvas, int cw, int ch, std::ptrdiff_t stride) {
        cv::Mat m(ch, cw, CV_8UC3, canvas, static_cast<size_t>(stride));
        draw_labels(m);
    };
}

// ProcessorConfig / SplitMode -> the C structs (the same mapping HipProcessingStages.hpp uses)
inline lvm_preprocess_params export_pre_params(const ProcessorConfig& cfg) {
    lvm_preprocess_params q{};
    q.downscale = cfg.preprocess.downscale; q.roi_enabled = cfg.preprocess.roiEnabled ? 1 : 0;
    q.roiX = cfg.preprocess.roiX; q.roiY = cfg.preprocess.roiY; q.roiW = cfg.preprocess.roiW; q.roiH = cfg.preprocess.roiH;
    q.grayscale = cfg.grayscale ? 1 : 0;
    return q;
}
inline lvm::MagnificationParams export_mag_params(const ProcessorConfig& cfg) {
    const MagnificationParams& p = cfg.magnification;
    lvm::MagnificationParams m;
    m.mode = static_cast<lvm::MagnificationMode>(static_cast<int>(p.mode));
    m.amplification = p.amplification; m.coWavelength = p.coWavelength; m.coLow = p.coLow; m.coHigh = p.coHigh;
    m.chromAttenuation = p.chromAttenuation; m.levels = p.levels; m.framerate = p.framerate;
    return m;
}
inline int export_split(SplitMode s) { return s == SplitMode::LeftRight ? LVM_SPLIT_LEFT_RIGHT : (s == SplitMode::TopBottom ? LVM_SPLIT_TOP_BOTTOM : LVM_SPLIT_NONE); }

}  // namespace livim
#endif
