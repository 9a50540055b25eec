// HipMagnificationProcessor.hpp -- the drop-in replacement of the reference's magnification stage.
//
// Compiles INSIDE the reference tree (needs its headers + OpenCV core): add this file and
// include/lvm_hip.h + lvm.hpp to the include path, link liblvm_hip.so, and change ONE line in
// src/processing/ChainBuilder.cpp:15
//
//     procs.push_back(std::make_unique<MagnificationProcessor>());
// ->  procs.push_back(std::make_unique<HipMagnificationProcessor>());
//
// Everything else (Qt UI, PlaybackController, ProcessingChain, Exporter, cv::Mat I/O) is untouched.
// Contract mirrored from processing/MagnificationProcessor.cpp:17-67:
//   * returns `in` itself when the stage is an identity for this frame (mode None, empty image,
//     frame too small, Color warm-up, Riesz first frame / gray input);
//   * otherwise returns a NEW Frame: metadata copied, image in a fresh buffer, format set;
//   * throws std::runtime_error on failure so ProcessingChain's catch block (ProcessingChain.cpp:50-62)
//     counts the error, calls reset() on every stage and shows the input frame;
//   * reset() drops all temporal state (MagnificationProcessor.cpp:10-15).
#pragma once
#include <cstring>
#include <memory>

#include <opencv2/core.hpp>

#include "lvm.hpp"
#include "processing/IProcessor.hpp"

namespace livim {

class HipMagnificationProcessor : public IProcessor {
public:
    explicit HipMagnificationProcessor(int device = 0) : mag_(device, 1) {}

    FrameRef process(const FrameRef& in, const ProcessorConfig& cfg) override {
        const MagnificationParams& p = cfg.magnification;
        lvm::MagnificationParams q;
        q.mode = static_cast<lvm::MagnificationMode>(static_cast<int>(p.mode));   // same enumerator order
        q.amplification = p.amplification;
        q.coWavelength = p.coWavelength;
        q.coLow = p.coLow;
        q.coHigh = p.coHigh;
        q.chromAttenuation = p.chromAttenuation;
        q.levels = p.levels;
        q.framerate = p.framerate;
        const cv::Mat& src = in->image;
        if (src.empty()) {                       // identity; also lets the core drop its state (:21-29)
            mag_.process(q, key(cfg.preprocess), nullptr, 0, 0, 3, 0, nullptr, 0);
            return in;
        }
        // the output frame lives in a recycled PAGE-LOCKED buffer: the last kernel writes it there directly (no download);
        // pageable memory when the pool cannot allocate
        const std::size_t row = static_cast<std::size_t>(src.cols) * static_cast<std::size_t>(src.channels());
        std::shared_ptr<std::uint8_t> buf = pool_.acquire(row * static_cast<std::size_t>(src.rows));
        cv::Mat dst = buf ? cv::Mat(src.rows, src.cols, src.type(), buf.get(), row) : cv::Mat(src.rows, src.cols, src.type());
        const bool produced = mag_.process(q, key(cfg.preprocess), src.data, src.cols, src.rows, src.channels(),
                                           static_cast<std::ptrdiff_t>(src.step), dst.data,
                                           static_cast<std::ptrdiff_t>(dst.step));
        if (!produced) return in;                // warm-up / unsupported input: emit the input unchanged (:61)
        auto out = std::make_shared<PinnedFrame>(*in);
        out->image = std::move(dst);             // fresh buffer; never aliases in->image (:63-66)
        out->keep = std::move(buf);              // returns to the pool with the frame
        out->format = src.channels() >= 3 ? PixelFormat::BGR8 : PixelFormat::Gray8;
        return out;
    }

    void reset() override { mag_.reset(); }

private:
    // Any value that changes iff PreprocessParams changes (operator==, IProcessor.hpp:36-39): FNV-1a
    // over the fields; the default-constructed struct maps to 0.
    static std::uint64_t key(const PreprocessParams& pp) {
        if (pp == PreprocessParams{}) return 0;
        std::uint64_t h = 1469598103934665603ull;
        auto mix = [&h](const void* d, std::size_t n) {
            const unsigned char* b = static_cast<const unsigned char*>(d);
            for (std::size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
        };
        const int ds = pp.downscale; const unsigned char roi = pp.roiEnabled ? 1 : 0;
        mix(&ds, sizeof ds); mix(&roi, 1);
        mix(&pp.roiX, sizeof pp.roiX); mix(&pp.roiY, sizeof pp.roiY); mix(&pp.roiW, sizeof pp.roiW); mix(&pp.roiH, sizeof pp.roiH);
        return h ? h : 1;
    }
    struct PinnedFrame : Frame { explicit PinnedFrame(const Frame& f) : Frame(f) {} std::shared_ptr<std::uint8_t> keep; };
    lvm::Magnifier mag_;
    lvm::PinnedPool pool_;
};

}  // namespace livim
