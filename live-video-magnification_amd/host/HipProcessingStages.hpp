// HipProcessingStages.hpp -- the reference's three per-frame stages as ONE device-side stage.
//
// The reference builds its chain in buildProcessors() (src/processing/ChainBuilder.cpp:9-17):
//
//     procs.push_back(std::make_unique<PreprocessProcessor>());
//     procs.push_back(std::make_unique<GrayscaleProcessor>());
//     procs.push_back(std::make_unique<MagnificationProcessor>());
//
// Replacing the three lines by
//
//     procs.push_back(std::make_unique<HipProcessingStages>());
//
// keeps runChainOnce / ProcessingChain / Exporter unchanged and moves crop + INTER_AREA decimation + gray +
// magnification onto the GPU: only the ROI rows are uploaded and only the (decimated) result comes back.
// Contracts mirrored:
//   * PreprocessProcessor.cpp:10-51: identity when no ROI and divisor 1; output Frame carries the new
//     width/height (:46-49);
//   * GrayscaleProcessor.cpp:7-16: identity unless cfg.grayscale and a 3-channel frame; format Gray8;
//   * MagnificationProcessor.cpp:17-67: passthrough hands the magnifier's INPUT on (here: the preprocessed
//     frame); state reset when PreprocessParams change (MagnifyCore.hpp:55-56);
//   * errors surface as std::runtime_error for ProcessingChain.cpp:50-62.
#pragma once
#include <memory>

#include <opencv2/core.hpp>

#include "lvm.hpp"
#include "processing/IProcessor.hpp"

namespace livim {

class HipProcessingStages : public IProcessor {
public:
    explicit HipProcessingStages(int device = 0) : mag_(device, 1) {}

    FrameRef process(const FrameRef& in, const ProcessorConfig& cfg) override {
        const cv::Mat& src = in->image;
        if (src.empty()) return in;                                   // every stage returns `in` on an empty image
        lvm_preprocess_params pre{};
        pre.downscale = cfg.preprocess.downscale;
        pre.roi_enabled = cfg.preprocess.roiEnabled ? 1 : 0;
        pre.roiX = cfg.preprocess.roiX; pre.roiY = cfg.preprocess.roiY;
        pre.roiW = cfg.preprocess.roiW; pre.roiH = cfg.preprocess.roiH;
        pre.grayscale = cfg.grayscale ? 1 : 0;
        const MagnificationParams& p = cfg.magnification;
        lvm::MagnificationParams q;
        q.mode = static_cast<lvm::MagnificationMode>(static_cast<int>(p.mode));
        q.amplification = p.amplification; q.coWavelength = p.coWavelength; q.coLow = p.coLow; q.coHigh = p.coHigh;
        q.chromAttenuation = p.chromAttenuation; q.levels = p.levels; q.framerate = p.framerate;
        int ow = 0, oh = 0, och = 0;
        lvm::Magnifier::chain_geometry(pre, src.cols, src.rows, src.channels(), &ow, &oh, &och);
        const bool stage_identity = ow == src.cols && oh == src.rows && och == src.channels();
        // (output in a recycled page-locked buffer: the download is a plain DMA, no runtime-side pinning per frame)
        const std::size_t row = static_cast<std::size_t>(ow) * static_cast<std::size_t>(och);
        std::shared_ptr<std::uint8_t> buf = pool_.acquire(row * static_cast<std::size_t>(oh));
        const int type = och == 1 ? CV_8UC1 : CV_8UC3;
        cv::Mat dst = buf ? cv::Mat(oh, ow, type, buf.get(), row) : cv::Mat(oh, ow, type);
        const bool produced = mag_.chain_process(pre, q, src.data, src.cols, src.rows, src.channels(),
                                                 static_cast<std::ptrdiff_t>(src.step), dst.data,
                                                 static_cast<std::ptrdiff_t>(dst.step));
        if (!produced && stage_identity) return in;                   // all three stages were identities
        auto out = std::make_shared<PinnedFrame>(*in);
        out->keep = std::move(buf);                                   // returns to the pool with the frame
        out->image = std::move(dst);
        out->width = ow; out->height = oh;                            // PreprocessProcessor.cpp:46-49
        out->format = och >= 3 ? PixelFormat::BGR8 : PixelFormat::Gray8;   // GrayscaleProcessor.cpp:14
        return out;
    }

    void reset() override { mag_.reset(); }

private:
    struct PinnedFrame : Frame { explicit PinnedFrame(const Frame& f) : Frame(f) {} std::shared_ptr<std::uint8_t> keep; };
    lvm::Magnifier mag_;
    lvm::PinnedPool pool_;
};

}  // namespace livim
