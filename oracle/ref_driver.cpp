// oracle/ref_driver.cpp -- C entry points over the REAL reference stage (TEST INFRASTRUCTURE).
//
// Built only where OpenCV 4 exists (oracle/Makefile target `ref_full`, probed at build time): the reference's own
// sources -- src/processing/MagnificationProcessor.cpp and src/processing/magnification/*.cpp -- are compiled where
// they lie under /root/reference together with this file into oracle/_ref/libref_magnify.so.  Nothing of the
// reference is copied; this file only adapts MagnificationProcessor::process (MagnificationProcessor.cpp:17-67) to
// plain pointers, with the same signature shape as lvmo_process so that tests and bench.py can swap checkers.
// NOT built in this image (no OpenCV: SURVEY.md 8c) and therefore untested here; oracle/pyoracle.py::RefOracle loads
// it when present and bench.py records the probe's result either way.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>

#include <opencv2/core.hpp>

#include "processing/MagnificationProcessor.hpp"

extern "C" {

struct ref_params {           // = lvmo_params / lvm_params
    std::int32_t mode, levels;
    double amplification, coWavelength, coLow, coHigh, chromAttenuation, framerate;
    std::uint64_t preprocess_key;
};

void* ref_create() { return new livim::MagnificationProcessor(); }
void ref_destroy(void* p) { delete static_cast<livim::MagnificationProcessor*>(p); }
void ref_reset(void* p) { static_cast<livim::MagnificationProcessor*>(p)->reset(); }

// returns 0 on success, -1 when the reference threw; *produced = 0 <=> the reference returned its input frame
int ref_process(void* p, const ref_params* prm, const std::uint8_t* in, int w, int h, int channels, std::ptrdiff_t in_stride,
                std::uint8_t* out, std::ptrdiff_t out_stride, int* produced) {
    try {
        auto f = std::make_shared<livim::Frame>();
        f->width = w; f->height = h;
        f->format = channels == 1 ? livim::PixelFormat::Gray8 : livim::PixelFormat::BGR8;
        f->image = cv::Mat(h, w, channels == 1 ? CV_8UC1 : CV_8UC3, const_cast<std::uint8_t*>(in), static_cast<size_t>(in_stride)).clone();
        livim::ProcessorConfig cfg;
        cfg.magnification.mode = static_cast<livim::MagnificationMode>(prm->mode);
        cfg.magnification.levels = prm->levels;
        cfg.magnification.amplification = prm->amplification; cfg.magnification.coWavelength = prm->coWavelength;
        cfg.magnification.coLow = prm->coLow; cfg.magnification.coHigh = prm->coHigh;
        cfg.magnification.chromAttenuation = prm->chromAttenuation; cfg.magnification.framerate = prm->framerate;
        // a preprocess key other than 0 stands for "some non-default PreprocessParams": any value that differs works
        if (prm->preprocess_key) { cfg.preprocess.roiEnabled = true; cfg.preprocess.roiX = static_cast<float>(prm->preprocess_key % 1000) * 1e-6f; }
        const livim::FrameRef in_ref = f;
        const livim::FrameRef r = static_cast<livim::MagnificationProcessor*>(p)->process(in_ref, cfg);
        *produced = r.get() != in_ref.get();
        if (*produced)
            for (int y = 0; y < h; ++y) std::memcpy(out + (size_t)y * out_stride, r->image.ptr(y), (size_t)w * channels);
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

}  // extern "C"
