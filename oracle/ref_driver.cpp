// oracle/ref_driver.cpp -- C entry points over the REAL reference stage (TEST INFRASTRUCTURE).
//
// Built only where OpenCV 4 exists (oracle/Makefile target `ref_full`, probed at build time): the reference's own
// sources -- src/processing/MagnificationProcessor.cpp and src/processing/magnification/*.cpp -- are compiled where
// they lie under /root/reference together with this file into oracle/_ref/libref_magnify.so.  Nothing of the
// reference is copied; this file only adapts MagnificationProcessor::process (MagnificationProcessor.cpp:17-67) to
// plain pointers, with the same signature shape as lvmo_process so that tests and bench.py can swap checkers.
// ref_recover_lab_lut hands out the forward Lab table of the OpenCV build it is linked with (round 3): the one residual of
// the restated colour path that cannot be pinned without OpenCV (its softfloat table build) then stops being a residual.
// NOT built in this image (no OpenCV: SURVEY.md 8c) and therefore untested here; oracle/pyoracle.py::RefOracle loads
// it when present and bench.py records the probe's result either way.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>

#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>

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

// OpenCV's forward Lab table, recovered through its public interface: at a grid node (channel values k / 32) every weight
// of the upper neighbours is 0, so cv::cvtColor returns the table entry itself (L = T * 100 / 16384, a = T / 64 - 128).
// out: 33 * 33 * 33 * 3 int16 in RGB2Labprev order, index 3 (p + 33 q + 1089 r) + channel (p, q, r = R, G, B grid index) --
// what lvm_set_lab_lut / lvmo_lab_lut_override take.  Returns 0 on success, -1 when the build does not interpolate (a value
// off the 1/16384 lattice): then the analytic flavour (lvm_debug_lab_analytic) is the one to compare with.
int ref_recover_lab_lut(std::int16_t* out) {
    try {
        cv::Mat img(33 * 33, 33, CV_32FC3), lab;
        for (int r = 0; r < 33; ++r)
            for (int q = 0; q < 33; ++q)
                for (int p = 0; p < 33; ++p) img.at<cv::Vec3f>(r * 33 + q, p) = cv::Vec3f(r / 32.0f, q / 32.0f, p / 32.0f);   // B, G, R
        cv::cvtColor(img, lab, cv::COLOR_BGR2Lab);
        for (int r = 0; r < 33; ++r)
            for (int q = 0; q < 33; ++q)
                for (int p = 0; p < 33; ++p) {
                    const cv::Vec3f v = lab.at<cv::Vec3f>(r * 33 + q, p);
                    const double t[3] = {v[0] * (16384.0 / 100.0), (v[1] + 128.0) * 64.0, (v[2] + 128.0) * 64.0};
                    for (int c = 0; c < 3; ++c) {
                        const double n = static_cast<double>(static_cast<long>(t[c] + (t[c] >= 0 ? 0.5 : -0.5)));
                        if (n - t[c] > 1e-3 || t[c] - n > 1e-3 || n < 0 || n > 16384) return -1;
                        out[3 * (p + 33 * (q + 33 * r)) + c] = static_cast<std::int16_t>(n);
                    }
                }
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

}  // extern "C"
