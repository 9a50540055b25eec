"""The four reference-side shims (host/HipMagnificationProcessor.hpp, HipProcessingStages.hpp,
HipBatchedProcessingChain.hpp and HipExportRunner.hpp with LVM_WITH_LIVIM_HEADERS) compiled against the REFERENCE'S OWN headers
(/root/reference/src: IProcessor.hpp, core/Frame.hpp, FrameQueue, LatestFrameMailbox, AtomicConfig, Instrumentation),
which are Qt-free.  OpenCV is not in this image, so <opencv2/core.hpp> is a small stand-in written here that declares just
the cv::Mat surface the shims and those headers use (constructors, data / rows / cols / step, channels(), type(), empty(),
clone(), ptr()) -- enough for the compiler to check every call the shims make into the reference's types: the override
signatures of IProcessor, the Frame fields, BoundedQueue::pop / stop, LatestFrameMailbox::publish, AtomicConfig::read,
Instrumentation's hooks.  Skipped where the reference checkout is absent (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "live-video-magnification_amd")
REF = "/root/reference/src"

CV_STUB = r'''
#pragma once
// stand-in for <opencv2/core.hpp> (TEST INFRASTRUCTURE): the part of cv::Mat the shims touch
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#define CV_8UC1 0
#define CV_8UC3 16
namespace cv {
struct Size { int width = 0, height = 0; Size() = default; Size(int w, int h) : width(w), height(h) {} };
struct MatStep { size_t v = 0; operator size_t() const { return v; } };
class Mat {
public:
    Mat() = default;
    Mat(int r, int c, int type) : rows(r), cols(c), type_(type) { alloc(); }
    Mat(int r, int c, int type, void* d, size_t s) : rows(r), cols(c), type_(type) { data = static_cast<unsigned char*>(d); step.v = s; }
    int channels() const { return type_ == CV_8UC3 ? 3 : 1; }
    int type() const { return type_; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Mat clone() const { Mat m(rows, cols, type_); for (int y = 0; y < rows; ++y) std::memcpy(m.ptr(y), ptr(y), (size_t)cols * channels()); return m; }
    unsigned char* ptr(int y) { return data + (size_t)y * step.v; }
    const unsigned char* ptr(int y) const { return data + (size_t)y * step.v; }
    unsigned char* data = nullptr;
    int rows = 0, cols = 0;
    MatStep step;
private:
    void alloc() { step.v = (size_t)cols * channels(); buf_ = std::shared_ptr<unsigned char>(new unsigned char[step.v * rows + 1], std::default_delete<unsigned char[]>()); data = buf_.get(); }
    int type_ = CV_8UC3;
    std::shared_ptr<unsigned char> buf_;
};
}  // namespace cv
'''

SRC = r'''
#define LVM_WITH_LIVIM_HEADERS
#include "HipMagnificationProcessor.hpp"
#include "HipProcessingStages.hpp"
#include "HipBatchedProcessingChain.hpp"
#include "HipExportRunner.hpp"
#include "HipMjpegWriter.hpp"
#include <cstdio>
#include <vector>
int main() {
    using namespace livim;
    try {
        std::vector<std::unique_ptr<IProcessor>> procs;                 // what buildProcessors() returns (ChainBuilder.cpp:9-17)
        procs.push_back(std::make_unique<HipMagnificationProcessor>());
        procs.push_back(std::make_unique<HipProcessingStages>());
        auto f = std::make_shared<Frame>();
        f->image = cv::Mat(48, 64, CV_8UC3); f->width = 64; f->height = 48;
        std::memset(f->image.data, 90, 48 * 64 * 3);
        ProcessorConfig cfg;
        cfg.magnification.mode = MagnificationMode::Laplace; cfg.magnification.levels = 2; cfg.magnification.amplification = 10;
        cfg.magnification.coWavelength = 100; cfg.magnification.coLow = 0.1; cfg.magnification.coHigh = 0.4;
        FrameRef in = f;
        for (auto& p : procs) { FrameRef out = p->process(in, cfg); std::printf("out %dx%d\n", out->image.cols, out->image.rows); p->reset(); }
        FrameQueue q0(4, OverflowPolicy::Block), q1(4, OverflowPolicy::Block);
        LatestFrameMailbox m0, m1;
        AtomicConfig<ProcessorConfig> ac(cfg);
        HipBatchedProcessingChain chain({&q0, &q1}, {&m0, &m1}, nullptr, &ac, 0);
        std::printf("sources %zu\n", chain.sources());
        // the export loop bound to IExportFrameSource / cv::Mat (Exporter.cpp:216-259): a two-frame source, a sink that counts
        struct TwoFrames : IExportFrameSource {
            int n = 0;
            bool open() override { return true; }
            int frameCount() const override { return 2; }
            cv::Size size() const override { return cv::Size(64, 48); }
            bool next(cv::Mat& out) override { if (n >= 2) return false; out = cv::Mat(48, 64, CV_8UC3); std::memset(out.data, 80 + 10 * n, 48 * 64 * 3); ++n; return true; }
            void close() override {}
        } two;
        ExportRequest req; req.config = cfg; req.split = SplitMode::LeftRight;
        std::atomic<bool> abort_flag{false}; std::atomic<int> done{0};
        LivimExportSink sink;
        sink.abort = &abort_flag; sink.frames_done = &done;
        sink.write_canvas = [](cv::Mat& canvas) { return canvas.cols == 128 && canvas.rows == 48; };
        LivimExportTraits::Source src{&two, cv::Mat()};
        HipExportLoop loop(0, 8);
        // request.textOverlay: the reference's label code as the loop's canvas drawer (a cv::Mat view around the raw canvas; HipExportOverlay.hpp)
        loop.set_canvas_drawer(export_canvas_drawer([](cv::Mat& canvas) { if (canvas.rows > 8 && canvas.cols > 40) std::memset(canvas.ptr(6) + 18, 255, 30); }));
        const auto written = loop.run(src, sink, export_pre_params(req.config), export_mag_params(req.config), export_split(req.split), 30.0);
        std::printf("export wrote %llu done %d\n", (unsigned long long)written, done.load());
        // ExportFormat::AviMjpg without the overlay: the canvases arrive as JPEG frames (lvm_export_frames_mjpeg), lvm::MjpegAviWriter is the container
        TwoFrames two_more;
        LivimExportTraits::Source src_j{&two_more, cv::Mat()};
        lvm::MjpegAviWriter avi;
        sink.write_jpeg = [&avi](const std::uint8_t* jpeg, std::size_t bytes, int cw, int ch) {
            return (avi.isOpened() || avi.open("/tmp/lvm_shim_test.avi", cw, ch, 30.0)) && avi.write(jpeg, bytes);
        };
        const auto jw = loop.run_mjpeg(src_j, sink, export_pre_params(req.config), export_mag_params(req.config), export_split(req.split), 30.0, 85);
        std::printf("mjpeg wrote %llu frames %u closed %d\n", (unsigned long long)jw, avi.frames(), (int)avi.close());
    } catch (const std::exception& e) { std::printf("exception: %s\n", e.what()); return 3; }
    return 0;
}
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout absent")
def test_shims_compile_against_the_reference_headers(tmp_path):
    inc = tmp_path / "opencv2"
    inc.mkdir()
    (inc / "core.hpp").write_text(CV_STUB)
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    cmd = ["g++", "-std=c++20", "-Wall", "-pthread", str(src), "-I", str(tmp_path), "-I", REF, "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(PKG, "host"), "-L", PKG, "-llvm_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    import torch
    if torch.cuda.is_available():
        assert run.returncode == 0 and "out 64x48" in run.stdout and "sources 2" in run.stdout and "export wrote 2 done 2" in run.stdout and "mjpeg wrote 2 frames 2 closed 1" in run.stdout, run.stdout + run.stderr
    else:                                                               # no device here: the constructors fail loudly
        assert run.returncode == 3 and "lvm_create failed" in run.stdout, run.stdout + run.stderr
