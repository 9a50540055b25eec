// HipBatchedProcessingChain.hpp -- ONE processing thread feeding N sources through ONE device context
// (the reference side of lvm_chain_process_batch).
//
// Generalises the reference's ProcessingChain (src/processing/ProcessingChain.hpp:19-44, ProcessingChain.cpp:34-71):
//
//   reference (one source)                                   here (N sources with the same geometry and configuration)
//   -------------------------------------------------------  ---------------------------------------------------------
//   in_->pop(in)                       (:37)                  one pop per source, in source order; any queue stopped => exit
//   cfg = config_->read()              (:39)                  the same: ONE snapshot per tick, shared by the N frames
//   cur = runChainOnce(chain_, in, *cfg, original)  (:43)     ONE lvm_chain_process_batch_ex call: crop + INTER_AREA + gray +
//                                                             magnification of the N frames, one launch per stage for all
//   out_->publish({processed = cur, original})      (:46-49)  one publish per source into ITS mailbox, processed/original
//                                                             of the same tick in one object
//   catch (std::exception): instr_->onProcessingError();      the same: count it, lvm_reset() (= reset() of every stage), publish
//     every stage reset(); publish {in, in}         (:50-58)  every source's INPUT frame as both panes
//   instr_->onProcessed(); recordLatency(now - captureTs)     once per frame of every source
//                                                   (:64-69)
//
// Frames are never pipelined across ticks (ProcessingChain.hpp:18-20) and never reordered: tick k processes frame k of
// every source.  The class is a template over a small traits type so that it compiles -- and is tested -- without OpenCV
// or Qt (tests/test_host_batched_chain.py); `livim::HipBatchedProcessingChain` at the end binds it to the reference's
// FrameQueue / LatestFrameMailbox / Frame / AtomicConfig and compiles only inside the reference tree.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <exception>
#include <memory>
#include <stdexcept>
#include <thread>
#include <utility>
#include <vector>

#include "lvm.hpp"

namespace lvm {

// What a traits type provides (see LivimTraits below and the mock in tests/test_host_batched_chain.py):
//   using FrameRef = ...;                       shared handle of an immutable input frame
//   struct View { const uint8_t* data; int w, h, channels; std::ptrdiff_t stride; };
//   static View view(const FrameRef&);
//   static FrameRef make_like(const FrameRef& in, int w, int h, int channels, uint8_t** data, std::ptrdiff_t* stride);
//                                               fresh frame carrying `in`'s metadata (seq, timestamps) and a new image
//   using Queue = ...;   bool pop(Queue&, FrameRef&);  void stop(Queue&);
//   using Mailbox = ...; void publish(Mailbox&, FrameRef processed, FrameRef original);
//   using Config = ...;  snapshot read(Config&): provides .pre (lvm_preprocess_params) and .mag (MagnificationParams)
//   using Instr = ...;   void on_error(Instr*); void on_processed(Instr*, const FrameRef&);   (Instr* may be null)
template <class T>
class BatchedChain {
public:
    using FrameRef = typename T::FrameRef;

    BatchedChain(std::vector<typename T::Queue*> in, std::vector<typename T::Mailbox*> out, typename T::Instr* instr,
                 typename T::Config* config, int device = 0)
        : in_(std::move(in)), out_(std::move(out)), instr_(instr), config_(config), mag_(device, static_cast<int>(in_.size())) {
        if (in_.empty() || in_.size() != out_.size()) throw std::invalid_argument("BatchedChain: one mailbox per input queue");
    }
    ~BatchedChain() { stop(); }
    BatchedChain(const BatchedChain&) = delete;
    BatchedChain& operator=(const BatchedChain&) = delete;

    void start() {                                       // ProcessingChain.cpp:22-26
        if (thread_.joinable()) return;
        stop_.store(false, std::memory_order_release);
        thread_ = std::thread([this] { run(); });
    }
    void stop() {                                        // :28-32
        stop_.store(true, std::memory_order_release);
        for (auto* q : in_) if (q) T::stop(*q);          // unblock a pop() that is waiting for the next frame
        if (thread_.joinable()) thread_.join();
    }
    std::size_t sources() const { return in_.size(); }
    std::uint64_t ticks() const { return ticks_.load(std::memory_order_acquire); }
    std::uint64_t errors() const { return errors_.load(std::memory_order_acquire); }

    // One tick on the calling thread (what run() loops over); false when a queue was stopped.  Public for tests and for
    // callers that own their thread.
    bool tick() {
        const std::size_t n = in_.size();
        std::vector<FrameRef> frames(n);
        for (std::size_t s = 0; s < n; ++s)
            if (!T::pop(*in_[s], frames[s])) return false;                    // :37
        const auto cfg = T::read(*config_);                                   // :39
        try {
            const auto v0 = T::view(frames[0]);
            std::vector<const std::uint8_t*> src(n);
            for (std::size_t s = 0; s < n; ++s) {
                const auto v = T::view(frames[s]);
                if (v.w != v0.w || v.h != v0.h || v.channels != v0.channels || v.stride != v0.stride || !v.data)
                    throw std::runtime_error("BatchedChain: the sources of one context must share geometry and row stride");
                src[s] = v.data;
            }
            int ow = 0, oh = 0, och = 0;
            Magnifier::chain_geometry(cfg.pre, v0.w, v0.h, v0.channels, &ow, &oh, &och);
            const bool stages_identity = ow == v0.w && oh == v0.h && och == v0.channels;   // Preprocess + Grayscale returned `in`
            // `original` is chain[0]'s output, tapped BEFORE GrayscaleProcessor (ChainBuilder.cpp:25): the source's own channel count
            const int tap_ch = v0.channels;
            std::vector<FrameRef> processed(n), original(n);
            std::vector<std::uint8_t*> dst(n), pre(n, nullptr);
            std::ptrdiff_t dstride = 0, pstride = 0;
            for (std::size_t s = 0; s < n; ++s) {
                processed[s] = T::make_like(frames[s], ow, oh, och, &dst[s], &dstride);
                if (stages_identity) original[s] = frames[s];                 // the magnifier saw the input frame itself
                else original[s] = T::make_like(frames[s], ow, oh, tap_ch, &pre[s], &pstride);
            }
            const lvm_params c = to_c(cfg.mag, 0);
            int produced = 0;
            const int rc = lvm_chain_process_batch_ex(mag_.handle(), &cfg.pre, &c, src.data(), v0.w, v0.h, v0.channels, v0.stride,
                                                      dst.data(), dstride, stages_identity ? nullptr : pre.data(), pstride, &produced);
            if (rc != LVM_OK) throw Error(rc, std::string("lvm: ") + lvm_last_error(mag_.handle()));
            for (std::size_t s = 0; s < n; ++s) {
                // MagnificationProcessor.cpp:61: on passthrough the chain hands the magnifier's INPUT on
                FrameRef cur = produced ? processed[s] : original[s];
                T::publish(*out_[s], std::move(cur), original[s]);            // :46-49
            }
        } catch (const std::exception&) {
            // a stage threw: reset the stateful stages and show the input (ProcessingChain.cpp:50-58)
            errors_.fetch_add(1, std::memory_order_acq_rel);
            T::on_error(instr_);
            (void)lvm_reset(mag_.handle());
            for (std::size_t s = 0; s < n; ++s) T::publish(*out_[s], frames[s], frames[s]);
        } catch (...) {
            // anything else: count it, reset, publish nothing (ProcessingChain.cpp:59-62)
            errors_.fetch_add(1, std::memory_order_acq_rel);
            T::on_error(instr_);
            (void)lvm_reset(mag_.handle());
        }
        for (std::size_t s = 0; s < n; ++s) T::on_processed(instr_, frames[s]);   // :64-69
        ticks_.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }

private:
    void run() {
        while (!stop_.load(std::memory_order_acquire))
            if (!tick()) break;
    }

    std::vector<typename T::Queue*> in_;
    std::vector<typename T::Mailbox*> out_;
    typename T::Instr* instr_;
    typename T::Config* config_;
    Magnifier mag_;
    std::thread thread_;
    std::atomic<bool> stop_{false};
    std::atomic<std::uint64_t> ticks_{0}, errors_{0};
};

}  // namespace lvm

// ---- binding to the reference's types (compiles inside the reference tree only) -------------------------------
#if defined(LVM_WITH_LIVIM_HEADERS)
#include <chrono>
#include <opencv2/core.hpp>

#include "core/AtomicConfig.hpp"
#include "core/Instrumentation.hpp"
#include "core/LatestFrameMailbox.hpp"
#include "core/PipelineTypes.hpp"
#include "processing/IProcessor.hpp"

namespace livim {

struct LivimBatchTraits {
    using FrameRef = livim::FrameRef;
    struct View { const std::uint8_t* data; int w, h, channels; std::ptrdiff_t stride; };
    static View view(const FrameRef& f) {
        const cv::Mat& m = f->image;
        return View{m.data, m.cols, m.rows, m.channels(), static_cast<std::ptrdiff_t>(m.step)};
    }
    static FrameRef make_like(const FrameRef& in, int w, int h, int channels, std::uint8_t** data, std::ptrdiff_t* stride) {
        auto out = std::make_shared<Frame>(*in);                              // metadata copied (MagnificationProcessor.cpp:63-66)
        out->image = cv::Mat(h, w, channels == 1 ? CV_8UC1 : CV_8UC3);        // fresh buffer, never aliases in->image
        out->width = w; out->height = h;                                      // PreprocessProcessor.cpp:46-49
        out->format = channels >= 3 ? PixelFormat::BGR8 : PixelFormat::Gray8; // GrayscaleProcessor.cpp:14
        *data = out->image.data; *stride = static_cast<std::ptrdiff_t>(out->image.step);
        return out;
    }
    using Queue = FrameQueue;
    static bool pop(Queue& q, FrameRef& f) { return q.pop(f); }
    static void stop(Queue& q) { q.stop(); }
    using Mailbox = LatestFrameMailbox;
    static void publish(Mailbox& m, FrameRef processed, FrameRef original) {
        auto pair = std::make_shared<DisplayFrame>();
        pair->processed = std::move(processed);
        pair->original = std::move(original);
        m.publish(std::move(pair));
    }
    using Config = AtomicConfig<ProcessorConfig>;
    struct Snapshot { lvm_preprocess_params pre; lvm::MagnificationParams mag; };
    static Snapshot read(Config& c) {
        const std::shared_ptr<const ProcessorConfig> cfg = c.read();
        Snapshot s{};
        s.pre.downscale = cfg->preprocess.downscale; s.pre.roi_enabled = cfg->preprocess.roiEnabled ? 1 : 0;
        s.pre.roiX = cfg->preprocess.roiX; s.pre.roiY = cfg->preprocess.roiY; s.pre.roiW = cfg->preprocess.roiW; s.pre.roiH = cfg->preprocess.roiH;
        s.pre.grayscale = cfg->grayscale ? 1 : 0;
        const MagnificationParams& p = cfg->magnification;
        s.mag.mode = static_cast<lvm::MagnificationMode>(static_cast<int>(p.mode));
        s.mag.amplification = p.amplification; s.mag.coWavelength = p.coWavelength; s.mag.coLow = p.coLow; s.mag.coHigh = p.coHigh;
        s.mag.chromAttenuation = p.chromAttenuation; s.mag.levels = p.levels; s.mag.framerate = p.framerate;
        return s;
    }
    using Instr = Instrumentation;
    static void on_error(Instr* i) { if (i) i->onProcessingError(); }
    static void on_processed(Instr* i, const FrameRef& f) {
        if (!i) return;
        i->onProcessed();
        i->recordLatency(std::chrono::duration<double, std::milli>(now() - f->captureTs).count());
    }
};

using HipBatchedProcessingChain = lvm::BatchedChain<LivimBatchTraits>;

}  // namespace livim
#endif
