// lvm.hpp -- header-only C++17 wrapper over the C ABI (include/lvm_hip.h): RAII context, the
// reference's parameter structs by the reference's names, exceptions instead of status codes.
// No OpenCV / Qt dependency; HipMagnificationProcessor.hpp builds the IProcessor shim on top.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "lvm_hip.h"

namespace lvm {

// reference: processing/IProcessor.hpp:10
enum class MagnificationMode { Laplace = LVM_MODE_LAPLACE, Phase = LVM_MODE_PHASE, Color = LVM_MODE_COLOR, None = LVM_MODE_NONE };

// reference: processing/IProcessor.hpp:14-23 (same field names, same defaults)
struct MagnificationParams {
    MagnificationMode mode = MagnificationMode::Laplace;
    double amplification = 0.0;
    double coWavelength = 0.0;
    double coLow = 0.0;
    double coHigh = 0.0;
    double chromAttenuation = 0.0;
    int levels = 4;
    double framerate = 30.0;
};

inline lvm_params to_c(const MagnificationParams& p, std::uint64_t preprocess_key) {
    lvm_params c{};
    c.mode = static_cast<std::int32_t>(p.mode);
    c.levels = p.levels;
    c.amplification = p.amplification;
    c.coWavelength = p.coWavelength;
    c.coLow = p.coLow;
    c.coHigh = p.coHigh;
    c.chromAttenuation = p.chromAttenuation;
    c.framerate = p.framerate;
    c.preprocess_key = preprocess_key;
    return c;
}

class Error : public std::runtime_error {
public:
    Error(int status, const std::string& what) : std::runtime_error(what), status_(status) {}
    int status() const { return status_; }
private:
    int status_;
};

// One magnifier instance == one reference MagnificationProcessor (processing/MagnificationProcessor.hpp:13-23).
class Magnifier {
public:
    explicit Magnifier(int device = 0, int n_streams = 1) {
        const int rc = lvm_create(device, n_streams, &ctx_);
        if (rc != LVM_OK) throw Error(rc, "lvm_create failed (no usable MI355X / HIP runtime; there is no CPU fallback)");
    }
    ~Magnifier() { lvm_destroy(ctx_); }
    Magnifier(const Magnifier&) = delete;
    Magnifier& operator=(const Magnifier&) = delete;
    Magnifier(Magnifier&& o) noexcept : ctx_(std::exchange(o.ctx_, nullptr)) {}

    // IProcessor::reset (processing/IProcessor.hpp:56-59)
    void reset() { check(lvm_reset(ctx_)); }

    // Host frames.  Returns false when the reference would return the input frame unchanged.
    bool process(const MagnificationParams& p, std::uint64_t preprocess_key, const std::uint8_t* in, int w, int h,
                 int channels, std::ptrdiff_t in_stride, std::uint8_t* out, std::ptrdiff_t out_stride) {
        const lvm_params c = to_c(p, preprocess_key);
        int produced = 0;
        check(lvm_process(ctx_, &c, in, w, h, channels, in_stride, out, out_stride, &produced));
        return produced != 0;
    }

    // Device frames of all streams, enqueued on `stream` (hipStream_t) without synchronisation.
    bool process_device(const MagnificationParams& p, std::uint64_t preprocess_key, const std::uint8_t* d_in, int w,
                        int h, int channels, std::ptrdiff_t in_stride, std::ptrdiff_t in_stream_stride,
                        std::uint8_t* d_out, std::ptrdiff_t out_stride, std::ptrdiff_t out_stream_stride,
                        void* stream = nullptr) {
        const lvm_params c = to_c(p, preprocess_key);
        int produced = 0;
        check(lvm_process_device(ctx_, &c, d_in, w, h, channels, in_stride, in_stream_stride, d_out, out_stride,
                                 out_stream_stride, &produced, stream));
        return produced != 0;
    }

    // Preprocess -> Grayscale -> Magnification on a host frame (runChainOnce, processing/ChainBuilder.cpp:19-29).
    // `out` must hold the geometry chain_geometry() reports; it receives the magnified frame, or the
    // preprocessed frame when the magnifier passes its input through (return value false).
    bool chain_process(const lvm_preprocess_params& pre, const MagnificationParams& p, const std::uint8_t* in, int w, int h,
                       int channels, std::ptrdiff_t in_stride, std::uint8_t* out, std::ptrdiff_t out_stride) {
        const lvm_params c = to_c(p, 0);
        int produced = 0;
        check(lvm_chain_process(ctx_, &pre, &c, in, w, h, channels, in_stride, out, out_stride, &produced));
        return produced != 0;
    }
    // size and channel count of the frame the chain emits for a w x h x channels input
    static void chain_geometry(const lvm_preprocess_params& pre, int w, int h, int channels, int* ow, int* oh, int* och) {
        if (lvm_preprocess_geometry(&pre, w, h, channels, nullptr, nullptr, nullptr, nullptr, ow, oh, och) != LVM_OK)
            throw Error(LVM_ERR_INVALID, "lvm_preprocess_geometry: invalid arguments");
    }

    void synchronize() { check(lvm_synchronize(ctx_)); }
    lvm_ctx* handle() const { return ctx_; }

private:
    void check(int rc) {
        if (rc != LVM_OK) throw Error(rc, std::string("lvm: ") + lvm_last_error(ctx_));
    }
    lvm_ctx* ctx_ = nullptr;
};

// Page-locked output buffers, recycled: what the shims hand lvm_process as `out`, so that its last kernel writes the frame straight
// into the caller's buffer over PCIe (no staging copy, no download; include/lvm_hip.h lvm_host_alloc).  A buffer returns to the pool
// when its last user drops it -- the same rule the reference's FramePool (core/FramePool.cpp:29-36) gives its pooled frames: nobody
// may keep a cv::Mat header of the pixels longer than the FrameRef.  A size change starts a new generation; the old buffers are freed
// as they come back.
class PinnedPool {
public:
    std::shared_ptr<std::uint8_t> acquire(std::size_t bytes) {
        if (!core_ || core_->bytes != bytes) { core_ = std::make_shared<Core>(); core_->bytes = bytes; }
        void* p = nullptr;
        {
            std::lock_guard<std::mutex> lg(core_->m);
            if (!core_->free.empty()) { p = core_->free.back(); core_->free.pop_back(); }
        }
        if (!p && lvm_host_alloc(bytes, &p) != LVM_OK) return nullptr;     // caller falls back to pageable memory
        std::shared_ptr<Core> core = core_;
        return std::shared_ptr<std::uint8_t>(static_cast<std::uint8_t*>(p), [core](std::uint8_t* q) {
            std::lock_guard<std::mutex> lg(core->m);
            core->free.push_back(q);
        });
    }
private:
    struct Core {
        std::mutex m; std::vector<void*> free; std::size_t bytes = 0;
        ~Core() { for (void* p : free) lvm_host_free(p); }
    };
    std::shared_ptr<Core> core_;
};

}  // namespace lvm
