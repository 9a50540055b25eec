// lvm_api.hip -- C ABI of liblvm_hip.so (see include/lvm_hip.h) and the host-side logic of
// the reference's MagnificationProcessor (processing/MagnificationProcessor.cpp:10-67):
// level clamping, structural-change reset (MagnifyCore.hpp:45-80), dispatch by mode and the
// passthrough rules.  No CPU fallback exists: without a usable HIP device every entry point
// fails with LVM_ERR_NO_DEVICE.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "lvm_internal.h"

struct lvm_ctx : lvm::Ctx {};

namespace lvm {

void prof_begin(Ctx* c, const char* name, hipStream_t s) {
    c->prof_skip = !c->prof_only.empty() && c->prof_only != name;     // lvm_profile_only: every other launch runs unbracketed
    if (c->prof_skip) return;
    int id = -1;
    for (size_t i = 0; i < c->prof_totals.size(); ++i)
        if (c->prof_totals[i].name == name) { id = (int)i; break; }
    if (id < 0) { ProfTotal t; t.name = name; c->prof_totals.push_back(t); id = (int)c->prof_totals.size() - 1; }
    ProfEvent e; e.name = id;
    (void)hipEventCreate(&e.e0); (void)hipEventCreate(&e.e1);
    (void)hipEventRecord(e.e0, s);
    c->prof_events.push_back(e);
}
void prof_end(Ctx* c, hipStream_t s) { if (!c->prof_skip) (void)hipEventRecord(c->prof_events.back().e1, s); }

// Waits for everything this context has enqueued -- on its own two streams and on caller streams: one event PER DISTINCT
// caller stream, re-recorded after every enqueue on it (an event outlives the stream it was recorded on).  Not a
// device-wide synchronisation: the reference runs a live chain and an export chain side by side
// (export/Exporter.cpp:204,231) and one context's reset must not stall the other.  A caller that cycles through more
// than kMaxCallerStreams streams gets hipDeviceSynchronize instead (correct, just wider).
constexpr size_t kMaxCallerStreams = 8;
void sync_streams(Ctx* c) {
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
    for (auto& e : c->caller_events) (void)hipEventSynchronize(e.second);
    // the device-wide wait has covered everything enqueued so far, on any stream: the flag starts afresh (it is not sticky)
    if (c->caller_overflow) { (void)hipDeviceSynchronize(); c->caller_overflow = false; }
    (void)hipGetLastError();
}
void mark_enqueued(Ctx* c, hipStream_t s) {
    if (s == c->own_stream) return;                       // the own stream is synchronised directly
    for (auto& e : c->caller_events)
        if (e.first == s) { if (hipEventRecord(e.second, s) != hipSuccess) { (void)hipGetLastError(); c->caller_overflow = true; } return; }
    hipEvent_t ev = nullptr;
    if (c->caller_events.size() >= kMaxCallerStreams) {
        // full: recycle a slot whose event has completed (its stream's work is done -- the stream may not even exist any
        // more; stream-per-call callers and framework stream pools would otherwise fill the list for good)
        for (size_t i = 0; i < c->caller_events.size(); ++i) {
            auto& e = c->caller_events[i];
            if (hipEventQuery(e.second) == hipSuccess) {
                (void)hipGetLastError();    // hipErrorNotReady of the earlier slots' queries must not surface as the next call's error
                if (hipEventRecord(e.second, s) != hipSuccess) {
                    // the slot no longer describes a live enqueue: drop it (its event with it) instead of keeping a stale stream
                    (void)hipGetLastError();
                    (void)hipEventDestroy(e.second);
                    c->caller_events.erase(c->caller_events.begin() + (ptrdiff_t)i);
                    c->caller_overflow = true;
                    return;
                }
                e.first = s;
                return;
            }
        }
        (void)hipGetLastError();        // hipErrorNotReady of the queries
        c->caller_overflow = true;
        return;
    }
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->caller_overflow = true; return; }
    if (hipEventRecord(ev, s) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(ev); c->caller_overflow = true; return; }
    c->caller_events.emplace_back(s, ev);
}

static void drop_state(Ctx* c) { delete c->state; c->state = nullptr; }

// (re)build the device layouts of the forward Lab table from c->lab_lut_compact
int upload_lab_lut(Ctx* c) {
    std::vector<uint32_t> ab; std::vector<int16_t> lcells;
    lab_lut_device_tables(c->lab_lut_compact.data(), ab, lcells);
    if (!c->d_lab_ab && hipMalloc((void**)&c->d_lab_ab, ab.size() * sizeof(uint32_t)) != hipSuccess) { c->d_lab_ab = nullptr; c->err = "hipMalloc (Lab table) failed"; return LVM_ERR_OOM; }
    if (!c->d_lab_Lcells && hipMalloc((void**)&c->d_lab_Lcells, lcells.size() * sizeof(int16_t)) != hipSuccess) { c->d_lab_Lcells = nullptr; c->err = "hipMalloc (Lab table) failed"; return LVM_ERR_OOM; }
    LVM_HIP_TRY(c, hipMemcpy(c->d_lab_ab, ab.data(), ab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    LVM_HIP_TRY(c, hipMemcpy(c->d_lab_Lcells, lcells.data(), lcells.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    c->lab_lut.ab = c->d_lab_ab; c->lab_lut.Lcells = c->d_lab_Lcells;
    return LVM_OK;
}

static int run_mode(Ctx* c, const lvm_params* p, int levels, const FrameIO& io, hipStream_t s, int* produced) {
    switch (p->mode) {                                                                  // MagnificationProcessor.cpp:48-60
    case LVM_MODE_LAPLACE: return laplace_process(c, *p, levels, io, s, produced);
    case LVM_MODE_PHASE:   return riesz_process(c, *p, levels, io, s, produced);
    case LVM_MODE_COLOR:   return color_process(c, *p, levels, io, s, produced);
    default: break;
    }
    return LVM_OK;
}

// A failed call (allocation failure, launch error) must not leave a half-built state installed: the next call
// would launch kernels on null buffers.  Drop it and disarm the tracker so that the next frame starts afresh,
// which is also what the reference's recovery path does (ProcessingChain.cpp:50-62 resets every stage).
static void tracker_disable(Ctx* c);
void fail_state(Ctx* c, hipStream_t s) {
    (void)hipStreamSynchronize(s);
    sync_streams(c);
    drop_state(c);
    tracker_disable(c);
}
static void tracker_disable(Ctx* c) { c->t_mode = LVM_MODE_NONE; c->t_levels = -1; c->t_channels = -1; c->t_w = c->t_h = 0; }

int ensure_float(Ctx* c, size_t count) {
    if (count > c->float_cap) {
        sync_streams(c);            // kernels of earlier calls may still be writing the kept frame
        if (c->d_float) (void)hipFree(c->d_float);
        c->d_float = nullptr; c->float_cap = 0;
        LVM_HIP_TRY(c, hipMalloc((void**)&c->d_float, count * sizeof(float)));
        c->float_cap = count;
    }
    c->float_count = count;
    return LVM_OK;
}

// MagnificationProcessor::process (MagnificationProcessor.cpp:17-67) on device buffers
static int process_device(Ctx* c, const lvm_params* p, const FrameIO& io, hipStream_t s, int* produced) {
    *produced = 0;
    if (p->mode == LVM_MODE_NONE || io.d_in == nullptr || io.w <= 0 || io.h <= 0) {      // :21-29
        if (c->t_mode != LVM_MODE_NONE) {
            // pipelined mode: the pending frame's output is owed to its caller (same rule as on a structural change)
            if (c->t_mode == LVM_MODE_LAPLACE) (void)laplace_flush(c, s);
            LVM_HIP_TRY(c, hipStreamSynchronize(s));
            sync_streams(c);
            drop_state(c); tracker_disable(c);
        }
        return LVM_OK;
    }
    if (p->mode < 0 || p->mode > LVM_MODE_NONE) { c->err = "invalid mode"; return LVM_ERR_INVALID; }
    if (io.channels != 1 && io.channels != 3) { c->err = "channels must be 1 or 3"; return LVM_ERR_INVALID; }
    if (io.d_out == nullptr) { c->err = "null output"; return LVM_ERR_INVALID; }
    const int maxLevels = max_levels(io.w, io.h);                                       // :32-33
    if (maxLevels < 1) return LVM_OK;
    int levels = p->levels < 1 ? 1 : (p->levels > maxLevels ? maxLevels : p->levels);   // :34
    if (levels > kMaxLevels) levels = kMaxLevels;
    const bool change = p->mode != c->t_mode || levels != c->t_levels || io.w != c->t_w || io.h != c->t_h ||
                        io.channels != c->t_channels || p->preprocess_key != c->t_pre;  // MagnifyCore.hpp:53-65
    if (change) {
        // buffers of the old geometry may still be in use by queued kernels
        if (c->t_mode == LVM_MODE_LAPLACE) (void)laplace_flush(c, s);   // pipelined mode: do not lose the pending frame
        LVM_HIP_TRY(c, hipStreamSynchronize(s));
        sync_streams(c);                                                 // earlier calls may have used other streams
        c->t_mode = p->mode; c->t_levels = levels; c->t_w = io.w; c->t_h = io.h;
        c->t_channels = io.channels; c->t_pre = p->preprocess_key;
        drop_state(c);                                                                  // :39-43
    }
    if (c->keep_float) {
        const int rc = ensure_float(c, (size_t)io.w * io.h * io.channels);
        if (rc != LVM_OK) return rc;
    }
    const int rc = run_mode(c, p, levels, io, s, produced);
    mark_enqueued(c, s);
    if (rc != LVM_OK) fail_state(c, s);
    return rc;
}

}  // namespace lvm

using lvm::Ctx;

extern "C" {

int lvm_create(int device, int n_streams, lvm_ctx** out) {
    if (!out || n_streams < 1) return LVM_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return LVM_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LVM_ERR_NO_DEVICE;
    lvm_ctx* c = new (std::nothrow) lvm_ctx();
    if (!c) return LVM_ERR_OOM;
    c->device = device;
    c->nstreams = n_streams;
    { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) c->num_cus = n; }
    float g[256] = {}, ig[4096] = {};
    lvm::build_lab_tables(g, ig, c->lab.fwd, c->lab.inv);
    for (int i = 0; i < 9; ++i) { c->lab.inv1024[i] = c->lab.inv[i] * 1024.0f; c->lab.inv4096[i] = c->lab.inv[i] * 4096.0f; }
    std::vector<uint32_t> steps(2 * lvm::kU8StepSlices);
    if (!lvm::build_u8_steps(ig, steps.data())) { delete c; return LVM_ERR_INVALID; }      // (cannot happen with OpenCV's spline: checked by the tests)
    bool ok = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_gamma_u8, sizeof(g)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_invgamma, sizeof(ig)) == hipSuccess;
    ok = ok && hipMemcpy(c->d_gamma_u8, g, sizeof(g), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(c->d_invgamma, ig, sizeof(ig), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_u8steps, steps.size() * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemcpy(c->d_u8steps, steps.data(), steps.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        // OpenCV's forward Lab table (lab_tables.cpp) and the closed form of its cell index (lab_lut.h)
        if (!lvm::lab_lut_fine_index_ok()) { lvm_destroy(c); return LVM_ERR_INVALID; }
        lvm::build_lab_lut_compact(c->lab_lut_compact);
        const int rc = lvm::upload_lab_lut(c);
        if (rc != LVM_OK) { lvm_destroy(c); return rc; }                 // LVM_ERR_OOM stays LVM_ERR_OOM
    }
    if (!ok) { lvm_destroy(c); return LVM_ERR_HIP; }
    c->lab.gamma_u8 = c->d_gamma_u8;
    c->lab.invgamma = c->d_invgamma;
    c->lab.u8steps = c->d_u8steps;
    c->lab.a255 = (float)(1.0 / 255.0f);
    *out = c;
    return LVM_OK;
}

void lvm_destroy(lvm_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    lvm::sync_streams(c);
    delete c->state; c->state = nullptr;
    for (auto& e : c->prof_events) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); }
    if (c->d_gamma_u8) (void)hipFree(c->d_gamma_u8);
    if (c->d_invgamma) (void)hipFree(c->d_invgamma);
    if (c->d_u8steps) (void)hipFree(c->d_u8steps);
    lvm::overlay_release(c);
    if (c->probe_running) { double m = 0; (void)lvm::clock_probe_stop(c, &m, nullptr); }
    if (c->h_probe) (void)hipHostFree(c->h_probe);
    if (c->d_lab_ab) (void)hipFree(c->d_lab_ab);
    if (c->d_lab_Lcells) (void)hipFree(c->d_lab_Lcells);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_float) (void)hipFree(c->d_float);
    lvm::preprocess_release(c);
    lvm::mjpeg_release(c);
    lvm::mjpeg_decode_release(c);
    if (c->d_pre_in) (void)hipFree(c->d_pre_in);
    if (c->d_pre_out) (void)hipFree(c->d_pre_out);
    if (c->d_chain_out) (void)hipFree(c->d_chain_out);
    c->d_pre_in = c->d_pre_out = c->d_chain_out = nullptr; c->pre_in_cap = c->pre_out_cap = c->chain_out_cap = 0;
    if (c->d_canvas) (void)hipFree(c->d_canvas);
    c->d_canvas = nullptr; c->canvas_cap = 0;
    if (c->d_pre_tap) (void)hipFree(c->d_pre_tap);
    c->d_pre_tap = nullptr; c->pre_tap_cap = 0;
    for (auto e : c->ev_up) (void)hipEventDestroy(e);
    for (auto e : c->ev_done) (void)hipEventDestroy(e);
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
    for (auto& e : c->caller_events) (void)hipEventDestroy(e.second);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int lvm_reset(lvm_ctx* c) {                       // MagnificationProcessor.cpp:10-15
    if (!c) return LVM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    // (an owed pipelined frame is discarded: reset() means "forget everything", MagnificationProcessor.cpp:10-15)
    lvm::sync_streams(c);
    lvm::drop_state(c);
    lvm::tracker_disable(c);
    c->t_pre = 0;
    c->err.clear();
    return LVM_OK;
}

int lvm_process_device(lvm_ctx* c, const lvm_params* p, const uint8_t* d_in, int w, int h, int channels,
                       ptrdiff_t in_stride, ptrdiff_t in_stream_stride, uint8_t* d_out, ptrdiff_t out_stride,
                       ptrdiff_t out_stream_stride, int* produced, void* hip_stream) {
    if (!c || !p || !produced) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    lvm::FrameIO io{d_in, in_stride, in_stream_stride, d_out, out_stride, out_stream_stride, w, h, channels};
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return lvm::process_device(c, p, io, s, produced);
}

int lvm_process_device_frames(lvm_ctx* c, const lvm_params* p, int n_frames, const uint8_t* d_in, int w, int h, int channels,
                              ptrdiff_t in_stride, ptrdiff_t in_stream_stride, ptrdiff_t in_frame_stride, uint8_t* d_out,
                              ptrdiff_t out_stride, ptrdiff_t out_stream_stride, ptrdiff_t out_frame_stride, int* produced,
                              void* hip_stream) {
    if (!c || !p || !produced || n_frames < 1) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    if (c->keep_float && w > 0 && h > 0 && (channels == 1 || channels == 3)) {   // the batched schedules keep the first frame of a batch
        const int rc = lvm::ensure_float(c, (size_t)w * h * channels);
        if (rc != LVM_OK) return rc;
    }
    int f = 0;
    while (f < n_frames) {
        lvm::FrameIO io{d_in ? d_in + (size_t)f * in_frame_stride : nullptr, in_stride, in_stream_stride,
                        d_out ? d_out + (size_t)f * out_frame_stride : nullptr, out_stride, out_stream_stride, w, h, channels};
        // temporal batch: same structural key as the tracked state, steady Laplace state, frames laid
        // out [frame][stream]; everything else (first frames, other modes, odd layouts) goes frame by frame
        const int left = n_frames - f;
        const int maxL = lvm::max_levels(w, h);
        const int lv = maxL < 1 ? 0 : (p->levels < 1 ? 1 : (p->levels > maxL ? maxL : p->levels));
        const bool same = p->mode == c->t_mode && lv == c->t_levels && w == c->t_w && h == c->t_h && channels == c->t_channels &&
                          p->preprocess_key == c->t_pre && d_in && d_out;
        const bool layout = in_frame_stride == in_stream_stride * c->nstreams && out_frame_stride == out_stream_stride * c->nstreams;
        if (left >= 2 && same && layout) {
            int rc = 1;
            if (p->mode == LVM_MODE_LAPLACE && lvm::laplace_can_batch(c)) rc = lvm::laplace_process_frames(c, *p, io, left, s);
            else if (p->mode == LVM_MODE_PHASE && channels >= 3 && lvm::riesz_can_batch(c, *p)) rc = lvm::riesz_process_frames(c, *p, io, left, s);
            else if (p->mode == LVM_MODE_COLOR && lvm::color_can_batch(c, *p, left < lvm::kColorBatchMax ? left : lvm::kColorBatchMax)) {
                const int nb = left < lvm::kColorBatchMax ? left : lvm::kColorBatchMax;      // the window ring keeps that many spare slots
                rc = lvm::color_process_frames(c, *p, io, nb, s);
                if (rc == LVM_OK) { lvm::mark_enqueued(c, s); for (int k = f; k < f + nb; ++k) produced[k] = 1; f += nb; continue; }
            }
            if (rc <= 0) {
                lvm::mark_enqueued(c, s);
                if (rc != LVM_OK) { lvm::fail_state(c, s); return rc; }
                for (int k = f; k < n_frames; ++k) produced[k] = 1;
                return LVM_OK;
            }
        }
        const int rc = lvm::process_device(c, p, io, s, &produced[f]);
        if (rc != LVM_OK) return rc;
        ++f;
    }
    return LVM_OK;
}


int lvm_preprocess_geometry(const lvm_preprocess_params* pp, int w, int h, int channels, int* rx, int* ry, int* rw, int* rh,
                            int* ow, int* oh, int* och) {
    if (!pp || w <= 0 || h <= 0 || (channels != 1 && channels != 3)) return LVM_ERR_INVALID;
    int v[7];
    lvm::preprocess_geometry(*pp, w, h, channels, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6]);
    int* dst[7] = {rx, ry, rw, rh, ow, oh, och};
    for (int i = 0; i < 7; ++i) if (dst[i]) *dst[i] = v[i];
    return LVM_OK;
}

int lvm_preprocess_device(lvm_ctx* c, const lvm_preprocess_params* pp, const uint8_t* d_in, int w, int h, int channels,
                          ptrdiff_t in_stride, ptrdiff_t in_stream_stride, uint8_t* d_out, ptrdiff_t out_stride,
                          ptrdiff_t out_stream_stride, void* hip_stream) {
    if (!c || !pp) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    const int rc = lvm::preprocess_device(c, *pp, d_in, w, h, channels, in_stride, in_stream_stride, d_out, out_stride, out_stream_stride, s);
    lvm::mark_enqueued(c, s);
    return rc;
}

int lvm_compose_geometry(int split, int ow, int oh, int pw, int ph, int* pane_w, int* pane_h, int* canvas_w, int* canvas_h) {
    if (split < LVM_SPLIT_NONE || split > LVM_SPLIT_TOP_BOTTOM) return LVM_ERR_INVALID;
    int v[4];
    (void)lvm::compose_geometry(split, ow, oh, pw, ph, &v[0], &v[1], &v[2], &v[3]);
    int* dst[4] = {pane_w, pane_h, canvas_w, canvas_h};
    for (int i = 0; i < 4; ++i) if (dst[i]) *dst[i] = v[i];
    return LVM_OK;
}

int lvm_compose_device(lvm_ctx* c, int split, const uint8_t* d_orig, int ow, int oh, int och, ptrdiff_t orig_stride,
                       ptrdiff_t orig_stream_stride, const uint8_t* d_proc, int pw, int ph, int pch, ptrdiff_t proc_stride,
                       ptrdiff_t proc_stream_stride, uint8_t* d_canvas, ptrdiff_t canvas_stride, ptrdiff_t canvas_stream_stride,
                       void* hip_stream) {
    if (!c) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    const int rc = lvm::compose_device(c, split, d_orig, ow, oh, och, orig_stride, orig_stream_stride, d_proc, pw, ph, pch, proc_stride,
                                       proc_stream_stride, d_canvas, canvas_stride, canvas_stream_stride, s);
    lvm::mark_enqueued(c, s);
    return rc;
}

// ---- spatial tiling of ONE Riesz stream (demonstrator; the production answer to several GPUs is one stream per GPU) ----------------------
static int tile_check(lvm_ctx* c, const lvm_params* p, int w, int h) {
    if (!c || !p) return LVM_ERR_INVALID;
    if (p->mode != LVM_MODE_PHASE || c->nstreams != 1 || w < 1 || h < 1) { c->err = "tiling: the Riesz mode on a 1-stream context"; return LVM_ERR_INVALID; }
    return LVM_OK;
}

int lvm_tile_riesz_stage1(lvm_ctx* c, const lvm_params* p, const uint8_t* d_in, int w, int h, ptrdiff_t in_stride, int* produced,
                          float* d_residual_out, int* residual_w, int* residual_h, void* hip_stream) {
    int rc = tile_check(c, p, w, h);
    if (rc != LVM_OK) return rc;
    if (!d_in || !produced || in_stride < (ptrdiff_t)w * 3) { c->err = "tiling: bad frame arguments"; return LVM_ERR_INVALID; }
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    // (no output in this stage: the frame argument stands in for the output pointer the per-frame path asks for and is never written)
    lvm::FrameIO io{d_in, in_stride, in_stride * h, const_cast<uint8_t*>(d_in), in_stride, in_stride * h, w, h, 3};
    c->tile_mode = 1;
    rc = lvm::process_device(c, p, io, s, produced);
    c->tile_mode = 0;
    if (rc != LVM_OK) return rc;
    return lvm::riesz_tile_residual(c, d_residual_out, residual_w, residual_h, s);
}

int lvm_tile_riesz_stage2(lvm_ctx* c, const lvm_params* p, const uint8_t* d_in, int w, int h, ptrdiff_t in_stride, const float* d_residual_in,
                          uint8_t* d_out, ptrdiff_t out_stride, void* hip_stream) {
    int rc = tile_check(c, p, w, h);
    if (rc != LVM_OK) return rc;
    if (!d_in || !d_out || !d_residual_in || in_stride < (ptrdiff_t)w * 3 || out_stride < (ptrdiff_t)w * 3) { c->err = "tiling: bad frame arguments"; return LVM_ERR_INVALID; }
    if (c->t_mode != LVM_MODE_PHASE || c->t_w != w || c->t_h != h) { c->err = "tiling: stage 2 without a matching stage 1"; return LVM_ERR_INVALID; }
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    lvm::FrameIO io{d_in, in_stride, in_stride * h, d_out, out_stride, out_stride * h, w, h, 3};
    rc = lvm::riesz_tile_finish(c, *p, io, d_residual_in, s);
    lvm::mark_enqueued(c, s);
    return rc;
}

int lvm_tile_riesz_planes(lvm_ctx* c, const lvm_params* p, const float* d_plane_in, int w, int h, float* d_plane_out, int* produced, void* hip_stream) {
    int rc = tile_check(c, p, w, h);
    if (rc != LVM_OK) return rc;
    if (!d_plane_in || !d_plane_out || !produced || p->levels < 2) { c->err = "tiling: planes in and out, at least two levels"; return LVM_ERR_INVALID; }
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    // the per-frame path with a plane where the frame would be: three "channels" so that the Riesz mode accepts it (MagnifyCore.hpp:212);
    // neither pointer is touched as bytes (tile_mode 2: riesz.hip rz_build / rz_collapse_out)
    lvm::FrameIO io{reinterpret_cast<const uint8_t*>(d_plane_in), (ptrdiff_t)w * 3, (ptrdiff_t)w * 3 * h, reinterpret_cast<uint8_t*>(d_plane_out), (ptrdiff_t)w * 3,
                    (ptrdiff_t)w * 3 * h, w, h, 3};
    c->tile_mode = 2; c->tile_plane_in = d_plane_in; c->tile_plane_out = d_plane_out;
    rc = lvm::process_device(c, p, io, s, produced);
    c->tile_mode = 0; c->tile_plane_in = nullptr; c->tile_plane_out = nullptr;
    return rc;
}

int lvm_export_set_overlay(lvm_ctx* c, int n_labels, const lvm_overlay_label* labels) {
    if (!c) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    return lvm::overlay_set(c, n_labels, labels);
}

int lvm_overlay_device(lvm_ctx* c, uint8_t* d_canvas, int canvas_w, int canvas_h, ptrdiff_t canvas_stride, ptrdiff_t frame_stride, int n_frames, void* hip_stream) {
    if (!c) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    const int rc = lvm::overlay_device(c, d_canvas, canvas_w, canvas_h, canvas_stride, frame_stride, n_frames, s);
    lvm::mark_enqueued(c, s);
    return rc;
}

// FNV-1a over the PreprocessParams fields the reference compares (IProcessor.hpp:36-39)
static uint64_t preprocess_key_of(const lvm_preprocess_params& pp) {
    uint64_t k = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { k ^= b[i]; k *= 1099511628211ull; } };
    const int32_t en = pp.roi_enabled ? 1 : 0;
    mix(&pp.downscale, 4); mix(&en, 4); mix(&pp.roiX, 4); mix(&pp.roiY, 4); mix(&pp.roiW, 4); mix(&pp.roiH, 4);
    return k;
}

int lvm_chain_process_batch(lvm_ctx* c, const lvm_preprocess_params* pp, const lvm_params* p, const uint8_t* const* in, int w, int h,
                            int channels, ptrdiff_t in_stride, uint8_t* const* out, ptrdiff_t out_stride, int* produced) {
    return lvm_chain_process_batch_ex(c, pp, p, in, w, h, channels, in_stride, out, out_stride, nullptr, 0, produced);
}

int lvm_chain_process_batch_ex(lvm_ctx* c, const lvm_preprocess_params* pp, const lvm_params* p, const uint8_t* const* in, int w, int h,
                               int channels, ptrdiff_t in_stride, uint8_t* const* out, ptrdiff_t out_stride, uint8_t* const* pre_out,
                               ptrdiff_t pre_stride, int* produced) {
    if (!c || !pp || !p || !produced || !in || !out) return LVM_ERR_INVALID;
    *produced = 0;
    const int NS = c->nstreams;
    if (w <= 0 || h <= 0 || (channels != 1 && channels != 3) || in_stride < (ptrdiff_t)w * channels) {
        c->err = "bad frame arguments"; return LVM_ERR_INVALID;
    }
    for (int s = 0; s < NS; ++s) if (!in[s] || !out[s]) { c->err = "null frame pointer"; return LVM_ERR_INVALID; }
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    int rx, ry, rw, rh, ow, oh, och;
    lvm::preprocess_geometry(*pp, w, h, channels, &rx, &ry, &rw, &rh, &ow, &oh, &och);
    if (out_stride < (ptrdiff_t)ow * och) { c->err = "output stride too small"; return LVM_ERR_INVALID; }
    const size_t roi_row = (size_t)rw * channels, roi_bytes = roi_row * rh;
    const size_t out_row = (size_t)ow * och, out_bytes = out_row * oh;
    auto reserve = [&](uint8_t*& ptr, size_t& cap, size_t need) -> int {
        if (need <= cap) return LVM_OK;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        LVM_HIP_TRY(c, hipMalloc((void**)&ptr, need));
        cap = need;
        return LVM_OK;
    };
    hipStream_t s = c->own_stream;
    LVM_HIP_TRY(c, hipStreamSynchronize(s));      // staging buffers may be replaced below
    int rc = reserve(c->d_pre_in, c->pre_in_cap, roi_bytes * NS); if (rc != LVM_OK) return rc;
    rc = reserve(c->d_pre_out, c->pre_out_cap, out_bytes * NS); if (rc != LVM_OK) return rc;
    rc = reserve(c->d_chain_out, c->chain_out_cap, out_bytes * NS); if (rc != LVM_OK) return rc;
    // only the ROI rows cross PCIe (the crop is the pitch of the 2-D copy)
    for (int k = 0; k < NS; ++k)
        LVM_HIP_TRY(c, hipMemcpy2DAsync(c->d_pre_in + (size_t)k * roi_bytes, roi_row, in[k] + (size_t)ry * in_stride + (size_t)rx * channels,
                                        (size_t)in_stride, roi_row, (size_t)rh, hipMemcpyHostToDevice, s));
    const uint8_t* mag_in = c->d_pre_in;
    const bool identity = ow == rw && oh == rh && och == channels;      // PreprocessProcessor.cpp:15, GrayscaleProcessor.cpp:8-9
    // runChainOnce's `original` is chain[0]'s output (ChainBuilder.cpp:25: the tap sits BEFORE GrayscaleProcessor): with grayscale on
    // a BGR source it is the cropped / decimated COLOUR frame, not the gray frame the magnifier sees
    const bool gray_tap = pre_out && och == 1 && channels == 3;
    const size_t tap_row = (size_t)ow * 3, tap_bytes = tap_row * oh;
    if (gray_tap) { rc = reserve(c->d_pre_tap, c->pre_tap_cap, tap_bytes * NS); if (rc != LVM_OK) return rc; }
    if (!identity) {
        lvm_preprocess_params q = *pp;
        q.roi_enabled = 0;                                               // already cropped by the copy
        rc = lvm::preprocess_device(c, q, c->d_pre_in, rw, rh, channels, (ptrdiff_t)roi_row, (ptrdiff_t)roi_bytes, c->d_pre_out,
                                    (ptrdiff_t)out_row, (ptrdiff_t)out_bytes, s, gray_tap ? c->d_pre_tap : nullptr, (ptrdiff_t)tap_row, (ptrdiff_t)tap_bytes);
        if (rc != LVM_OK) { (void)hipStreamSynchronize(s); return rc; }
        mag_in = c->d_pre_out;
    }
    lvm_params mp = *p;
    mp.preprocess_key = preprocess_key_of(*pp);
    lvm::FrameIO io{mag_in, (ptrdiff_t)out_row, (ptrdiff_t)out_bytes, c->d_chain_out, (ptrdiff_t)out_row, (ptrdiff_t)out_bytes, ow, oh, och};
    const int saved_depth = c->pipeline_depth;
    c->pipeline_depth = 0;
    rc = lvm::process_device(c, &mp, io, s, produced);
    c->pipeline_depth = saved_depth;
    if (rc != LVM_OK) { (void)hipStreamSynchronize(s); return rc; }
    const uint8_t* res = *produced ? c->d_chain_out : mag_in;
    for (int k = 0; k < NS; ++k)
        LVM_HIP_TRY(c, hipMemcpy2DAsync(out[k], (size_t)out_stride, res + (size_t)k * out_bytes, out_row, out_row, (size_t)oh, hipMemcpyDeviceToHost, s));
    if (pre_out) {     // the pre-magnification tap (runChainOnce's `original`, ChainBuilder.cpp:19-29): the display's left pane
        const uint8_t* tsrc = gray_tap ? c->d_pre_tap : mag_in;
        const size_t trow = gray_tap ? tap_row : out_row, tbytes = gray_tap ? tap_bytes : out_bytes;
        if (pre_stride < (ptrdiff_t)trow) { c->err = "pre_out stride too small"; (void)hipStreamSynchronize(s); return LVM_ERR_INVALID; }
        for (int k = 0; k < NS; ++k)
            if (pre_out[k])
                LVM_HIP_TRY(c, hipMemcpy2DAsync(pre_out[k], (size_t)pre_stride, tsrc + (size_t)k * tbytes, trow, trow, (size_t)oh, hipMemcpyDeviceToHost, s));
    }
    LVM_HIP_TRY(c, hipStreamSynchronize(s));
    return LVM_OK;
}

// runChainOnce for the LIVE DISPLAY (display half): host frame in, both frames the display shows left in DEVICE
// memory -- see include/lvm_hip.h.  The reference publishes {processed, original} to the display's mailbox (ProcessingChain.cpp:46-49)
// and DisplayWidget::uploadFrame (ui/DisplayWidget.cpp:133-152) sends both to GL textures from HOST memory every frame; with the two
// destinations being mapped GL pixel-unpack buffers (host/HipDisplayPresenter.hpp) the frames never come back over PCIe.
int lvm_chain_present(lvm_ctx* c, const lvm_preprocess_params* pp, const lvm_params* p, const uint8_t* in, int w, int h, int channels,
                      ptrdiff_t in_stride, uint8_t* d_proc, ptrdiff_t proc_stride, uint8_t* d_orig, ptrdiff_t orig_stride, int* produced) {
    if (!c || !pp || !p || !produced || !in) return LVM_ERR_INVALID;
    *produced = 0;
    if (c->nstreams != 1) { c->err = "lvm_chain_present needs a 1-stream context"; return LVM_ERR_INVALID; }
    if (w <= 0 || h <= 0 || (channels != 1 && channels != 3) || in_stride < (ptrdiff_t)w * channels) { c->err = "bad frame arguments"; return LVM_ERR_INVALID; }
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    int rx, ry, rw, rh, ow, oh, och;
    lvm::preprocess_geometry(*pp, w, h, channels, &rx, &ry, &rw, &rh, &ow, &oh, &och);
    const size_t roi_row = (size_t)rw * channels, roi_bytes = roi_row * rh;
    const size_t out_row = (size_t)ow * och, out_bytes = out_row * oh;
    const size_t tap_row = (size_t)ow * channels;                       // the tap keeps the source's channel count (ChainBuilder.cpp:25)
    if (d_proc && proc_stride < (ptrdiff_t)out_row) { c->err = "proc stride too small"; return LVM_ERR_INVALID; }
    if (d_orig && orig_stride < (ptrdiff_t)tap_row) { c->err = "orig stride too small"; return LVM_ERR_INVALID; }
    auto reserve = [&](uint8_t*& ptr, size_t& cap, size_t need) -> int {
        if (need <= cap) return LVM_OK;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        LVM_HIP_TRY(c, hipMalloc((void**)&ptr, need));
        cap = need;
        return LVM_OK;
    };
    hipStream_t s = c->own_stream;
    LVM_HIP_TRY(c, hipStreamSynchronize(s));      // staging buffers may be replaced below
    int rc = reserve(c->d_pre_in, c->pre_in_cap, roi_bytes); if (rc != LVM_OK) return rc;
    rc = reserve(c->d_pre_out, c->pre_out_cap, out_bytes); if (rc != LVM_OK) return rc;
    rc = reserve(c->d_chain_out, c->chain_out_cap, out_bytes); if (rc != LVM_OK) return rc;
    LVM_HIP_TRY(c, hipMemcpy2DAsync(c->d_pre_in, roi_row, in + (size_t)ry * in_stride + (size_t)rx * channels, (size_t)in_stride, roi_row, (size_t)rh,
                                    hipMemcpyHostToDevice, s));
    const bool identity = ow == rw && oh == rh && och == channels;
    const bool gray_tap = d_orig && och == 1 && channels == 3;
    const uint8_t* mag_in = c->d_pre_in;
    if (!identity) {
        lvm_preprocess_params q = *pp;
        q.roi_enabled = 0;                                               // already cropped by the copy
        // with grayscale on a BGR source the kernel writes the colour tap straight into the caller's `original` buffer
        rc = lvm::preprocess_device(c, q, c->d_pre_in, rw, rh, channels, (ptrdiff_t)roi_row, (ptrdiff_t)roi_bytes, c->d_pre_out, (ptrdiff_t)out_row,
                                    (ptrdiff_t)out_bytes, s, gray_tap ? d_orig : nullptr, orig_stride, (ptrdiff_t)orig_stride * oh);
        if (rc != LVM_OK) { (void)hipStreamSynchronize(s); return rc; }
        mag_in = c->d_pre_out;
    }
    if (d_orig && !gray_tap)        // no gray stage in between: the tap IS the frame the magnifier sees
        LVM_HIP_TRY(c, hipMemcpy2DAsync(d_orig, (size_t)orig_stride, mag_in, out_row, out_row, (size_t)oh, hipMemcpyDeviceToDevice, s));
    lvm_params mp = *p;
    mp.preprocess_key = preprocess_key_of(*pp);
    // the last kernel writes the processed frame straight into the caller's buffer
    uint8_t* dst = d_proc ? d_proc : c->d_chain_out;
    const ptrdiff_t dstride = d_proc ? proc_stride : (ptrdiff_t)out_row;
    lvm::FrameIO io{mag_in, (ptrdiff_t)out_row, (ptrdiff_t)out_bytes, dst, dstride, dstride * oh, ow, oh, och};
    const int saved_depth = c->pipeline_depth;
    c->pipeline_depth = 0;
    rc = lvm::process_device(c, &mp, io, s, produced);
    c->pipeline_depth = saved_depth;
    if (rc != LVM_OK) { (void)hipStreamSynchronize(s); return rc; }
    if (!*produced && d_proc)       // passthrough: the chain hands the magnifier's INPUT on (MagnificationProcessor.cpp:61) -- that is what the display shows
        LVM_HIP_TRY(c, hipMemcpy2DAsync(d_proc, (size_t)proc_stride, mag_in, out_row, out_row, (size_t)oh, hipMemcpyDeviceToDevice, s));
    LVM_HIP_TRY(c, hipStreamSynchronize(s));
    return LVM_OK;
}

int lvm_export_geometry(const lvm_preprocess_params* pp, int split, int w, int h, int channels, int* cw, int* ch) {
    if (!pp || w <= 0 || h <= 0 || (channels != 1 && channels != 3) || !cw || !ch) return LVM_ERR_INVALID;
    if (split < LVM_SPLIT_NONE || split > LVM_SPLIT_TOP_BOTTOM) return LVM_ERR_INVALID;
    int rx, ry, rw, rh, ow, oh, och, pw, ph;
    lvm::preprocess_geometry(*pp, w, h, channels, &rx, &ry, &rw, &rh, &ow, &oh, &och);
    (void)lvm::compose_geometry(split, ow, oh, ow, oh, &pw, &ph, cw, ch);      // 0 x 0 where Exporter::compose returns an empty Mat
    return LVM_OK;
}

// Exporter::run's loop body for a batch of host frames (export/Exporter.cpp:216-259); see include/lvm_hip.h.
// Round 5: a three-stage pipeline over three queues -- the frames go through in sub-batches of LVM_EXPORT_CHUNK (2) frames, and while
// sub-batch k is preprocessed / magnified / composed on the context's stream, sub-batch k + 1 is uploaded on `up_stream` and the
// canvases of sub-batch k - 1 are downloaded on `down_stream`: PCIe runs in both directions at once (measured on this box,
// tools/ubench_pcie.hip: 52-57 GB/s one way, 90-97 GB/s with both directions busy).  The magnifier still sees every frame in order
// with the state of its predecessor; a sub-batch is one temporal batch (lvm_process_device_frames), so the frames are the ones a
// single 32-frame batch -- or 32 per-frame calls -- gives.
// lvm_export_frames and lvm_export_frames_mjpeg: the same three-queue loop; with `mj` the canvases stay on the device and are encoded there
struct MjpegSink { int quality; uint8_t* out; size_t capacity; size_t* offsets; };
// ... and with `js` the frames arrive as JPEG (lvm_export_mjpeg_frames): decoded on the device by the upload queue, the ROI is a view into
// the decoded frame instead of the pitch of a copy
struct JpegSource { const uint8_t* bytes; const size_t* offsets; };
static int export_frames_impl(lvm_ctx* c, const lvm_preprocess_params* pp, const lvm_params* p, int split, int n_frames,
                              const uint8_t* const* frames, int w, int h, int channels, ptrdiff_t in_stride, uint8_t* const* canvases,
                              ptrdiff_t canvas_stride, int* produced, const MjpegSink* mj, const JpegSource* js = nullptr) {
    if (!c || !pp || !p || (!frames && !js) || (!canvases && !mj) || !produced || n_frames < 1) return LVM_ERR_INVALID;
    if (mj && (!mj->out || !mj->offsets)) return LVM_ERR_INVALID;
    if (js && (!js->bytes || !js->offsets || channels != 3)) return LVM_ERR_INVALID;
    if (js) in_stride = (ptrdiff_t)w * 3;
    if (c->nstreams != 1) { c->err = "lvm_export_frames needs a 1-stream context"; return LVM_ERR_INVALID; }
    if (w <= 0 || h <= 0 || (channels != 1 && channels != 3) || in_stride < (ptrdiff_t)w * channels) { c->err = "bad frame arguments"; return LVM_ERR_INVALID; }
    for (int k = 0; k < n_frames; ++k) { produced[k] = 0; if ((!js && !frames[k]) || (!mj && !canvases[k])) { c->err = "null frame pointer"; return LVM_ERR_INVALID; } }
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    int rx, ry, rw, rh, ow, oh, och, pw, ph, cw, chh;
    lvm::preprocess_geometry(*pp, w, h, channels, &rx, &ry, &rw, &rh, &ow, &oh, &och);
    if (split < LVM_SPLIT_NONE || split > LVM_SPLIT_TOP_BOTTOM) { c->err = "invalid split mode"; return LVM_ERR_INVALID; }
    if (!lvm::compose_geometry(split, ow, oh, ow, oh, &pw, &ph, &cw, &chh) || cw <= 0 || chh <= 0) {
        c->err = "lvm_export_frames: empty canvas (Exporter::compose returns an empty Mat for this geometry)"; return LVM_ERR_INVALID;
    }
    if (!mj && canvas_stride < (ptrdiff_t)cw * 3) { c->err = "canvas stride too small"; return LVM_ERR_INVALID; }
    const size_t roi_row = (size_t)rw * channels, roi_bytes = roi_row * rh;
    const size_t out_row = (size_t)ow * och, out_bytes = out_row * oh;
    const size_t can_row = (size_t)cw * 3, can_bytes = can_row * chh;
    auto reserve = [&](uint8_t*& ptr, size_t& cap, size_t need) -> int {
        if (need <= cap) return LVM_OK;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        LVM_HIP_TRY(c, hipMalloc((void**)&ptr, need));
        cap = need;
        return LVM_OK;
    };
    hipStream_t s = c->own_stream;
    LVM_HIP_TRY(c, hipStreamSynchronize(s));      // staging buffers may be replaced below
    // the pane Exporter::compose labels "Original" is runChainOnce's tap: chain[0]'s output, i.e. the frame BEFORE GrayscaleProcessor
    // (ChainBuilder.cpp:25).  With grayscale on a BGR source that is the cropped / decimated colour frame.
    const bool gray_tap = och == 1 && channels == 3 && split != LVM_SPLIT_NONE;
    const size_t tap_row = (size_t)ow * 3, tap_bytes = tap_row * oh;
    // where stage 1 leaves frame k: the cropped copy (host frames), or the ROI inside the decoded frame (JPEG frames)
    const size_t full_row = (size_t)w * channels, full_bytes = full_row * h;
    const size_t src_row = js ? full_row : roi_row, src_fbytes = js ? full_bytes : roi_bytes;
    int rc = reserve(c->d_pre_in, c->pre_in_cap, src_fbytes * n_frames); if (rc != LVM_OK) return rc;
    const uint8_t* src_base = js ? c->d_pre_in + (size_t)ry * full_row + (size_t)rx * channels : c->d_pre_in;
    rc = reserve(c->d_pre_out, c->pre_out_cap, out_bytes * n_frames); if (rc != LVM_OK) return rc;
    rc = reserve(c->d_chain_out, c->chain_out_cap, out_bytes * n_frames); if (rc != LVM_OK) return rc;
    rc = reserve(c->d_canvas, c->canvas_cap, can_bytes * n_frames); if (rc != LVM_OK) return rc;
    if (gray_tap) { rc = reserve(c->d_pre_tap, c->pre_tap_cap, tap_bytes * n_frames); if (rc != LVM_OK) return rc; }
    int chunk = mj ? 4 : 2;      // (JPEG frames: nothing large to download, the encoder's launches want more frames each)
    if (const char* e = std::getenv(mj ? "LVM_EXPORT_MJPEG_CHUNK" : "LVM_EXPORT_CHUNK")) { const int v = std::atoi(e); if (v >= 1) chunk = v; }
    const int nchunks = (n_frames + chunk - 1) / chunk;

    if (!c->up_stream) LVM_HIP_TRY(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    if (!c->down_stream) LVM_HIP_TRY(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    while ((int)c->ev_up.size() < nchunks) { hipEvent_t e = nullptr; LVM_HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_up.push_back(e); }
    while ((int)c->ev_done.size() < nchunks) { hipEvent_t e = nullptr; LVM_HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); c->ev_done.push_back(e); }
    if (mj) { rc = lvm::mjpeg_begin(c, cw, chh, mj->quality, chunk < n_frames ? chunk : n_frames, (size_t)n_frames, mj->capacity, c->down_stream); if (rc != LVM_OK) return rc; }
    auto drain = [&]() { (void)hipStreamSynchronize(c->up_stream); (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(c->down_stream); if (mj) lvm::mjpeg_abort(c); };
#define LVM_EXPORT_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e_); drain(); return LVM_ERR_HIP; } } while (0)
    const bool identity = ow == rw && oh == rh && och == channels;      // PreprocessProcessor.cpp:15, GrayscaleProcessor.cpp:8-9
    const uint8_t* mag_base = identity ? src_base : c->d_pre_out;
    const size_t mag_row = identity ? src_row : out_row, mag_fbytes = identity ? src_fbytes : out_bytes;
    lvm_params mp = *p;
    mp.preprocess_key = preprocess_key_of(*pp);
    const int saved_depth = c->pipeline_depth;
    if (js) {
        // JPEG frames: only the compressed bytes cross PCIe, and ALL frames of the call are decoded by one set of launches -- Huffman decoding
        // is serial inside a restart interval (a lane each), its time is a latency that does not grow with the number of frames
        rc = lvm::mjpeg_decode_begin(c, js->bytes, js->offsets, n_frames, w, h, c->up_stream);
        if (rc == LVM_OK) rc = lvm::mjpeg_decode_enqueue(c, 0, n_frames, c->d_pre_in, (ptrdiff_t)full_row, (ptrdiff_t)full_bytes, c->up_stream);
        if (rc != LVM_OK) { drain(); return rc; }
    }
    for (int q = 0; q < nchunks; ++q) {
        const int f0 = q * chunk, nf = (f0 + chunk <= n_frames) ? chunk : n_frames - f0;
        // stage 1 (up_stream): only the ROI rows cross PCIe (the crop is the pitch of the 2-D copy)
        if (!js) {
            for (int k = f0; k < f0 + nf; ++k)
                LVM_EXPORT_TRY(hipMemcpy2DAsync(c->d_pre_in + (size_t)k * roi_bytes, roi_row, frames[k] + (size_t)ry * in_stride + (size_t)rx * channels,
                                                (size_t)in_stride, roi_row, (size_t)rh, hipMemcpyHostToDevice, c->up_stream));
        }
        LVM_EXPORT_TRY(hipEventRecord(c->ev_up[q], c->up_stream));
        // stage 2 (the context's stream): Preprocess + Grayscale, the magnifier as one temporal batch, compose
        LVM_EXPORT_TRY(hipStreamWaitEvent(s, c->ev_up[q], 0));
        if (!identity) {
            lvm_preprocess_params qp = *pp;
            qp.roi_enabled = 0;                                          // already cropped by the copy
            for (int k = f0; k < f0 + nf; ++k) {                         // (stateless: a frame of the batch is one more "stream" of a 1-stream context)
                rc = lvm::preprocess_device(c, qp, src_base + (size_t)k * src_fbytes, rw, rh, channels, (ptrdiff_t)src_row, (ptrdiff_t)src_fbytes,
                                            c->d_pre_out + (size_t)k * out_bytes, (ptrdiff_t)out_row, (ptrdiff_t)out_bytes, s,
                                            gray_tap ? c->d_pre_tap + (size_t)k * tap_bytes : nullptr, (ptrdiff_t)tap_row, (ptrdiff_t)tap_bytes);
                if (rc != LVM_OK) { drain(); return rc; }
            }
        }
        c->pipeline_depth = 0;                                           // (the synchronous surface completes its own frames)
        rc = lvm_process_device_frames(c, &mp, nf, mag_base + (size_t)f0 * mag_fbytes, ow, oh, och, (ptrdiff_t)mag_row, (ptrdiff_t)mag_fbytes, (ptrdiff_t)mag_fbytes,
                                       c->d_chain_out + (size_t)f0 * out_bytes, (ptrdiff_t)out_row, (ptrdiff_t)out_bytes, (ptrdiff_t)out_bytes, produced + f0, s);
        c->pipeline_depth = saved_depth;
        if (rc != LVM_OK) { drain(); return rc; }
        for (int k = f0; k < f0 + nf; ++k) {
            const uint8_t* seen = mag_base + (size_t)k * mag_fbytes;                                // what the magnifier saw
            const bool pr = produced[k] != 0;
            const uint8_t* proc = pr ? c->d_chain_out + (size_t)k * out_bytes : seen;               // MagnificationProcessor.cpp:61
            const uint8_t* orig = gray_tap ? c->d_pre_tap + (size_t)k * tap_bytes : seen;           // ChainBuilder.cpp:25
            rc = lvm::compose_device(c, split, orig, ow, oh, gray_tap ? 3 : och, (ptrdiff_t)(gray_tap ? tap_row : mag_row), (ptrdiff_t)(gray_tap ? tap_bytes : mag_fbytes),
                                     proc, ow, oh, och, (ptrdiff_t)(pr ? out_row : mag_row), (ptrdiff_t)(pr ? out_bytes : mag_fbytes), c->d_canvas + (size_t)k * can_bytes,
                                     (ptrdiff_t)can_row, (ptrdiff_t)can_bytes, s);
            if (rc != LVM_OK) { drain(); return rc; }
        }
        // drawLabel on every canvas of the sub-batch (Exporter.cpp:74-77, :82-85), from the tables of lvm_export_set_overlay
        rc = lvm::overlay_device(c, c->d_canvas + (size_t)f0 * can_bytes, cw, chh, (ptrdiff_t)can_row, (ptrdiff_t)can_bytes, nf, s);
        if (rc != LVM_OK) { drain(); return rc; }
        LVM_EXPORT_TRY(hipEventRecord(c->ev_done[q], s));
        LVM_EXPORT_TRY(hipStreamWaitEvent(c->down_stream, c->ev_done[q], 0));
        if (mj) {       // stage 3 stays on the device (down_stream, next to the magnifier of the following sub-batch): the canvases become JPEG
                        // frames behind the earlier ones; nothing to download until the end
            rc = lvm::mjpeg_encode_device(c, c->d_canvas + (size_t)f0 * can_bytes, (ptrdiff_t)can_row, (ptrdiff_t)can_bytes, nf, f0, mj->capacity, c->down_stream);
            if (rc == LVM_OK && q >= 2) rc = lvm::mjpeg_drain(c, mj->out, (size_t)q - 1);       // the frames of sub-batch q - 2 come down meanwhile
            if (rc != LVM_OK) { drain(); return rc; }
            continue;
        }
        // stage 3 (down_stream): the canvases of this sub-batch
        for (int k = f0; k < f0 + nf; ++k)
            LVM_EXPORT_TRY(hipMemcpy2DAsync(canvases[k], (size_t)canvas_stride, c->d_canvas + (size_t)k * can_bytes, can_row, can_row, (size_t)chh,
                                            hipMemcpyDeviceToHost, c->down_stream));
    }
#undef LVM_EXPORT_TRY
    lvm::mark_enqueued(c, s);
    int drc = LVM_OK;
    if (js) drc = lvm::mjpeg_decode_finish(c, c->up_stream);                                    // a malformed input frame fails the call
    if (mj) {
        rc = lvm::mjpeg_finish(c, (size_t)n_frames, mj->out, mj->offsets, c->down_stream);       // waits for the encoder, then the compressed bytes come down in one copy
        (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize(c->up_stream);
        return drc != LVM_OK ? drc : rc;
    }
    LVM_HIP_TRY(c, hipStreamSynchronize(c->down_stream));     // (the last canvases: everything on `s` and `up_stream` precedes them)
    LVM_HIP_TRY(c, hipStreamSynchronize(s));
    return drc;
}

int lvm_export_frames(lvm_ctx* c, const lvm_preprocess_params* pp, const lvm_params* p, int split, int n_frames,
                      const uint8_t* const* frames, int w, int h, int channels, ptrdiff_t in_stride, uint8_t* const* canvases,
                      ptrdiff_t canvas_stride, int* produced) {
    if (!canvases) return LVM_ERR_INVALID;
    return export_frames_impl(c, pp, p, split, n_frames, frames, w, h, channels, in_stride, canvases, canvas_stride, produced, nullptr);
}

int lvm_export_frames_mjpeg(lvm_ctx* c, const lvm_preprocess_params* pp, const lvm_params* p, int split, int n_frames,
                            const uint8_t* const* frames, int w, int h, int channels, ptrdiff_t in_stride, int quality,
                            uint8_t* out, size_t out_capacity, size_t* offsets, int* produced) {
    const MjpegSink mj{quality, out, out_capacity, offsets};
    return export_frames_impl(c, pp, p, split, n_frames, frames, w, h, channels, in_stride, nullptr, 0, produced, &mj);
}

int lvm_mjpeg_decode_device(lvm_ctx* c, const uint8_t* jpegs, const size_t* offsets, int n_frames, int w, int h, uint8_t* d_bgr, ptrdiff_t stride,
                            ptrdiff_t frame_stride) {
    if (!c || !jpegs || !offsets || !d_bgr || n_frames < 1) return LVM_ERR_INVALID;
    if (stride < (ptrdiff_t)w * 3 || (n_frames > 1 && frame_stride < (ptrdiff_t)h * stride)) { c->err = "bad frame arguments"; return LVM_ERR_INVALID; }
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    const int rc = lvm::mjpeg_decode_device(c, jpegs, offsets, n_frames, w, h, d_bgr, stride, frame_stride, c->own_stream);
    if (rc == LVM_OK) lvm::mark_enqueued(c, c->own_stream);
    return rc;
}

int lvm_export_mjpeg_frames(lvm_ctx* c, const lvm_preprocess_params* pp, const lvm_params* p, int split, int n_frames, const uint8_t* jpegs,
                            const size_t* in_offsets, int w, int h, int quality, uint8_t* out, size_t out_capacity, size_t* offsets, int* produced) {
    const MjpegSink mj{quality, out, out_capacity, offsets};
    const JpegSource js{jpegs, in_offsets};
    return export_frames_impl(c, pp, p, split, n_frames, nullptr, w, h, 3, (ptrdiff_t)w * 3, nullptr, 0, produced, &mj, &js);
}

int lvm_mjpeg_set_restart_interval(lvm_ctx* c, int mcus) {
    if (!c || mcus < 0) return LVM_ERR_INVALID;
    lvm::mjpeg_set_restart(c, mcus);
    return LVM_OK;
}

size_t lvm_mjpeg_bound(int w, int h) { return (w < 1 || h < 1) ? 0 : lvm::mjpeg_bound(w, h); }

int lvm_mjpeg_encode_device(lvm_ctx* c, const uint8_t* d_bgr, int w, int h, ptrdiff_t stride, ptrdiff_t frame_stride, int n_frames, int quality,
                            uint8_t* out, size_t out_capacity, size_t* offsets) {
    if (!c || !d_bgr || !out || !offsets || n_frames < 1) return LVM_ERR_INVALID;
    if (stride < (ptrdiff_t)w * 3 || (n_frames > 1 && frame_stride < (ptrdiff_t)h * stride)) { c->err = "bad frame arguments"; return LVM_ERR_INVALID; }
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->own_stream;
    const int per = n_frames < 8 ? n_frames : 8;                 // scratch (coefficients, bit buffers) for eight frames at a time
    int rc = lvm::mjpeg_begin(c, w, h, quality, per, (size_t)n_frames, out_capacity, s);
    if (rc != LVM_OK) return rc;
    for (int f0 = 0; f0 < n_frames; f0 += per) {
        const int nf = f0 + per <= n_frames ? per : n_frames - f0;
        rc = lvm::mjpeg_encode_device(c, d_bgr + (size_t)f0 * frame_stride, stride, frame_stride, nf, f0, out_capacity, s);
        if (rc == LVM_OK && f0 >= 2 * per) rc = lvm::mjpeg_drain(c, out, (size_t)(f0 / per) - 1);
        if (rc != LVM_OK) { (void)hipStreamSynchronize(s); lvm::mjpeg_abort(c); return rc; }
    }
    lvm::mark_enqueued(c, s);
    return lvm::mjpeg_finish(c, (size_t)n_frames, out, offsets, s);
}

int lvm_chain_process(lvm_ctx* c, const lvm_preprocess_params* pp, const lvm_params* p, const uint8_t* in, int w, int h, int channels,
                      ptrdiff_t in_stride, uint8_t* out, ptrdiff_t out_stride, int* produced) {
    if (!c || !produced) return LVM_ERR_INVALID;
    *produced = 0;
    if (c->nstreams != 1) { c->err = "lvm_chain_process needs a 1-stream context"; return LVM_ERR_INVALID; }
    return lvm_chain_process_batch(c, pp, p, &in, w, h, channels, in_stride, &out, out_stride, produced);
}

// The device-visible alias of a page-locked host buffer (hipHostMalloc / hipHostRegister memory; lvm_host_alloc), or null.
static uint8_t* pinned_alias(const void* p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
    return static_cast<uint8_t*>(a.devicePointer);
}

// Round 5: frames in PAGE-LOCKED memory (lvm_host_alloc, i.e. a FramePool built on it) take no staging copy at all -- the first kernel
// reads the input straight over PCIe and the last kernel writes the output frame straight into the caller's buffer ("zero copy").
// Measured on this box (tools/ubench_pcie.hip): a kernel reads / writes a page-locked 1080p frame at 55 GB/s (112 us) while the DMA
// engine needs 120 us up and 166 us down for ONE frame in flight, and splitting a frame into row chunks to overlap copy and kernel
// costs ~20 us of queue hand-offs per chunk.  The input alias is used when exactly ONE kernel reads the u8 frame (Lab modes on BGR
// frames with the table flavour: the conversion kernel; everything downstream reads its integer planes); the colour mode and gray frames
// read their input again in the output pass and keep the upload.  The output alias is used in every mode: every mode's last kernel
// writes each output byte once.  Pageable frames take the copies as before.  LVM_ZERO_COPY=0 switches the aliases off (A/B, tests).
int lvm_process(lvm_ctx* c, const lvm_params* p, const uint8_t* in, int w, int h, int channels, ptrdiff_t in_stride,
                uint8_t* out, ptrdiff_t out_stride, int* produced) {
    if (!c || !p || !produced) return LVM_ERR_INVALID;
    *produced = 0;
    if (c->nstreams != 1) { c->err = "lvm_process needs a 1-stream context"; return LVM_ERR_INVALID; }
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    if (p->mode == LVM_MODE_NONE || !in || w <= 0 || h <= 0) {
        lvm::FrameIO io{nullptr, 0, 0, nullptr, 0, 0, w, h, channels};
        return lvm::process_device(c, p, io, c->own_stream, produced);
    }
    if (!out || (channels != 1 && channels != 3) || in_stride < (ptrdiff_t)w * channels ||
        out_stride < (ptrdiff_t)w * channels) { c->err = "bad frame arguments"; return LVM_ERR_INVALID; }
    const size_t row = (size_t)w * channels, bytes = row * h;
    static const bool zero_copy = [] { const char* e = std::getenv("LVM_ZERO_COPY"); return !(e && std::atoi(e) == 0); }();
    const bool one_reader = channels == 3 && !c->lab_analytic && (p->mode == LVM_MODE_LAPLACE || p->mode == LVM_MODE_PHASE);
    const uint8_t* in_alias = (zero_copy && one_reader) ? pinned_alias(in) : nullptr;
    uint8_t* out_alias = zero_copy ? pinned_alias(out) : nullptr;
    if (bytes > c->stage_cap && (!in_alias || !out_alias)) {
        lvm::sync_streams(c);
        if (c->d_in) (void)hipFree(c->d_in);
        if (c->d_out) (void)hipFree(c->d_out);
        c->d_in = c->d_out = nullptr; c->stage_cap = 0;
        LVM_HIP_TRY(c, hipMalloc((void**)&c->d_in, bytes));
        LVM_HIP_TRY(c, hipMalloc((void**)&c->d_out, bytes));
        c->stage_cap = bytes;
    }
    hipStream_t s = c->own_stream;
    if (!in_alias) LVM_HIP_TRY(c, hipMemcpy2DAsync(c->d_in, row, in, (size_t)in_stride, row, (size_t)h, hipMemcpyHostToDevice, s));
    lvm::FrameIO io{in_alias ? in_alias : c->d_in, in_alias ? in_stride : (ptrdiff_t)row, in_alias ? in_stride * h : (ptrdiff_t)bytes,
                    out_alias ? out_alias : c->d_out, out_alias ? out_stride : (ptrdiff_t)row, out_alias ? out_stride * h : (ptrdiff_t)bytes, w, h, channels};
    const int saved_depth = c->pipeline_depth;
    c->pipeline_depth = 0;                       // the synchronous surface completes its own frame
    const int rc = lvm::process_device(c, p, io, s, produced);
    c->pipeline_depth = saved_depth;
    if (rc != LVM_OK) { (void)hipStreamSynchronize(s); return rc; }
    if (*produced && !out_alias)
        LVM_HIP_TRY(c, hipMemcpy2DAsync(out, (size_t)out_stride, c->d_out, row, row, (size_t)h, hipMemcpyDeviceToHost, s));
    LVM_HIP_TRY(c, hipStreamSynchronize(s));      // (polling an event instead measured the same 310 us per 1080p frame: nothing to gain)
    return LVM_OK;
}

// Page-locked host frames for the reference's FramePool (core/FramePool.cpp:29-36 allocates the pooled cv::Mat
// buffers the chain hands to process()): frames living in such memory cross PCIe by DMA at link speed in
// lvm_process / lvm_chain_process with no staging copy anywhere (hipMemcpy2DAsync sees the registration).
int lvm_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return LVM_ERR_INVALID;
    *out = nullptr;
    if (hipHostMalloc(out, bytes, 0) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return LVM_ERR_OOM; }
    return LVM_OK;
}
void lvm_host_free(void* p) { if (p) (void)hipHostFree(p); }

int lvm_set_max_frames(lvm_ctx* c, int n_frames) {
    if (!c || n_frames < 1) return LVM_ERR_INVALID;
    c->max_frames = n_frames;
    return LVM_OK;
}

int lvm_synchronize(lvm_ctx* c) {
    if (!c) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipStreamSynchronize(c->own_stream));
    return LVM_OK;
}

const char* lvm_last_error(lvm_ctx* c) { return c ? c->err.c_str() : "null context"; }

int lvm_debug_keep_float(lvm_ctx* c, int on) { if (!c) return LVM_ERR_INVALID; c->keep_float = on != 0; return LVM_OK; }

int lvm_debug_exact_lab(lvm_ctx* c, int on) { if (!c) return LVM_ERR_INVALID; c->exact_lab = on != 0; return LVM_OK; }

int lvm_debug_sweep_u8_steps(lvm_ctx* c, uint32_t first_bits, uint64_t count, uint64_t* mismatches, uint32_t* first_bad_bits) {
    if (!c || !mismatches || (uint64_t)first_bits + count > (1ull << 32)) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long bad = 0, fb = 0;
    const int rc = lvm::sweep_u8_steps(c, first_bits, count, &bad, &fb, c->own_stream);
    if (rc != LVM_OK) return rc;
    *mismatches = bad;
    if (first_bad_bits) *first_bad_bits = bad ? (uint32_t)fb : 0u;
    return LVM_OK;
}

int lvm_debug_clock_probe_start(lvm_ctx* c, double max_seconds) {
    if (!c) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    return lvm::clock_probe_start(c, max_seconds);
}

int lvm_debug_clock_probe_stop(lvm_ctx* c, double* mhz, double* seconds) {
    if (!c) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    return lvm::clock_probe_stop(c, mhz, seconds);
}

int lvm_debug_lab_analytic(lvm_ctx* c, int on) { if (!c) return LVM_ERR_INVALID; c->lab_analytic = on != 0; return LVM_OK; }

int lvm_get_lab_lut(lvm_ctx* c, int16_t* dst) {
    if (!c || !dst) return LVM_ERR_INVALID;
    std::memcpy(dst, c->lab_lut_compact.data(), c->lab_lut_compact.size() * sizeof(int16_t));
    return LVM_OK;
}

int lvm_set_lab_lut(lvm_ctx* c, const int16_t* src) {
    if (!c || !src) return LVM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    for (size_t i = 0; i < (size_t)LVM_LAB_LUT_ENTRIES; ++i)
        if (src[i] < 0 || src[i] > 16384) { c->err = "lvm_set_lab_lut: entry outside [0, 16384]"; return LVM_ERR_INVALID; }
    lvm::sync_streams(c);                        // kernels in flight read the old table (this context's buffers: no device-wide wait)
    c->lab_lut_compact.assign(src, src + LVM_LAB_LUT_ENTRIES);
    return lvm::upload_lab_lut(c);
}

int lvm_debug_read_float(lvm_ctx* c, float* dst, size_t count) {
    if (!c || !dst) return LVM_ERR_INVALID;
    if (!c->d_float || count > c->float_count) { c->err = "no float frame kept"; return LVM_ERR_INVALID; }
    lvm::sync_streams(c);
    LVM_HIP_TRY(c, hipMemcpy(dst, c->d_float, count * sizeof(float), hipMemcpyDeviceToHost));
    return LVM_OK;
}

int lvm_profile_enable(lvm_ctx* c, int on) {
    if (!c) return LVM_ERR_INVALID;
    c->profiling = on != 0;
    return LVM_OK;
}

int lvm_profile_only(lvm_ctx* c, const char* name) {
    if (!c) return LVM_ERR_INVALID;
    (void)lvm_profile_collect(c);
    c->prof_totals.clear();
    c->prof_only = name ? name : "";
    return LVM_OK;
}

int lvm_profile_collect(lvm_ctx* c) {
    if (!c) return LVM_ERR_INVALID;
    lvm::sync_streams(c);
    for (auto& e : c->prof_events) {
        (void)hipEventSynchronize(e.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.e0, e.e1) == hipSuccess) { c->prof_totals[e.name].ms += ms; c->prof_totals[e.name].n += 1; }
        (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1);
    }
    c->prof_events.clear();
    return (int)c->prof_totals.size();
}

int lvm_profile_entry(lvm_ctx* c, int idx, char* name, size_t cap, double* total_ms, long long* launches) {
    if (!c || idx < 0 || idx >= (int)c->prof_totals.size()) return LVM_ERR_INVALID;
    const auto& t = c->prof_totals[idx];
    if (name && cap) { std::snprintf(name, cap, "%s", t.name.c_str()); }
    if (total_ms) *total_ms = t.ms;
    if (launches) *launches = t.n;
    return LVM_OK;
}

int lvm_set_pipeline(lvm_ctx* c, int depth) {
    if (!c || depth < 0 || depth > 1) return LVM_ERR_INVALID;
    if (depth != c->pipeline_depth) {
        lvm::sync_streams(c);
        if (c->t_mode == LVM_MODE_LAPLACE) (void)lvm::laplace_flush(c, c->own_stream);
        lvm::sync_streams(c);
            c->pipeline_depth = depth;
    }
    return LVM_OK;
}

int lvm_flush(lvm_ctx* c, void* hip_stream) {
    if (!c) return LVM_ERR_INVALID;
    LVM_HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    if (c->t_mode == LVM_MODE_LAPLACE) return lvm::laplace_flush(c, s);
    return LVM_OK;
}

// compulsory traffic only -- every input byte read once, every output byte
// written once, every live persistent state word read once and written once.
double lvm_algorithmic_bytes(int mode, int w, int h, int channels, int levels, double framerate) {
    const int maxL = lvm::max_levels(w, h);
    if (maxL < 1) return 0.0;
    int L = levels < 1 ? 1 : (levels > maxL ? maxL : levels);
    double n[lvm::kMaxLevels + 2];
    int lw = w, lh = h;
    for (int l = 0; l <= L && l <= lvm::kMaxLevels; ++l) { n[l] = (double)lw * lh; lw = (lw + 1) / 2; lh = (lh + 1) / 2; }
    const double io = 2.0 * channels * n[0];
    if (mode == LVM_MODE_LAPLACE) {
        double s = 0; for (int l = 1; l <= L - 1; ++l) s += n[l];
        return io + 16.0 * channels * s;             // 2 low-pass states x 4 B x (R + W) per channel
    }
    if (mode == LVM_MODE_PHASE) {
        double s = 0; for (int l = 0; l <= L - 2; ++l) s += n[l];
        return io + 88.0 * s;                        // 11 live floats per band pixel, R + W
    }
    if (mode == LVM_MODE_COLOR) {
        const int T = lvm::optimal_buffer_size((int)framerate);
        return io + 4.0 * channels * n[L] * T + 4.0 * channels * n[L];
    }
    return 0.0;
}

}  // extern "C"
