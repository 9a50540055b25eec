/*
 * lvm_hip.h -- C ABI of liblvm_hip.so, the MI355X (gfx950) Eulerian video-magnification core.
 *
 * Drop-in boundary for the reference's magnification stage.  Every entry point cites the
 * reference interface it replaces (paths relative to the reference's src/):
 *
 *   processing/IProcessor.hpp:50-60            IProcessor::process / reset   (the surface)
 *   processing/MagnificationProcessor.cpp:17-67  process(): clamp levels, structural reset,
 *                                              dispatch by mode, passthrough on !produced
 *   processing/MagnificationProcessor.cpp:10-15  reset()
 *   processing/magnification/MagnifyCore.hpp:83,163,209  magnifyMotion/Color/Riesz
 *
 * Plain pointers and sizes only; no torch / OpenCV / Qt types.  A context is
 * thread-compatible (one thread at a time per context, any number of contexts concurrently --
 * the reference runs a live chain and an export chain side by side, export/Exporter.cpp:204).
 * All entry points return 0 on success, a negative lvm_status otherwise;
 * lvm_last_error(ctx) gives the message.  After an error the context accepts lvm_reset()
 * and continues (the reference's recovery path: processing/ProcessingChain.cpp:50-62).
 */
#ifndef LVM_HIP_H
#define LVM_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* processing/IProcessor.hpp:10 -- numeric order of `enum class MagnificationMode` */
enum lvm_mode { LVM_MODE_LAPLACE = 0, LVM_MODE_PHASE = 1, LVM_MODE_COLOR = 2, LVM_MODE_NONE = 3 };

enum lvm_status {
    LVM_OK = 0,
    LVM_ERR_INVALID = -1,   /* bad argument */
    LVM_ERR_HIP = -2,       /* HIP runtime error (message in lvm_last_error) */
    LVM_ERR_NO_DEVICE = -3, /* no gfx950 device / HIP runtime unavailable */
    LVM_ERR_OOM = -4
};

/* processing/IProcessor.hpp:14-23 (MagnificationParams, algorithm units) plus a key that
 * stands for PreprocessParams equality (IProcessor.hpp:26-41): the reference drops temporal
 * state when the ROI/downscale changes even at equal size (MagnifyCore.hpp:55-56).          */
typedef struct lvm_params {
    int32_t  mode;              /* lvm_mode */
    int32_t  levels;            /* clamped to [1, maxLevels(w,h)] like MagnificationProcessor.cpp:34 */
    double   amplification;     /* alpha */
    double   coWavelength;      /* Laplace: lambda_c (UI% * 10); Phase: 100 - UI% */
    double   coLow;             /* Laplace: IIR blend in [0,1]; Phase/Color: Hz */
    double   coHigh;
    double   chromAttenuation;  /* Laplace only: gain on Lab a,b of the motion image */
    double   framerate;         /* Color ideal filter + Riesz Butterworth */
    uint64_t preprocess_key;    /* any value that changes iff PreprocessParams changes */
} lvm_params;

typedef struct lvm_ctx lvm_ctx;

/* Replaces constructing a MagnificationProcessor (processing/ChainBuilder.cpp:15).
 * n_streams >= 1 independent streams share one context: every per-stream buffer (temporal
 * state, pyramids) is laid out [stream][...] and every kernel launch covers all of them.
 * n_streams == 1 is the drop-in case.                                                       */
int  lvm_create(int device, int n_streams, lvm_ctx** out);
void lvm_destroy(lvm_ctx* ctx);

/* IProcessor::reset (MagnificationProcessor.cpp:10-15): next frame behaves as the first. */
int  lvm_reset(lvm_ctx* ctx);

/* IProcessor::process on host memory (MagnificationProcessor.cpp:17-67), n_streams == 1.
 * in : h rows of w*channels uint8 (BGR interleaved or gray), in_stride bytes per row.
 * out: caller-allocated, same geometry; written only when *produced != 0.
 * *produced == 0 <=> the reference returns the INPUT frame (mode None, frame too small,
 * Color warm-up < 2 columns, Riesz first frame / re-init, Riesz on < 3 channels).
 * Synchronous: H2D, kernels and D2H have completed on return; `in` is not retained.        */
int  lvm_process(lvm_ctx* ctx, const lvm_params* p, const uint8_t* in, int w, int h,
                 int channels, ptrdiff_t in_stride, uint8_t* out, ptrdiff_t out_stride,
                 int* produced);

/* Same contract on DEVICE memory for all n_streams streams at once: stream s reads
 * d_in + s*in_stream_stride and writes d_out + s*out_stream_stride (bytes).  Work is enqueued
 * on `hip_stream` (a hipStream_t; NULL = the context's own stream) and NOT synchronised:
 * *produced is decided on the host before any kernel runs.                                 */
int  lvm_process_device(lvm_ctx* ctx, const lvm_params* p, const uint8_t* d_in, int w, int h,
                        int channels, ptrdiff_t in_stride, ptrdiff_t in_stream_stride,
                        uint8_t* d_out, ptrdiff_t out_stride, ptrdiff_t out_stream_stride,
                        int* produced, void* hip_stream);

/* n_frames CONSECUTIVE frames of every stream in one call -- the export / offline case
 * (export/Exporter.cpp:216-259 feeds its chain frame after frame from a file).  Frame f of stream s
 * is read at d_in + f*in_frame_stride + s*in_stream_stride and written likewise.  Semantically this
 * IS n_frames calls of lvm_process_device in order (produced[f] per frame); when the mode has a
 * temporally batched schedule (Laplace: stateless kernels take the frames as a batch dimension, the
 * IIR kernels walk over them in order with their state in registers) the frames share launches.  */
int  lvm_process_device_frames(lvm_ctx* ctx, const lvm_params* p, int n_frames, const uint8_t* d_in,
                               int w, int h, int channels, ptrdiff_t in_stride,
                               ptrdiff_t in_stream_stride, ptrdiff_t in_frame_stride, uint8_t* d_out,
                               ptrdiff_t out_stride, ptrdiff_t out_stream_stride,
                               ptrdiff_t out_frame_stride, int* produced, void* hip_stream);

/* Largest n_frames the caller will pass to lvm_process_device_frames: the per-batch pyramids are then
 * allocated once when a mode's state is created, never inside a steady-state call (an export knows its
 * batch size up front, export/Exporter.cpp:216-259).  Without the hint the buffers grow on demand.   */
int  lvm_set_max_frames(lvm_ctx* ctx, int n_frames);

/* Page-locked host memory for frame buffers (what the reference's core/FramePool.cpp:29-36 would allocate
 * its pooled cv::Mat storage from).  lvm_process takes NO staging copy for such frames: the first kernel reads
 * the input and the last kernel writes the output frame directly over PCIe (the input alias only where exactly
 * one kernel reads the u8 frame: Laplace / Riesz on BGR frames); lvm_export_frames / lvm_chain_process copy them
 * by DMA at link speed.  Pageable frames work everywhere (staged copies; the HIP runtime pins them on the fly). */
int  lvm_host_alloc(size_t bytes, void** out);
void lvm_host_free(void* p);

/* ---- the two uint8 stages in front of the magnifier -------------------------
 * processing/IProcessor.hpp:26-41 PreprocessParams + ProcessorConfig::grayscale (IProcessor.hpp:45).  */
typedef struct lvm_preprocess_params {
    int32_t downscale;      /* 1 / 2 / 4 / 8, clamped to [1, 8] like PreprocessProcessor.cpp:14 */
    int32_t roi_enabled;
    float   roiX, roiY, roiW, roiH;   /* fractions of the full frame */
    int32_t grayscale;      /* ProcessorConfig::grayscale: BGR -> Gray8 after the decimation */
} lvm_preprocess_params;

/* Geometry of PreprocessProcessor::process + GrayscaleProcessor::process for a w x h x channels frame:
 * the clamped ROI rectangle (PreprocessProcessor.cpp:19-31) and the size / channel count of the frame
 * handed to the magnifier (:36-39, GrayscaleProcessor.cpp:8-15).                                      */
int  lvm_preprocess_geometry(const lvm_preprocess_params* pp, int w, int h, int channels, int* roi_x,
                             int* roi_y, int* roi_w, int* roi_h, int* out_w, int* out_h, int* out_channels);
/* PreprocessProcessor::process (processing/PreprocessProcessor.cpp:10-51: crop + cv::resize INTER_AREA)
 * followed by GrayscaleProcessor::process (processing/GrayscaleProcessor.cpp:7-16) on DEVICE memory, all
 * n_streams streams, one kernel: the full frame is read once, the small frame written once.  d_out has
 * the geometry lvm_preprocess_geometry reports.  Enqueued on hip_stream, not synchronised.            */
int  lvm_preprocess_device(lvm_ctx* ctx, const lvm_preprocess_params* pp, const uint8_t* d_in, int w, int h,
                           int channels, ptrdiff_t in_stride, ptrdiff_t in_stream_stride, uint8_t* d_out,
                           ptrdiff_t out_stride, ptrdiff_t out_stream_stride, void* hip_stream);
/* runChainOnce (processing/ChainBuilder.cpp:19-29) for the three stages Preprocess -> Grayscale ->
 * Magnification on host memory, n_streams == 1: only the ROI rows cross PCIe, crop / decimation / gray /
 * magnification run on the device, the (small) result comes back.  `out` must hold out_h rows of
 * out_w * out_channels bytes (lvm_preprocess_geometry).  It is always written: the magnified frame
 * when *produced != 0, otherwise the preprocessed frame (the reference's passthrough hands the
 * magnifier's INPUT on, MagnificationProcessor.cpp:61).  The structural key of the magnifier follows
 * the PreprocessParams (MagnifyCore.hpp:55-56), p->preprocess_key is ignored here.                  */
int  lvm_chain_process(lvm_ctx* ctx, const lvm_preprocess_params* pp, const lvm_params* p, const uint8_t* in,
                       int w, int h, int channels, ptrdiff_t in_stride, uint8_t* out, ptrdiff_t out_stride,
                       int* produced);

/* The same for all n_streams streams of a context at once (one processing thread feeding N
 * cameras): in[s] / out[s] are the host frames of stream s, all with the same geometry and parameters; one
 * preprocess launch and one launch per magnifier stage cover every stream.  *produced applies to all streams
 * (they share mode, geometry and history length).                                                     */
int  lvm_chain_process_batch(lvm_ctx* ctx, const lvm_preprocess_params* pp, const lvm_params* p,
                             const uint8_t* const* in, int w, int h, int channels, ptrdiff_t in_stride,
                             uint8_t* const* out, ptrdiff_t out_stride, int* produced);

/* The same plus runChainOnce's `original` (processing/ChainBuilder.cpp:19-29: the tap after chain[0] = PreprocessProcessor,
 * BEFORE GrayscaleProcessor; what the display shows in its left pane, core/LatestFrameMailbox.hpp:13-16): pre_out[s] (may
 * be NULL, entries may be NULL) receives the cropped / decimated frame of stream s -- out_w x out_h like out[s], but with
 * the SOURCE's channel count: with grayscale on a BGR source the tap is the colour frame (3 bytes per pixel,
 * pre_stride >= 3 * out_w) while out[s] is gray.                                                                 */
int  lvm_chain_process_batch_ex(lvm_ctx* ctx, const lvm_preprocess_params* pp, const lvm_params* p,
                                const uint8_t* const* in, int w, int h, int channels, ptrdiff_t in_stride,
                                uint8_t* const* out, ptrdiff_t out_stride, uint8_t* const* pre_out,
                                ptrdiff_t pre_stride, int* produced);

/* ---- display hand-off on the device (the display half) ---------------------------------
 * runChainOnce (processing/ChainBuilder.cpp:19-29) for the LIVE path, n_streams == 1: host frame in, the two frames the display
 * shows left in DEVICE memory.  The reference's ProcessingChain publishes {processed, original} (ProcessingChain.cpp:46-49) and
 * DisplayWidget::uploadFrame (ui/DisplayWidget.cpp:133-152) uploads both from host memory to GL textures on every frame; here
 *   d_proc  (out_w x out_h x out_channels, lvm_preprocess_geometry) receives the magnified frame or, on passthrough
 *           (*produced == 0), the frame the magnifier saw -- i.e. always what the display's main pane shows
 *           (MagnificationProcessor.cpp:61);
 *   d_orig  (out_w x out_h x the SOURCE's channels) receives runChainOnce's `original`: PreprocessProcessor's output, tapped before
 *           GrayscaleProcessor (ChainBuilder.cpp:25);
 * either may be NULL.  Both are caller-owned device pointers -- e.g. the mapped pointers of two GL pixel-unpack buffers registered
 * with hipGraphicsGLRegisterBuffer (host/HipDisplayPresenter.hpp does that), from which glTexSubImage2D then copies device to
 * device: only the ROI rows of the input cross PCIe, nothing comes back.  Synchronous (complete on return).            */
int  lvm_chain_present(lvm_ctx* ctx, const lvm_preprocess_params* pp, const lvm_params* p, const uint8_t* in, int w, int h,
                       int channels, ptrdiff_t in_stride, uint8_t* d_proc, ptrdiff_t proc_stride, uint8_t* d_orig,
                       ptrdiff_t orig_stride, int* produced);

/* ---- export hand-off on the device ---------------------------------------------------
 * export/ExportTypes.hpp:11 -- numeric order of `enum class SplitMode`                                        */
enum lvm_split { LVM_SPLIT_NONE = 0, LVM_SPLIT_LEFT_RIGHT = 1, LVM_SPLIT_TOP_BOTTOM = 2 };
/* Pane and canvas size of Exporter::compose (export/Exporter.cpp:53-88) for an original of ow x oh and a processed
 * frame of pw x ph: panes are cropped to the common EVEN size (:63-64; split None: the processed frame's own even
 * size, :56).  Returns LVM_OK; all four outputs are 0 when the reference returns an empty Mat (:57, :65).     */
int  lvm_compose_geometry(int split, int ow, int oh, int pw, int ph, int* pane_w, int* pane_h, int* canvas_w,
                          int* canvas_h);
/* Exporter::compose on DEVICE memory for all n_streams streams: toBgr (gray -> b = g = r, Exporter.cpp:22-34), crop,
 * and the side-by-side / stacked / single-pane BGR canvas (3 bytes per pixel, canvas_stride >= 3 * canvas_w).  d_orig
 * may be NULL (the reference falls back to the processed frame, :62).  Enqueued on hip_stream, not synchronised.  The
 * text overlay (:36-50) is lvm_overlay_device's job (below). */
int  lvm_compose_device(lvm_ctx* ctx, int split, const uint8_t* d_orig, int ow, int oh, int och,
                        ptrdiff_t orig_stride, ptrdiff_t orig_stream_stride, const uint8_t* d_proc, int pw, int ph,
                        int pch, ptrdiff_t proc_stride, ptrdiff_t proc_stream_stride, uint8_t* d_canvas,
                        ptrdiff_t canvas_stride, ptrdiff_t canvas_stream_stride, void* hip_stream);

/* Spatial tiling of ONE Riesz stream over several devices: a CORRECTNESS DEMONSTRATOR, not a throughput path (DESIGN.md section 7:
 * a frame needs a handful of small exchanges, each a latency, against ~230 us of work on one MI355X -- independent streams, one per GPU, are
 * the production answer).  The frame is cut into horizontal stripes; rank r runs a stripe context on its rows extended by a halo (tiling.py:
 * 64 rows for 2 fine levels) with `levels` = F + 1, F = the fine levels it owns entirely:
 *   lvm_tile_riesz_stage1   Lab, pyramid, phase, temporal filters, normalize + amplify of the stripe (MagnifyCore.hpp:218-267, RieszPyramid.cpp
 *                           :114-144) -- everything up to the collapse; copies the stripe's residual octave (octave F of the full pyramid,
 *                           residual_w x residual_h floats) to d_residual_out.  The owned rows of it go to the rank that holds the coarse levels
 *                           (RCCL send / hipMemcpyPeer: the caller's exchange).
 *   lvm_tile_riesz_planes   that rank: the Riesz chain on a FLOAT PLANE -- the gathered octave F as level 0 of a pyramid with the remaining
 *                           levels -- collapsed back into a float plane: res_F of the full pyramid (RieszPyramid.cpp:304-325).  Rows of it go back.
 *   lvm_tile_riesz_stage2   collapse of the fine levels from the received rows of res_F instead of the stripe's own residual, Lab2BGR, u8
 *                           (MagnifyCore.hpp:269-277) for the extended stripe; the owned rows are the result.
 * Rows closer than the halo to an artificial stripe edge are computed from reflected instead of real neighbours and discarded; with the halo
 * of tiling.py the owned rows are BIT-IDENTICAL to the unsplit context's (tests/test_tiling.py).  produced follows the per-frame rules (first
 * frame: 0 -- every rank passes its rows through).  All calls enqueue on hip_stream without synchronising.                          */
int  lvm_tile_riesz_stage1(lvm_ctx* ctx, const lvm_params* p, const uint8_t* d_in, int w, int h, ptrdiff_t in_stride, int* produced,
                           float* d_residual_out, int* residual_w, int* residual_h, void* hip_stream);
int  lvm_tile_riesz_planes(lvm_ctx* ctx, const lvm_params* p, const float* d_plane_in, int w, int h, float* d_plane_out, int* produced,
                           void* hip_stream);
int  lvm_tile_riesz_stage2(lvm_ctx* ctx, const lvm_params* p, const uint8_t* d_in, int w, int h, ptrdiff_t in_stride,
                           const float* d_residual_in, uint8_t* d_out, ptrdiff_t out_stride, void* hip_stream);

/* The export's text overlay (drawLabel, export/Exporter.cpp:36-50: the rectangle behind a caption darkened with
 * cv::addWeighted(roi, 0.35, black, 0.65), then cv::putText(FONT_HERSHEY_SIMPLEX, white, LINE_AA); called from compose, :74-77 / :82-85)
 * as PER-PIXEL TABLES.  Both steps read-modify-write single pixels with the same arithmetic for B, G and R, so what a label does to a
 * canvas pixel is a function of that pixel's byte; the label itself depends only on the canvas size, i.e. it is fixed for an export.
 * The reference-side shim renders each label ONCE with the reference's own calls onto 256 constant canvases
 * (value v = 0..255) and reads the function off them -- exact for whatever the linked OpenCV draws; nothing of cv::putText is restated
 * in this library.  Pixels with equal functions share a class:
 *   x, y, w, h   the rectangle on the canvas (`bg` of drawLabel after its clip, :43-44)
 *   cls[h][w]    class of every pixel;   fn[n_classes][256]   new byte = fn[cls][old byte]
 * lvm_export_set_overlay copies up to 4 labels into device memory (0 labels: overlay off); lvm_export_frames / _mjpeg /
 * lvm_export_mjpeg_frames then apply them to every composed canvas ON THE DEVICE, so an export with `textOverlay` keeps the device and the
 * Motion-JPEG paths.  A label that does not fit the canvas of a later call fails that call (LVM_ERR_INVALID: it was rendered for another
 * canvas size).  lvm_overlay_device: the same on caller-owned device canvases (enqueued, not synchronised).                        */
typedef struct lvm_overlay_label {
    int32_t x, y, w, h;
    int32_t n_classes;
    const uint16_t* cls;
    const uint8_t* fn;
} lvm_overlay_label;
int  lvm_export_set_overlay(lvm_ctx* ctx, int n_labels, const lvm_overlay_label* labels);
int  lvm_overlay_device(lvm_ctx* ctx, uint8_t* d_canvas, int canvas_w, int canvas_h, ptrdiff_t canvas_stride, ptrdiff_t frame_stride,
                        int n_frames, void* hip_stream);

/* The loop body of Exporter::run (export/Exporter.cpp:216-259) for n_frames CONSECUTIVE host frames of a 1-stream context:
 *   runChainOnce (ChainBuilder.cpp:19-29: PreprocessProcessor -> GrayscaleProcessor -> MagnificationProcessor) on every frame in
 *   order, then Exporter::compose(original, processed, split) (Exporter.cpp:53-88) -- `original` = PreprocessProcessor's output
 *   (the tap of ChainBuilder.cpp:25 sits BEFORE GrayscaleProcessor: a colour pane even when the chain grays), `processed` = the
 *   magnifier's output or, on passthrough (produced[i] == 0), the frame it saw (MagnificationProcessor.cpp:61).
 * frames[i] -> canvases[i]: canvas_w x canvas_h x 3 bytes (lvm_export_geometry), row stride canvas_stride.  Only the ROI rows of
 * the inputs and the canvases cross PCIe, in both directions at once: the frames go through in sub-batches of 8 (each ONE temporal
 * batch of the magnifier, lvm_process_device_frames), sub-batch k + 1 uploading and sub-batch k - 1 downloading while sub-batch k is
 * magnified and composed on the device (lvm_compose_device), the labels of lvm_export_set_overlay applied there too.
 * cv::VideoWriter::write (:259) stays on the host: host/HipExportRunner.hpp is the reference-side loop around this call.  Synchronous. */
int  lvm_export_geometry(const lvm_preprocess_params* pp, int split, int w, int h, int channels, int* canvas_w, int* canvas_h);
int  lvm_export_frames(lvm_ctx* ctx, const lvm_preprocess_params* pp, const lvm_params* p, int split, int n_frames,
                       const uint8_t* const* frames, int w, int h, int channels, ptrdiff_t in_stride,
                       uint8_t* const* canvases, ptrdiff_t canvas_stride, int* produced);

/* Motion-JPEG encode on the device (the encode half): what cv::VideoWriter::write(canvas) does for
 * ExportFormat::AviMjpg, the reference's AVI export format and the fallback of every other one (export/Exporter.cpp:107-117, :259).
 * Baseline JPEG (ITU-T T.81), 8 bit, YCbCr 4:2:0 (JFIF), Annex K Huffman tables, restart intervals of 8 MCUs, libjpeg's quality scale
 * (1..100).  Frames: BGR, 1..8192 x 1..16384.
 *   lvm_mjpeg_bound            bytes that always suffice for ONE encoded frame of this size (a bound; typical frames are 5-20 % of it)
 *   lvm_mjpeg_encode_device    n_frames device-resident BGR frames -> out[offsets[i] .. offsets[i + 1]) = frame i, a complete JPEG;
 *                              offsets has n_frames + 1 entries.  LVM_ERR_INVALID when out_capacity is too small.  Synchronous.
 *   lvm_export_frames_mjpeg    lvm_export_frames with the canvases encoded on the device: only the ROI rows go up and only the compressed
 *                              frames come down.  host/HipMjpegWriter.hpp wraps the frames into the AVI container.                       */
/*   lvm_mjpeg_decode_device    the other direction (cv::VideoCapture::read on an AVI / Motion-JPEG file, source/FileSource.cpp:99): n_frames
 *                              JPEG frames in host memory (jpegs[offsets[i] .. offsets[i + 1])) -> BGR frames of w x h in DEVICE memory.
 *                              Baseline 4:2:0 in one scan, any tables; frames with restart intervals decode a lane per interval, frames
 *                              without through self-synchronising lanes of 1024 bits (a call's time is mostly latency: pass many frames); anything else, a size other
 *                              than w x h or a malformed stream is LVM_ERR_INVALID (lvm_last_error says which frame and why).  Synchronous. */
/*   lvm_export_mjpeg_frames    lvm_export_frames_mjpeg with JPEG frames IN as well (an AVI / Motion-JPEG source file): decode, chain, compose and
 *                              encode all on the device, only compressed bytes cross PCIe in either direction (file -> file export).           */
/*   lvm_mjpeg_set_restart_interval   MCUs (16 x 16 pixels, raster order) per restart interval of the frames this context encodes from now on;
 *                              0 = the default, 8 (pass (w + 15) / 16 for one interval per MCU row).  Entropy coding is serial inside an
 *                              interval: short intervals cost < 0.5 % of bytes and make the frames decode in parallel --
 *                              lvm_mjpeg_decode_device gives every interval a lane.                                                              */
size_t lvm_mjpeg_bound(int w, int h);
int  lvm_mjpeg_set_restart_interval(lvm_ctx* ctx, int mcus);
int  lvm_export_mjpeg_frames(lvm_ctx* ctx, const lvm_preprocess_params* pp, const lvm_params* p, int split, int n_frames, const uint8_t* jpegs,
                             const size_t* in_offsets, int w, int h, int quality, uint8_t* out, size_t out_capacity, size_t* offsets, int* produced);
int  lvm_mjpeg_decode_device(lvm_ctx* ctx, const uint8_t* jpegs, const size_t* offsets, int n_frames, int w, int h, uint8_t* d_bgr,
                             ptrdiff_t stride, ptrdiff_t frame_stride);
int  lvm_mjpeg_encode_device(lvm_ctx* ctx, const uint8_t* d_bgr, int w, int h, ptrdiff_t stride, ptrdiff_t frame_stride, int n_frames,
                             int quality, uint8_t* out, size_t out_capacity, size_t* offsets);
int  lvm_export_frames_mjpeg(lvm_ctx* ctx, const lvm_preprocess_params* pp, const lvm_params* p, int split, int n_frames,
                             const uint8_t* const* frames, int w, int h, int channels, ptrdiff_t in_stride, int quality,
                             uint8_t* out, size_t out_capacity, size_t* offsets, int* produced);

/* Cross-frame software pipeline for lvm_process_device (throughput mode, default depth 0).
 * depth 1 (implemented for the Laplace mode; other modes ignore it): a call enqueues the
 * down-sweep of ITS frame on an internal second stream concurrently with the up-sweep + output of
 * the PREVIOUS frame, so the output of call t is written (into the d_out given at call t) when
 * call t+1 -- or lvm_flush -- has been enqueued.  d_in / d_out of a call must stay valid until
 * then.  Results are identical to depth 0; only the schedule differs.  lvm_process (host path) and
 * the reference shim always run at depth 0.                                                   */
int  lvm_set_pipeline(lvm_ctx* ctx, int depth);
/* Enqueue whatever the pipeline still holds on `hip_stream` (NULL = the context's own stream). */
int  lvm_flush(lvm_ctx* ctx, void* hip_stream);

/* Wait for everything enqueued by lvm_process_device on the context's own stream. */
int  lvm_synchronize(lvm_ctx* ctx);

const char* lvm_last_error(lvm_ctx* ctx);

/* processing/magnification/SpatialFilter.cpp:5-11 calculateMaxLevels */
int  lvm_max_levels(int w, int h);
/* processing/magnification/TemporalFilter.cpp:82-94 getOptimalBufferSize */
int  lvm_optimal_buffer_size(int fps);
/* processing/magnification/TemporalFilter.cpp:280-297 butterworth(N = 2, Wn) */
void lvm_butterworth2(double Wn, double a[3], double b[3]);

/* ---- instrumentation (not part of the reference surface) -------------------------------- */
/* Keep the pre-quantisation float frame of stream 0 (parity metric).
 * lvm_debug_read_float copies w*h*channels floats (interleaved like the u8 output).        */
int  lvm_debug_keep_float(lvm_ctx* ctx, int on);
int  lvm_debug_read_float(lvm_ctx* ctx, float* dst, size_t count);
/* Colour arithmetic.  The forward conversion is what cv::cvtColor(COLOR_BGR2Lab) on CV_32F computes in OpenCV 4
 * (MagnifyCore.hpp:90,219): RGB2Labfloat's trilinear interpolation in a 33 x 33 x 33 int16 table -- integer work, bit-exact.
 * lvm_debug_exact_lab(1): every float operation around it (inverse conversion, pyramid taps, Riesz amplify) in OpenCV's
 * operation order with true divisions -- bit-faithful to the CPU oracle on the emulation build, used by the kernel-logic
 * tests; 0 (default): reciprocal multiplies / fma chains / hardware transcendentals where the value feeds
 * well-conditioned math.  lvm_debug_lab_analytic(1): the cube-root form RGB2Lab_f computes when OpenCV's interpolation is
 * switched off (implies the exact operation order).                                                                    */
int  lvm_debug_exact_lab(lvm_ctx* ctx, int on);
/* The output quantiser of the Lab modes -- u8 = saturate_cast<uchar>(cvRound(255 * invGamma(clip01(c)) + 1/255)) per channel
 * (Lab2RGBfloat + convertTo, MagnifyCore.hpp:152-153, :275-276) -- runs as a 4096-slice step table in the default flavour
 * (lab_tables.cpp build_u8_steps).  This entry compares the table with the operations it replaces for every float whose bit
 * pattern lies in [first_bits, first_bits + count) ON THE DEVICE: *mismatches = patterns where the bytes differ, *first_bad_bits =
 * the smallest of them.  (0, 1 << 32) sweeps every binary32 value: negative, above 1, infinities, NaN included.             */
/* Measurement aid (bench.py `clock_mhz`): a one-lane kernel on the context's auxiliary stream reads the shader-clock counter
 * (s_memtime) and the constant 100 MHz counter (s_memrealtime), waits -- beside whatever runs on the other streams -- until _stop is
 * called (or max_seconds, <= 5, have passed), and reads them again: *mhz = the AVERAGE shader clock over that interval, *seconds its
 * length.  One probe at a time per context.                                                                                        */
int  lvm_debug_clock_probe_start(lvm_ctx* ctx, double max_seconds);
int  lvm_debug_clock_probe_stop(lvm_ctx* ctx, double* mhz, double* seconds);
int  lvm_debug_sweep_u8_steps(lvm_ctx* ctx, uint32_t first_bits, uint64_t count, uint64_t* mismatches, uint32_t* first_bad_bits);
int  lvm_debug_lab_analytic(lvm_ctx* ctx, int on);
/* The forward table: LVM_LAB_LUT_ENTRIES int16 values in OpenCV's RGB2Labprev order, index 3 (p + 33 q + 1089 r) + channel
 * with p, q, r the R, G, B grid indices and entries L / 100 * 16384, (a + 128) / 256 * 16384, (b + 128) / 256 * 16384.
 * lvm_create builds it by restating initLabTabs (color_lab.cpp); lvm_set_lab_lut installs another one, e.g. the table of
 * a real OpenCV build recovered by converting the 33^3 node colours (oracle/ref_driver.cpp) -- the context must be idle. */
#define LVM_LAB_LUT_ENTRIES (33 * 33 * 33 * 3)
int  lvm_get_lab_lut(lvm_ctx* ctx, int16_t* dst);
int  lvm_set_lab_lut(lvm_ctx* ctx, const int16_t* src);
/* Per-kernel timing with HIP events recorded on the launch stream.  While enabled every
 * kernel launch is bracketed by two events; lvm_profile_collect synchronises and folds them
 * into per-kernel totals, readable with lvm_profile_entry (idx = 0..n-1).                   */
int  lvm_profile_enable(lvm_ctx* ctx, int on);
int  lvm_profile_collect(lvm_ctx* ctx);
/* Clears the totals and restricts the bracketing to launches with this report name (NULL or "": all launches).  With
 * one kernel bracketed its neighbours run back to back, as they do outside the profiling pass: the VALU-bound kernels
 * take ~10 % longer that way than with an event gap on both sides (the clock recovers in the gaps), and that is the
 * duration rocprofv3 reports for them.                                                                              */
int  lvm_profile_only(lvm_ctx* ctx, const char* name);
int  lvm_profile_entry(lvm_ctx* ctx, int idx, char* name, size_t name_cap, double* total_ms,
                       long long* launches);
/* Algorithmic bytes per frame per stream for the current geometry/mode. */
double lvm_algorithmic_bytes(int mode, int w, int h, int channels, int levels, double framerate);

#ifdef __cplusplus
}
#endif
#endif
