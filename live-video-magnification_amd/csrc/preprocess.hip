// preprocess.hip -- the two uint8 stages in front of the magnifier, on the device.
//
// Replaces PreprocessProcessor::process (reference: processing/PreprocessProcessor.cpp:10-51: ROI crop
// in normalised coordinates, cv::resize(INTER_AREA) by 1/2/4/8) and GrayscaleProcessor::process
// (processing/GrayscaleProcessor.cpp:7-16: cv::cvtColor(BGR2GRAY)).  One kernel does crop + decimation +
// gray per output pixel, so the full-resolution frame is read once and only the small frame is written:
//   * integer decimation (ROI size divisible by the divisor): block sum, (sum + 2) >> 2 for 2x2,
//     saturate_cast<uchar>(sum * (1.f / area)) otherwise (OpenCV resizeAreaFast_);
//   * any other size: the fractional-cell tables of OpenCV's general INTER_AREA path, float sums in table
//     order (resizeArea_), tables built on the host per geometry and cached in the context;
//   * gray: 15-bit fixed point (b*3735 + g*19235 + r*9798 + 2^14) >> 15, applied to the decimated bytes
//     exactly as the reference's stage order does.
// Byte/integer work: results are bit-exact against the oracle (tests/test_preprocess_*.py).
#include <cmath>
#include <cstring>
#include <vector>

#include "lvm_internal.h"

namespace lvm {

struct AreaTab { int si, di; float alpha; };

struct PreArgs {
    const uint8_t* in; long in_stride, in_sstride;     // already offset to the ROI origin
    uint8_t* out; long out_stride, out_sstride;
    int ow, oh, cn, ocn;                               // output size, input channels, output channels
    int sx, sy;                                        // integer decimation factors (fast path)
    uint8_t* tap; long tap_stride, tap_sstride;        // optional: the 3-channel frame BEFORE BGR2GRAY (runChainOnce's `original`, ChainBuilder.cpp:25)
    const AreaTab* xtab; const int* xk;                // general path: entries of output column dx are xtab[xk[dx] .. xk[dx+1])
    const AreaTab* ytab; const int* yk;
};

__device__ __forceinline__ uint8_t gray15(int b, int g, int r) { return (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15); }

// MODE 0 = crop only, 1 = integer decimation, 2 = fractional area tables.  One thread per output pixel;
// blockIdx.z = stream.
template <int MODE, int CN>
__global__ __launch_bounds__(256) void k_preprocess(PreArgs a) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= a.ow || dy >= a.oh) return;
    const uint8_t* src = a.in + (size_t)blockIdx.z * a.in_sstride;
    uint8_t* q = a.out + (size_t)blockIdx.z * a.out_sstride + (size_t)dy * a.out_stride + (size_t)dx * a.ocn;
    int v[CN];
    if (MODE == 0) {
        const uint8_t* p = src + (size_t)dy * a.in_stride + (size_t)dx * CN;
#pragma unroll
        for (int c = 0; c < CN; ++c) v[c] = p[c];
    } else if (MODE == 1) {
        int sum[CN];
#pragma unroll
        for (int c = 0; c < CN; ++c) sum[c] = 0;
        for (int yy = 0; yy < a.sy; ++yy) {
            const uint8_t* p = src + (size_t)(dy * a.sy + yy) * a.in_stride + (size_t)(dx * a.sx) * CN;
            for (int xx = 0; xx < a.sx; ++xx) {
#pragma unroll
                for (int c = 0; c < CN; ++c) sum[c] += p[xx * CN + c];
            }
        }
        const bool fast2 = a.sx == 2 && a.sy == 2;
        const float scale = 1.f / (float)(a.sx * a.sy);
#pragma unroll
        for (int c = 0; c < CN; ++c) v[c] = fast2 ? ((sum[c] + 2) >> 2) : (int)sat_u8((float)sum[c] * scale);
    } else {
        float sum[CN];
        const int x0 = a.xk[dx], x1 = a.xk[dx + 1], y0 = a.yk[dy], y1 = a.yk[dy + 1];
        for (int j = y0; j < y1; ++j) {
            const float beta = a.ytab[j].alpha;
            const uint8_t* S = src + (size_t)a.ytab[j].si * a.in_stride;
            float buf[CN];
#pragma unroll
            for (int c = 0; c < CN; ++c) buf[c] = 0.f;
            for (int k = x0; k < x1; ++k) {
                const float alpha = a.xtab[k].alpha;
                const uint8_t* p = S + (size_t)a.xtab[k].si * CN;
#pragma unroll
                for (int c = 0; c < CN; ++c) buf[c] = buf[c] + (float)p[c] * alpha;
            }
#pragma unroll
            for (int c = 0; c < CN; ++c) sum[c] = (j == y0) ? beta * buf[c] : sum[c] + beta * buf[c];
        }
#pragma unroll
        for (int c = 0; c < CN; ++c) v[c] = (int)sat_u8(sum[c]);
    }
    if (CN == 3 && a.ocn == 1) {
        q[0] = gray15(v[0], v[1], v[2]);
        if (a.tap) {        // PreprocessProcessor's own output: what the chain hands on as `original` before GrayscaleProcessor runs
            uint8_t* t = a.tap + (size_t)blockIdx.z * a.tap_sstride + (size_t)dy * a.tap_stride + (size_t)dx * 3;
            t[0] = (uint8_t)v[0]; t[1] = (uint8_t)v[1]; t[2] = (uint8_t)v[2];
        }
    } else {
#pragma unroll
        for (int c = 0; c < CN; ++c) q[c] = (uint8_t)v[c];
    }
}

// ---- host ----------------------------------------------------------------------------------------
// PreprocessProcessor.cpp:13-31, :36-39; GrayscaleProcessor.cpp:8-9
void preprocess_geometry(const lvm_preprocess_params& pp, int w, int h, int channels, int* rx, int* ry, int* rw, int* rh,
                         int* ow, int* oh, int* och) {
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const int divisor = clampi(pp.downscale, 1, 8);
    int x = 0, y = 0, cw = w, chh = h;
    if (pp.roi_enabled) {
        x = (int)std::lround((double)pp.roiX * w);
        y = (int)std::lround((double)pp.roiY * h);
        cw = (int)std::lround((double)pp.roiW * w);
        chh = (int)std::lround((double)pp.roiH * h);
        x = clampi(x, 0, w - 1); y = clampi(y, 0, h - 1);
        cw = clampi(cw, 1, w - x); chh = clampi(chh, 1, h - y);
    }
    *rx = x; *ry = y; *rw = cw; *rh = chh;
    *ow = divisor > 1 ? (cw / divisor > 1 ? cw / divisor : 1) : cw;
    *oh = divisor > 1 ? (chh / divisor > 1 ? chh / divisor : 1) : chh;
    *och = (pp.grayscale && channels != 1) ? 1 : channels;
}

// fractional source cells of one axis (OpenCV computeResizeAreaTab), plus the first entry of every output index
static void area_table(int ssize, int dsize, double scale, std::vector<AreaTab>& tab, std::vector<int>& first) {
    tab.clear(); first.assign((size_t)dsize + 1, 0);
    for (int dx = 0; dx < dsize; ++dx) {
        first[(size_t)dx] = (int)tab.size();
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, dx, (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; ++sx) tab.push_back({sx, dx, (float)(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3) tab.push_back({sx2, dx, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
    }
    first[(size_t)dsize] = (int)tab.size();
}

struct PreTables {                       // cached per (roi size, output size)
    int rw = 0, rh = 0, ow = 0, oh = 0;
    AreaTab *xtab = nullptr, *ytab = nullptr; int *xk = nullptr, *yk = nullptr;
    void release() { for (void* p : {(void*)xtab, (void*)ytab, (void*)xk, (void*)yk}) if (p) (void)hipFree(p); xtab = ytab = nullptr; xk = yk = nullptr; rw = rh = ow = oh = 0; }
};

void preprocess_release(Ctx* c) {
    PreTables* t = static_cast<PreTables*>(c->pre_tables);
    if (t) { t->release(); delete t; c->pre_tables = nullptr; }
}

int preprocess_device(Ctx* c, const lvm_preprocess_params& pp, const uint8_t* d_in, int w, int h, int channels, ptrdiff_t in_stride,
                      ptrdiff_t in_sstride, uint8_t* d_out, ptrdiff_t out_stride, ptrdiff_t out_sstride, hipStream_t s,
                      uint8_t* d_tap, ptrdiff_t tap_stride, ptrdiff_t tap_sstride) {
    if (!d_in || !d_out || w <= 0 || h <= 0 || (channels != 1 && channels != 3)) { c->err = "preprocess: bad frame arguments"; return LVM_ERR_INVALID; }
    int rx, ry, rw, rh, ow, oh, och;
    preprocess_geometry(pp, w, h, channels, &rx, &ry, &rw, &rh, &ow, &oh, &och);
    if (out_stride < (ptrdiff_t)ow * och) { c->err = "preprocess: output stride too small"; return LVM_ERR_INVALID; }
    PreArgs a{};
    a.in = d_in + (size_t)ry * in_stride + (size_t)rx * channels; a.in_stride = (long)in_stride; a.in_sstride = (long)in_sstride;
    a.out = d_out; a.out_stride = (long)out_stride; a.out_sstride = (long)out_sstride;
    a.ow = ow; a.oh = oh; a.cn = channels; a.ocn = och;
    a.tap = (channels == 3 && och == 1) ? d_tap : nullptr; a.tap_stride = (long)tap_stride; a.tap_sstride = (long)tap_sstride;
    const int divisor = pp.downscale < 1 ? 1 : (pp.downscale > 8 ? 8 : pp.downscale);
    int mode = 0;
    if (divisor > 1) {
        // cv::resize(INTER_AREA): integer factors take resizeAreaFast_, anything else the fractional tables
        const double scx = (double)rw / ow, scy = (double)rh / oh;
        const int isx = (int)std::lrint(scx), isy = (int)std::lrint(scy);
        if (std::fabs(scx - isx) < 2.220446049250313e-16 && std::fabs(scy - isy) < 2.220446049250313e-16) {
            mode = 1; a.sx = isx; a.sy = isy;
        } else {
            mode = 2;
            PreTables* t = static_cast<PreTables*>(c->pre_tables);
            if (!t) { t = new PreTables(); c->pre_tables = t; }
            if (t->rw != rw || t->rh != rh || t->ow != ow || t->oh != oh) {
                LVM_HIP_TRY(c, hipStreamSynchronize(s));            // the old tables may still be in use
                t->release();
                std::vector<AreaTab> xt, yt; std::vector<int> xf, yf;
                area_table(rw, ow, scx, xt, xf);
                area_table(rh, oh, scy, yt, yf);
                LVM_HIP_TRY(c, hipMalloc((void**)&t->xtab, xt.size() * sizeof(AreaTab)));
                LVM_HIP_TRY(c, hipMalloc((void**)&t->ytab, yt.size() * sizeof(AreaTab)));
                LVM_HIP_TRY(c, hipMalloc((void**)&t->xk, xf.size() * sizeof(int)));
                LVM_HIP_TRY(c, hipMalloc((void**)&t->yk, yf.size() * sizeof(int)));
                LVM_HIP_TRY(c, hipMemcpy(t->xtab, xt.data(), xt.size() * sizeof(AreaTab), hipMemcpyHostToDevice));
                LVM_HIP_TRY(c, hipMemcpy(t->ytab, yt.data(), yt.size() * sizeof(AreaTab), hipMemcpyHostToDevice));
                LVM_HIP_TRY(c, hipMemcpy(t->xk, xf.data(), xf.size() * sizeof(int), hipMemcpyHostToDevice));
                LVM_HIP_TRY(c, hipMemcpy(t->yk, yf.data(), yf.size() * sizeof(int), hipMemcpyHostToDevice));
                t->rw = rw; t->rh = rh; t->ow = ow; t->oh = oh;
            }
            a.xtab = t->xtab; a.xk = t->xk; a.ytab = t->ytab; a.yk = t->yk;
        }
    }
    const dim3 grid((ow + 63) / 64, (oh + 3) / 4, c->nstreams), blk(256);
    if (channels == 3) {
        auto k = mode == 0 ? k_preprocess<0, 3> : (mode == 1 ? k_preprocess<1, 3> : k_preprocess<2, 3>);
        LVM_LAUNCH(c, "preprocess", k, grid, blk, s, a);
    } else {
        auto k = mode == 0 ? k_preprocess<0, 1> : (mode == 1 ? k_preprocess<1, 1> : k_preprocess<2, 1>);
        LVM_LAUNCH(c, "preprocess", k, grid, blk, s, a);
    }
    LVM_HIP_TRY(c, hipGetLastError());
    return LVM_OK;
}

}  // namespace lvm
