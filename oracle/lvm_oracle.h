/*
 * lvm_oracle.h -- CPU ORACLE for the Eulerian video-magnification hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, OpenCV-free restatement of the
 * reference's per-frame magnifiers (reference: src/processing/MagnificationProcessor.cpp:17-67,
 * src/processing/magnification/{MagnifyCore.hpp,SpatialFilter.cpp,TemporalFilter.cpp,
 * RieszPyramid.cpp,ComplexMat.hpp}).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (liblvm_hip.so) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" at the OpenCV boundary.  The reference has no tests,
 * golden frames or fixtures, and its arithmetic lives in OpenCV 4 (vcpkg, un-vendored, absent
 * here).  OpenCV primitives are restated from their published semantics (see DESIGN.md
 * "Oracle").  The OpenCV-free slices of the reference (butterworth, getOptimalBufferSize)
 * ARE pinned: they are compiled from /root/reference in place (oracle/Makefile ->
 * oracle/_ref/) and checked against this restatement and tests/golden/.
 */
#ifndef LVM_ORACLE_H
#define LVM_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* reference: IProcessor.hpp:10 (same numeric order as the enum class) */
enum { LVMO_MODE_LAPLACE = 0, LVMO_MODE_PHASE = 1, LVMO_MODE_COLOR = 2, LVMO_MODE_NONE = 3 };

/* reference: IProcessor.hpp:14-23 (+ a 64-bit key standing for PreprocessParams equality,
 * IProcessor.hpp:26-41, which StructuralTracker compares: MagnifyCore.hpp:55-56) */
typedef struct lvmo_params {
    int32_t  mode;
    int32_t  levels;
    double   amplification;
    double   coWavelength;
    double   coLow;
    double   coHigh;
    double   chromAttenuation;
    double   framerate;
    uint64_t preprocess_key;
} lvmo_params;

typedef struct lvmo_ctx lvmo_ctx;

lvmo_ctx* lvmo_create(void);
void      lvmo_destroy(lvmo_ctx*);
void      lvmo_reset(lvmo_ctx*);                     /* MagnificationProcessor.cpp:10-15 */
void      lvmo_set_threads(int n);                   /* OpenMP threads for the timed baseline */
/* MagnificationProcessor.cpp:17-67.  *produced==0 => caller shows the input frame. */
int lvmo_process(lvmo_ctx*, const lvmo_params*, const uint8_t* in, int w, int h, int channels,
                 ptrdiff_t in_stride, uint8_t* out, ptrdiff_t out_stride, int* produced);
/* pre-quantisation float frame of the last produced output (interleaved, w*h*channels) */
const float* lvmo_last_float(lvmo_ctx*, int* w, int* h, int* c);
/* Colour mode: min/max used by the final rescale (MagnifyCore.hpp:200-203) */
void lvmo_last_minmax(lvmo_ctx*, double* mn, double* mx);

/* ---- primitives (exported for unit tests) ---------------------------------------------- */
int  lvmo_max_levels(int w, int h);                                   /* SpatialFilter.cpp:5-11 */
int  lvmo_optimal_buffer_size(int fps);                               /* TemporalFilter.cpp:82-94 */
void lvmo_butterworth2(double Wn, double a[3], double b[3]);          /* TemporalFilter.cpp:280-297 (N=2) */
void lvmo_laplace_gains(int w, int h, int levels, double amplification, double coWavelength,
                        float* gains /* levels+1 */);                 /* MagnifyCore.hpp:114-134 */
void lvmo_pyr_down(const float* src, int w, int h, int cn, float* dst);
void lvmo_pyr_up(const float* src, int w, int h, int cn, float* dst, int dw, int dh);
void lvmo_bgr2lab(const float* src, int npix, float* dst);
void lvmo_lab2bgr(const float* src, int npix, float* dst);
void lvmo_filter2d(const float* src, int w, int h, const float* k, int kw, int kh, float* dst);
void lvmo_gauss_kernel(int n, double sigma, float* k);
void lvmo_sep_filter(const float* src, int w, int h, const float* k, int n, float* dst);
void lvmo_resize_linear(const float* src, int w, int h, int cn, float* dst, int dw, int dh);
void lvmo_dft_rows(const float* src, int rows, int n, float* dst);    /* DFT_ROWS|DFT_SCALE, CCS */
void lvmo_idft_rows(const float* src, int rows, int n, float* dst);   /* DFT_ROWS|DFT_SCALE, CCS */
void lvmo_mul_spectrums_rows(const float* a, const float* b, int rows, int n, float* dst);
void lvmo_ideal_filter(const float* win, int rows, int cols, int cn, double lo, double hi,
                       double fps, float* dst, int full);             /* TemporalFilter.cpp:24-57 */
void lvmo_riesz_kernels(float lp[81], float hp[81]);                  /* RieszPyramid.cpp:146-167 */
float lvmo_cube_root(float v);
const float* lvmo_gamma_tab(int inverse);  /* 1024*4 spline coefficients */
/* the output quantiser cvRound(255 * invGamma(clip01(c)) + 1/255) of one channel, and its exhaustive sweep (lvm_oracle.c) */
uint8_t lvmo_u8_of_linear(float c);
void lvmo_u8_of_linear_sweep(uint32_t first_bits, uint64_t count, uint64_t* descents, uint64_t* steps, uint64_t* jumps, uint32_t* thr /* [256] */);

/* ---- the two stages in front of the magnifier --------------------------------
 * PreprocessProcessor::process (processing/PreprocessProcessor.cpp:10-51): ROI crop in normalised
 * coordinates + cv::resize(INTER_AREA) by 1/2/4/8, and GrayscaleProcessor::process
 * (processing/GrayscaleProcessor.cpp:7-16): cv::cvtColor(BGR2GRAY) on the uint8 frame.            */
typedef struct lvmo_pre_params {
    int32_t downscale;      /* clamped to [1, 8] like PreprocessProcessor.cpp:14 */
    int32_t roi_enabled;
    float   roiX, roiY, roiW, roiH;
    int32_t grayscale;      /* ProcessorConfig::grayscale */
} lvmo_pre_params;
/* geometry of the stage output: ROI rectangle inside the source and the output size/channels */
void lvmo_preprocess_geometry(const lvmo_pre_params* pp, int w, int h, int channels, int* rx, int* ry,
                              int* rw, int* rh, int* ow, int* oh, int* och);
/* out: ow*oh*och bytes, contiguous rows */
void lvmo_preprocess(const lvmo_pre_params* pp, const uint8_t* in, int w, int h, int channels,
                     ptrdiff_t in_stride, uint8_t* out);
/* cv::resize(INTER_AREA) on uint8, scale >= 1 in both directions (fast integer path or the general
 * area tables), and the fixed-point BGR2GRAY */
void lvmo_resize_area_u8(const uint8_t* src, int w, int h, int cn, ptrdiff_t stride, uint8_t* dst, int dw, int dh);
void lvmo_bgr2gray_u8(const uint8_t* src, int npix, uint8_t* dst);
/* the decimation table of the general path: returns the number of entries written (<= cap) */
typedef struct lvmo_area_tab { int si, di; float alpha; } lvmo_area_tab;
int  lvmo_area_table(int ssize, int dsize, double scale, lvmo_area_tab* tab, int cap);

/* Forward Lab flavour of the oracle: 1 (default) OpenCV 4's default trilinear-LUT path (RGB2Labfloat::useInterpolation,
 * what cv::cvtColor(COLOR_BGR2Lab) on CV_32F runs), 0 the analytic float path (OpenCV with interpolation disabled) */
void lvmo_set_lab_lut(int on);
/* the 33^3 x 3 int16 table in RGB2Labprev order (3 (p + 33 q + 1089 r) + channel); override with a real build's table */
void lvmo_lab_lut_table(int16_t* out);
void lvmo_lab_lut_override(const int16_t* tab);

/* Unpinned OpenCV build choices as switches (bit mask, 0 = the restatement the parity tests use; lvm_oracle.c "unpinned OpenCV
 * build choices").  tests/test_oracle_variants.py runs BASELINE configs under every bit and records the envelope. */
enum {
    LVMO_VAR_PYR_SIMD        = 1,    /* pyrDown / pyrUp sums in the association of OpenCV's SIMD loops (PyrDownVecH/V, PyrUpVecV) instead of the scalar loops' */
    LVMO_VAR_FILTER_UNFUSED  = 2,    /* filter2D / sepFilter2D taps as multiply + add (SSE2 baseline dispatch) instead of fma (AVX2 / NEON) */
    LVMO_VAR_ADDW_FUSED      = 4,    /* Laplace IIR a*alpha + b*beta as addWeighted's SIMD form fma(a, alpha, b*beta) instead of two products + sum */
    LVMO_VAR_MUL_F32         = 8,    /* Riesz IIR Mat * double with the scalar narrowed to float first instead of a float64 product rounded once */
    LVMO_VAR_GAMMA_F32       = 16,   /* forward Lab table: applyGamma as three binary32 operations (the rounds 1-3 restatement) instead of softdouble pow rounded once */
    LVMO_VAR_LUT_NUDGE_UP    = 32,   /* forward Lab table: every interior gamma node one binary32 step up (bounds a last-bit disagreement of pow) */
    LVMO_VAR_LUT_NUDGE_DOWN  = 64,   /* ... one step down */
    LVMO_VAR_SPLINE_CV3      = 128,  /* gamma / inverse-gamma splines in the OpenCV 3.x form of splineBuild (forward sweep to n-2, x 0.3333333333333333f) instead of
                                        OpenCV 4's (sweep to n-1, / 3): what rounds 1-4 restated */
    LVMO_VAR_DFT_F32         = 256,  /* Color band-pass: both transforms in binary32 (radix-2 FFT for power-of-two windows) instead of float64 direct sums */
    LVMO_VAR_FILTER_DFT      = 512   /* filter2D with kw * kh >= 50 through its DFT path (crossCorr; builds without SSE3 = every ARM build): the 9 x 9 Riesz kernels as
                                        float64 sums rounded once instead of binary32 fma chains */
};
void     lvmo_set_variant(unsigned mask);
unsigned lvmo_get_variant(void);

/* Exporter::compose (export/Exporter.cpp:53-88) without the text overlay */
int lvmo_compose_geometry(int split, int ow, int oh, int pw, int ph, int* cw, int* ch);
int lvmo_compose(int split, const uint8_t* orig, int ow, int oh, int och, ptrdiff_t ostride, const uint8_t* proc, int pw, int ph,
                 int pch, ptrdiff_t pstride, uint8_t* canvas, ptrdiff_t cstride);

#ifdef __cplusplus
}
#endif
#endif
