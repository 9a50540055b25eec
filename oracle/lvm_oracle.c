/*
 * lvm_oracle.c -- CPU ORACLE (test infrastructure; see lvm_oracle.h for the rules).
 *
 * Plain-C restatement of the reference hot path.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference/src/processing/).  OpenCV
 * primitives are restated from the published semantics of OpenCV 4.x imgproc/core
 * ("[cv]" comments).  PARITY UNPINNED at the OpenCV boundary (no OpenCV in this image,
 * no golden frames in the reference); pinned for butterworth/getOptimalBufferSize.
 *
 * Arithmetic rules: float32 storage everywhere (CV_32F), every Mat-level operation is
 * rounded on its own (the reference runs one cv:: call per operation, so nothing can
 * fuse) => build with -ffp-contract=off.  Coefficient design is float64.
 */
#include "lvm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_D 3.1415926535897932384626433832795 /* CV_PI */

/* ------------------------------------------------------------------------------------- */
/* small helpers                                                                           */
/* ------------------------------------------------------------------------------------- */
typedef struct Img { int w, h, c; float* d; } Img;

static Img img_alloc(int w, int h, int c) {
    Img m; m.w = w; m.h = h; m.c = c;
    m.d = (float*)calloc((size_t)w * h * c > 0 ? (size_t)w * h * c : 1, sizeof(float));
    return m;
}
static void img_free(Img* m) { free(m->d); m->d = NULL; m->w = m->h = m->c = 0; }
static Img img_clone(const Img* s) {
    Img m = img_alloc(s->w, s->h, s->c);
    memcpy(m.d, s->d, (size_t)s->w * s->h * s->c * sizeof(float));
    return m;
}
static size_t img_count(const Img* m) { return (size_t)m->w * m->h * m->c; }

/* [cv] borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p; else p = 2 * len - 2 - p;
    }
    return p;
}
/* [cv] cvRound: round half to even; saturate_cast<uchar> */
static inline uint8_t sat_u8(float v) {
    if (!(v == v)) return 0;
    float r = nearbyintf(v); /* default rounding mode = nearest-even */
    if (r < 0.f) return 0;
    if (r > 255.f) return 255;
    return (uint8_t)r;
}

void lvmo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- unpinned OpenCV build choices as SWITCHES (tests/test_oracle_variants.py) -------------
 * No OpenCV exists in this image, so a handful of choices that depend on the OpenCV build (CPU dispatch of its SIMD
 * loops, scalar type of Mat * double) cannot be checked.  Each is a bit here; 0 = the restatement every parity test
 * runs against.  The variant tests run whole clips under every bit and report how far the frames move: that envelope,
 * not an argument, is what bounds "oracle vs a real OpenCV build".                                                    */
static unsigned g_var = 0;
static void lab_lut_drop(void);
static void lab_tabs_drop(void);
void lvmo_set_variant(unsigned mask) {
    const unsigned lut_bits = LVMO_VAR_GAMMA_F32 | LVMO_VAR_LUT_NUDGE_UP | LVMO_VAR_LUT_NUDGE_DOWN;
    if ((mask ^ g_var) & lut_bits) lab_lut_drop();
    if ((mask ^ g_var) & LVMO_VAR_SPLINE_CV3) lab_tabs_drop();       /* the gamma splines are rebuilt on the next use */
    g_var = mask;
}
unsigned lvmo_get_variant(void) { return g_var; }
/* one filter / pyramid tap: s + k * v.  [cv] v_muladd == v_fma in the AVX2 / AVX-512 / NEON dispatches (one rounding);
 * the baseline SSE2 dispatch rounds the product first (LVMO_VAR_FILTER_UNFUSED).                                      */
static inline float tap(float k, float v, float s) { return (g_var & LVMO_VAR_FILTER_UNFUSED) ? s + k * v : fmaf(k, v, s); }

/* ------------------------------------------------------------------------------------- */
/* scalar helpers of the reference that need no OpenCV                                     */
/* ------------------------------------------------------------------------------------- */
/* SpatialFilter.cpp:5-11 */
int lvmo_max_levels(int w, int h) {
    if (w > 5 && h > 5) return 1 + lvmo_max_levels((1 + w) / 2, (1 + h) / 2);
    return 0;
}
/* TemporalFilter.cpp:82-94 */
int lvmo_optimal_buffer_size(int fps) {
    unsigned int r = (unsigned int)(2 * fps > 16 ? 2 * fps : 16);
    r--; r |= r >> 1; r |= r >> 2; r |= r >> 4; r |= r >> 8; r |= r >> 16; r++;
    return (int)r;
}

/* TemporalFilter.cpp:97-297 specialised to N == 2 (the only order the hot path uses,
 * TemporalFilter.cpp:326).  Same pipeline: analog prototype poles -> polynomial ->
 * toLowpass(w0) -> bilinear(fs = 2) -> normalise, in float64.  For N = 2 the prototype
 * denominator is s^2 + sqrt(2) s + 1 and the numerator is 1.                             */
void lvmo_butterworth2(double Wn, double a[3], double b[3]) {
    const double fs = 2.0;
    const double w0 = 2.0 * fs * tan(PI_D * Wn / fs);           /* :283 */
    /* prototypeAnalogButterworth(2) (:266-274): poles exp(j(2k-1)pi/4)*j, k=1,2;
     * polynomialCoefficients (:107-143) => a = [1, -(p1+p2), p1*p2], b = [1]           */
    const double p1r = -sin(1.0 / 4.0 * PI_D), p1i = cos(1.0 / 4.0 * PI_D);
    const double p2r = -sin(3.0 / 4.0 * PI_D), p2i = cos(3.0 / 4.0 * PI_D);
    double A0 = 1.0, A1 = -(p1r + p2r), A2 = p1r * p2r - p1i * p2i;
    double B0 = 1.0;
    /* toLowpass (:230-262): d = 3, n = 1, M = 3, pwo = [w0^2, w0, 1], start1 = 0,
     * start2 = 2: b[0] *= pwo[0]/pwo[2]; a[k] *= pwo[0]/pwo[k]; pwo[k]==0 => coefficient 0;
     * then normalize by a[0].                                                            */
    double pwo[3] = { pow(w0, 2.0), pow(w0, 1.0), pow(w0, 0.0) };
    double bl = (pwo[2] == 0.0) ? 0.0 : B0 * (pwo[0] / pwo[2]);
    double al[3];
    double ain[3] = { A0, A1, A2 };
    for (int k = 0; k < 3; ++k) al[k] = (pwo[k] == 0.0) ? 0.0 : ain[k] * (pwo[0] / pwo[k]);
    {
        double lead = al[0];
        for (int k = 0; k < 3; ++k) al[k] = (lead == 0.0) ? 0.0 : al[k] / lead;
        bl = (lead == 0.0) ? 0.0 : bl / lead;
    }
    /* bilinearTransform (:185-226) with D = 2, N = 0, M = 2:
     *   bprime[j] = sum_{l==j} C(2,l) * b[0]                          (i = 0 only)
     *   aprime[j] = sum_{i,k,l: k+l==j} C(i,k) C(2-i,l) a[2-i] (2fs)^i (-1)^k           */
    double bp[3], ap[3];
    static const double C2[3] = { 1.0, 2.0, 1.0 };
    for (int j = 0; j < 3; ++j) bp[j] = C2[j] * bl;
    for (int j = 0; j < 3; ++j) {
        double val = 0.0;
        for (int i = 0; i <= 2; ++i)
            for (int k = 0; k <= i; ++k)
                for (int l = 0; l <= 2 - i; ++l)
                    if (k + l == j) {
                        double cik = (i == 2 && k == 1) ? 2.0 : 1.0;
                        double cml = (2 - i == 2 && l == 1) ? 2.0 : 1.0;
                        val += cik * cml * al[2 - i] * pow(2.0 * fs, (double)i) * pow(-1.0, (double)k);
                    }
        ap[j] = val;
    }
    {
        double lead = ap[0];
        for (int k = 0; k < 3; ++k) {
            a[k] = (lead == 0.0) ? 0.0 : ap[k] / lead;
            b[k] = (lead == 0.0) ? 0.0 : bp[k] / lead;
        }
    }
}

/* MagnifyCore.hpp:114-134: per-level gain; gains[levels] and gains[0] are 0.              */
void lvmo_laplace_gains(int w, int h, int levels, double amplification, double coWavelength,
                        float* gains) {
    const float delta = (float)(coWavelength / (8.0 * (1.0 + amplification)));
    const float exaggeration = 2.0f;
    float lambda = (float)(sqrt((double)(w * w + h * h)) / 3.0);
    for (int l = levels; l >= 0; --l) {
        const float currAlpha = (float)((lambda / (delta * 8.0) - 1.0) * exaggeration);
        const float amp = (float)amplification;
        gains[l] = (l == levels || l == 0) ? 0.0f : (amp < currAlpha ? amp : currAlpha);
        lambda = (float)(lambda / 2.0);
    }
}

/* ------------------------------------------------------------------------------------- */
/* [cv] pyrDown / pyrUp (imgproc/pyramids.cpp, float path, BORDER_REFLECT_101)             */
/* ------------------------------------------------------------------------------------- */
/* horizontal: row[x] = s[2x]*6 + (s[2x-1]+s[2x+1])*4 + s[2x-2] + s[2x+2]
 * vertical:   dst    = (r2*6 + (r1+r3)*4 + r0 + r4) * (1/256)                             */
void lvmo_pyr_down(const float* src, int w, int h, int cn, float* dst) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    float* rows = (float*)malloc((size_t)h * dw * cn * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float* s = src + (size_t)y * w * cn;
        float* r = rows + (size_t)y * dw * cn;
        for (int x = 0; x < dw; ++x) {
            const int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = 2 * x,
                      x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
            for (int c = 0; c < cn; ++c) {
                const float t0 = s[x0 * cn + c], t1 = s[x1 * cn + c], t2 = s[x2 * cn + c], t3 = s[x3 * cn + c], t4 = s[x4 * cn + c];
                /* [cv] scalar loop (and the form every parity test uses); LVMO_VAR_PYR_SIMD: PyrDownVecH's
                 * v_muladd(r2, 6, v_muladd(r1 + r3, 4, r0 + r4)) */
                r[x * cn + c] = (g_var & LVMO_VAR_PYR_SIMD) ? fmaf(t2, 6.f, fmaf(t1 + t3, 4.f, t0 + t4))
                                                            : t2 * 6.f + (t1 + t3) * 4.f + t0 + t4;
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
        const float* r0 = rows + (size_t)reflect101(2 * y - 2, h) * dw * cn;
        const float* r1 = rows + (size_t)reflect101(2 * y - 1, h) * dw * cn;
        const float* r2 = rows + (size_t)(2 * y) * dw * cn;
        const float* r3 = rows + (size_t)reflect101(2 * y + 1, h) * dw * cn;
        const float* r4 = rows + (size_t)reflect101(2 * y + 2, h) * dw * cn;
        float* d = dst + (size_t)y * dw * cn;
        if (g_var & LVMO_VAR_PYR_SIMD)         /* [cv] PyrDownVecV<float>: v_muladd(r1 + r3 + r2, 4, r0 + r4 + (r2 + r2)) * scale */
            for (int i = 0; i < dw * cn; ++i) d[i] = fmaf(r1[i] + r3[i] + r2[i], 4.f, r0[i] + r4[i] + (r2[i] + r2[i])) * (1.f / 256.f);
        else
        for (int i = 0; i < dw * cn; ++i)
            d[i] = (r2[i] * 6.f + (r1[i] + r3[i]) * 4.f + r0[i] + r4[i]) * (1.f / 256.f);
    }
    free(rows);
}

/* horizontal (per source row): even[2x] = s[x-1] + s[x]*6 + s[x+1], odd[2x+1] = (s[x]+s[x+1])*4;
 *   x == 0: even = s[0]*6 + s[1]*2, odd = (s[0]+s[1])*4;  x == w-1: even = s[w-2] + s[w-1]*7,
 *   odd = s[w-1]*8.  vertical (source rows y-1,y,y+1 with row -1 -> 1, row h -> h-1):
 *   dst[2y] = (r0 + r1*6 + r2)/64, dst[2y+1] = ((r1+r2)*4)/64.  dsize may be 2n or 2n-1; in
 *   the odd case the last odd row/col is not produced (odd row is written first to the same
 *   dst row and then overwritten by the even row).                                        */
void lvmo_pyr_up(const float* src, int w, int h, int cn, float* dst, int dw, int dh) {
    const int rw = 2 * w; /* full-width horizontal buffer */
    float* rows = (float*)malloc((size_t)h * rw * cn * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float* s = src + (size_t)y * w * cn;
        float* r = rows + (size_t)y * rw * cn;
        for (int c = 0; c < cn; ++c) {
            if (w == 1) { r[c] = r[cn + c] = s[c] * 8.f; continue; }
            r[c] = s[c] * 6.f + s[cn + c] * 2.f;
            r[cn + c] = (s[c] + s[cn + c]) * 4.f;
            const int sx = w - 1;
            r[(2 * sx) * cn + c] = s[(sx - 1) * cn + c] + s[sx * cn + c] * 7.f;
            r[(2 * sx + 1) * cn + c] = s[sx * cn + c] * 8.f;
            for (int x = 1; x < w - 1; ++x) {
                r[(2 * x) * cn + c] = s[(x - 1) * cn + c] + s[x * cn + c] * 6.f + s[(x + 1) * cn + c];
                r[(2 * x + 1) * cn + c] = (s[x * cn + c] + s[(x + 1) * cn + c]) * 4.f;
            }
        }
    }
    const int ncol = (dw < rw ? dw : rw) * cn;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const int ym = reflect101(2 * (y - 1), 2 * h) / 2, yp = reflect101(2 * (y + 1), 2 * h) / 2;
        const float* r0 = rows + (size_t)ym * rw * cn;
        const float* r1 = rows + (size_t)y * rw * cn;
        const float* r2 = rows + (size_t)yp * rw * cn;
        const int y0 = 2 * y, y1 = (2 * y + 1 < dh - 1) ? 2 * y + 1 : dh - 1;
        if (y0 >= dh) continue;
        float* d0 = dst + (size_t)y0 * dw * cn;
        float* d1 = dst + (size_t)y1 * dw * cn;
        for (int i = 0; i < ncol; ++i) {
            const float t1 = ((r1[i] + r2[i]) * 4.f) * (1.f / 64.f);
            /* LVMO_VAR_PYR_SIMD: [cv] PyrUpVecV<float>: scale * (v_muladd(6, r1, r0) + r2); the odd row's scale * 4 * (r1 + r2)
             * equals ((r1 + r2) * 4) * scale (power-of-two factors) */
            const float t0 = (g_var & LVMO_VAR_PYR_SIMD) ? (fmaf(6.f, r1[i], r0[i]) + r2[i]) * (1.f / 64.f)
                                                         : (r0[i] + r1[i] * 6.f + r2[i]) * (1.f / 64.f);
            d1[i] = t1;
            d0[i] = t0;
        }
        if (dw > rw) /* dsize = 2n+1: extra column copies the last odd column [cv] */
            for (int c = 0; c < cn; ++c) {
                d1[(dw - 1) * cn + c] = d1[(rw - 1) * cn + c];
                d0[(dw - 1) * cn + c] = d0[(rw - 1) * cn + c];
            }
    }
    if (dh > 2 * h) /* extra row copies row 2h-2 [cv] */
        memcpy(dst + (size_t)(dh - 1) * dw * cn, dst + (size_t)(2 * h - 2) * dw * cn,
               (size_t)dw * cn * sizeof(float));
    free(rows);
}

/* ------------------------------------------------------------------------------------- */
/* [cv] cvtColor float BGR<->Lab (imgproc/color_lab.cpp, analytic float path)              */
/* ------------------------------------------------------------------------------------- */
#define GAMMA_TAB_SIZE 1024
static float g_gamma_tab[GAMMA_TAB_SIZE * 4], g_invgamma_tab[GAMMA_TAB_SIZE * 4];
static float g_fwd[9], g_inv[9];
static float g_gamma_u8[256];
static int g_lab_ready = 0;
static void lab_tabs_drop(void) { g_lab_ready = 0; }

/* [cv] splineBuild: natural cubic spline through f[0..n] (n intervals), binary32 arithmetic.
 * Default (round 5) = the OpenCV 4 form (color_lab.cpp, softfloat): the forward sweep runs over i = 1 .. n-1 and the back
 * substitution DIVIDES by 3:  b = f[i+1] - f[i] - (cn + c*2)/3,  d = (cn - c)/3.
 * LVMO_VAR_SPLINE_CV3 = the OpenCV 3.x form rounds 1-4 restated: forward sweep i = 1 .. n-2 (entry n-1 of the static, hence
 * zero, table is read unwritten) and multiplications by 0.3333333333333333f.  The two differ in the last bit of b / d at a
 * fraction of the knots and in the top few knots (x -> 1): tests/test_oracle_variants.py bounds what that does to a frame. */
static void spline_build(const float* f, int n, float* tab) {
    const int cv3 = (g_var & LVMO_VAR_SPLINE_CV3) != 0;
    float cn = 0.f;
    tab[0] = tab[1] = 0.f;
    tab[(n - 1) * 4] = tab[(n - 1) * 4 + 1] = 0.f;
    for (int i = 1; i < (cv3 ? n - 1 : n); ++i) {
        float t = 3.f * (f[i + 1] - 2.f * f[i] + f[i - 1]);
        float l = 1.f / (4.f - tab[(i - 1) * 4]);
        tab[i * 4] = l;
        tab[i * 4 + 1] = (t - tab[(i - 1) * 4 + 1]) * l;
    }
    for (int i = n - 1; i >= 0; --i) {
        float c = tab[i * 4 + 1] - tab[i * 4] * cn;
        float b = cv3 ? f[i + 1] - f[i] - (cn + c * 2.f) * 0.3333333333333333f : f[i + 1] - f[i] - (cn + c * 2.f) / 3.f;
        float d = cv3 ? (cn - c) * 0.3333333333333333f : (cn - c) / 3.f;
        tab[i * 4] = f[i]; tab[i * 4 + 1] = b; tab[i * 4 + 2] = c; tab[i * 4 + 3] = d;
        cn = c;
    }
}
/* [cv] splineInterpolate */
static inline float spline_interp(float x, const float* tab, int n) {
    int ix = (int)x;
    if (ix < 0) ix = 0;
    if (ix > n - 1) ix = n - 1;
    x -= (float)ix;
    tab += ix * 4;
    return ((tab[3] * x + tab[2]) * x + tab[1]) * x + tab[0];
}

/* [cv] cv::cubeRoot (core/mathfuncs.cpp): exponent split + quartic rational polynomial */
float lvmo_cube_root(float value) {
    union { float f; int32_t i; uint32_t u; } v, m;
    v.f = value;
    int32_t ix = v.i & 0x7fffffff;
    uint32_t s = v.u & 0x80000000u;
    int ex = (ix >> 23) - 127;
    int shx = ex % 3;
    shx -= shx >= 0 ? 3 : 0;
    ex = (ex - shx) / 3;
    v.i = (ix & ((1 << 23) - 1)) | ((shx + 127) << 23);
    double fr = v.f;
    fr = (((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr +
             119.1654824285581628956914143) * fr + 13.43250139086239872172837314) * fr +
           0.1636161226585754240958355063) /
          ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr +
             168.5254414101568283957668343) * fr + 33.9905941350215598754191872) * fr + 1.0));
    m.f = value;
    v.f = (float)fr;
    v.u = (v.u + ((uint32_t)ex << 23) + s) & ((m.u * 2u) != 0 ? 0xffffffffu : 0u);
    return v.f;
}

static void lab_init(void) {
    if (g_lab_ready) return;
    static const double sRGB2XYZ_D65[9] = { 0.412453, 0.357580, 0.180423, 0.212671, 0.715160,
                                            0.072169, 0.019334, 0.119193, 0.950227 };
    static const double XYZ2sRGB_D65[9] = { 3.240479, -1.53715, -0.498535, -0.969256, 1.875991,
                                            0.041556, 0.055648, -0.204043, 1.057311 };
    static const double D65[3] = { 0.950456, 1.0, 1.088754 };
    float f[GAMMA_TAB_SIZE + 1], g[GAMMA_TAB_SIZE + 1];
    /* [cv] initLabTabs: g[i] = applyGamma(x), ig[i] = applyInvGamma(x), x = softfloat(i) / 1024; both promote x to
     * softdouble, use the BINARY32 constants 809/20000, 7827/2500000, 323/25, 12/5, 11/200 promoted likewise, evaluate pow
     * in binary64 and round once (round 4: rounds 1-3 used the binary64 constants 0.04045, 12.92, 2.4, 0.055, which moves
     * some knots by one binary32 step) */
    {
        const double thr = (double)(809.f / 20000.f), ithr = (double)(7827.f / 2500000.f), low = (double)(323.f / 25.f),
                     power = (double)(12.f / 5.f), shift = (double)(11.f / 200.f);
        for (int i = 0; i <= GAMMA_TAB_SIZE; ++i) {
            const double x = (double)((float)i * (1.0f / GAMMA_TAB_SIZE));
            f[i] = (float)(x <= thr ? x / low : pow((x + shift) / (1.0 + shift), power));
            g[i] = (float)(x <= ithr ? x * low : pow(x, 1.0 / power) * (1.0 + shift) - shift);
        }
    }
    spline_build(f, GAMMA_TAB_SIZE, g_gamma_tab);
    spline_build(g, GAMMA_TAB_SIZE, g_invgamma_tab);
    /* forward: coeffs[j + (blueIdx^2)] = scale*M[j], [j+1] = scale*M[j+1], [j+blueIdx] =
     * scale*M[j+2] with blueIdx = 0 (BGR input): index 0 multiplies B.                    */
    for (int i = 0; i < 3; ++i) {
        float sc = (float)(1.0 / D65[i]);
        if (i == 1) sc = 1.f;
        g_fwd[i * 3 + 2] = sc * (float)sRGB2XYZ_D65[i * 3 + 0];
        g_fwd[i * 3 + 1] = sc * (float)sRGB2XYZ_D65[i * 3 + 1];
        g_fwd[i * 3 + 0] = sc * (float)sRGB2XYZ_D65[i * 3 + 2];
    }
    /* inverse: coeffs[i + (blueIdx^2)*3] = M[i]*wp[i], [i+3] = M[i+3]*wp[i],
     * [i + blueIdx*3] = M[i+6]*wp[i]; row 0 produces dst[0] = B.                          */
    for (int i = 0; i < 3; ++i) {
        float wp = (float)D65[i];
        g_inv[i + 6] = (float)XYZ2sRGB_D65[i] * wp;
        g_inv[i + 3] = (float)XYZ2sRGB_D65[i + 3] * wp;
        g_inv[i + 0] = (float)XYZ2sRGB_D65[i + 6] * wp;
    }
    /* the hot path only feeds u8/255 into the forward gamma: tabulate it */
    const float a255 = (float)(1.0 / 255.0f);
    for (int i = 0; i < 256; ++i) {
        float v = (float)i * a255;
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        g_gamma_u8[i] = spline_interp(v * (float)GAMMA_TAB_SIZE, g_gamma_tab, GAMMA_TAB_SIZE);
    }
    g_lab_ready = 1;
}
const float* lvmo_gamma_tab(int inverse) { lab_init(); return inverse ? g_invgamma_tab : g_gamma_tab; }

static inline float clip01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }

/* [cv] RGB2Lab_f::operator() scalar path, srgb = true, blueIdx = 0 */
static inline void bgr2lab_px(float s0, float s1, float s2, float* o) {
    const float _a = 16.0f / 116.0f;
    float B = spline_interp(clip01(s0) * (float)GAMMA_TAB_SIZE, g_gamma_tab, GAMMA_TAB_SIZE);
    float G = spline_interp(clip01(s1) * (float)GAMMA_TAB_SIZE, g_gamma_tab, GAMMA_TAB_SIZE);
    float R = spline_interp(clip01(s2) * (float)GAMMA_TAB_SIZE, g_gamma_tab, GAMMA_TAB_SIZE);
    float X = B * g_fwd[0] + G * g_fwd[1] + R * g_fwd[2];
    float Y = B * g_fwd[3] + G * g_fwd[4] + R * g_fwd[5];
    float Z = B * g_fwd[6] + G * g_fwd[7] + R * g_fwd[8];
    float FX = X > 0.008856f ? lvmo_cube_root(X) : (7.787f * X + _a);
    float FY = Y > 0.008856f ? lvmo_cube_root(Y) : (7.787f * Y + _a);
    float FZ = Z > 0.008856f ? lvmo_cube_root(Z) : (7.787f * Z + _a);
    o[0] = Y > 0.008856f ? (116.f * FY - 16.f) : (903.3f * Y);
    o[1] = 500.f * (FX - FY);
    o[2] = 200.f * (FY - FZ);
}
/* ---- OpenCV 4's DEFAULT forward float path: the trilinear-interpolated 33^3 int16 LUT ---------------------------
 * [cv] color_lab.cpp: RGB2Labfloat with useInterpolation (= sRGB + default coefficients + default white point, the
 * case cv::cvtColor(COLOR_BGR2Lab) on CV_32F is -- MagnifyCore.hpp:90,219), initLabTabs (LAB_LUT_DIM = 33,
 * lab_base_shift = 14, trilinear_shift = 4), trilinearInterpolate.  This IS what the reference computes, so it is the
 * oracle's default since round 3; lvmo_set_lab_lut(0) selects the analytic form (OpenCV with interpolation disabled).
 * UNPINNED like the rest of the OpenCV boundary (restated from the published source, no OpenCV here to check against).
 * Table build: OpenCV uses its softfloat type, i.e. IEEE binary32 operations rounded one by one; restated on native floats
 * (-ffp-contract=off).  softfloat's pow(x, y) = exp(y * log(x)) with log, product and exp each rounded to binary32 (its
 * log / exp work in binary64 inside and round once: the C library's double log / exp stand in for them), its cbrt is the
 * cv::cubeRoot polynomial, mulAdd is a fused multiply-add.  A real build's table can be installed with
 * lvmo_lab_lut_override (oracle/ref_driver.cpp recovers it from cvtColor on the 33^3 node colours).                  */
enum { LAB_LUT_DIM = 33, LAB_BASE_SHIFT = 14, LAB_BASE = 1 << LAB_BASE_SHIFT, LAB_LUT_SHIFT = 5, TRI_SHIFT = 4 };
enum { LAB_LUT_ENTRIES = 3 * LAB_LUT_DIM * LAB_LUT_DIM * LAB_LUT_DIM };
static int16_t* g_lab_lut = NULL;               /* RGB2Labprev order: [3 (p + 33 q + 1089 r) + ch], p / q / r = R / G / B grid index
                                                   (OpenCV then replicates the 8 corners of every cell: same values) */
static int g_lab_use_lut = 1;
void lvmo_set_lab_lut(int on) { g_lab_use_lut = on != 0; }
static void lab_lut_drop(void) { if (g_lab_lut) { free(g_lab_lut); g_lab_lut = NULL; } }
/* [cv] applyGamma(softfloat x): "softdouble xd = x; return xd <= gammaThreshold ? xd / gammaLowScale :
 * pow((xd + gammaXshift) / (softdouble::one() + gammaXshift), softdouble(gammaPower))" -- the argument is PROMOTED, the
 * constants are binary32 values (softfloat(809) / softfloat(20000) ...) promoted likewise, pow runs in binary64 and the
 * result is rounded ONCE to binary32 (round 4; rounds 1-3 restated it as three binary32 operations, which moves 24 of the
 * 33 gamma nodes by up to 3e-7 relative and ~20 of the 107 811 table entries by one unit: LVMO_VAR_GAMMA_F32 keeps that
 * form as a variant).  softdouble's pow is exp(y log x) evaluated with ~1e-15 relative error; the C library's pow stands
 * in for it (a different binary32 result needs the binary64 value within ~1e-15 of a rounding boundary: p ~ 1e-8 per node). */
static float lut_apply_gamma(float x) {
    const float thr = 809.f / 20000.f, low = 323.f / 25.f, shift = 11.f / 200.f, power = 12.f / 5.f;
    if (g_var & LVMO_VAR_GAMMA_F32) {
        if (x <= thr) return x / low;
        const float base = (x + shift) / (1.f + shift);
        const float lg = (float)log((double)base);
        const float pr = power * lg;
        return (float)exp((double)pr);
    }
    const double xd = (double)x;
    float g = (float)(xd <= (double)thr ? xd / (double)low : pow((xd + (double)shift) / (1.0 + (double)shift), (double)power));
    /* the residual of the stand-in, as switches: every interior gamma node one binary32 step up / down */
    if (x > 0.f && x < 1.f) {
        if (g_var & LVMO_VAR_LUT_NUDGE_UP) g = nextafterf(g, 2.f);
        if (g_var & LVMO_VAR_LUT_NUDGE_DOWN) g = nextafterf(g, -1.f);
    }
    return g;
}
static void lab_lut_init(void) {
    if (g_lab_lut) return;
    static const double M[9] = { 0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227 };
    static const double D65[3] = { 0.950456, 1.0, 1.088754 };
    float C[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = (float)(M[i * 3 + j] * (i == 1 ? 1.0 : 1.0 / D65[i]));
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f, f9033 = 24389.f / 27.f;
    int16_t* t = (int16_t*)malloc(sizeof(int16_t) * LAB_LUT_ENTRIES);
    for (int r = 0; r < LAB_LUT_DIM; ++r)
        for (int q = 0; q < LAB_LUT_DIM; ++q)
            for (int p = 0; p < LAB_LUT_DIM; ++p) {
                const float R = lut_apply_gamma((float)p / 32.f), G = lut_apply_gamma((float)q / 32.f), B = lut_apply_gamma((float)r / 32.f);
                const float X = R * C[0] + G * C[1] + B * C[2];
                const float Y = R * C[3] + G * C[4] + B * C[5];
                const float Z = R * C[6] + G * C[7] + B * C[8];
                const float FX = X > lthresh ? lvmo_cube_root(X) : fmaf(X, lscale, lbias);
                const float FY = Y > lthresh ? lvmo_cube_root(Y) : fmaf(Y, lscale, lbias);
                const float FZ = Z > lthresh ? lvmo_cube_root(Z) : fmaf(Z, lscale, lbias);
                const float L = Y > lthresh ? (116.f * FY - 16.f) : (Y * f9033);
                const float a = 500.f * (FX - FY), b = 200.f * (FY - FZ);
                int16_t* e = t + 3 * (p + LAB_LUT_DIM * (q + LAB_LUT_DIM * r));
                e[0] = (int16_t)lrintf((float)LAB_BASE * L / 100.f);
                e[1] = (int16_t)lrintf((float)LAB_BASE * (a + 128.f) / 256.f);
                e[2] = (int16_t)lrintf((float)LAB_BASE * (b + 128.f) / 256.f);
            }
    g_lab_lut = t;
}
void lvmo_lab_lut_table(int16_t* out) { lab_lut_init(); memcpy(out, g_lab_lut, sizeof(int16_t) * LAB_LUT_ENTRIES); }
void lvmo_lab_lut_override(const int16_t* tab) {     /* NULL: back to the restated table */
    lab_lut_drop();
    if (!tab) return;
    g_lab_lut = (int16_t*)malloc(sizeof(int16_t) * LAB_LUT_ENTRIES);
    memcpy(g_lab_lut, tab, sizeof(int16_t) * LAB_LUT_ENTRIES);
}
/* [cv] RGB2Labfloat::operator(), useInterpolation branch (scalar form; the SIMD form computes the same integers) */
static inline void bgr2lab_lut_px(float s0, float s1, float s2, float* o) {
    const int c[3] = { (int)lrintf(clip01(s2) * (float)LAB_BASE), (int)lrintf(clip01(s1) * (float)LAB_BASE), (int)lrintf(clip01(s0) * (float)LAB_BASE) };   /* R, G, B */
    int t[3], w1[3];
    for (int k = 0; k < 3; ++k) {
        t[k] = c[k] >> (LAB_BASE_SHIFT - LAB_LUT_SHIFT);                              /* cell origin */
        w1[k] = (c[k] >> (LAB_BASE_SHIFT - 8 - 1)) & ((1 << TRI_SHIFT) - 1);          /* weight of the upper neighbour, 0..15 */
    }
    int acc[3] = { 0, 0, 0 };
    for (int corner = 0; corner < 8; ++corner) {
        int idx[3], wgt = 1;
        for (int k = 0; k < 3; ++k) {
            const int up = (corner >> k) & 1;
            idx[k] = t[k] + up; if (idx[k] > LAB_LUT_DIM - 1) idx[k] = LAB_LUT_DIM - 1;
            wgt *= up ? w1[k] : (1 << TRI_SHIFT) - w1[k];
        }
        const int16_t* e = g_lab_lut + 3 * (idx[0] + LAB_LUT_DIM * (idx[1] + LAB_LUT_DIM * idx[2]));
        acc[0] += e[0] * wgt; acc[1] += e[1] * wgt; acc[2] += e[2] * wgt;
    }
    for (int k = 0; k < 3; ++k) acc[k] = (acc[k] + (1 << (3 * TRI_SHIFT - 1))) >> (3 * TRI_SHIFT);      /* CV_DESCALE */
    o[0] = (float)acc[0] * (1.0f / LAB_BASE) * 100.0f;
    o[1] = (float)acc[1] * (1.0f / LAB_BASE) * 256.0f - 128.0f;
    o[2] = (float)acc[2] * (1.0f / LAB_BASE) * 256.0f - 128.0f;
}
void lvmo_bgr2lab(const float* src, int npix, float* dst) {
    lab_init();
    if (g_lab_use_lut) {
        lab_lut_init();
#pragma omp parallel for schedule(static)
        for (int i = 0; i < npix; ++i) {
            float o[3];
            bgr2lab_lut_px(src[i * 3], src[i * 3 + 1], src[i * 3 + 2], o);
            dst[i * 3] = o[0]; dst[i * 3 + 1] = o[1]; dst[i * 3 + 2] = o[2];
        }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < npix; ++i) {
        float o[3];
        bgr2lab_px(src[i * 3], src[i * 3 + 1], src[i * 3 + 2], o);
        dst[i * 3] = o[0]; dst[i * 3 + 1] = o[1]; dst[i * 3 + 2] = o[2];
    }
}
/* [cv] Lab2RGBfloat::process + Lab2RGB_f, srgb = true, blueIdx = 0 */
static inline void lab2bgr_px(float li, float ai, float bi, float* o) {
    const float lThresh = 0.008856f * 903.3f;
    const float fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    float y, fy;
    if (li <= lThresh) { y = li / 903.3f; fy = 7.787f * y + 16.0f / 116.0f; }
    else { fy = (li + 16.0f) / 116.0f; y = fy * fy * fy; }
    float fxz[2] = { ai / 500.0f + fy, fy - bi / 200.0f };
    for (int j = 0; j < 2; ++j) {
        if (fxz[j] <= fThresh) fxz[j] = (fxz[j] - 16.0f / 116.0f) / 7.787f;
        else fxz[j] = fxz[j] * fxz[j] * fxz[j];
    }
    const float x = fxz[0], z = fxz[1];
    float c0 = g_inv[0] * x + g_inv[1] * y + g_inv[2] * z;
    float c1 = g_inv[3] * x + g_inv[4] * y + g_inv[5] * z;
    float c2 = g_inv[6] * x + g_inv[7] * y + g_inv[8] * z;
    o[0] = spline_interp(clip01(c0) * (float)GAMMA_TAB_SIZE, g_invgamma_tab, GAMMA_TAB_SIZE);
    o[1] = spline_interp(clip01(c1) * (float)GAMMA_TAB_SIZE, g_invgamma_tab, GAMMA_TAB_SIZE);
    o[2] = spline_interp(clip01(c2) * (float)GAMMA_TAB_SIZE, g_invgamma_tab, GAMMA_TAB_SIZE);
}
void lvmo_lab2bgr(const float* src, int npix, float* dst) {
    lab_init();
#pragma omp parallel for schedule(static)
    for (int i = 0; i < npix; ++i) {
        float o[3];
        lab2bgr_px(src[i * 3], src[i * 3 + 1], src[i * 3 + 2], o);
        dst[i * 3] = o[0]; dst[i * 3 + 1] = o[1]; dst[i * 3 + 2] = o[2];
    }
}

/* The output quantiser of the Lab modes as a function of ONE clipped linear channel value c: what lab2bgr_px's last three lines and
 * float_to_u8(255, 1/255) do to it (MagnifyCore.hpp:152-153, :275-276).  lvmo_u8_of_linear_sweep walks over every float whose bit
 * pattern lies in [first, first + count) -- non-negative floats order like their patterns -- and reports: descents (places where the
 * byte DROPS as c grows: the step-table form of the library's output kernels is exact iff there are none), steps (places where it
 * rises), jumps (rises by more than one level), and thr[k] = the smallest pattern whose byte is >= k (k = 1..255; 0xffffffff if none).
 * tests/test_u8_steps.py runs it over [0, 1.0f]: 1 065 353 217 floats, a few seconds with OpenMP. */
uint8_t lvmo_u8_of_linear(float c) {
    lab_init();
    const float o = spline_interp(clip01(c) * (float)GAMMA_TAB_SIZE, g_invgamma_tab, GAMMA_TAB_SIZE);
    return sat_u8(o * 255.0f + (float)(1.0 / 255.0f));
}
void lvmo_u8_of_linear_sweep(uint32_t first, uint64_t count, uint64_t* descents, uint64_t* steps, uint64_t* jumps, uint32_t* thr /* [256] */) {
    lab_init();
    const float a255 = (float)(1.0 / 255.0f);
    const uint64_t CH = 1u << 20;
    const uint64_t nch = (count + CH - 1) / CH;
    uint64_t nd = 0, ns = 0, nj = 0;
    int* first_v = (int*)malloc(sizeof(int) * (nch ? nch : 1));
    int* last_v = (int*)malloc(sizeof(int) * (nch ? nch : 1));
    uint32_t* t_all = (uint32_t*)malloc(sizeof(uint32_t) * 256 * (nch ? nch : 1));
#pragma omp parallel for schedule(dynamic) reduction(+ : nd, ns, nj)
    for (int64_t k = 0; k < (int64_t)nch; ++k) {
        uint32_t* t = t_all + 256 * k;
        for (int i = 0; i < 256; ++i) t[i] = 0xffffffffu;
        const uint64_t b0 = first + (uint64_t)k * CH, b1 = (b0 + CH < first + count) ? b0 + CH : first + count;
        int prev = -1;
        for (uint64_t b = b0; b < b1; ++b) {
            const uint32_t bits = (uint32_t)b;
            float c; memcpy(&c, &bits, 4);
            const float o = spline_interp(clip01(c) * (float)GAMMA_TAB_SIZE, g_invgamma_tab, GAMMA_TAB_SIZE);
            const int g = sat_u8(o * 255.0f + a255);
            if (prev < 0) first_v[k] = g;
            else { if (g < prev) ++nd; if (g > prev) ++ns; if (g > prev + 1) ++nj; }
            if (t[g] == 0xffffffffu) t[g] = bits;
            prev = g;
        }
        last_v[k] = prev;
    }
    for (int i = 0; i < 256; ++i) thr[i] = 0xffffffffu;
    for (uint64_t k = 0; k < nch; ++k) {
        if (k > 0) { if (first_v[k] < last_v[k - 1]) ++nd; if (first_v[k] > last_v[k - 1]) ++ns; if (first_v[k] > last_v[k - 1] + 1) ++nj; }
        for (int i = 0; i < 256; ++i) if (t_all[256 * k + i] < thr[i]) thr[i] = t_all[256 * k + i];
    }
    /* thr[k] so far = first pattern whose byte EQUALS k; with no descents that is also the first pattern whose byte is >= k */
    free(first_v); free(last_v); free(t_all);
    *descents = nd; *steps = ns; *jumps = nj;
}

/* ------------------------------------------------------------------------------------- */
/* [cv] filter2D (direct path: correlation, centre anchor, REFLECT_101, non-zero taps in    */
/* row-major order), sepFilter2D / GaussianBlur, getGaussianKernel, resize.                 */
/* Accumulation is s = fma(k, v, s): OpenCV's FilterVec_32f / RowVec_32f / SymmColumnVec_32f */
/* inner loops are v_muladd == v_fma, a true FMA in the AVX2 dispatch every current x86 host */
/* takes (the SSE-only dispatch would round the product first; unpinned either way).         */
/* ------------------------------------------------------------------------------------- */
/* LVMO_VAR_FILTER_DFT -- the OTHER path of cv::filter2D.  filter.dispatch.cpp sends a kernel of kw * kh >= dft_filter_size elements
 * through crossCorr (templmatch.cpp) instead of the FilterEngine; dft_filter_size is 130 where checkHardwareSupport(CV_CPU_SSE3)
 * holds for 8U / 32F images and 50 everywhere else -- i.e. on every ARM build (the reference builds for macOS, CMakeLists.txt:10,
 * :142-163) the 9 x 9 kernels of RieszPyramid.cpp:227-232, :316-319 (81 taps) take it; the 1 x 5 / 5 x 1 Riesz kernels and
 * sepFilter2D do not.  crossCorr pads the image block with copyMakeBorder(borderType) -- the same REFLECT_101 samples -- and, for images
 * deeper than CV_8S, transforms in CV_64F ("maxDepth = depth > CV_8S ? CV_64F : ..."): block sizes 256 - 9 + 1 rounded up to
 * getOptimalDFTSize, spectra multiplied with mulSpectrums(conj), inverse transform scaled, ONE convertTo(CV_32F) at the end.  In
 * float64 the FFT's error against the exact sum is ~1e-14 absolute on planes of magnitude <= 100 -- far below the binary32 rounding
 * that follows -- so the path is restated as what it computes: the float64 sum of the 81 exact products, rounded once
 * (tests/test_oracle_variants.py checks this model against a numpy float64 FFT correlation with crossCorr's block sizes).           */
static void filter2d_f64_sum(const float* src, int w, int h, const float* k, int kw, int kh, float* dst) {
    const int ax = kw / 2, ay = kh / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            double s = 0.0;
            for (int i = 0; i < kh; ++i) {
                const float* row = src + (size_t)reflect101(y + i - ay, h) * w;
                for (int j = 0; j < kw; ++j) s += (double)k[i * kw + j] * (double)row[reflect101(x + j - ax, w)];
            }
            dst[(size_t)y * w + x] = (float)s;
        }
}
void lvmo_filter2d(const float* src, int w, int h, const float* k, int kw, int kh, float* dst) {
    if ((g_var & LVMO_VAR_FILTER_DFT) && kw * kh >= 50) { filter2d_f64_sum(src, w, h, k, kw, kh, dst); return; }
    const int ax = kw / 2, ay = kh / 2;
    int nt = 0;
    int* tx = (int*)malloc(sizeof(int) * kw * kh);
    int* ty = (int*)malloc(sizeof(int) * kw * kh);
    float* tk = (float*)malloc(sizeof(float) * kw * kh);
    for (int i = 0; i < kh; ++i)
        for (int j = 0; j < kw; ++j)
            if (k[i * kw + j] != 0.f) { tx[nt] = j - ax; ty[nt] = i - ay; tk[nt] = k[i * kw + j]; ++nt; }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            float s = 0.f;
            if (x >= ax && x < w - ax && y >= ay && y < h - ay) {
                for (int t = 0; t < nt; ++t) s = tap(tk[t], src[(size_t)(y + ty[t]) * w + x + tx[t]], s);
            } else {
                for (int t = 0; t < nt; ++t)
                    s = tap(tk[t], src[(size_t)reflect101(y + ty[t], h) * w + reflect101(x + tx[t], w)], s);
            }
            dst[(size_t)y * w + x] = s;
        }
    }
    free(tx); free(ty); free(tk);
}
/* [cv] getGaussianKernel(n, sigma, CV_32F): exp(-x^2/(2 sigma^2)) in double, normalised */
void lvmo_gauss_kernel(int n, double sigma, float* k) {
    double* t = (double*)malloc(sizeof(double) * n);
    const double scale2X = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        double x = i - (n - 1) * 0.5;
        t[i] = exp(scale2X * x * x);
        sum += t[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) k[i] = (float)(t[i] * sum);
    free(t);
}
/* [cv] sepFilter2D with a symmetric odd kernel on both axes: RowFilter (s = k0*S0; s = fma(kj,Sj,s),
 * left to right) then SymmColumnFilter (s = kc*S[c]; s = fma(kj, S[c+j] + S[c-j], s)).    */
void lvmo_sep_filter(const float* src, int w, int h, const float* k, int n, float* dst) {
    const int r = n / 2;
    float* tmp = (float*)malloc((size_t)w * h * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float* s = src + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            float acc = k[0] * s[reflect101(x - r, w)];
            for (int j = 1; j < n; ++j) acc = tap(k[j], s[reflect101(x - r + j, w)], acc);
            tmp[(size_t)y * w + x] = acc;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            float acc = k[r] * tmp[(size_t)y * w + x];
            for (int j = 1; j <= r; ++j)
                acc = tap(k[r + j], tmp[(size_t)reflect101(y + j, h) * w + x] +
                                     tmp[(size_t)reflect101(y - j, h) * w + x], acc);
            dst[(size_t)y * w + x] = acc;
        }
    }
    free(tmp);
}
/* [cv] resize INTER_LINEAR float: fx = (dx+0.5)*scale-0.5, floor, clamp; horizontal pass
 * then vertical: D = S0*b0 + S1*b1.  Equal sizes => plain copy.                            */
void lvmo_resize_linear(const float* src, int w, int h, int cn, float* dst, int dw, int dh) {
    if (w == dw && h == dh) { memcpy(dst, src, (size_t)w * h * cn * sizeof(float)); return; }
    const double scale_x = 1. / ((double)dw / w), scale_y = 1. / ((double)dh / h);
    int* xofs = (int*)malloc(sizeof(int) * dw); float* xa = (float*)malloc(sizeof(float) * dw);
    int* yofs = (int*)malloc(sizeof(int) * dh); float* ya = (float*)malloc(sizeof(float) * dh);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= w - 1) { fx = 0; sx = w - 1; }
        xofs[dx] = sx; xa[dx] = fx;
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= h - 1) { fy = 0; sy = h - 1; }
        yofs[dy] = sy; ya[dy] = fy;
    }
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        const int sy0 = yofs[dy], sy1 = sy0 + 1 < h ? sy0 + 1 : h - 1;
        const float b1 = ya[dy], b0 = 1.f - b1;
        const float* r0 = src + (size_t)sy0 * w * cn;
        const float* r1 = src + (size_t)sy1 * w * cn;
        float* d = dst + (size_t)dy * dw * cn;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx0 = xofs[dx], sx1 = sx0 + 1 < w ? sx0 + 1 : w - 1;
            const float a1 = xa[dx], a0 = 1.f - a1;
            for (int c = 0; c < cn; ++c) {
                const float h0 = r0[sx0 * cn + c] * a0 + r0[sx1 * cn + c] * a1;
                const float h1 = r1[sx0 * cn + c] * a0 + r1[sx1 * cn + c] * a1;
                d[dx * cn + c] = h0 * b0 + h1 * b1;
            }
        }
    }
    free(xofs); free(xa); free(yofs); free(ya);
}

/* ------------------------------------------------------------------------------------- */
/* [cv] dft/idft(DFT_ROWS|DFT_SCALE) on real rows, CCS-packed; mulSpectrums(DFT_ROWS).     */
/* The transform itself is evaluated as direct float64 sums and rounded to float32 at the  */
/* Mat boundary (OpenCV's float32 FFT differs from this by its own rounding noise).        */
/* ------------------------------------------------------------------------------------- */
static void twiddles(int n, double* cs, double* sn) {
    for (int k = 0; k < n; ++k) { cs[k] = cos(2.0 * PI_D * k / n); sn[k] = sin(2.0 * PI_D * k / n); }
}
/* packed index x -> (bin, is_imag).  [Re0, Re1, Im1, Re2, Im2, ..., Re(n/2) if n even] */
static inline void ccs_index(int x, int n, int* bin, int* im) {
    if (x == 0) { *bin = 0; *im = 0; return; }
    if (n % 2 == 0 && x == n - 1) { *bin = n / 2; *im = 0; return; }
    *bin = (x + 1) / 2; *im = (x % 2 == 0);
}
static float dft_elem(const float* row, int n, int x, const double* cs, const double* sn) {
    int bin, im; ccs_index(x, n, &bin, &im);
    double acc = 0;
    for (int t = 0; t < n; ++t) {
        int idx = (int)(((long long)bin * t) % n);
        acc += im ? -(double)row[t] * sn[idx] : (double)row[t] * cs[idx];
    }
    return (float)(acc / n);
}
void lvmo_dft_rows(const float* src, int rows, int n, float* dst) {
    double* cs = (double*)malloc(sizeof(double) * n); double* sn = (double*)malloc(sizeof(double) * n);
    twiddles(n, cs, sn);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int x = 0; x < n; ++x) dst[(size_t)r * n + x] = dft_elem(src + (size_t)r * n, n, x, cs, sn);
    free(cs); free(sn);
}
static float idft_elem(const float* X, int n, int t, const double* cs, const double* sn) {
    double acc = X[0];
    const int half = (n - 1) / 2; /* number of complex bins with both parts */
    for (int k = 1; k <= half; ++k) {
        int idx = (int)(((long long)k * t) % n);
        acc += 2.0 * ((double)X[2 * k - 1] * cs[idx] - (double)X[2 * k] * sn[idx]);
    }
    if (n % 2 == 0) acc += (t % 2 ? -1.0 : 1.0) * (double)X[n - 1];
    return (float)(acc / n);
}
void lvmo_idft_rows(const float* src, int rows, int n, float* dst) {
    double* cs = (double*)malloc(sizeof(double) * n); double* sn = (double*)malloc(sizeof(double) * n);
    twiddles(n, cs, sn);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r)
        for (int t = 0; t < n; ++t) dst[(size_t)r * n + t] = idft_elem(src + (size_t)r * n, n, t, cs, sn);
    free(cs); free(sn);
}
/* [cv] mulSpectrums(a, b, dst, DFT_ROWS), conjB = false, float */
void lvmo_mul_spectrums_rows(const float* a, const float* b, int rows, int n, float* dst) {
    for (int r = 0; r < rows; ++r) {
        const float* A = a + (size_t)r * n; const float* B = b + (size_t)r * n; float* D = dst + (size_t)r * n;
        D[0] = A[0] * B[0];
        int j1 = n - 1;
        if (n % 2 == 0) { D[n - 1] = A[n - 1] * B[n - 1]; j1 = n - 2; }
        for (int j = 1; j + 1 <= j1; j += 2) {
            float re = A[j] * B[j] - A[j + 1] * B[j + 1];
            float im = A[j + 1] * B[j] + A[j] * B[j + 1];
            D[j] = re; D[j + 1] = im;
        }
    }
}

/* LVMO_VAR_DFT_F32: the two transforms in BINARY32 the way a float32 FFT accumulates them -- power-of-two lengths as an
 * iterative radix-2 decimation-in-time FFT (twiddles = float(cos), float(sin) of a binary64 angle, like OpenCV's table; butterflies
 * in binary32; scale 1/n applied at the end in binary32), other lengths as direct sums with binary32 accumulators.  Not OpenCV's
 * exact factorisation (radix 4 / 2 / 3 / 5 mixes): a SWITCH that bounds what binary32 transform noise does to a colour frame. */
static void fft32(float* re, float* im, int n, int inverse) {
    for (int i = 1, j = 0; i < n; ++i) {                       /* bit reversal */
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { float t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        for (int k = 0; k < len / 2; ++k) {
            const double ang = 2.0 * PI_D * k / len * (inverse ? 1.0 : -1.0);
            const float wr = (float)cos(ang), wi = (float)sin(ang);
            for (int i = k; i < n; i += len) {
                const int j = i + len / 2;
                const float tr = re[j] * wr - im[j] * wi, ti = re[j] * wi + im[j] * wr;
                re[j] = re[i] - tr; im[j] = im[i] - ti;
                re[i] = re[i] + tr; im[i] = im[i] + ti;
            }
        }
    }
}
/* forward: real row -> CCS-packed, scaled 1/n; inverse: CCS-packed -> real row, scaled 1/n */
static void dft_row_f32(const float* row, int n, float* X) {
    float* re = (float*)malloc(sizeof(float) * 2 * n); float* im = re + n;
    if ((n & (n - 1)) == 0) {
        for (int t = 0; t < n; ++t) { re[t] = row[t]; im[t] = 0.f; }
        fft32(re, im, n, 0);
    } else {
        for (int k = 0; k < n; ++k) {
            float ar = 0.f, ai = 0.f;
            for (int t = 0; t < n; ++t) {
                const double ang = -2.0 * PI_D * (double)(((long long)k * t) % n) / n;
                ar += row[t] * (float)cos(ang); ai += row[t] * (float)sin(ang);
            }
            re[k] = ar; im[k] = ai;
        }
    }
    const float sc = 1.f / (float)n;
    for (int x = 0; x < n; ++x) { int bin, isim; ccs_index(x, n, &bin, &isim); X[x] = (isim ? im[bin] : re[bin]) * sc; }
    free(re);
}
static void idft_row_f32(const float* Y, int n, float* out) {
    float* re = (float*)malloc(sizeof(float) * 2 * n); float* im = re + n;
    const int half = (n - 1) / 2;
    re[0] = Y[0]; im[0] = 0.f;
    for (int k = 1; k <= half; ++k) { re[k] = Y[2 * k - 1]; im[k] = Y[2 * k]; re[n - k] = Y[2 * k - 1]; im[n - k] = -Y[2 * k]; }
    if (n % 2 == 0) { re[n / 2] = Y[n - 1]; im[n / 2] = 0.f; }
    const float sc = 1.f / (float)n;
    if ((n & (n - 1)) == 0) {
        fft32(re, im, n, 1);
        for (int t = 0; t < n; ++t) out[t] = re[t] * sc;
    } else {
        for (int t = 0; t < n; ++t) {
            float a = 0.f;
            for (int k = 0; k < n; ++k) {
                const double ang = 2.0 * PI_D * (double)(((long long)k * t) % n) / n;
                a += re[k] * (float)cos(ang) - im[k] * (float)sin(ang);
            }
            out[t] = a * sc;
        }
    }
    free(re);
}

/* TemporalFilter.cpp:24-80: idealFilter + createIdealBandpassFilter.
 * win: rows x cols x cn interleaved (rows = pixels, cols = frames).  dst same shape.
 * full != 0: literal dft -> mask -> mulSpectrums -> idft over every packed element.
 * full == 0: skips packed elements whose mask pair is all-zero (they contribute exactly
 * +/-0 to every later sum) -- bit-identical, much faster; tests cross-check both.        */
void lvmo_ideal_filter(const float* win, int rows, int cols, int cn, double lo, double hi,
                       double fps, float* dst, int full) {
    if (lo == 0.00) lo += 0.01;                                  /* :26-27 */
    const int n = cols;
    const float width = (float)n;                                /* :61 */
    const double fl = 2 * lo * width / fps, fh = 2 * hi * width / fps; /* :65-66 */
    float* mask = (float*)malloc(sizeof(float) * n);
    for (int x = 0; x < n; ++x) mask[x] = (x >= fl && x <= fh) ? 1.0f : 0.0f; /* :70-77 */
    double* cs = (double*)malloc(sizeof(double) * n); double* sn = (double*)malloc(sizeof(double) * n);
    twiddles(n, cs, sn);
    /* which packed elements are needed */
    char* need = (char*)calloc(n, 1);
    if (full) memset(need, 1, n);
    else {
        need[0] = mask[0] != 0.f;
        int j1 = n - 1;
        if (n % 2 == 0) { need[n - 1] = mask[n - 1] != 0.f; j1 = n - 2; }
        for (int j = 1; j + 1 <= j1; j += 2) {
            const char v = (char)((mask[j] != 0.f) | (mask[j + 1] != 0.f));
            need[j] = v;
            need[j + 1] = v;
        }
    }
#pragma omp parallel
    {
        float* row = (float*)malloc(sizeof(float) * n * 3);
        float* X = row + n; float* Y = row + 2 * n;
#pragma omp for schedule(static)
        for (int r = 0; r < rows; ++r) {
            for (int c = 0; c < cn; ++c) {
                for (int t = 0; t < n; ++t) row[t] = win[((size_t)r * n + t) * cn + c];
                if (g_var & LVMO_VAR_DFT_F32) {                                                 /* binary32 transforms (switch) */
                    float* o = (float*)malloc(sizeof(float) * n);
                    dft_row_f32(row, n, X);
                    lvmo_mul_spectrums_rows(X, mask, 1, n, Y);
                    idft_row_f32(Y, n, o);
                    for (int t = 0; t < n; ++t) dst[((size_t)r * n + t) * cn + c] = o[t];
                    free(o);
                    continue;
                }
                for (int x = 0; x < n; ++x) X[x] = need[x] ? dft_elem(row, n, x, cs, sn) : 0.f; /* :43 */
                lvmo_mul_spectrums_rows(X, mask, 1, n, Y);                                     /* :48 */
                for (int t = 0; t < n; ++t) dst[((size_t)r * n + t) * cn + c] = idft_elem(Y, n, t, cs, sn); /* :49 */
            }
        }
        free(row);
    }
    /* :55 normalize(dst, dst, 0, 1, NORM_MINMAX): global min/max over all channels;
     * [cv] scale = 1/(max-min) (0 if max-min <= DBL_EPSILON), shift = -min*scale,
     * convertTo with float(scale), float(shift).                                          */
    const size_t cnt = (size_t)rows * n * cn;
    double mn = dst[0], mx = dst[0];
    for (size_t i = 1; i < cnt; ++i) { if (dst[i] < mn) mn = dst[i]; if (dst[i] > mx) mx = dst[i]; }
    const double scale = (mx - mn > 2.220446049250313e-16) ? 1. / (mx - mn) : 0.;
    const double shift = 0. - mn * scale;
    const float fs = (float)scale, fsh = (float)shift;
    for (size_t i = 0; i < cnt; ++i) dst[i] = dst[i] * fs + fsh;
    free(mask); free(cs); free(sn); free(need);
}

/* ------------------------------------------------------------------------------------- */
/* Riesz constants: RieszPyramid.cpp:146-167                                               */
/* ------------------------------------------------------------------------------------- */
static const float k_lp9[81] = {
    -0.0001f, -0.0007f, -0.0023f, -0.0046f, -0.0057f, -0.0046f, -0.0023f, -0.0007f, -0.0001f,
    -0.0007f, -0.0030f, -0.0047f, -0.0025f, -0.0003f, -0.0025f, -0.0047f, -0.0030f, -0.0007f,
    -0.0023f, -0.0047f, 0.0054f, 0.0272f, 0.0387f, 0.0272f, 0.0054f, -0.0047f, -0.0023f,
    -0.0046f, -0.0025f, 0.0272f, 0.0706f, 0.0910f, 0.0706f, 0.0272f, -0.0025f, -0.0046f,
    -0.0057f, -0.0003f, 0.0387f, 0.0910f, 0.1138f, 0.0910f, 0.0387f, -0.0003f, -0.0057f,
    -0.0046f, -0.0025f, 0.0272f, 0.0706f, 0.0910f, 0.0706f, 0.0272f, -0.0025f, -0.0046f,
    -0.0023f, -0.0047f, 0.0054f, 0.0272f, 0.0387f, 0.0272f, 0.0054f, -0.0047f, -0.0023f,
    -0.0007f, -0.0030f, -0.0047f, -0.0025f, -0.0003f, -0.0025f, -0.0047f, -0.0030f, -0.0007f,
    -0.0001f, -0.0007f, -0.0023f, -0.0046f, -0.0057f, -0.0046f, -0.0023f, -0.0007f, -0.0001f };
static const float k_hp9[81] = {
    0.0000f, 0.0003f, 0.0011f, 0.0022f, 0.0027f, 0.0022f, 0.0011f, 0.0003f, 0.0000f,
    0.0003f, 0.0020f, 0.0059f, 0.0103f, 0.0123f, 0.0103f, 0.0059f, 0.0020f, 0.0003f,
    0.0011f, 0.0059f, 0.0151f, 0.0249f, 0.0292f, 0.0249f, 0.0151f, 0.0059f, 0.0011f,
    0.0022f, 0.0103f, 0.0249f, 0.0402f, 0.0469f, 0.0402f, 0.0249f, 0.0103f, 0.0022f,
    0.0027f, 0.0123f, 0.0292f, 0.0469f, -0.9455f, 0.0469f, 0.0292f, 0.0123f, 0.0027f,
    0.0022f, 0.0103f, 0.0249f, 0.0402f, 0.0469f, 0.0402f, 0.0249f, 0.0103f, 0.0022f,
    0.0011f, 0.0059f, 0.0151f, 0.0249f, 0.0292f, 0.0249f, 0.0151f, 0.0059f, 0.0011f,
    0.0003f, 0.0020f, 0.0059f, 0.0103f, 0.0123f, 0.0103f, 0.0059f, 0.0020f, 0.0003f,
    0.0000f, 0.0003f, 0.0011f, 0.0022f, 0.0027f, 0.0022f, 0.0011f, 0.0003f, 0.0000f };
void lvmo_riesz_kernels(float lp[81], float hp[81]) {
    memcpy(lp, k_lp9, sizeof(k_lp9)); memcpy(hp, k_hp9, sizeof(k_hp9));
}

/* ------------------------------------------------------------------------------------- */
/* context / state (MagnifyCore.hpp:24-80, MagnificationProcessor.hpp:13-23)               */
/* ------------------------------------------------------------------------------------- */
#define MAXLV 32
typedef struct RieszLevel {            /* RieszPyramid.hpp:19-52 */
    int w, h;
    float *lowpass, *r1, *r2;          /* itsLowpass, itsRiesz (real, imag) */
    float *amp, *ampBlur;              /* itsAmplitude, itsAmplitudeBlurred */
    float *pdc, *pds;                  /* itsPhaseDiff (cos, sin) */
    float *loc, *los, *hic, *his;      /* itsLowpassIIR, itsHighpassIIR */
} RieszLevel;
typedef struct RieszPyr { int numLevels; RieszLevel lv[MAXLV]; } RieszPyr;
typedef struct RieszFilt {             /* TemporalFilter.hpp:38-67 */
    double freq, fps, a[3], b[3];
    int nlv;
    float *r0c[MAXLV], *r0s[MAXLV], *r1c[MAXLV], *r1s[MAXLV], *phc[MAXLV], *phs[MAXLV];
    int n[MAXLV];
} RieszFilt;

struct lvmo_ctx {
    /* StructuralTracker (MagnifyCore.hpp:45-80) */
    int t_mode, t_levels, t_channels, t_w, t_h; uint64_t t_pre;
    /* MotionState */
    int m_n; Img m_hi[MAXLV], m_lo[MAXLV];
    /* ColorState: window rows x cols x cn (interleaved like the reference Mat) */
    float* c_win; int c_rows, c_cols, c_cn;
    /* RieszState */
    int r_init; RieszPyr r_cur, r_old; RieszFilt r_lo, r_hi;
    /* last float frame */
    Img last; double last_min, last_max;
};

static void motion_reset(lvmo_ctx* c) {
    for (int i = 0; i < c->m_n; ++i) { img_free(&c->m_hi[i]); img_free(&c->m_lo[i]); }
    c->m_n = 0;
}
static void color_reset(lvmo_ctx* c) { free(c->c_win); c->c_win = NULL; c->c_rows = c->c_cols = c->c_cn = 0; }
static void rlevel_free(RieszLevel* l) {
    free(l->lowpass); free(l->r1); free(l->r2); free(l->amp); free(l->ampBlur); free(l->pdc); free(l->pds);
    free(l->loc); free(l->los); free(l->hic); free(l->his);
    memset(l, 0, sizeof(*l));
}
static void rfilt_free(RieszFilt* f) {
    for (int i = 0; i < f->nlv; ++i) { free(f->r0c[i]); free(f->r0s[i]); free(f->r1c[i]); free(f->r1s[i]); free(f->phc[i]); free(f->phs[i]); }
    memset(f, 0, sizeof(*f));
}
static void riesz_reset(lvmo_ctx* c) {
    if (!c->r_init) return;
    for (int i = 0; i < c->r_cur.numLevels; ++i) { rlevel_free(&c->r_cur.lv[i]); rlevel_free(&c->r_old.lv[i]); }
    rfilt_free(&c->r_lo); rfilt_free(&c->r_hi);
    c->r_cur.numLevels = c->r_old.numLevels = 0;
    c->r_init = 0;
}
static void tracker_disable(lvmo_ctx* c) { c->t_mode = LVMO_MODE_NONE; c->t_levels = -1; c->t_channels = -1; c->t_w = c->t_h = 0; }

lvmo_ctx* lvmo_create(void) {
    lvmo_ctx* c = (lvmo_ctx*)calloc(1, sizeof(lvmo_ctx));
    tracker_disable(c); c->t_pre = 0;
    lab_init();
    return c;
}
void lvmo_reset(lvmo_ctx* c) { /* MagnificationProcessor.cpp:10-15 */
    motion_reset(c); color_reset(c); riesz_reset(c);
    tracker_disable(c); c->t_pre = 0;
}
void lvmo_destroy(lvmo_ctx* c) { if (!c) return; lvmo_reset(c); img_free(&c->last); free(c); }
const float* lvmo_last_float(lvmo_ctx* c, int* w, int* h, int* ch) { *w = c->last.w; *h = c->last.h; *ch = c->last.c; return c->last.d; }
void lvmo_last_minmax(lvmo_ctx* c, double* mn, double* mx) { *mn = c->last_min; *mx = c->last_max; }

static void set_last(lvmo_ctx* c, const Img* m) { img_free(&c->last); c->last = img_clone(m); }

/* convertTo(CV_32F, alpha) from 8U: float(src)*float(alpha) [cv] */
static Img u8_to_float(const uint8_t* in, int w, int h, int cn, ptrdiff_t stride, float alpha) {
    Img m = img_alloc(w, h, cn);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = in + (size_t)y * stride;
        float* d = m.d + (size_t)y * w * cn;
        for (int i = 0; i < w * cn; ++i) d[i] = (float)s[i] * alpha;
    }
    return m;
}
/* convertTo(CV_8U, alpha, beta): saturate(round_half_even(src*float(alpha) + float(beta))) [cv] */
static void float_to_u8(const Img* m, float alpha, float beta, uint8_t* out, ptrdiff_t stride) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < m->h; ++y) {
        const float* s = m->d + (size_t)y * m->w * m->c;
        uint8_t* d = out + (size_t)y * stride;
        for (int i = 0; i < m->w * m->c; ++i) d[i] = sat_u8(s[i] * alpha + beta);
    }
}

/* ------------------------------------------------------------------------------------- */
/* Laplace motion: MagnifyCore.hpp:83-160, SpatialFilter.cpp:25-38,52-61, TemporalFilter.cpp:9-22 */
/* ------------------------------------------------------------------------------------- */
static void build_laplace_pyr(const Img* img, int levels, Img* pyr /* levels+1 */) {
    Img cur = img_clone(img);
    for (int l = 0; l < levels; ++l) {
        Img down = img_alloc((cur.w + 1) / 2, (cur.h + 1) / 2, cur.c);
        lvmo_pyr_down(cur.d, cur.w, cur.h, cur.c, down.d);               /* :31 */
        Img up = img_alloc(cur.w, cur.h, cur.c);
        lvmo_pyr_up(down.d, down.w, down.h, down.c, up.d, cur.w, cur.h); /* :32 */
        const size_t n = img_count(&cur);
        for (size_t i = 0; i < n; ++i) up.d[i] = cur.d[i] - up.d[i];     /* :33 */
        pyr[l] = up;
        img_free(&cur);
        cur = down;
    }
    pyr[levels] = cur;                                                   /* :37 */
}

static int magnify_motion(lvmo_ctx* c, const uint8_t* in, int w, int h, int channels, ptrdiff_t stride,
                          const lvmo_params* p, int levels, uint8_t* out, ptrdiff_t ostride) {
    const int color = channels >= 3;
    Img input = u8_to_float(in, w, h, channels, stride, (float)(1.0 / 255.0f)); /* :89,:92 */
    if (color) lvmo_bgr2lab(input.d, w * h, input.d);                    /* :90 */
    Img pyr[MAXLV];
    build_laplace_pyr(&input, levels, pyr);                              /* :96 */
    Img output;
    if (c->m_n == 0) {                                                   /* :98-103 */
        for (int l = 0; l <= levels; ++l) { c->m_hi[l] = img_clone(&pyr[l]); c->m_lo[l] = img_clone(&pyr[l]); }
        c->m_n = levels + 1;
        output = img_clone(&input);
    } else {
        double cLo = p->coLow, cHi = p->coHigh;
        if (cLo == 0) cLo = 0.01;                                        /* TemporalFilter.cpp:11-12 */
        const float aHi = (float)(1 - cHi), bHi = (float)cHi, aLo = (float)(1 - cLo), bLo = (float)cLo;
        Img motion[MAXLV];
        for (int l = 0; l < levels; ++l) {                               /* :106-109 */
            const size_t n = img_count(&pyr[l]);
            motion[l] = img_alloc(pyr[l].w, pyr[l].h, pyr[l].c);
            float *hi = c->m_hi[l].d, *lo = c->m_lo[l].d; const float* s = pyr[l].d; float* m = motion[l].d;
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; ++i) {
                /* [cv] MatExpr a*alpha + b*beta = addWeighted; scalar loop a*alpha + b*beta (two products, one sum);
                 * LVMO_VAR_ADDW_FUSED: the SIMD loop's v_fma(a, alpha, v_fma(b, beta, gamma = 0)) */
                const int fw = (g_var & LVMO_VAR_ADDW_FUSED) != 0;
                const float t1 = fw ? fmaf(hi[i], aHi, s[i] * bHi) : hi[i] * aHi + s[i] * bHi;   /* TemporalFilter.cpp:16 */
                const float t2 = fw ? fmaf(lo[i], aLo, s[i] * bLo) : lo[i] * aLo + s[i] * bLo;   /* :17 */
                hi[i] = t1; lo[i] = t2;
                m[i] = t1 - t2;                                          /* :21 */
            }
        }
        motion[levels] = img_clone(&pyr[levels]);                        /* :112 */
        float gains[MAXLV];
        lvmo_laplace_gains(w, h, levels, p->amplification, p->coWavelength, gains); /* :114-134 */
        for (int l = levels; l >= 0; --l) {
            const size_t n = img_count(&motion[l]);
            for (size_t i = 0; i < n; ++i) motion[l].d[i] = motion[l].d[i] * gains[l];
        }
        /* buildImgFromLaplacePyr: SpatialFilter.cpp:52-61 */
        Img cur = img_clone(&motion[levels]);
        for (int l = levels - 1; l >= 0; --l) {
            Img up = img_alloc(motion[l].w, motion[l].h, motion[l].c);
            lvmo_pyr_up(cur.d, cur.w, cur.h, cur.c, up.d, up.w, up.h);
            const size_t n = img_count(&up);
            for (size_t i = 0; i < n; ++i) up.d[i] = up.d[i] + motion[l].d[i];
            img_free(&cur); cur = up;
        }
        if (cur.c > 2) {                                                 /* :140-146 */
            const float ca = (float)p->chromAttenuation;
            const size_t n = (size_t)cur.w * cur.h;
            for (size_t i = 0; i < n; ++i) { cur.d[i * 3 + 1] = cur.d[i * 3 + 1] * ca; cur.d[i * 3 + 2] = cur.d[i * 3 + 2] * ca; }
        }
        output = img_alloc(w, h, channels);
        { const size_t n = img_count(&output); for (size_t i = 0; i < n; ++i) output.d[i] = input.d[i] + cur.d[i]; } /* :148 */
        img_free(&cur);
        for (int l = 0; l <= levels; ++l) img_free(&motion[l]);
    }
    if (color) lvmo_lab2bgr(output.d, w * h, output.d);                  /* :152 */
    set_last(c, &output);
    float_to_u8(&output, 255.0f, (float)(1.0 / 255.0), out, ostride);    /* :153,:156 */
    img_free(&output); img_free(&input);
    for (int l = 0; l <= levels; ++l) img_free(&pyr[l]);
    return 1;
}

/* ------------------------------------------------------------------------------------- */
/* Colour: MagnifyCore.hpp:163-206, SpatialFilter.cpp:13-23,40-50,63-89                    */
/* ------------------------------------------------------------------------------------- */
static int magnify_color(lvmo_ctx* c, const uint8_t* in, int w, int h, int channels, ptrdiff_t stride,
                         const lvmo_params* p, int levels, uint8_t* out, ptrdiff_t ostride) {
    Img input = u8_to_float(in, w, h, channels, stride, 1.0f);          /* :169 (unscaled) */
    /* buildGaussPyrFromImg: only the smallest level is used (:175) */
    Img cur = img_clone(&input);
    for (int l = 0; l < levels; ++l) {
        Img down = img_alloc((cur.w + 1) / 2, (cur.h + 1) / 2, cur.c);
        lvmo_pyr_down(cur.d, cur.w, cur.h, cur.c, down.d);
        img_free(&cur); cur = down;
    }
    const int sw = cur.w, sh = cur.h, rows = sw * sh, cn = channels;
    /* img2tempMat (SpatialFilter.cpp:63-84) */
    const int maxImages = lvmo_optimal_buffer_size((int)p->framerate);  /* :176 */
    if (c->c_cols == 0) {
        c->c_win = (float*)malloc((size_t)rows * cn * sizeof(float));
        memcpy(c->c_win, cur.d, (size_t)rows * cn * sizeof(float));
        c->c_rows = rows; c->c_cols = 1; c->c_cn = cn;
    } else {
        const int oc = c->c_cols, nc = oc + 1;
        float* nw = (float*)malloc((size_t)rows * nc * cn * sizeof(float));
        for (int r = 0; r < rows; ++r) {
            memcpy(nw + (size_t)r * nc * cn, c->c_win + (size_t)r * oc * cn, (size_t)oc * cn * sizeof(float));
            memcpy(nw + ((size_t)r * nc + oc) * cn, cur.d + (size_t)r * cn, (size_t)cn * sizeof(float));
        }
        free(c->c_win); c->c_win = nw; c->c_cols = nc;
    }
    if (c->c_cols > maxImages && maxImages > 0) {                        /* drop the oldest column */
        const int oc = c->c_cols, nc = oc - 1;
        float* nw = (float*)malloc((size_t)rows * nc * cn * sizeof(float));
        for (int r = 0; r < rows; ++r)
            memcpy(nw + (size_t)r * nc * cn, c->c_win + ((size_t)r * oc + 1) * cn, (size_t)nc * cn * sizeof(float));
        free(c->c_win); c->c_win = nw; c->c_cols = nc;
    }
    img_free(&cur);
    if (c->c_cols < 2) { img_free(&input); return 0; }                  /* :180 */
    const int T = c->c_cols;
    float* filt = (float*)malloc((size_t)rows * T * cn * sizeof(float));
    lvmo_ideal_filter(c->c_win, rows, T, cn, p->coLow, p->coHigh, p->framerate, filt, 0); /* :183 */
    /* :185 filteredMat * amplification (MatExpr scale: float(alpha)); :190-192 column 1 */
    const float amp = (float)p->amplification;
    const int pos = 1 < T - 1 ? 1 : T - 1;
    Img small = img_alloc(sw, sh, cn);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < cn; ++k) small.d[(size_t)r * cn + k] = filt[((size_t)r * T + pos) * cn + k] * amp;
    free(filt);
    /* buildImgFromGaussPyr (SpatialFilter.cpp:40-50): levels x pyrUp (2x), then bilinear resize */
    Img up = small;
    for (int l = 0; l < levels; ++l) {
        Img nx = img_alloc(up.w * 2, up.h * 2, cn);
        lvmo_pyr_up(up.d, up.w, up.h, cn, nx.d, nx.w, nx.h);
        img_free(&up); up = nx;
    }
    Img colorImg = img_alloc(w, h, cn);
    lvmo_resize_linear(up.d, up.w, up.h, cn, colorImg.d, w, h);
    img_free(&up);
    const size_t n = img_count(&input);
    for (size_t i = 0; i < n; ++i) colorImg.d[i] = input.d[i] + colorImg.d[i];       /* :197 */
    double mn = colorImg.d[0], mx = colorImg.d[0];                                    /* :200-201 */
    for (size_t i = 1; i < n; ++i) { if (colorImg.d[i] < mn) mn = colorImg.d[i]; if (colorImg.d[i] > mx) mx = colorImg.d[i]; }
    c->last_min = mn; c->last_max = mx;
    set_last(c, &colorImg);
    float_to_u8(&colorImg, (float)(255.0 / (mx - mn)), (float)(-mn * 255.0 / (mx - mn)), out, ostride); /* :202 */
    img_free(&colorImg); img_free(&input);
    return 1;
}

/* ------------------------------------------------------------------------------------- */
/* Riesz: MagnifyCore.hpp:209-279, RieszPyramid.cpp, TemporalFilter.cpp:299-362            */
/* ------------------------------------------------------------------------------------- */
static float* falloc(size_t n) { return (float*)calloc(n ? n : 1, sizeof(float)); }

/* RieszPyramidLevel::build (RieszPyramid.cpp:66-78): takes ownership of `octave` */
static void rlevel_build(RieszLevel* L, float* octave, int w, int h) {
    static const float realK[5] = { -0.2f, -0.48f, 0.f, 0.48f, 0.2f };
    L->w = w; L->h = h;
    free(L->lowpass); L->lowpass = octave;
    if (!L->r1) L->r1 = falloc((size_t)w * h);
    if (!L->r2) L->r2 = falloc((size_t)w * h);
    lvmo_filter2d(octave, w, h, realK, 5, 1, L->r1);
    lvmo_filter2d(octave, w, h, realK, 1, 5, L->r2);
}
/* subsample (RieszPyramid.cpp:254-278) */
static float* subsample(const float* img, int w, int h, int* ow, int* oh) {
    const int sw = w / 2 + (w % 2), sh = h / 2 + (h % 2);
    float* t = falloc((size_t)sw * sh);
    for (int y = 0; y < h; y += 2) for (int x = 0; x < w; x += 2) t[x / 2 + (y / 2) * sw] = img[x + (size_t)y * w];
    *ow = sw; *oh = sh; return t;
}
/* RieszPyramid::buildPyramid (RieszPyramid.cpp:215-238) */
static void rpyr_build(RieszPyr* P, const float* frame, int w, int h) {
    const int max = P->numLevels - 1;
    if (max == -1) return;
    float lp2[81];
    for (int i = 0; i < 81; ++i) lp2[i] = 2.0f * k_lp9[i];              /* 2.0 * lowPassFilter */
    float* octave = falloc((size_t)w * h);
    memcpy(octave, frame, (size_t)w * h * sizeof(float));
    int ow = w, oh = h;
    for (int i = 0; i < max; ++i) {
        float* hp = falloc((size_t)ow * oh);
        float* lp = falloc((size_t)ow * oh);
        lvmo_filter2d(octave, ow, oh, k_hp9, 9, 9, hp);                 /* :227 */
        lvmo_filter2d(octave, ow, oh, lp2, 9, 9, lp);                   /* :232 */
        rlevel_build(&P->lv[i], hp, ow, oh);                            /* :229 */
        int nw, nh;
        float* sub = subsample(lp, ow, oh, &nw, &nh);                   /* :234 */
        free(lp); free(octave);
        octave = sub; ow = nw; oh = nh;
    }
    rlevel_build(&P->lv[max], octave, ow, oh);                          /* :237 */
}
/* RieszPyramid::init (RieszPyramid.cpp:192-213): build, then zero everything else --
 * INCLUDING itsRiesz, which buildPyramid had just computed (reference behaviour).         */
static void rpyr_init(RieszPyr* P, const float* frame, int w, int h, int levels) {
    P->numLevels = levels;
    memset(P->lv, 0, sizeof(RieszLevel) * levels);
    rpyr_build(P, frame, w, h);
    for (int i = 0; i < levels; ++i) {
        RieszLevel* L = &P->lv[i];
        const size_t n = (size_t)L->w * L->h;
        memset(L->r1, 0, n * sizeof(float)); memset(L->r2, 0, n * sizeof(float));
        L->pdc = falloc(n); L->pds = falloc(n); L->loc = falloc(n); L->los = falloc(n);
        L->hic = falloc(n); L->his = falloc(n); L->amp = falloc(n); L->ampBlur = falloc(n);
    }
}
/* arcCos (RieszPyramid.cpp:8-23): out-of-range input returns -1.0 / +1.0 (not pi / 0) */
static inline float arc_cos(float x) { if (x < -1.0) return -1.0f; if (x > 1.0) return 1.0f; return acosf(x); }

/* RieszPyramidLevel::computePhaseDifferenceAndAmplitude (RieszPyramid.cpp:81-111) */
static void rlevel_phase(RieszLevel* L, const RieszLevel* P, const float* gk13) {
    const size_t n = (size_t)L->w * L->h;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const float p = L->lowpass[i], r1 = L->r1[i], r2 = L->r2[i];
        const float Pp = P->lowpass[i], R1 = P->r1[i], R2 = P->r2[i];
        const float q0 = (p * Pp + r1 * R1) + r2 * R2;                  /* :82-84 */
        const float np = p * (-1.f);
        const float q1 = R1 * np + r1 * Pp;                             /* :86 */
        const float q2 = R2 * np + r2 * Pp;
        const float xy = q1 * q1 + q2 * q2;                             /* :89 square() */
        const float ampq = sqrtf(q0 * q0 + xy);                         /* :91 */
        const float phi = arc_cos(q0 / ampq);                           /* :93-97 */
        const float sxy = sqrtf(xy);                                    /* :99-100 */
        float dc = (q1 / sxy) * phi, ds = (q2 / sxy) * phi;             /* :102-104 */
        if (dc != dc) dc = 0.f;                                         /* :105-106 patchNaNs */
        if (ds != ds) ds = 0.f;
        L->pdc[i] = dc; L->pds[i] = ds;
        L->amp[i] = sqrtf(ampq);                                        /* :108 */
    }
    lvmo_sep_filter(L->amp, L->w, L->h, gk13, 13, L->ampBlur);          /* :110 */
}
static void rfilt_init(RieszFilt* f, double frq, double fps, const RieszPyr* P) { /* TemporalFilter.cpp:299-317 */
    memset(f, 0, sizeof(*f));
    f->freq = frq; f->fps = fps; f->nlv = P->numLevels;
    for (int l = 0; l < f->nlv; ++l) {
        const size_t n = (size_t)P->lv[l].w * P->lv[l].h;
        f->n[l] = (int)n;
        f->r0c[l] = falloc(n); f->r0s[l] = falloc(n); f->r1c[l] = falloc(n); f->r1s[l] = falloc(n);
        f->phc[l] = falloc(n); f->phs[l] = falloc(n);
    }
}
static void rfilt_coeffs(RieszFilt* f) {                                /* :324-327 */
    const double Wn = f->fps == 0.0 ? 0.0 : f->freq / (f->fps / 2.0);
    lvmo_butterworth2(Wn, f->a, f->b);
}
static void rfilt_reset_mat(RieszFilt* f) {                             /* :353-362 */
    for (int l = 0; l < f->nlv; ++l) {
        const size_t b = (size_t)f->n[l] * sizeof(float);
        memset(f->r0c[l], 0, b); memset(f->r0s[l], 0, b); memset(f->r1c[l], 0, b); memset(f->r1s[l], 0, b);
        memset(f->phc[l], 0, b); memset(f->phs[l], 0, b);
    }
}
/* [cv] multiply(Mat32f, double scalar): evaluated in float64, rounded once to float32 */
static inline float mul_sd(float x, double s) { return (g_var & LVMO_VAR_MUL_F32) ? x * (float)s : (float)((double)x * s); }   /* LVMO_VAR_MUL_F32: a build that narrows the scalar first */
/* RieszTemporalFilter::IIRTemporalFilter (TemporalFilter.cpp:340-351), one component */
static void rfilt_iir_comp(float* ph, float* r0, float* r1, const float* d, float* res, size_t n,
                           const double* a, const double* b) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const float phase = ph[i] + d[i];                               /* :343 */
        const float y = mul_sd(phase, b[0]) + r0[i];                    /* :345 */
        const float nr0 = (mul_sd(phase, b[1]) + r1[i]) - mul_sd(y, a[1]); /* :347-348 */
        const float nr1 = mul_sd(phase, b[2]) - mul_sd(y, a[2]);        /* :350 */
        ph[i] = phase; r0[i] = nr0; r1[i] = nr1; res[i] = y;
    }
}
/* RieszPyramidLevel::normalize + amplify (RieszPyramid.cpp:114-144) */
static void rlevel_amplify(RieszLevel* L, double alpha, double threshold, const float* gk13) {
    const size_t n = (size_t)L->w * L->h;
    float* tc = falloc(n); float* ts = falloc(n); float* bc = falloc(n); float* bs = falloc(n);
    for (size_t i = 0; i < n; ++i) {
        tc[i] = (L->hic[i] - L->loc[i]) * L->amp[i];                    /* :118-120 */
        ts[i] = (L->his[i] - L->los[i]) * L->amp[i];
    }
    lvmo_sep_filter(tc, L->w, L->h, gk13, 13, bc);                      /* :121-124 */
    lvmo_sep_filter(ts, L->w, L->h, gk13, 13, bs);
    const float fa = (float)alpha, thr = (float)threshold;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const float c = bc[i] / L->ampBlur[i], s = bs[i] / L->ampBlur[i]; /* :125-126 */
        const float magV = sqrtf(c * c + s * s);                        /* :133-134 */
        float magV2 = magV * fa;                                        /* :135 */
        /* :136 cv::threshold(THRESH_TRUNC) on CV_32F is v_min(src, thresh) in OpenCV's vector loop (minps: the
         * SECOND operand when the first is NaN), so a NaN magnitude (0/0 in a flat black region) becomes thr;
         * only the <= 3-pixel scalar tail of a row would keep the NaN.  The vector semantics are restated. */
        magV2 = magV2 < thr ? magV2 : thr;
        const float cp = cosf(magV2), sp = sinf(magV2);                 /* :138 cosSin */
        float pair = (L->r1[i] * c + L->r2[i] * s) / magV;              /* :139-140 */
        if (pair != pair) pair = 0.f;                                   /* :141 */
        L->lowpass[i] = L->lowpass[i] * cp - pair * sp;                 /* :143 */
    }
    free(tc); free(ts); free(bc); free(bs);
}
/* RieszPyramid::collapsePyramid (RieszPyramid.cpp:304-325) incl. injectZerosEven (:280-302) */
static float* rpyr_collapse(const RieszPyr* P) {
    const int count = P->numLevels - 1;
    float lp2[81];
    for (int i = 0; i < 81; ++i) lp2[i] = 2.0f * k_lp9[i];
    int rw = P->lv[count].w, rh = P->lv[count].h;
    float* result = falloc((size_t)rw * rh);
    memcpy(result, P->lv[count].lowpass, (size_t)rw * rh * sizeof(float));
    for (int i = count - 1; i >= 0; --i) {
        const RieszLevel* L = &P->lv[i];
        const size_t n = (size_t)L->w * L->h;
        float* up0 = falloc(n);
        /* resize INTER_NEAREST then keep only even (x,y): up0(2i,2j) = result(i,j) */
        for (int y = 0; y < L->h; y += 2)
            for (int x = 0; x < L->w; x += 2) {
                int sx = (int)floor(x * ((double)rw / L->w)); if (sx > rw - 1) sx = rw - 1;
                int sy = (int)floor(y * ((double)rh / L->h)); if (sy > rh - 1) sy = rh - 1;
                up0[x + (size_t)y * L->w] = result[sx + (size_t)sy * rw];
            }
        float* lp = falloc(n); float* hp = falloc(n);
        lvmo_filter2d(up0, L->w, L->h, lp2, 9, 9, lp);                  /* :316 */
        lvmo_filter2d(L->lowpass, L->w, L->h, k_hp9, 9, 9, hp);         /* :319 */
        for (size_t k = 0; k < n; ++k) lp[k] = lp[k] + hp[k];           /* :322 */
        free(up0); free(hp); free(result);
        result = lp; rw = L->w; rh = L->h;
    }
    return result;
}
/* RieszPyramidLevel::operator= (RieszPyramid.cpp:52-64): IIR outputs are not copied */
static void rlevel_copy(RieszLevel* d, const RieszLevel* s) {
    const size_t b = (size_t)s->w * s->h * sizeof(float);
    memcpy(d->lowpass, s->lowpass, b); memcpy(d->r1, s->r1, b); memcpy(d->r2, s->r2, b);
    memcpy(d->pdc, s->pdc, b); memcpy(d->pds, s->pds, b); memcpy(d->amp, s->amp, b); memcpy(d->ampBlur, s->ampBlur, b);
}

static int magnify_riesz(lvmo_ctx* c, const uint8_t* in, int w, int h, int channels, ptrdiff_t stride,
                         const lvmo_params* p, int levels, uint8_t* out, ptrdiff_t ostride) {
    if (channels < 3) return 0;                                         /* :212 */
    const double PI_PERCENT = PI_D / 100.0;
    Img buf = u8_to_float(in, w, h, 3, stride, (float)(1.0 / 255.0));   /* :218 */
    lvmo_bgr2lab(buf.d, w * h, buf.d);                                  /* :219 */
    const size_t n0 = (size_t)w * h;
    float* Lpl = falloc(n0);
    for (size_t i = 0; i < n0; ++i) Lpl[i] = buf.d[i * 3];              /* :220-222 */
    if (!c->r_init || isnan(c->r_lo.a[0]) || isnan(c->r_hi.a[0])) {     /* :226-240 */
        riesz_reset(c);
        rpyr_init(&c->r_cur, Lpl, w, h, levels);
        rpyr_init(&c->r_old, Lpl, w, h, levels);
        rfilt_init(&c->r_lo, p->coLow, p->framerate, &c->r_cur);
        rfilt_init(&c->r_hi, p->coHigh, p->framerate, &c->r_cur);
        rfilt_coeffs(&c->r_lo); rfilt_coeffs(&c->r_hi);
        c->r_init = 1;
        free(Lpl); img_free(&buf);
        return 0;
    }
    if (c->r_lo.freq != p->coLow) {                                     /* :243-248 */
        c->r_lo.freq = p->coLow; rfilt_coeffs(&c->r_lo);
        rfilt_reset_mat(&c->r_lo); rfilt_reset_mat(&c->r_hi);
        rpyr_build(&c->r_old, Lpl, w, h);
    }
    if (c->r_hi.freq != p->coHigh) {                                    /* :249-254 */
        c->r_hi.freq = p->coHigh; rfilt_coeffs(&c->r_hi);
        rfilt_reset_mat(&c->r_hi); rfilt_reset_mat(&c->r_lo);
        rpyr_build(&c->r_old, Lpl, w, h);
    }
    float gk13[13];
    lvmo_gauss_kernel(13, 3.0, gk13);
    rpyr_build(&c->r_cur, Lpl, w, h);                                   /* :256 */
    for (int l = 0; l < levels - 1; ++l) rlevel_phase(&c->r_cur.lv[l], &c->r_old.lv[l], gk13); /* :257 */
    for (int l = 0; l < levels - 1; ++l) {                              /* :259-264 */
        RieszLevel* L = &c->r_cur.lv[l];
        const size_t n = (size_t)L->w * L->h;
        rfilt_iir_comp(c->r_lo.phc[l], c->r_lo.r0c[l], c->r_lo.r1c[l], L->pdc, L->loc, n, c->r_lo.a, c->r_lo.b);
        rfilt_iir_comp(c->r_lo.phs[l], c->r_lo.r0s[l], c->r_lo.r1s[l], L->pds, L->los, n, c->r_lo.a, c->r_lo.b);
        rfilt_iir_comp(c->r_hi.phc[l], c->r_hi.r0c[l], c->r_hi.r1c[l], L->pdc, L->hic, n, c->r_hi.a, c->r_hi.b);
        rfilt_iir_comp(c->r_hi.phs[l], c->r_hi.r0s[l], c->r_hi.r1s[l], L->pds, L->his, n, c->r_hi.a, c->r_hi.b);
    }
    for (int l = 0; l < levels; ++l) rlevel_copy(&c->r_old.lv[l], &c->r_cur.lv[l]); /* :267 */
    for (int l = levels - 2; l >= 0; --l)                               /* :269, RieszPyramid.cpp:248-252 */
        rlevel_amplify(&c->r_cur.lv[l], p->amplification, p->coWavelength * PI_PERCENT, gk13);
    float* mag = rpyr_collapse(&c->r_cur);                              /* :270 */
    for (size_t i = 0; i < n0; ++i) buf.d[i * 3] = mag[i];              /* :273-274 */
    free(mag); free(Lpl);
    lvmo_lab2bgr(buf.d, w * h, buf.d);                                  /* :275 */
    set_last(c, &buf);
    float_to_u8(&buf, 255.0f, (float)(1.0 / 255.0), out, ostride);      /* :276 */
    img_free(&buf);
    return 1;
}

/* ------------------------------------------------------------------------------------- */
/* MagnificationProcessor::process (MagnificationProcessor.cpp:17-67)                      */
/* ------------------------------------------------------------------------------------- */
int lvmo_process(lvmo_ctx* c, const lvmo_params* p, const uint8_t* in, int w, int h, int channels,
                 ptrdiff_t in_stride, uint8_t* out, ptrdiff_t out_stride, int* produced) {
    *produced = 0;
    if (p->mode == LVMO_MODE_NONE || in == NULL || w <= 0 || h <= 0) {  /* :21-29 */
        if (c->t_mode != LVMO_MODE_NONE) { motion_reset(c); color_reset(c); riesz_reset(c); tracker_disable(c); }
        return 0;
    }
    if (p->mode < 0 || p->mode > LVMO_MODE_NONE) return -1;
    if (channels != 1 && channels != 3) return -1;
    const int maxLevels = lvmo_max_levels(w, h);                        /* :32-33 */
    if (maxLevels < 1) return 0;
    int levels = p->levels < 1 ? 1 : (p->levels > maxLevels ? maxLevels : p->levels); /* :34 */
    if (levels > MAXLV - 2) levels = MAXLV - 2;
    /* StructuralTracker::update (MagnifyCore.hpp:53-65) */
    const int change = p->mode != c->t_mode || levels != c->t_levels || w != c->t_w || h != c->t_h ||
                       channels != c->t_channels || p->preprocess_key != c->t_pre;
    if (change) {
        c->t_mode = p->mode; c->t_levels = levels; c->t_w = w; c->t_h = h; c->t_channels = channels; c->t_pre = p->preprocess_key;
        motion_reset(c); color_reset(c); riesz_reset(c);                /* :39-43 */
    }
    int ok = 0;
    switch (p->mode) {                                                  /* :48-60 */
    case LVMO_MODE_LAPLACE: ok = magnify_motion(c, in, w, h, channels, in_stride, p, levels, out, out_stride); break;
    case LVMO_MODE_COLOR:   ok = magnify_color(c, in, w, h, channels, in_stride, p, levels, out, out_stride); break;
    case LVMO_MODE_PHASE:   ok = magnify_riesz(c, in, w, h, channels, in_stride, p, levels, out, out_stride); break;
    default: break;
    }
    *produced = ok;
    return 0;
}

/* =====================================================================================================
 * Preprocess + Grayscale (the stages in front of the magnifier).
 * PARITY UNPINNED like the rest of the OpenCV boundary: cv::resize(INTER_AREA) and cvtColor(BGR2GRAY) are
 * restated from OpenCV 4.x (imgproc/src/resize.cpp resizeAreaFast_ / resizeArea_, color_rgb.simd.hpp
 * RGB2Gray<uchar>); the reference holds no fixture for them.
 * ===================================================================================================== */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* PreprocessProcessor.cpp:13-31 and :36-39; GrayscaleProcessor.cpp:8-9 */
void lvmo_preprocess_geometry(const lvmo_pre_params* pp, int w, int h, int channels, int* rx, int* ry,
                              int* rw, int* rh, int* ow, int* oh, int* och) {
    const int divisor = clampi(pp->downscale, 1, 8);
    int x = 0, y = 0, cw = w, chh = h;
    if (pp->roi_enabled) {
        x = (int)lround((double)pp->roiX * w);
        y = (int)lround((double)pp->roiY * h);
        cw = (int)lround((double)pp->roiW * w);
        chh = (int)lround((double)pp->roiH * h);
        x = clampi(x, 0, w - 1);
        y = clampi(y, 0, h - 1);
        cw = clampi(cw, 1, w - x);
        chh = clampi(chh, 1, h - y);
    }
    *rx = x; *ry = y; *rw = cw; *rh = chh;
    if (divisor > 1) {
        *ow = cw / divisor > 1 ? cw / divisor : 1;
        *oh = chh / divisor > 1 ? chh / divisor : 1;
    } else { *ow = cw; *oh = chh; }
    *och = (pp->grayscale && channels != 1) ? 1 : channels;
}

/* computeResizeAreaTab (resize.cpp): source cells [dx*scale, (dx+1)*scale) with fractional end weights */
int lvmo_area_table(int ssize, int dsize, double scale, lvmo_area_tab* tab, int cap) {
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        const double fsx1 = dx * scale;
        const double fsx2 = fsx1 + scale;
        const double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
        sx1 = sx1 < sx2 ? sx1 : sx2;
        if (sx1 - fsx1 > 1e-3) {
            if (k < cap) { tab[k].di = dx; tab[k].si = sx1 - 1; tab[k].alpha = (float)((sx1 - fsx1) / cellWidth); }
            k++;
        }
        for (int sx = sx1; sx < sx2; sx++) {
            if (k < cap) { tab[k].di = dx; tab[k].si = sx; tab[k].alpha = (float)(1.0 / cellWidth); }
            k++;
        }
        if (fsx2 - sx2 > 1e-3) {
            double a = fsx2 - sx2; a = a < 1.0 ? a : 1.0; a = a < cellWidth ? a : cellWidth;
            if (k < cap) { tab[k].di = dx; tab[k].si = sx2; tab[k].alpha = (float)(a / cellWidth); }
            k++;
        }
    }
    return k;
}

static uint8_t sat_u8f(float v) {              /* saturate_cast<uchar>(float) = cvRound, round-half-even */
    const long r = lrintf(v);
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

void lvmo_resize_area_u8(const uint8_t* src, int w, int h, int cn, ptrdiff_t stride, uint8_t* dst, int dw, int dh) {
    const double scale_x = (double)w / dw, scale_y = (double)h / dh;
    const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);       /* saturate_cast<int>(double) */
    const int fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    if (fast) {
        /* resizeAreaFast_: integer sum of the iscale_x x iscale_y block; 2x2 takes the (sum + 2) >> 2 shortcut
         * of ResizeAreaFastVec, everything else saturate_cast<uchar>(sum * (1.f / area)) */
        const int area = iscale_x * iscale_y;
        const float scale = 1.f / area;
        const int fast2 = iscale_x == 2 && iscale_y == 2;
        for (int dy = 0; dy < dh; dy++)
            for (int dx = 0; dx < dw; dx++)
                for (int c = 0; c < cn; c++) {
                    int sum = 0;
                    for (int sy = 0; sy < iscale_y; sy++)
                        for (int sx = 0; sx < iscale_x; sx++)
                            sum += src[(size_t)(dy * iscale_y + sy) * stride + (size_t)(dx * iscale_x + sx) * cn + c];
                    dst[((size_t)dy * dw + dx) * cn + c] = fast2 ? (uint8_t)((sum + 2) >> 2) : sat_u8f((float)sum * scale);
                }
        return;
    }
    /* resizeArea_<uchar, float>: per source row a horizontal weighted sum (table order), rows accumulated
     * with their vertical weights; an output row is emitted when the table moves on to the next one */
    lvmo_area_tab* xtab = (lvmo_area_tab*)malloc(sizeof(lvmo_area_tab) * (size_t)(w * 2 + 2));
    lvmo_area_tab* ytab = (lvmo_area_tab*)malloc(sizeof(lvmo_area_tab) * (size_t)(h * 2 + 2));
    const int xn = lvmo_area_table(w, dw, scale_x, xtab, w * 2 + 2);
    const int yn = lvmo_area_table(h, dh, scale_y, ytab, h * 2 + 2);
    float* buf = (float*)malloc(sizeof(float) * (size_t)dw * cn);
    float* sum = (float*)malloc(sizeof(float) * (size_t)dw * cn);
    int prev_dy = ytab[0].di;
    for (int i = 0; i < dw * cn; i++) sum[i] = 0.f;
    for (int j = 0; j < yn; j++) {
        const float beta = ytab[j].alpha;
        const int dy = ytab[j].di, sy = ytab[j].si;
        const uint8_t* S = src + (size_t)sy * stride;
        for (int i = 0; i < dw * cn; i++) buf[i] = 0.f;
        for (int k = 0; k < xn; k++) {
            const float alpha = xtab[k].alpha;
            for (int c = 0; c < cn; c++) buf[xtab[k].di * cn + c] += S[xtab[k].si * cn + c] * alpha;
        }
        if (dy != prev_dy) {
            for (int i = 0; i < dw * cn; i++) { dst[(size_t)prev_dy * dw * cn + i] = sat_u8f(sum[i]); sum[i] = beta * buf[i]; }
            prev_dy = dy;
        } else {
            for (int i = 0; i < dw * cn; i++) sum[i] += beta * buf[i];
        }
    }
    for (int i = 0; i < dw * cn; i++) dst[(size_t)prev_dy * dw * cn + i] = sat_u8f(sum[i]);
    free(xtab); free(ytab); free(buf); free(sum);
}

/* RGB2Gray<uchar> (OpenCV 4.x): 15-bit fixed point, BY15 = 3735, GY15 = 19235, RY15 = 9798 */
void lvmo_bgr2gray_u8(const uint8_t* src, int npix, uint8_t* dst) {
    for (int i = 0; i < npix; i++)
        dst[i] = (uint8_t)((src[3 * i] * 3735 + src[3 * i + 1] * 19235 + src[3 * i + 2] * 9798 + (1 << 14)) >> 15);
}

void lvmo_preprocess(const lvmo_pre_params* pp, const uint8_t* in, int w, int h, int channels,
                     ptrdiff_t in_stride, uint8_t* out) {
    int rx, ry, rw, rh, ow, oh, och;
    lvmo_preprocess_geometry(pp, w, h, channels, &rx, &ry, &rw, &rh, &ow, &oh, &och);
    const int divisor = clampi(pp->downscale, 1, 8);
    const uint8_t* crop = in + (size_t)ry * in_stride + (size_t)rx * channels;
    uint8_t* tmp = (uint8_t*)malloc((size_t)ow * oh * channels);
    if (divisor > 1) lvmo_resize_area_u8(crop, rw, rh, channels, in_stride, tmp, ow, oh);          /* PreprocessProcessor.cpp:36-40 */
    else for (int y = 0; y < oh; y++) memcpy(tmp + (size_t)y * ow * channels, crop + (size_t)y * in_stride, (size_t)ow * channels);   /* :42 */
    if (och != channels) lvmo_bgr2gray_u8(tmp, ow * oh, out);                                       /* GrayscaleProcessor.cpp:13 */
    else memcpy(out, tmp, (size_t)ow * oh * channels);
    free(tmp);
}

/* ---- export pane composition: Exporter::compose (export/Exporter.cpp:53-88) with toBgr (:22-34) -------------------
 * split: 0 None, 1 LeftRight, 2 TopBottom (export/ExportTypes.hpp:11).  The text overlay (:36-50) is not restated.
 * Returns 0 and leaves the canvas untouched when the reference returns an empty Mat (:57, :65).                     */
int lvmo_compose_geometry(int split, int ow, int oh, int pw, int ph, int* cw, int* ch) {
    int w, h;
    if (split == 0) { w = pw & ~1; h = ph & ~1; }                                      /* :56 */
    else { w = (ow < pw ? ow : pw) & ~1; h = (oh < ph ? oh : ph) & ~1; }               /* :63-64 */
    if (w <= 0 || h <= 0) { *cw = 0; *ch = 0; return 0; }
    *cw = split == 1 ? 2 * w : w;                                                      /* :71 */
    *ch = split == 2 ? 2 * h : h;                                                      /* :79 */
    return 1;
}
static void compose_pane(const uint8_t* src, int cn, ptrdiff_t stride, int w, int h, uint8_t* dst, ptrdiff_t dstride) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++)
                dst[(size_t)y * dstride + 3 * x + c] = cn == 3 ? src[(size_t)y * stride + 3 * x + c] : src[(size_t)y * stride + x];   /* GRAY2BGR: b = g = r */
}
int lvmo_compose(int split, const uint8_t* orig, int ow, int oh, int och, ptrdiff_t ostride, const uint8_t* proc, int pw, int ph,
                 int pch, ptrdiff_t pstride, uint8_t* canvas, ptrdiff_t cstride) {
    if (!orig) { orig = proc; ow = pw; oh = ph; och = pch; ostride = pstride; }         /* :62 */
    int cw, ch;
    if (!lvmo_compose_geometry(split, ow, oh, pw, ph, &cw, &ch)) return 0;
    if (split == 0) { compose_pane(proc, pch, pstride, cw, ch, canvas, cstride); return 1; }   /* :58 */
    const int w = split == 1 ? cw / 2 : cw, h = split == 2 ? ch / 2 : ch;
    compose_pane(orig, och, ostride, w, h, canvas, cstride);                            /* :73, :81 */
    if (split == 1) compose_pane(proc, pch, pstride, w, h, canvas + 3 * w, cstride);    /* :74 */
    else compose_pane(proc, pch, pstride, w, h, canvas + (size_t)h * cstride, cstride); /* :82 */
    return 1;
}
