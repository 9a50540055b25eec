// lab_tables.cpp -- host-side constant tables and scalar helpers of liblvm_hip.so.
//
// * sRGB gamma / inverse-gamma tables and the BGR<->XYZ coefficient matrices that OpenCV 4's
//   float cvtColor(COLOR_BGR2Lab / COLOR_Lab2BGR) uses (reference call sites:
//   processing/magnification/MagnifyCore.hpp:90,152,219,275).  OpenCV evaluates both gammas
//   through 1024-knot natural cubic splines (splineBuild/splineInterpolate, color_lab.cpp);
//   the hot path only ever feeds u8/255 into the forward gamma, so that side collapses to a
//   256-entry table.
// * scalar helpers the reference computes on the host:
//   calculateMaxLevels (SpatialFilter.cpp:5-11), getOptimalBufferSize (TemporalFilter.cpp:82-94),
//   butterworth(2, Wn) (TemporalFilter.cpp:280-297).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "lvm_hip.h"
#include "lab_lut.h"

namespace lvm {

namespace {
constexpr int kTab = 1024;

// natural cubic spline through f[0..n]; 4 coefficients per interval (float arithmetic)
// OpenCV's splineBuild never writes tab[4 (n - 1)] and tab[4 (n - 1) + 1] in its first sweep and READS them in the second: its
// tables are static arrays, so it reads zeros.  Written here, because the caller's array need not be (until round 4 it was an
// uninitialised stack array of lvm_create: whatever the stack held went into the top knots of the inverse-gamma table -- harmless for
// the usual small junk, every pixel NaN when it happened to be a NaN pattern: one GPU test run in four inside a full pytest session,
// never in isolation, found with -ftrivial-auto-var-init=pattern on the emulation build, tools/emu_uninit.sh).
// Round 5: the OpenCV 4 form of the function (color_lab.cpp, softfloat): the forward sweep runs to i = n - 1 and the back substitution
// divides by 3 -- rounds 1-4 restated the 3.x form (sweep to n - 2, multiplications by 0.3333333333333333f), which the oracle keeps as
// the switch LVMO_VAR_SPLINE_CV3 (tests/test_oracle_variants.py: what the difference does to a frame).
void spline_build(const float* f, int n, float* tab) {
    float cn = 0.f;
    tab[0] = tab[1] = 0.f;
    tab[(n - 1) * 4] = tab[(n - 1) * 4 + 1] = 0.f;
    for (int i = 1; i < n; ++i) {
        const float t = 3.f * (f[i + 1] - 2.f * f[i] + f[i - 1]);
        const float l = 1.f / (4.f - tab[(i - 1) * 4]);
        tab[i * 4] = l;
        tab[i * 4 + 1] = (t - tab[(i - 1) * 4 + 1]) * l;
    }
    for (int i = n - 1; i >= 0; --i) {
        const float c = tab[i * 4 + 1] - tab[i * 4] * cn;
        const float b = f[i + 1] - f[i] - (cn + c * 2.f) / 3.f;
        const float d = (cn - c) / 3.f;
        tab[i * 4] = f[i];
        tab[i * 4 + 1] = b;
        tab[i * 4 + 2] = c;
        tab[i * 4 + 3] = d;
        cn = c;
    }
}
float spline_eval(float x, const float* tab, int n) {
    int ix = (int)x;
    ix = ix < 0 ? 0 : (ix > n - 1 ? n - 1 : ix);
    x -= (float)ix;
    const float* t = tab + ix * 4;
    return ((t[3] * x + t[2]) * x + t[1]) * x + t[0];
}
}  // namespace

void build_lab_tables(float gamma_u8[256], float invgamma[4096], float fwd[9], float inv[9]) {
    static const double M[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160,
                                0.072169, 0.019334, 0.119193, 0.950227};
    static const double Mi[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991,
                                 0.041556, 0.055648, -0.204043, 1.057311};
    static const double D65[3] = {0.950456, 1.0, 1.088754};
    float f[kTab + 1], g[kTab + 1];
    std::vector<float> gam_v((size_t)kTab * 4, 0.f);     // (not static: two threads may create contexts at the same time)
    float* gam = gam_v.data();
    // color_lab.cpp applyGamma / applyInvGamma: the argument and the binary32 constants (809/20000, 7827/2500000, 323/25,
    // 12/5, 11/200 as softfloat quotients) are promoted to softdouble, pow runs in binary64, ONE rounding to binary32
    const double thr = (double)(809.f / 20000.f), ithr = (double)(7827.f / 2500000.f), low = (double)(323.f / 25.f),
                 power = (double)(12.f / 5.f), shift = (double)(11.f / 200.f);
    for (int i = 0; i <= kTab; ++i) {
        const double x = (double)((float)i * (1.0f / kTab));
        f[i] = (float)(x <= thr ? x / low : std::pow((x + shift) / (1.0 + shift), power));
        g[i] = (float)(x <= ithr ? x * low : std::pow(x, 1.0 / power) * (1.0 + shift) - shift);
    }
    spline_build(f, kTab, gam);
    spline_build(g, kTab, invgamma);
    const float a255 = (float)(1.0 / 255.0f);
    for (int i = 0; i < 256; ++i) {
        float v = (float)i * a255;
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        gamma_u8[i] = spline_eval(v * (float)kTab, gam, kTab);
    }
    // blueIdx = 0: column 0 of the forward matrix multiplies B, row 0 of the inverse gives B
    for (int i = 0; i < 3; ++i) {
        const float sc = (i == 1) ? 1.f : (float)(1.0 / D65[i]);
        fwd[i * 3 + 2] = sc * (float)M[i * 3 + 0];
        fwd[i * 3 + 1] = sc * (float)M[i * 3 + 1];
        fwd[i * 3 + 0] = sc * (float)M[i * 3 + 2];
        const float wp = (float)D65[i];
        inv[i + 6] = (float)Mi[i] * wp;
        inv[i + 3] = (float)Mi[i + 3] * wp;
        inv[i + 0] = (float)Mi[i + 6] * wp;
    }
}

// ---- the output quantiser as a step function (round 6) ----------------------------------------------------------------------
// The last thing the Lab modes do to a pixel is u8 = saturate_cast<uchar>(cvRound(255 * invGamma(clip01(c)) + 1/255)) per channel
// (Lab2RGBfloat's splineInterpolate, then convertTo(CV_8U, 255, 1.0/255): MagnifyCore.hpp:152-153, :275-276) -- a monotone step
// function of the clipped linear value c with exactly 255 steps (monotone over EVERY float in [0, 1]: checked exhaustively by
// tests/test_u8_steps.py on the CPU and by lvm_debug_sweep_u8_steps on the GPU; the cubic's rounding noise never straddles a
// rounding boundary twice).  Its 255 thresholds, found by bisection against the exact code above, replace spline evaluation + scale +
// round + clamp in the output kernels by one table hit and one compare: [0, 1) is cut into 4096 slices of 1/4096 (the smallest gap
// between two thresholds is 1.24 slices, in the linear segment of the sRGB curve), entry i = { the threshold inside slice i scaled
// by 4096, or +inf ; the value at the slice's start }.  Bit-exact u8 by construction.
namespace {
inline int u8_of_linear(float c, const float* invgamma, float a255) {
    const float cc = c < 0.f ? 0.f : (c > 1.f ? 1.f : c);
    const float o = spline_eval(cc * (float)kTab, invgamma, kTab);
    const float v = o * 255.0f + a255;
    const float r = std::nearbyint(v);           // cvRound: round half to even (the default rounding mode)
    return r < 0.f ? 0 : (r > 255.f ? 255 : (int)r);
}
}  // namespace

bool build_u8_steps(const float invgamma[4096], uint32_t steps[2 * kU8StepSlices]) {
    const float a255 = (float)(1.0 / 255.0f);
    uint32_t one_bits; { const float one = 1.0f; std::memcpy(&one_bits, &one, 4); }
    auto at_bits = [&](uint32_t b) { float c; std::memcpy(&c, &b, 4); return u8_of_linear(c, invgamma, a255); };
    if (at_bits(0) != 0 || at_bits(one_bits) != 255) return false;
    float thr[256];                              // thr[k] = the smallest c with u8(c) >= k (non-negative floats order like their bit patterns)
    for (int k = 1; k <= 255; ++k) {
        uint32_t lo = 0, hi = one_bits;          // u8(lo) < k <= u8(hi)
        while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (at_bits(mid) >= k) hi = mid; else lo = mid; }
        std::memcpy(&thr[k], &hi, 4);
        if (at_bits(hi) != k || at_bits(hi - 1) != k - 1) return false;      // a step of more than one level, or not monotone here
        if (k > 1 && !(thr[k] > thr[k - 1])) return false;
    }
    const uint32_t inf_bits = 0x7f800000u;
    for (int i = 0; i < kU8StepSlices; ++i) { steps[2 * i] = inf_bits; steps[2 * i + 1] = 0; }
    int k = 1;
    for (int i = 0; i < kU8StepSlices; ++i) {
        steps[2 * i + 1] = (uint32_t)(k - 1);                               // value on [i / 4096, first threshold inside the slice)
        int inside = 0;
        while (k <= 255 && thr[k] * (float)kU8StepSlices < (float)(i + 1)) {   // (the scaling is a power of two: exact)
            const float t = thr[k] * (float)kU8StepSlices;
            std::memcpy(&steps[2 * i], &t, 4);
            ++k; ++inside;
        }
        if (inside > 1) return false;
    }
    // the kernels clamp to the largest float below 4096: it must already give what 1.0 (and everything above) gives
    if (k != 256) return false;
    return true;
}

// ---- OpenCV 4's RGB2Lab interpolation table (color_lab.cpp initLabTabs, the enableRGB2LabInterpolation block) ----------
// OpenCV builds the 33^3 table with its softfloat type, i.e. IEEE binary32 operations rounded one by one; the same
// sequence is restated here on native floats (this translation unit is built with -ffp-contract=off):
//   R, G, B = applyGamma(p / 32): x <= 809/20000 ? x / (323/25) : pow((x + 11/200) / (1 + 11/200), 12/5), evaluated in
//   softdouble on the promoted argument and binary32 constants and rounded once (sf_gamma below; the C library's pow stands
//   in for softdouble's, whose ~1e-15 relative error changes a binary32 result with p ~ 1e-8 per node);
//   X, Y, Z = R C0 + G C1 + B C2 (coefficients = double(sRGB2XYZ_D65 * 1 / D65) rounded to binary32);
//   f(t) = t > 216/24389 ? cbrt(t) : fma(t, 841/108, 16/116)   (softfloat cbrt = the cv::cubeRoot polynomial);
//   L = Y > 216/24389 ? 116 fY - 16 : Y * (24389/27);  a = 500 (fX - fY);  b = 200 (fY - fZ);
//   entries cvRound(16384 L / 100), cvRound(16384 (a + 128) / 256), cvRound(16384 (b + 128) / 256).
// UNPINNED: no OpenCV exists in this image to compare the table with; a real build's table can be
// recovered by converting the 33^3 node colours (oracle/ref_driver.cpp does that where OpenCV exists) and handed to
// lvm_set_lab_lut().  The oracle (oracle/lvm_oracle.c lab_lut_init) restates the same sequence independently;
// tests/test_lab_lut.py compares the two tables entry by entry.
namespace {
float cube_root_f32(float value) {              // cv::cubeRoot / softfloat f32_cbrt
    uint32_t vi; std::memcpy(&vi, &value, 4);
    const uint32_t ix = vi & 0x7fffffffu, s = vi & 0x80000000u;
    int ex = (int)(ix >> 23) - 127;
    int shx = ex % 3;
    shx -= shx >= 0 ? 3 : 0;
    ex = (ex - shx) / 3;
    const uint32_t fb = (ix & ((1u << 23) - 1)) | ((uint32_t)(shx + 127) << 23);
    float ff; std::memcpy(&ff, &fb, 4);
    double fr = (double)ff;
    fr = ((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr + 119.1654824285581628956914143) * fr +
           13.43250139086239872172837314) * fr + 0.1636161226585754240958355063) /
         ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr + 168.5254414101568283957668343) * fr +
           33.9905941350215598754191872) * fr + 1.0);
    const float rf = (float)fr;
    uint32_t r; std::memcpy(&r, &rf, 4);
    r = (r + ((uint32_t)ex << 23) + s) & ((vi * 2u) != 0u ? 0xffffffffu : 0u);
    float out; std::memcpy(&out, &r, 4);
    return out;
}
// applyGamma(softfloat): "softdouble xd = x; xd <= gammaThreshold ? xd / gammaLowScale : pow((xd + gammaXshift) / (one + gammaXshift),
// softdouble(gammaPower))" -- promoted argument, promoted binary32 constants, binary64 pow, one rounding (round 4; the three
// binary32 operations of rounds 1-3 moved ~20 of the 107 811 entries by one unit)
float sf_gamma(float x) {
    const float thr = 809.f / 20000.f, low = 323.f / 25.f, shift = 11.f / 200.f, power = 12.f / 5.f;
    const double xd = (double)x;
    return (float)(xd <= (double)thr ? xd / (double)low : std::pow((xd + (double)shift) / (1.0 + (double)shift), (double)power));
}
}  // namespace

// compact[((r * 33 + q) * 33 + p) * 3 + ch], p = R index (fastest), q = G, r = B
void build_lab_lut_compact(std::vector<int16_t>& compact) {
    static const double M[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    static const double D65[3] = {0.950456, 1.0, 1.088754};
    float C[9];
    for (int i = 0; i < 3; ++i) {
        const double sw = i == 1 ? 1.0 : 1.0 / D65[i];
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = (float)(M[i * 3 + j] * sw);
    }
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f, l9033 = 24389.f / 27.f;
    float gam[33];
    for (int p = 0; p < 33; ++p) gam[p] = sf_gamma((float)p / 32.f);
    compact.assign((size_t)33 * 33 * 33 * 3, 0);
    for (int r = 0; r < 33; ++r)
        for (int q = 0; q < 33; ++q)
            for (int p = 0; p < 33; ++p) {
                const float R = gam[p], G = gam[q], B = gam[r];
                const float X = R * C[0] + G * C[1] + B * C[2];
                const float Y = R * C[3] + G * C[4] + B * C[5];
                const float Z = R * C[6] + G * C[7] + B * C[8];
                const float FX = X > lthresh ? cube_root_f32(X) : std::fmaf(X, lscale, lbias);
                const float FY = Y > lthresh ? cube_root_f32(Y) : std::fmaf(Y, lscale, lbias);
                const float FZ = Z > lthresh ? cube_root_f32(Z) : std::fmaf(Z, lscale, lbias);
                const float L = Y > lthresh ? (116.f * FY - 16.f) : (Y * l9033);
                const float a = 500.f * (FX - FY), b = 200.f * (FY - FZ);
                int16_t* e = &compact[(((size_t)r * 33 + q) * 33 + p) * 3];
                e[0] = (int16_t)std::lrintf(16384.f * L / 100.f);
                e[1] = (int16_t)std::lrintf(16384.f * (a + 128.f) / 256.f);
                e[2] = (int16_t)std::lrintf(16384.f * (b + 128.f) / 256.f);
            }
}

// device layouts (lab_lut.h): every entry as 2 v + 1 (the rounding constant of CV_DESCALE rides in the table).
// ab[n] = (2 a + 1) | (2 b + 1) << 16 of node n = p + 33 q + 1089 r, padded by one B plane + 35 zero entries (upper neighbours of
// the edge nodes carry weight 0 and are read unclamped); lcells[8 c + 4 dp + 2 dq + dr] = 2 L + 1 of node (p + dp, q + dq, r + dr),
// indices clamped to 32, c = the cell's origin node (or its position in 2 x 2 x 2 blocks, LVM_LUT_LCELL_BLOCKED)
static size_t lcell_index(int p, int q, int r) {
#if LVM_LUT_LCELL_BLOCKED
    return (((size_t)(p >> 1) + 17 * (q >> 1) + 289 * (r >> 1)) << 3) | (p & 1) | ((q & 1) << 1) | ((r & 1) << 2);
#else
    return (size_t)p + 33 * q + 1089 * r;
#endif
}
void lab_lut_device_tables(const int16_t* compact, std::vector<uint32_t>& ab, std::vector<int16_t>& lcells) {
    ab.assign((size_t)kLabAbWords, 0u);
    lcells.assign((size_t)kLabLCells * 8, 0);
    auto at = [&](int p, int q, int r, int ch) {
        p = p > 32 ? 32 : p; q = q > 32 ? 32 : q; r = r > 32 ? 32 : r;
        return (uint32_t)(2 * (int)compact[(((size_t)r * 33 + q) * 33 + p) * 3 + ch] + 1);         // <= 2 * 16384 + 1: fits uint16
    };
    for (int r = 0; r < 33; ++r)
        for (int q = 0; q < 33; ++q)
            for (int p = 0; p < 33; ++p) {
                const size_t n = (size_t)p + 33 * q + 1089 * r;
                ab[n] = at(p, q, r, 1) | (at(p, q, r, 2) << 16);
                for (int dp = 0; dp < 2; ++dp)
                    for (int dq = 0; dq < 2; ++dq)
                        for (int dr = 0; dr < 2; ++dr) lcells[lcell_index(p, q, r) * 8 + 4 * dp + 2 * dq + dr] = (int16_t)(uint16_t)at(p + dp, q + dq, r + dr, 0);
            }
}
// lab_lut.h takes cell and weight of a u8 channel value from (514 u + 4) >> 8 instead of rounding float(u) * a255 * 16384:
// the identity is checked for all 256 values whenever a context is created
bool lab_lut_fine_index_ok() {
    const float a255 = (float)(1.0 / 255.0f);
    for (int u = 0; u < 256; ++u) {
        float v = (float)u * a255;
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        const long c = std::lrintf(v * 16384.f);
        if ((c >> 5) != (long)((u * 514 + 4) >> 8)) return false;
    }
    return true;
}
int max_levels(int w, int h) {
    int n = 0;
    while (w > 5 && h > 5) { w = (1 + w) / 2; h = (1 + h) / 2; ++n; }
    return n;
}

int optimal_buffer_size(int fps) {
    unsigned r = (unsigned)(2 * fps > 16 ? 2 * fps : 16);
    r--; r |= r >> 1; r |= r >> 2; r |= r >> 4; r |= r >> 8; r |= r >> 16; r++;
    return (int)r;
}

// Order-2 digital Butterworth low-pass, same construction as the reference: analog prototype
// s^2 + sqrt2 s + 1 -> low-pass at w0 = 2 fs tan(pi Wn / fs) -> bilinear transform (fs = 2)
// -> normalise by a0.  Degenerate inputs (w0 == 0) produce all-zero coefficients like the
// reference does.
void butterworth2(double Wn, double a[3], double b[3]) {
    const double kPi = 3.1415926535897932384626433832795;
    const double fs = 2.0;
    const double w0 = 2.0 * fs * std::tan(kPi * Wn / fs);
    const double p1r = -std::sin(0.25 * kPi), p1i = std::cos(0.25 * kPi);
    const double p2r = -std::sin(0.75 * kPi), p2i = std::cos(0.75 * kPi);
    const double proto[3] = {1.0, -(p1r + p2r), p1r * p2r - p1i * p2i};
    const double pw[3] = {std::pow(w0, 2.0), std::pow(w0, 1.0), std::pow(w0, 0.0)};
    double al[3], bl = (pw[2] == 0.0) ? 0.0 : 1.0 * (pw[0] / pw[2]);
    for (int k = 0; k < 3; ++k) al[k] = (pw[k] == 0.0) ? 0.0 : proto[k] * (pw[0] / pw[k]);
    const double lead = al[0];
    for (int k = 0; k < 3; ++k) al[k] = (lead == 0.0) ? 0.0 : al[k] / lead;
    bl = (lead == 0.0) ? 0.0 : bl / lead;
    static const double C2[3] = {1.0, 2.0, 1.0};
    double ap[3], bp[3];
    for (int j = 0; j < 3; ++j) bp[j] = C2[j] * bl;
    for (int j = 0; j < 3; ++j) {
        double v = 0.0;
        for (int i = 0; i <= 2; ++i)
            for (int k = 0; k <= i; ++k)
                for (int l = 0; l <= 2 - i; ++l)
                    if (k + l == j) {
                        const double cik = (i == 2 && k == 1) ? 2.0 : 1.0;
                        const double cml = (2 - i == 2 && l == 1) ? 2.0 : 1.0;
                        v += cik * cml * al[2 - i] * std::pow(2.0 * fs, (double)i) * std::pow(-1.0, (double)k);
                    }
        ap[j] = v;
    }
    const double l2 = ap[0];
    for (int k = 0; k < 3; ++k) {
        a[k] = (l2 == 0.0) ? 0.0 : ap[k] / l2;
        b[k] = (l2 == 0.0) ? 0.0 : bp[k] / l2;
    }
}

}  // namespace lvm

extern "C" {
int lvm_max_levels(int w, int h) { return lvm::max_levels(w, h); }
int lvm_optimal_buffer_size(int fps) { return lvm::optimal_buffer_size(fps); }
void lvm_butterworth2(double Wn, double a[3], double b[3]) { lvm::butterworth2(Wn, a, b); }
}
