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
#include <cstring>

#include "lvm_hip.h"

namespace lvm {

namespace {
constexpr int kTab = 1024;

// natural cubic spline through f[0..n]; 4 coefficients per interval (float arithmetic)
void spline_build(const float* f, int n, float* tab) {
    float cn = 0.f;
    tab[0] = tab[1] = 0.f;
    for (int i = 1; i < n - 1; ++i) {
        const float t = 3.f * (f[i + 1] - 2.f * f[i] + f[i - 1]);
        const float l = 1.f / (4.f - tab[(i - 1) * 4]);
        tab[i * 4] = l;
        tab[i * 4 + 1] = (t - tab[(i - 1) * 4 + 1]) * l;
    }
    for (int i = n - 1; i >= 0; --i) {
        const float c = tab[i * 4 + 1] - tab[i * 4] * cn;
        const float b = f[i + 1] - f[i] - (cn + c * 2.f) * 0.3333333333333333f;
        const float d = (cn - c) * 0.3333333333333333f;
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
    static float gam[kTab * 4];
    for (int i = 0; i <= kTab; ++i) {
        const double x = (double)i / kTab;
        f[i] = (float)(x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4));
        g[i] = (float)(x <= 0.0031308 ? x * 12.92 : 1.055 * std::pow(x, 1.0 / 2.4) - 0.055);
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
