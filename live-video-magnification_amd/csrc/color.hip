// color.hip -- Gaussian-pyramid + ideal temporal band-pass colour magnification on gfx950.
//
// Replaces magcore::magnifyColor (reference: processing/magnification/MagnifyCore.hpp:163-206)
// with buildGaussPyrFromImg / buildImgFromGaussPyr / img2tempMat / tempMat2img
// (SpatialFilter.cpp:13-23, 40-50, 63-89) and idealFilter / createIdealBandpassFilter
// (TemporalFilter.cpp:24-80).
//
// HBM layout per context: G_l[plane][h_l][w_l] (l = 1..L, unscaled [0,255] floats), the rolling
// window win[row][slot] with row = plane * n_L + pixel (time-contiguous per row, ring buffer of
// capacity >= getOptimalBufferSize(fps) + 33 slots: one row's history is one contiguous run and a
// temporal batch of up to 32 frames appends 32 columns), the up-chain images up_k[plane] of size
// (w_L 2^k) x (h_L 2^k), two min/max pairs per stream and the bilinear resize tables.
//
// Launch sequence (one frame, or a batch of <= 32 frames of every stream):
//   k_down0_rows | k_down0_v4, k_pyr_down_rows | _multi   u8 -> float -> Gaussian pyramid (pyramid.h)
//   k_col_append                   smallest level -> window slots; resets the min/max cells
//   k_col_dft | k_col_dft_thin     per row: packed real DFT of the needed elements (float64 sums), 0/1 mask applied
//                                  as a packed complex spectrum (mulSpectrums), inverse transform of all T samples
//                                  -> global min/max, keep column 1   (wave per row | thread per row, narrow bands)
//   k_col_norm                     min-max normalise column 1, x alpha -> up_0
//   k_pyr_up x (L-1)               up_k -> up_{k+1}
//   k_col_out_rows<false>          last pyrUp + bilinear resize + input add -> global min/max   (wave strips;
//   k_col_out_rows<true>           same values again -> u8 with the min/max rescale             k_col_out* = tiles)
#include <cmath>
#include <cstdlib>

#include <vector>

#include "pyramid.h"

namespace lvm {

// monotonic float <-> int key so that integer atomicMin/Max order floats
__device__ __forceinline__ int fkey(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float fkey_inv(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// Per stream (and frame of a batch): ideal-filter range (1) and output range (2).  Each extreme is kept
// in kMMCells cells so that the ~2000 workgroups of a launch do not serialise on one atomic address
// (~12 ns per contended atomic on MI355X); readers fold the cells.
constexpr int kMMCells = 32;
struct MinMax { int mn1[kMMCells], mx1[kMMCells], mn2[kMMCells], mx2[kMMCells]; };
__device__ __forceinline__ float mm_min(const int* cells) { int k = cells[0]; for (int i = 1; i < kMMCells; ++i) k = cells[i] < k ? cells[i] : k; return fkey_inv(k); }
__device__ __forceinline__ float mm_max(const int* cells) { int k = cells[0]; for (int i = 1; i < kMMCells; ++i) k = cells[i] > k ? cells[i] : k; return fkey_inv(k); }

__device__ __forceinline__ void block_minmax(float mn, float mx, int* gmn, int* gmx) {
    __shared__ float s_mn[256], s_mx[256];
    const int t = threadIdx.x;
    s_mn[t] = mn; s_mx[t] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
            s_mn[t] = s_mn[t + s] < s_mn[t] ? s_mn[t + s] : s_mn[t];
            s_mx[t] = s_mx[t + s] > s_mx[t] ? s_mx[t + s] : s_mx[t];
        }
        __syncthreads();
    }
    if (t == 0) {
        const int cell = (blockIdx.x + blockIdx.y * gridDim.x) % kMMCells;
        atomicMin(gmn + cell, fkey(s_mn[0])); atomicMax(gmx + cell, fkey(s_mx[0]));
    }
}
// Per-wave variant for kernels whose waves may belong to different frames: every wave folds its 64 values
// through LDS (one barrier for the workgroup) and its first lane issues the two atomics into the output
// range cells of frame `z`.
__device__ __forceinline__ void block_minmax_waves(float mn, float mx, MinMax* mm, int z, bool active) {
    __shared__ float s_mn[256], s_mx[256];
    const int t = threadIdx.x;
    s_mn[t] = mn; s_mx[t] = mx;
    __syncthreads();
    if ((t & 63) == 0 && active) {
        float a = s_mn[t], b = s_mx[t];
        for (int i = 1; i < 64; ++i) { a = s_mn[t + i] < a ? s_mn[t + i] : a; b = s_mx[t + i] > b ? s_mx[t + i] : b; }
        const int cell = (int)((blockIdx.x * 4u + (unsigned)(t >> 6)) % kMMCells);
        atomicMin(mm[z].mn2 + cell, fkey(a)); atomicMax(mm[z].mx2 + cell, fkey(b));
    }
}

// Barrier-free variant: six ds_bpermute butterfly steps per extreme fold the wave (no LDS memory, no workgroup barrier at the
// tail of every workgroup: the LDS scan above cost the min/max pass 10 of its 140 us per 32 frames), lane 0 issues the atomics.
__device__ __forceinline__ void wave_minmax(float mn, float mx, MinMax* mm, int z, bool active) {
    const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) {
        const float on = __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ k) << 2, __float_as_int(mn)));
        const float ox = __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ k) << 2, __float_as_int(mx)));
        mn = on < mn ? on : mn; mx = ox > mx ? ox : mx;
    }
    if (lane == 0 && active) {
        const int cell = (int)((blockIdx.x * 4u + (unsigned)(threadIdx.x >> 6)) % kMMCells);
        atomicMin(mm[z].mn2 + cell, fkey(mn)); atomicMax(mm[z].mx2 + cell, fkey(mx));
    }
}

// img2tempMat (SpatialFilter.cpp:63-84): one window column per frame.  Window layout: win[row][slot]
// (time-contiguous per row, ring of `cap` slots) so that a wave reads one row's history as one
// contiguous run.
// grid (row blocks, streams, frames): frame f of the batch goes to ring slot (slot + f) mod cap
__global__ __launch_bounds__(256) void k_col_append(const float* __restrict__ GL, float* __restrict__ win, int rows,
                                                    int rows_per_stream, int cap, int slot, MinMax* mm, int nstreams) {
    const int r = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, f = blockIdx.z;
    int sl = slot + f; if (sl >= cap) sl -= cap;
    if (r < rows) win[((size_t)b * rows_per_stream + r) * cap + sl] = GL[((size_t)f * nstreams + b) * rows + r];
    if (blockIdx.x == 0 && threadIdx.x < kMMCells) {
        MinMax& m = mm[f * nstreams + b];
        m.mn1[threadIdx.x] = m.mn2[threadIdx.x] = fkey(INFINITY); m.mx1[threadIdx.x] = m.mx2[threadIdx.x] = fkey(-INFINITY);
    }
}
// ring growth: copy the n live columns of every row into a larger ring, oldest column first
__global__ __launch_bounds__(256) void k_col_regrow(const float* __restrict__ ow, int ocap, int slot0, int n,
                                                    float* __restrict__ nw, int ncap, int rows) {
    const int r = blockIdx.x, t0 = threadIdx.x;
    if (r >= rows) return;
    for (int t = t0; t < n; t += 256) nw[(size_t)r * ncap + t] = ow[(size_t)r * ocap + (slot0 + t) % ocap];
}

// idealFilter (TemporalFilter.cpp:24-57): one WAVE per window row.  n = window length, slot0 = ring
// index of the oldest column.  Packed (CCS) element x: [Re0, Re1, Im1, ..., Re(n/2) if n even].
// Entry list (built once per workgroup): the DC element, every complex bin whose mask pair is not
// all-zero, the Nyquist element -- only those can be non-zero after mulSpectrums with the 0/1 mask.
// Forward: lane j owns entry j and runs the float64 sum over t = 0..n-1 in order (samples are LDS
// broadcasts).  Inverse: lane owns output samples t, t+64, ... and sums the entries in ascending
// bin order.  Both are the oracle's summation orders, so results are order-faithful.
constexpr int kDftMaxN = 1024;             // window lengths above this use k_col_dft_serial
constexpr int kTwAllMax = 512;             // window lengths up to this get their twiddle tables built once for every warm-up length
constexpr int kDftRows = 4;                // rows (waves) per workgroup
struct DftEntry { short bin; short kind; };   // kind 0 = complex bin, 1 = DC, 2 = Nyquist
__global__ __launch_bounds__(256) void k_col_dft(const float* __restrict__ win, int slot0, int n, int cap,
                                                 int rows_per_stream, int live_per_stream, double fl, double fh,
                                                 const double* __restrict__ tw, float* __restrict__ col1, MinMax* mm) {
    __shared__ double s_tw[2 * kDftMaxN];
    __shared__ float s_row[kDftRows][kDftMaxN];
    __shared__ float s_y[kDftRows][kDftMaxN + 2];
    __shared__ DftEntry s_ent[kDftMaxN / 2 + 2];
    __shared__ int s_ne;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, fr = blockIdx.z;              // stream, frame of the batch
    slot0 += fr; if (slot0 >= cap) slot0 -= cap;            // frame fr sees the window shifted by fr columns
    col1 += (size_t)fr * gridDim.y * rows_per_stream;
    mm += fr * gridDim.y;
    auto m = [&](int x) { return (x >= fl && x <= fh) ? 1.0f : 0.0f; };              // :70-77
    for (int i = threadIdx.x; i < 2 * n; i += 256) s_tw[i] = tw[i];
    if (threadIdx.x == 0) {
        int ne = 0;
        const int half = (n - 1) / 2;
        if (m(0) != 0.f) { s_ent[ne].bin = 0; s_ent[ne].kind = 1; ++ne; }
        for (int k = 1; k <= half; ++k)
            if (m(2 * k - 1) != 0.f || m(2 * k) != 0.f) { s_ent[ne].bin = (short)k; s_ent[ne].kind = 0; ++ne; }
        if (n % 2 == 0 && m(n - 1) != 0.f) { s_ent[ne].bin = (short)(n / 2); s_ent[ne].kind = 2; ++ne; }
        s_ne = ne;
    }
    __syncthreads();
    const int ne = s_ne;
    const double* cs = s_tw;
    const double* sn = s_tw + n;
    float vmin = INFINITY, vmax = -INFINITY;
    const int iters = (live_per_stream + gridDim.x * kDftRows - 1) / (gridDim.x * kDftRows);
    for (int it = 0; it < iters; ++it) {
        const int r = (it * gridDim.x + blockIdx.x) * kDftRows + wv;
        const bool live = r < live_per_stream;
        const size_t grow = (size_t)b * rows_per_stream + (live ? r : 0);
        if (live) {
            const float* wr = win + grow * cap;
            for (int t = lane; t < n; t += 64) { int slot = slot0 + t; if (slot >= cap) slot -= cap; s_row[wv][t] = wr[slot]; }
        }
        __syncthreads();
        if (live) {
            for (int j = lane; j < ne; j += 64) {                                     // dft + mulSpectrums
                const int k = s_ent[j].bin, kind = s_ent[j].kind;
                double ar = 0, ai = 0;
                int idx = 0;
                for (int t = 0; t < n; ++t) {
                    const double v = (double)s_row[wv][t];
                    ar += v * cs[idx];
                    ai += -v * sn[idx];
                    idx += k; if (idx >= n) idx -= n;
                }
                const float a = (float)(ar / n), bq = (float)(ai / n);
                if (kind == 0) {
                    const float ma = m(2 * k - 1), mb = m(2 * k);
                    s_y[wv][2 * j] = a * ma - bq * mb;
                    s_y[wv][2 * j + 1] = bq * ma + a * mb;
                } else {
                    s_y[wv][2 * j] = a * (kind == 1 ? m(0) : m(n - 1));
                    s_y[wv][2 * j + 1] = 0.f;
                }
            }
        }
        __syncthreads();
        if (live) {
            const bool has_dc = ne > 0 && s_ent[0].kind == 1;
            const bool has_ny = ne > 0 && s_ent[ne - 1].kind == 2;
            const int j0 = has_dc ? 1 : 0, j1 = has_ny ? ne - 1 : ne;
            for (int t = lane; t < n; t += 64) {                                      // idft, every sample
                double acc = has_dc ? s_y[wv][0] : 0.f;
                for (int j = j0; j < j1; ++j) {
                    const int k = s_ent[j].bin;
                    const int idx = (int)(((unsigned)k * (unsigned)t) % (unsigned)n);
                    acc += 2.0 * ((double)s_y[wv][2 * j] * cs[idx] - (double)s_y[wv][2 * j + 1] * sn[idx]);
                }
                if (n % 2 == 0) acc += (t % 2 ? -1.0 : 1.0) * (double)(has_ny ? s_y[wv][2 * (ne - 1)] : 0.f);
                const float v = (float)(acc / n);
                vmin = v < vmin ? v : vmin; vmax = v > vmax ? v : vmax;
                if (t == 1) col1[grow] = v;                                           // MagnifyCore.hpp:190-192
            }
        }
        __syncthreads();
    }
    block_minmax(vmin, vmax, mm[b].mn1, mm[b].mx1);
}

// Narrow pass bands (at most kThinBins non-zero spectrum entries -- the usual pulse band is 1-3 bins):
// one THREAD per window row.  With so few entries the wave-per-row kernel runs its forward sums on 1-3
// lanes of 64; here every lane runs the (in-order, float64) sums of its own row for all entries while
// it walks once over the window, then the inverse sums of all n samples.  Entry list built on the host.
constexpr int kThinBins = 8;
struct ThinBins { int ne; int bin[kThinBins]; int kind[kThinBins]; float ma[kThinBins], mb[kThinBins]; };
__global__ __launch_bounds__(256) void k_col_dft_thin(const float* __restrict__ win, int slot0, int n, int cap,
                                                      int rows_per_stream, int live_per_stream,
                                                      const double* __restrict__ tw, float* __restrict__ col1, MinMax* mm,
                                                      ThinBins eb) {
    const int r = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, fr = blockIdx.z;
    slot0 += fr; if (slot0 >= cap) slot0 -= cap;
    col1 += (size_t)fr * gridDim.y * rows_per_stream;
    mm += fr * gridDim.y;
    const bool live = r < live_per_stream;
    const size_t grow = (size_t)b * rows_per_stream + (live ? r : 0);
    // twiddles in LDS: the table index depends on the running bin phase, so a global (scalar) load per
    // sample would put its latency on every step of the serial sums
    __shared__ double s_tw[2 * kDftMaxN];
    for (int i = threadIdx.x; i < 2 * n; i += 256) s_tw[i] = tw[i];
    __syncthreads();
    const double* cs = s_tw;
    const double* sn = s_tw + n;
    const bool pow2 = (n & (n - 1)) == 0;            // x / 2^k == x * 2^-k exactly: skip the float64 division
    const double inv_n = 1.0 / (double)n;
    float vmin = INFINITY, vmax = -INFINITY;
    if (live) {
        const float* wr = win + grow * cap;
        double ar[kThinBins], ai[kThinBins];
        int idx[kThinBins];
#pragma unroll
        for (int j = 0; j < kThinBins; ++j) { ar[j] = ai[j] = 0; idx[j] = 0; }
        constexpr int U = 8;                         // samples per batch; the next batch is in flight while this one is summed
        auto fetch = [&](int t0, float (&v)[U]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int slot = slot0 + (t0 + u < n ? t0 + u : n - 1);
                if (slot >= cap) slot -= cap;
                v[u] = wr[slot];
            }
        };
        float cur8[U], nxt8[U];
        fetch(0, cur8);
        for (int t0 = 0; t0 < n; t0 += U) {                                           // dft: sums in sample order
            fetch(t0 + U < n ? t0 + U : t0, nxt8);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t0 + u >= n) break;
                const double v = (double)cur8[u];
#pragma unroll
                for (int j = 0; j < kThinBins; ++j) {
                    if (j >= eb.ne) break;
                    ar[j] += v * cs[idx[j]];
                    ai[j] += -v * sn[idx[j]];
                    idx[j] += eb.bin[j]; if (idx[j] >= n) idx[j] -= n;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) cur8[u] = nxt8[u];
        }
        float yre[kThinBins], yim[kThinBins];
#pragma unroll
        for (int j = 0; j < kThinBins; ++j) {                                         // mulSpectrums with the packed mask
            yre[j] = yim[j] = 0.f;
            if (j >= eb.ne) continue;
            const float a = (float)(pow2 ? ar[j] * inv_n : ar[j] / n), bq = (float)(pow2 ? ai[j] * inv_n : ai[j] / n);
            if (eb.kind[j] == 0) { yre[j] = a * eb.ma[j] - bq * eb.mb[j]; yim[j] = bq * eb.ma[j] + a * eb.mb[j]; }
            else yre[j] = a * eb.ma[j];
            idx[j] = 0;
        }
        const bool has_dc = eb.ne > 0 && eb.kind[0] == 1;
        const bool has_ny = eb.ne > 0 && eb.kind[eb.ne - 1] == 2;
        float yny = 0.f;
#pragma unroll
        for (int j = 0; j < kThinBins; ++j) if (has_ny && j == eb.ne - 1) yny = yre[j];
        auto idft = [&](auto scale) __attribute__((always_inline)) {                  // idft of every sample
            for (int t = 0; t < n; ++t) {
                double acc = has_dc ? yre[0] : 0.f;
#pragma unroll
                for (int j = 0; j < kThinBins; ++j) {
                    if (j >= eb.ne) break;
                    if (eb.kind[j] != 0) continue;
                    acc += 2.0 * ((double)yre[j] * cs[idx[j]] - (double)yim[j] * sn[idx[j]]);
                    idx[j] += eb.bin[j]; if (idx[j] >= n) idx[j] -= n;
                }
                if (n % 2 == 0) acc += (t % 2 ? -1.0 : 1.0) * (double)yny;
                const float v = (float)scale(acc);
                vmin = v < vmin ? v : vmin; vmax = v > vmax ? v : vmax;
                if (t == 1) col1[grow] = v;                                           // MagnifyCore.hpp:190-192
            }
        };
        if (pow2) idft([&](double x) { return x * inv_n; });                          // (uniform branch: no division in the loop)
        else idft([&](double x) { return x / n; });
    }
    block_minmax(vmin, vmax, mm[b].mn1, mm[b].mx1);
}

// The same narrow-band filter with EIGHT lanes per window row (at most four entries, i.e. eight real sums).  One thread per row is a
// chain of 2 n dependent float64 steps behind an LDS look-up each, and a 32-frame batch of the smallest 1080p level is 49 k rows:
// 192 workgroups, one wave per SIMD, nothing to hide the latencies behind (39 us per batch).  Here lane s of a row runs forward sum
// s (entry s / 2, real or imaginary part: the in-order float64 sum over the window, as before), the eight partial results meet in
// LDS, and every lane runs the inverse sums of the samples t = s, s + 8, ... (each sample's sum over the entries in ascending order,
// as before).  Same values, eight times the threads.
constexpr int kThin8Bins = 4, kThin8Rows = 32, kThin8MaxN = 256;   // windows up to 256 frames (NMAX = 128 / 256: 21 / 40 KB of LDS per workgroup;
                                                                    // with 21 KB the 1536 workgroups of a 32-frame 1080p batch are resident at once)
template <int NMAX>
__global__ __launch_bounds__(256) void k_col_dft_thin8(const float* __restrict__ win, int slot0, int n, int cap,
                                                       int rows_per_stream, int live_per_stream,
                                                       const double* __restrict__ tw, float* __restrict__ col1, MinMax* mm,
                                                       ThinBins eb) {
    const int sub = threadIdx.x & 7, lr = threadIdx.x >> 3;
    const int r = blockIdx.x * kThin8Rows + lr, b = blockIdx.y, fr = blockIdx.z;
    slot0 += fr; if (slot0 >= cap) slot0 -= cap;
    col1 += (size_t)fr * gridDim.y * rows_per_stream;
    mm += fr * gridDim.y;
    const bool live = r < live_per_stream;
    const size_t grow = (size_t)b * rows_per_stream + (live ? r : 0);
    __shared__ double s_tw[2 * NMAX];
    __shared__ float s_row[kThin8Rows][NMAX + 1];
    __shared__ float s_sum[kThin8Rows][8];
    for (int i = threadIdx.x; i < 2 * n; i += 256) s_tw[i] = tw[i];
    {
        const float* wr = win + grow * cap;
        for (int t0 = 0; t0 < n; t0 += 64) {                       // eight loads in flight per lane (a rolled loop serialises the round trips)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 8 * u + sub;
                int slot = slot0 + (t < n ? t : n - 1); if (slot >= cap) slot -= cap;
                v[u] = wr[slot];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int t = t0 + 8 * u + sub; if (t < n) s_row[lr][t] = v[u]; }
        }
    }
    __syncthreads();
    const double* cs = s_tw;
    const double* sn = s_tw + n;
    const bool pow2 = (n & (n - 1)) == 0;
    const double inv_n = 1.0 / (double)n;
    {   // forward: the 2 ne sums of the 32 rows on the first 64 ne threads -- ne whole waves run the n-step chains, the others wait at
        // the barrier without issuing (dft + the 1 / n of DFT_SCALE)
        const int nch = 2 * eb.ne;
        if ((int)threadIdx.x < kThin8Rows * nch) {
            const int frow = threadIdx.x / nch, ch = threadIdx.x - frow * nch;
            const int bin = eb.bin[ch >> 1];
            const bool im = (ch & 1) != 0;
            double acc = 0;
            int idx = 0;
            // eight products at a time (their LDS reads and multiplies do not depend on the running sum), then the eight in-order adds
            for (int t0 = 0; t0 < n; t0 += 8) {
                double pr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int t = t0 + u < n ? t0 + u : n - 1;
                    const double v = (double)s_row[frow][t];
                    pr[u] = im ? -v * sn[idx] : v * cs[idx];
                    idx += bin; if (idx >= n) idx -= n;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (t0 + u < n) acc += pr[u];
            }
            s_sum[frow][ch] = (float)(pow2 ? acc * inv_n : acc / n);
        }
    }
    __syncthreads();
    float yre[kThin8Bins], yim[kThin8Bins];
#pragma unroll
    for (int j = 0; j < kThin8Bins; ++j) {                                            // mulSpectrums with the packed mask
        yre[j] = yim[j] = 0.f;
        if (j >= eb.ne) continue;
        const float a = s_sum[lr][2 * j], bq = s_sum[lr][2 * j + 1];
        if (eb.kind[j] == 0) { yre[j] = a * eb.ma[j] - bq * eb.mb[j]; yim[j] = bq * eb.ma[j] + a * eb.mb[j]; }
        else yre[j] = a * eb.ma[j];
    }
    const bool has_dc = eb.ne > 0 && eb.kind[0] == 1;
    const bool has_ny = eb.ne > 0 && eb.kind[eb.ne - 1] == 2;
    float yny = 0.f;
#pragma unroll
    for (int j = 0; j < kThin8Bins; ++j) if (has_ny && j == eb.ne - 1) yny = yre[j];
    float vmin = INFINITY, vmax = -INFINITY;
    if (live) {
        int idx[kThin8Bins], step[kThin8Bins];
#pragma unroll
        for (int j = 0; j < kThin8Bins; ++j) {
            const int bin = j < eb.ne ? eb.bin[j] : 0;
            idx[j] = (int)(((unsigned)bin * (unsigned)sub) % (unsigned)n);
            step[j] = (int)(((unsigned)bin * 8u) % (unsigned)n);
        }
#pragma unroll 4
        for (int t = sub; t < n; t += 8) {                                            // idft of the samples t = sub (mod 8)
            double acc = has_dc ? yre[0] : 0.f;
#pragma unroll
            for (int j = 0; j < kThin8Bins; ++j) {
                if (j >= eb.ne) break;
                if (eb.kind[j] != 0) continue;
                acc += 2.0 * ((double)yre[j] * cs[idx[j]] - (double)yim[j] * sn[idx[j]]);
                idx[j] += step[j]; if (idx[j] >= n) idx[j] -= n;
            }
            if (n % 2 == 0) acc += (t % 2 ? -1.0 : 1.0) * (double)yny;
            const float v = (float)(pow2 ? acc * inv_n : acc / n);
            vmin = v < vmin ? v : vmin; vmax = v > vmax ? v : vmax;
            if (t == 1) col1[grow] = v;                                               // MagnifyCore.hpp:190-192
        }
    }
    block_minmax(vmin, vmax, mm[b].mn1, mm[b].mx1);
}

// Serial fallback for very long windows (n > kDftMaxN): one thread per row, same arithmetic.
__global__ __launch_bounds__(256) void k_col_dft_serial(const float* __restrict__ win, int slot0, int n, int cap,
                                                        int rows_per_stream, int live_per_stream, double fl, double fh,
                                                        const double* __restrict__ tw, float* __restrict__ Y, float* __restrict__ col1,
                                                        MinMax* mm) {
    const int r = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    const bool live = r < live_per_stream;
    const size_t grow = (size_t)b * rows_per_stream + (live ? r : 0);
    const double* cs = tw;
    const double* sn = tw + n;
    const int half = (n - 1) / 2;
    auto m = [&](int x) { return (x >= fl && x <= fh) ? 1.0f : 0.0f; };
    float vmin = INFINITY, vmax = -INFINITY;
    if (live) {
        const float* wr = win + grow * cap;
        float* Yr = Y + grow * cap;
        auto dot = [&](int bin, bool im) {
            double acc = 0;
            int idx = 0, slot = slot0;
            for (int t = 0; t < n; ++t) {
                const double v = (double)wr[slot];
                acc += im ? -v * sn[idx] : v * cs[idx];
                idx += bin; if (idx >= n) idx -= n;
                if (++slot >= cap) slot -= cap;
            }
            return (float)(acc / n);
        };
        const float m0 = m(0);
        const float y0 = m0 != 0.f ? dot(0, false) * m0 : 0.f;
        for (int k = 1; k <= half; ++k) {
            const float ma = m(2 * k - 1), mb = m(2 * k);
            if (ma == 0.f && mb == 0.f) continue;
            const float a = dot(k, false), bq = dot(k, true);
            Yr[2 * k - 1] = a * ma - bq * mb;
            Yr[2 * k] = bq * ma + a * mb;
        }
        float yl = 0.f;
        const float ml = (n % 2 == 0) ? m(n - 1) : 0.f;
        if (ml != 0.f) yl = dot(n / 2, false) * ml;
        for (int t = 0; t < n; ++t) {
            double acc = y0;
            for (int k = 1; k <= half; ++k) {
                if (m(2 * k - 1) == 0.f && m(2 * k) == 0.f) continue;
                const int idx = (int)(((unsigned long long)k * (unsigned long long)t) % (unsigned long long)n);
                acc += 2.0 * ((double)Yr[2 * k - 1] * cs[idx] - (double)Yr[2 * k] * sn[idx]);
            }
            if (n % 2 == 0) acc += (t % 2 ? -1.0 : 1.0) * (double)yl;
            const float v = (float)(acc / n);
            vmin = v < vmin ? v : vmin; vmax = v > vmax ? v : vmax;
            if (t == 1) col1[grow] = v;
        }
    }
    block_minmax(vmin, vmax, mm[b].mn1, mm[b].mx1);
}

// normalize(0, 1, NORM_MINMAX) (TemporalFilter.cpp:55) of column 1, x amplification (MagnifyCore.hpp:185)
// grid (row blocks, streams * frames)
__global__ __launch_bounds__(256) void k_col_norm(const float* __restrict__ col1, float* __restrict__ up0, int rows,
                                                  int rows_per_stream, const MinMax* mm, float amp) {
    const int r = blockIdx.x * 256 + threadIdx.x, z = blockIdx.y;
    if (r >= rows) return;
    col1 += (size_t)z * rows_per_stream; up0 += (size_t)z * rows;
    const double mn = (double)mm_min(mm[z].mn1), mx = (double)mm_max(mm[z].mx1);
    const double scale = (mx - mn > 2.220446049250313e-16) ? 1. / (mx - mn) : 0.;
    const double shift = 0. - mn * scale;
    const float fs = (float)scale, fsh = (float)shift;
    up0[r] = (col1[r] * fs + fsh) * amp;
}

// k_col_norm and the FIRST pyrUp of the up chain in one launch: k_pyr_up_rows' blocks of 4 x 2 outputs (pyramid.h) whose source values
// are normalised on the fly -- (col1 * fs + fsh) * amp, the same three operations per value -- instead of read from a normalised copy.
// One dependent launch less per batch (6 us of launch + latency for 49 k values); plane = (frame * streams + stream) * C + channel.
__global__ __launch_bounds__(256) void k_pyr_up_rows_norm(const float* __restrict__ col1, int rows_per_stream, int C, int sw, int sh,
                                                          float* __restrict__ dst, int ngroups, const MinMax* mm, float amp) {
    __shared__ float s_sc[2];
    const int plane = blockIdx.y, z = plane / C, ch = plane - z * C;
    if (threadIdx.x == 0) {
        const double mn = (double)mm_min(mm[z].mn1), mx = (double)mm_max(mm[z].mx1);
        const double scale = (mx - mn > 2.220446049250313e-16) ? 1. / (mx - mn) : 0.;
        const double shift = 0. - mn * scale;
        s_sc[0] = (float)scale; s_sc[1] = (float)shift;
    }
    __syncthreads();
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    const float fs = s_sc[0], fsh = s_sc[1];
    const int dw = 2 * sw, gw = dw >> 2;
    const int j = gi / gw, gx = (gi - j * gw) * 4;
    const float* sp = col1 + (size_t)z * rows_per_stream + (size_t)ch * sw * sh;
    float* dp = dst + (size_t)plane * (size_t)dw * (2 * sh);
    const int i0 = gx >> 1;
    const int cm1 = i0 > 0 ? i0 - 1 : 0, cp1 = i0 + 1 < sw ? i0 + 1 : sw - 1, cp2 = i0 + 2 < sw ? i0 + 2 : sw - 1;
    const bool f0 = i0 == 0, l0 = i0 == sw - 1, l1 = i0 + 1 == sw - 1;
    float h[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        int sy = j - 1 + q; sy = sy < 0 ? 1 : (sy >= sh ? sh - 1 : sy);
        const float* r = sp + (size_t)sy * sw;
        const float sm1 = (r[cm1] * fs + fsh) * amp, s0 = (r[i0] * fs + fsh) * amp, s1 = (r[cp1] * fs + fsh) * amp, s2 = (r[cp2] * fs + fsh) * amp;
        h[q][0] = sel(f0, s0 * 6.f + s1 * 2.f, sel(l0, sm1 + s0 * 7.f, sm1 + s0 * 6.f + s1));
        h[q][1] = sel(l0, s0 * 8.f, (s0 + s1) * 4.f);
        h[q][2] = sel(l1, s0 + s1 * 7.f, s0 + s1 * 6.f + s2);
        h[q][3] = sel(l1, s1 * 8.f, (s1 + s2) * 4.f);
    }
    float e[4], o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e[k] = (h[0][k] + h[1][k] * 6.f + h[2][k]) * (1.f / 64.f);
        o[k] = ((h[1][k] + h[2][k]) * 4.f) * (1.f / 64.f);
    }
    float* d0 = dp + (size_t)(2 * j) * dw + gx;
    *reinterpret_cast<float4*>(d0) = make_float4(e[0], e[1], e[2], e[3]);
    *reinterpret_cast<float4*>(d0 + dw) = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- last pyrUp + resize(INTER_LINEAR) + input add (SpatialFilter.cpp:45-48, MagnifyCore.hpp:197-203)
constexpr int CT_W = 64, CT_H = 16;                 // output tile
constexpr int CU_W = 100, CU_H = 28;                // max extent of the pyrUp'ed tile (scale < 1.5)
constexpr int CV_W = 54, CV_H = 18;                 // max extent of its source tile

// row slots of k_col_out_strips: the inverse of the vertical resize map with its blend weight, padded so that the kernel's
// look-ahead needs no bounds checks (one s_load_dwordx8 = the four slots of an iteration)
struct __attribute__((aligned(8))) YSlot { int row; float b1; };
struct YSlot4 { YSlot s[4]; };
constexpr int kYSlotPad = 8, kYSlotTail = 40;
struct OutArgs {
    const uint8_t* in; long in_stride, in_sstride;
    uint8_t* out; long out_stride, out_sstride;
    int w, h;
    const float* V; int vw, vh;                     // source of the last pyrUp (planes); U = 2vw x 2vh
    const float* U2; int w2, h2;                    // k_col_out_strips: source of the pyrUp BEFORE the last one (vw = 2 w2, vh = 2 h2); V is not read
    const int* xofs; const float* xa; const int* yofs; const float* ya;
    const struct YSlot* yslot;                      // k_col_out_strips: entry kYSlotPad + s = {the output row gy with yofs[gy] == s or -1, ya[gy]}
    MinMax* mm;
    int tiles_x, tiles_y;
    float* dbg;
};

template <int C, bool WRITE>
__global__ __launch_bounds__(256) void k_col_out(OutArgs a) {
    __shared__ float sv[CV_H][CV_W + 1];
    __shared__ float hv[CV_H][CU_W + 1];
    __shared__ float su[CU_H][CU_W + 1];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
    const int xe = (x0 + CT_W < a.w ? x0 + CT_W : a.w) - 1, ye = (y0 + CT_H < a.h ? y0 + CT_H : a.h) - 1;
    const int uw = 2 * a.vw, uh = 2 * a.vh;
    // U-tile extent needed by the bilinear taps of this output tile
    const int ux0 = a.xofs[x0], uy0 = a.yofs[y0];
    int ux1 = a.xofs[xe] + 1, uy1 = a.yofs[ye] + 1;
    ux1 = ux1 < uw ? ux1 : uw - 1; uy1 = uy1 < uh ? uy1 : uh - 1;
    const int nux = ux1 - ux0 + 1, nuy = uy1 - uy0 + 1;
    // V-tile extent needed by pyrUp for those U pixels
    const int vx0 = (ux0 >> 1) - 1, vy0 = (uy0 >> 1) - 1;
    const int nvx = (ux1 >> 1) + 1 - vx0 + 1, nvy = (uy1 >> 1) + 1 - vy0 + 1;
    float val[4][C];
    for (int c = 0; c < C; ++c) {
        const float* V = a.V + ((size_t)b * C + c) * a.vw * a.vh;
        __syncthreads();
        for (int i = threadIdx.x; i < nvy * nvx; i += 256) {
            const int ly = i / nvx, lx = i - ly * nvx;
            int gy = vy0 + ly; gy = gy < 0 ? 1 : (gy >= a.vh ? a.vh - 1 : gy);
            int gx = vx0 + lx; gx = gx < 0 ? 0 : (gx >= a.vw ? a.vw - 1 : gx);
            sv[ly][lx] = V[(size_t)gy * a.vw + gx];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nvy * nux; i += 256) {
            const int ly = i / nux, x = i - ly * nux;
            hv[ly][x] = pyrup_h(&sv[ly][0], ux0 + x, vx0, a.vw);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nuy * nux; i += 256) {
            const int y = i / nux, x = i - y * nux;
            const int gy = uy0 + y, lj = (gy >> 1) - vy0;
            su[y][x] = ((gy & 1) == 0) ? (hv[lj - 1][x] + hv[lj][x] * 6.f + hv[lj + 1][x]) * (1.f / 64.f)
                                       : ((hv[lj][x] + hv[lj + 1][x]) * 4.f) * (1.f / 64.f);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + k * 256;
            const int y = i / CT_W, x = i - y * CT_W;
            const int gx = x0 + x, gy = y0 + y;
            float v = 0.f;
            if (gx < a.w && gy < a.h) {
                // resize INTER_LINEAR: horizontal then vertical, D = S0*b0 + S1*b1
                const int sx0 = a.xofs[gx] - ux0, sx1 = (a.xofs[gx] + 1 < uw ? a.xofs[gx] + 1 : uw - 1) - ux0;
                const int sy0 = a.yofs[gy] - uy0, sy1 = (a.yofs[gy] + 1 < uh ? a.yofs[gy] + 1 : uh - 1) - uy0;
                const float a1 = a.xa[gx], a0 = 1.f - a1, b1 = a.ya[gy], b0 = 1.f - b1;
                const float h0 = su[sy0][sx0] * a0 + su[sy0][sx1] * a1;
                const float h1 = su[sy1][sx0] * a0 + su[sy1][sx1] * a1;
                v = h0 * b0 + h1 * b1;
            }
            val[k][c] = v;
        }
    }
    float vmin = INFINITY, vmax = -INFINITY;
    double mn = 0, mx = 0;
    float osc = 0.f, osh = 0.f;
    if (WRITE) {   // convertTo(CV_8U, 255/(max-min), -min*255/(max-min)) (MagnifyCore.hpp:202)
        mn = (double)mm_min(a.mm[b].mn2); mx = (double)mm_max(a.mm[b].mx2);
        osc = (float)(255.0 / (mx - mn)); osh = (float)(-mn * 255.0 / (mx - mn));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + k * 256;
        const int y = i / CT_W, x = i - y * CT_W;
        const int gx = x0 + x, gy = y0 + y;
        if (gx >= a.w || gy >= a.h) continue;
        const uint8_t* p = a.in + (size_t)b * a.in_sstride + (size_t)gy * a.in_stride + (size_t)gx * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float o = (float)p[c] * 1.0f + val[k][c];                 // :169 (unscaled input), :197
            if (WRITE) {
                a.out[(size_t)b * a.out_sstride + (size_t)gy * a.out_stride + (size_t)gx * C + c] = sat_u8(o * osc + osh);
                if (a.dbg && b == 0) a.dbg[((size_t)gy * a.w + gx) * C + c] = o;
            } else {
                vmin = o < vmin ? o : vmin; vmax = o > vmax ? o : vmax;
            }
        }
    }
    if (!WRITE) block_minmax(vmin, vmax, a.mm[b].mn2, a.mm[b].mx2);
}

// Vectorised variant for the common geometry: 3 channels, dword-aligned 4-pixel groups and a width
// the up-chain reproduces exactly (w == 2 * vw, so the horizontal resize taps are the identity).
// The pyrUp horizontal pass reads V straight from global memory (4 floats per 4 outputs), its result
// is the only LDS tile; the vertical pyrUp pass and the vertical bilinear taps are evaluated per pixel.
constexpr int CO_ROWS = 18;            // source rows of V a 16-row output tile can touch (scale < 1.5)
template <bool WRITE>
__global__ __launch_bounds__(256) void k_col_out_v4(OutArgs a) {
    __shared__ __attribute__((aligned(16))) float hv[3][CO_ROWS][CT_W];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
    const int ye = (y0 + CT_H < a.h ? y0 + CT_H : a.h) - 1;
    const int uh = 2 * a.vh;
    const int uy0 = a.yofs[y0];
    int uy1 = a.yofs[ye] + 1; uy1 = uy1 < uh ? uy1 : uh - 1;
    const int vy0 = (uy0 >> 1) - 1;
    const int nvy = (uy1 >> 1) + 1 - vy0 + 1;
    for (int i = threadIdx.x; i < 3 * nvy * 16; i += 256) {
        const int c = i / (nvy * 16), rem = i - c * (nvy * 16);
        const int ly = rem >> 4, g = rem & 15;
        const int gx0 = x0 + 4 * g;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gx0 < a.w) {
            int gy = vy0 + ly; gy = gy < 0 ? 1 : (gy >= a.vh ? a.vh - 1 : gy);
            const float* row = a.V + ((size_t)b * 3 + c) * ((size_t)a.vw * a.vh) + (size_t)gy * a.vw;
            const int i0 = gx0 >> 1;
            const float sm1 = row[i0 > 0 ? i0 - 1 : 0], s0 = row[i0], s1 = row[i0 + 1 < a.vw ? i0 + 1 : a.vw - 1],
                        s2 = row[i0 + 2 < a.vw ? i0 + 2 : a.vw - 1];
            o.x = (i0 == 0) ? s0 * 6.f + s1 * 2.f : ((i0 == a.vw - 1) ? sm1 + s0 * 7.f : sm1 + s0 * 6.f + s1);
            o.y = (i0 == a.vw - 1) ? s0 * 8.f : (s0 + s1) * 4.f;
            o.z = (i0 + 1 == a.vw - 1) ? s0 + s1 * 7.f : s0 + s1 * 6.f + s2;
            o.w = (i0 + 1 == a.vw - 1) ? s1 * 8.f : (s1 + s2) * 4.f;
        }
        *reinterpret_cast<float4*>(&hv[c][ly][4 * g]) = o;
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4, xg = threadIdx.x & 15;
    const int gx = x0 + 4 * xg, gy = y0 + ty;
    float vmin = INFINITY, vmax = -INFINITY;
    if (gx < a.w && gy < a.h) {
        const int sy0 = a.yofs[gy], sy1 = sy0 + 1 < uh ? sy0 + 1 : uh - 1;
        const float b1 = a.ya[gy], b0 = 1.f - b1;
        auto urow = [&](int c, int uy) {       // pyrUp vertical pass for U row uy, 4 columns
            const int lj = (uy >> 1) - vy0;
            const float4 r0 = *reinterpret_cast<const float4*>(&hv[c][lj - 1][4 * xg]);
            const float4 r1 = *reinterpret_cast<const float4*>(&hv[c][lj][4 * xg]);
            const float4 r2 = *reinterpret_cast<const float4*>(&hv[c][lj + 1][4 * xg]);
            float4 u;
            if ((uy & 1) == 0) {
                u.x = (r0.x + r1.x * 6.f + r2.x) * (1.f / 64.f); u.y = (r0.y + r1.y * 6.f + r2.y) * (1.f / 64.f);
                u.z = (r0.z + r1.z * 6.f + r2.z) * (1.f / 64.f); u.w = (r0.w + r1.w * 6.f + r2.w) * (1.f / 64.f);
            } else {
                u.x = ((r1.x + r2.x) * 4.f) * (1.f / 64.f); u.y = ((r1.y + r2.y) * 4.f) * (1.f / 64.f);
                u.z = ((r1.z + r2.z) * 4.f) * (1.f / 64.f); u.w = ((r1.w + r2.w) * 4.f) * (1.f / 64.f);
            }
            return u;
        };
        float val[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float4 u0 = urow(c, sy0), u1 = urow(c, sy1);
            // resize INTER_LINEAR: horizontal taps are (1, 0) here: h = S*1 + S'*0; vertical D = S0*b0 + S1*b1
            val[c][0] = u0.x * b0 + u1.x * b1; val[c][1] = u0.y * b0 + u1.y * b1;
            val[c][2] = u0.z * b0 + u1.z * b1; val[c][3] = u0.w * b0 + u1.w * b1;
        }
        const Px4 pin = *reinterpret_cast<const Px4*>(a.in + (size_t)b * a.in_sstride + (size_t)gy * a.in_stride + (size_t)gx * 3);
        int Bv[4], Gv[4], Rv[4];
        unpack_px4(pin, Bv, Gv, Rv);
        float osc = 0.f, osh = 0.f;
        if (WRITE) {   // convertTo(CV_8U, 255/(max-min), -min*255/(max-min)) (MagnifyCore.hpp:202)
            const double mn = (double)mm_min(a.mm[b].mn2), mx = (double)mm_max(a.mm[b].mx2);
            osc = (float)(255.0 / (mx - mn)); osh = (float)(-mn * 255.0 / (mx - mn));
        }
        uint32_t ob[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float o0 = (float)Bv[k] * 1.0f + val[0][k], o1 = (float)Gv[k] * 1.0f + val[1][k], o2 = (float)Rv[k] * 1.0f + val[2][k];
            if (WRITE) {
                ob[3 * k] = sat_u8(o0 * osc + osh); ob[3 * k + 1] = sat_u8(o1 * osc + osh); ob[3 * k + 2] = sat_u8(o2 * osc + osh);
                if (a.dbg && b == 0) { float* d = a.dbg + ((size_t)gy * a.w + gx + k) * 3; d[0] = o0; d[1] = o1; d[2] = o2; }
            } else {
                vmin = fminf(vmin, fminf(o0, fminf(o1, o2))); vmax = fmaxf(vmax, fmaxf(o0, fmaxf(o1, o2)));
            }
        }
        if (WRITE) {
            Px4 q;
            q.a = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
            q.b = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
            q.c = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
            *reinterpret_cast<Px4*>(a.out + (size_t)b * a.out_sstride + (size_t)gy * a.out_stride + (size_t)gx * 3) = q;
        }
    }
    if (!WRITE) block_minmax(vmin, vmax, a.mm[b].mn2, a.mm[b].mx2);
}

// Barrier-free strip form of k_col_out_v4 (same preconditions, same arithmetic).  Every wave owns a
// strip of 256 x `rows` output pixels, one lane per group of 4 pixels.  Output row gy blends the two
// pyrUp rows sy0 = yofs[gy] and sy0 + 1, and those two rows need exactly three consecutive rows of the
// horizontal pass of V starting at js = (sy0 - 1) >> 1: the lane computes that horizontal pass straight
// from global memory and keeps the three rows in registers, advancing the window as js grows (js and
// the row parity are uniform across the wave: no divergence, no LDS, no barrier until the final
// min/max reduction of the workgroup).
struct HRow3 { float4 c[3]; };
template <bool WRITE, bool DBG>     // DBG: also store the float frame (compile time: no per-pixel branch in the production kernel)
__global__ __launch_bounds__(256) void k_col_out_rows(OutArgs a, int strips_x, int strips_y, int ntasks, int rows) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + wave;
    float vmin = INFINITY, vmax = -INFINITY;
    int b = 0;
    if (task < ntasks) {
        b = task / (strips_x * strips_y);
        const int r = task - b * (strips_x * strips_y);
        const int ty = r / strips_x, tx = r - ty * strips_x;
        const int gx = tx * 256 + 4 * lane, y0 = ty * rows;
        if (gx < a.w) {
            const int uh = 2 * a.vh;
            const uint8_t* src = a.in + (size_t)b * a.in_sstride;
            uint8_t* dst = a.out + (size_t)b * a.out_sstride;
            const unsigned xoff = (unsigned)gx * 3u;
            const float* pl = a.V + (size_t)b * 3 * ((size_t)a.vw * a.vh);
            const size_t pstride = (size_t)a.vw * a.vh;
            const int i0 = gx >> 1;
            const unsigned cm1 = 4u * (i0 > 0 ? i0 - 1 : 0), c00 = 4u * i0, cp1 = 4u * (i0 + 1 < a.vw ? i0 + 1 : a.vw - 1),
                           cp2 = 4u * (i0 + 2 < a.vw ? i0 + 2 : a.vw - 1);
            auto hrow = [&](int vy) __attribute__((always_inline)) {       // horizontal pyrUp pass of V row vy (border map -1 -> 1, vh -> vh - 1)
                HRow3 o;
                vy = vy < 0 ? 1 : (vy >= a.vh ? a.vh - 1 : vy);
                const char* row = reinterpret_cast<const char*>(pl + (size_t)vy * a.vw);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const char* rc = row + c * pstride * sizeof(float);
                    const float sm1 = *reinterpret_cast<const float*>(rc + cm1), s0 = *reinterpret_cast<const float*>(rc + c00),
                                s1 = *reinterpret_cast<const float*>(rc + cp1), s2 = *reinterpret_cast<const float*>(rc + cp2);
                    const bool f0 = i0 == 0, l0 = i0 == a.vw - 1, l1 = i0 + 1 == a.vw - 1;
                    o.c[c].x = sel(f0, s0 * 6.f + s1 * 2.f, sel(l0, sm1 + s0 * 7.f, sm1 + s0 * 6.f + s1));
                    o.c[c].y = sel(l0, s0 * 8.f, (s0 + s1) * 4.f);
                    o.c[c].z = sel(l1, s0 + s1 * 7.f, s0 + s1 * 6.f + s2);
                    o.c[c].w = sel(l1, s1 * 8.f, (s1 + s2) * 4.f);
                }
                return o;
            };
            float osc = 0.f, osh = 0.f;
            if (WRITE) {   // convertTo(CV_8U, 255/(max-min), -min*255/(max-min)) (MagnifyCore.hpp:202)
                const double mn = (double)mm_min(a.mm[b].mn2), mx = (double)mm_max(a.mm[b].mx2);
                osc = (float)(255.0 / (mx - mn)); osh = (float)(-mn * 255.0 / (mx - mn));
            }
            const int yend = y0 + rows < a.h ? y0 + rows : a.h;
            int js = (a.yofs[y0] - 1) >> 1;
            HRow3 A = hrow(js), B = hrow(js + 1), C = hrow(js + 2);
            // the input pixels of the NEXT row are fetched before the current row's arithmetic (the Laplace first kernel's fix
            // for the same per-row dependent load): a wave keeps a row of loads in flight
            Px4 pnext = *reinterpret_cast<const Px4*>(src + (size_t)y0 * a.in_stride + xoff);
            for (int gy = y0; gy < yend; ++gy) {
                const int sy0 = a.yofs[gy];
                const float b1 = a.ya[gy], b0 = 1.f - b1;
                const bool clamp1 = sy0 + 1 > uh - 1;                // sy1 = sy0 (last row)
                const int jn = (sy0 - 1) >> 1;
                const Px4 pin = pnext;
                if (gy + 1 < yend) pnext = *reinterpret_cast<const Px4*>(src + (size_t)(gy + 1) * a.in_stride + xoff);
                while (js < jn) { A = B; B = C; C = hrow(js + 3); ++js; }
                float val[3][4];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* pa = &A.c[c].x; const float* pb = &B.c[c].x; const float* pc = &C.c[c].x;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float u0, u1;
                        if ((sy0 & 1) == 0) {             // rows 2j (window j-1, j, j+1) and 2j+1 (j, j+1)
                            u0 = (pa[k] + pb[k] * 6.f + pc[k]) * (1.f / 64.f);
                            u1 = ((pb[k] + pc[k]) * 4.f) * (1.f / 64.f);
                        } else {                          // rows 2j+1 (window j, j+1) and 2j+2 (j, j+1, j+2)
                            u0 = ((pa[k] + pb[k]) * 4.f) * (1.f / 64.f);
                            u1 = (pa[k] + pb[k] * 6.f + pc[k]) * (1.f / 64.f);
                        }
                        if (clamp1) u1 = u0;
                        // resize INTER_LINEAR: horizontal taps are (1, 0) here; vertical D = S0*b0 + S1*b1
                        val[c][k] = u0 * b0 + u1 * b1;
                    }
                }
                int Bv[4], Gv[4], Rv[4];
                unpack_px4(pin, Bv, Gv, Rv);
                float ov[12];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float o0 = (float)Bv[k] * 1.0f + val[0][k], o1 = (float)Gv[k] * 1.0f + val[1][k], o2 = (float)Rv[k] * 1.0f + val[2][k];
                    if (WRITE) {
                        ov[3 * k] = o0 * osc + osh; ov[3 * k + 1] = o1 * osc + osh; ov[3 * k + 2] = o2 * osc + osh;
                        if (DBG && a.dbg && b == 0) { float* d = a.dbg + ((size_t)gy * a.w + gx + k) * 3; d[0] = o0; d[1] = o1; d[2] = o2; }
                    } else {
                        vmin = fminf(vmin, fminf(o0, fminf(o1, o2))); vmax = fmaxf(vmax, fmaxf(o0, fmaxf(o1, o2)));
                    }
                }
                if (WRITE) {
                    Px4 q;
                    q.a = pack_u8x4(ov[0], ov[1], ov[2], ov[3]); q.b = pack_u8x4(ov[4], ov[5], ov[6], ov[7]); q.c = pack_u8x4(ov[8], ov[9], ov[10], ov[11]);
                    *reinterpret_cast<Px4*>(dst + (size_t)gy * a.out_stride + xoff) = q;
                }
            }
        }
    }
    if (!WRITE) {
        // all four waves of a workgroup belong to the same frame when strips_x * strips_y is a multiple of 4 or
        // the workgroup does not straddle a frame boundary; otherwise reduce per wave into that wave's frame
        block_minmax_waves(vmin, vmax, a.mm, b, task < ntasks);
    }
}

// ---- strip output kernels, second form (round 3): both pyrUps inside, packed FP32, loads a whole iteration ahead -------------
// What the measurements of k_col_out_rows said (1080p, 32 frames per launch, min/max pass):
//   * its ISA: ~290 vector + ~150 scalar instructions and ~36 scalar branches per row of 4-pixel groups;
//   * with the loads replaced by constants it takes 63 us, the loads alone need ~80 us (400 MB) -- and the kernel takes 140:
//     compute and memory time ADD UP.  The loop fetched one row ahead and every wait in it was s_waitcnt vmcnt(0): the window
//     advance was a data-dependent `while` around loads (unknown number of loads in flight -> the compiler drains them all), and
//     with 4-5 resident waves the bytes in flight per SIMD (~5 KB) were a third of what 5 TB/s x 2-3 us of loaded latency asks for;
//   * a FIFO of prefetch registers does not help: the v_mov that shifts it is a use of the register in flight.
// This form therefore
//   * makes the V rows (source of the last pyrUp) from three register-resident rows of the level-2 image U2 instead of loading
//     them (k_pyr_up_rows' arithmetic and order): the last generic pyrUp launch and its 25 MB per frame written once and read
//     twice are gone, and the only bulk loads left are the input rows;
//   * walks WINDOW POSITIONS: position js (horizontal passes of V rows js .. js + 2 in A, B, C) serves the output rows whose
//     first source row sy0 = yofs[gy] is 2 js + 1 (rows O(A, B), E(A, B, C) of the up-sampled image) or 2 js + 2 (E(A, B, C),
//     O(B, C)): at most one each, the row map being strictly increasing (host-checked); yinv[s] is that row.  One iteration =
//     two positions = four row slots, and its loads -- the four input rows and the one U2 row of the NEXT iteration -- are
//     issued at its top, unconditionally (a slot without a row re-reads row y0), into registers that were copied out just
//     before: the wait at the top of an iteration is legitimately "everything issued one iteration ago", and the loads fly for
//     a whole iteration (four rows of arithmetic x the resident waves: 2-3 us);
//   * keeps a lane's 4 outputs per channel as two float pairs, (x, z) = the even pyrUp columns (three-term formula) and
//     (y, w) = the odd ones (two-term formula): every step is one v_pk_{add,mul}_f32;
//   * evaluates every row of the up-sampled image once (O(B, C) of a position is O(A, B) of the next), selects row parity by
//     code position instead of per value, and confines the border formulas to the strips that contain a border lane (i0 and
//     vw are even: only the first and the last 4-pixel group of a row have one; U2: also column vw - 4).
typedef float f2 __attribute__((vector_size(8)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 v = {a, b}; return v; }
__device__ __forceinline__ f2 bc2(float a) { f2 v = {a, a}; return v; }
struct HRowP { f2 xz[3], yw[3]; };                  // horizontal pyrUp pass of one V row at the lane's 4 output columns
struct URowP { f2 xz[3], yw[3]; };                  // one row of the up-sampled image, same layout
struct U2Raw { float um[3], u0[3], u1[3], u2[3]; }; // U2 columns k0 - 1 .. k0 + 2 (clamped) per channel
struct U2H { f2 a[3], b[3]; };                      // U2's horizontal pass at level-1 columns (i0 - 1, i0) and (i0 + 1, i0 + 2)

// The U2 window (three rows of U2's horizontal pass = 36 floats per lane) lives in LDS, not in registers: every lane owns a private
// column of a ring of three row slots (logical U2 row r in slot (r + 3) % 3: no sharing, no barrier, the shift is an index step),
// read twice and written once per iteration through lgkmcnt -- with it in VGPRs the kernel needed 190-200 of them (two waves per
// SIMD, or scratch spills that count on vmcnt and drain the prefetches).
constexpr int kU2RingFloats = 3 * 6 * 2 * 256;      // [slot][a0 a1 a2 b0 b1 b2][x, y][thread]
template <bool WRITE, bool DBG, bool BORDER>
__device__ __forceinline__ void col_out_strip(const OutArgs& a, int b, int tx, int ty, int rows, int lane, float& vmin, float& vmax, f2* ring) {
    const int gx = tx * 256 + 4 * lane, y0 = ty * rows;
    if (gx >= a.w) return;
    const int uh = 2 * a.vh;
    const const_tab<int> yofs = as_const_tab(a.yofs);              // (scalar loads also in the storing pass)
    const const_tab<YSlot> yslot = as_const_tab(a.yslot) + kYSlotPad;
    const uint8_t* src = a.in + (size_t)b * a.in_sstride;
    uint8_t* dst = a.out + (size_t)b * a.out_sstride;
    const unsigned xoff = (unsigned)gx * 3u;
    const int i0 = gx >> 1;                                        // even
    const bool f0 = BORDER && i0 == 0, l1 = BORDER && i0 + 2 == a.vw;
    // k_pyr_up_rows' arithmetic for the lane's four level-1 columns i0 - 1 .. i0 + 2 out of U2 columns k0 - 1 .. k0 + 2
    const int k0 = i0 >> 1;
    // All bulk accesses go through buffer resources built from wave-uniform values (frame base pointers): resource + per-lane 32-bit
    // offset + scalar 32-bit row offset, i.e. no 64-bit address arithmetic in vector registers.  (With generic pointers hipcc 7.2
    // kept the lanes' column offsets as 64-bit VGPR pairs whose zero high halves it re-materialised by copying a register that no
    // longer held zero on one path: memory faults on the GPU -- the faulting address had the right low and a foreign high dword --
    // while the CPU emulation build of the same source was clean under ASan and a guard-page allocator.)
    const unsigned w2b = (unsigned)a.w2 * 4u, pstride2 = w2b * (unsigned)a.h2;
    const BufRsrc ru2 = buf_rsrc(reinterpret_cast<const char*>(a.U2) + (size_t)b * 3 * (size_t)pstride2, 3u * pstride2);
    const unsigned in_stride = (unsigned)a.in_stride, out_stride = (unsigned)a.out_stride;    // (host-checked: positive, frame < 2 GB)
    const BufRsrc rin = buf_rsrc(src, in_stride * (unsigned)(a.h - 1) + (unsigned)a.w * 3u);
    const BufRsrc rout = buf_rsrc(dst, out_stride * (unsigned)(a.h - 1) + (unsigned)a.w * 3u);
    const bool g0 = BORDER && k0 == 0, gl0 = BORDER && k0 == a.w2 - 1, gl1 = BORDER && k0 + 1 == a.w2 - 1;
    const unsigned dm1 = 4u * (unsigned)(k0 > 0 ? k0 - 1 : 0), d00 = 4u * (unsigned)k0, dp1 = 4u * (unsigned)(k0 + 1 < a.w2 ? k0 + 1 : a.w2 - 1),
                   dp2 = 4u * (unsigned)(k0 + 2 < a.w2 ? k0 + 2 : a.w2 - 1);
    auto u2load = [&](int j) __attribute__((always_inline)) {     // vertical border map of pyrUp: row -1 -> 1, row h2 -> h2 - 1
        U2Raw r;
        j = j < 0 ? 1 : (j >= a.h2 ? a.h2 - 1 : j);
        const unsigned row = (unsigned)j * w2b;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned rc = row + c * pstride2;
            r.um[c] = buf_ld_f32(ru2, dm1, rc); r.u0[c] = buf_ld_f32(ru2, d00, rc);
            r.u1[c] = buf_ld_f32(ru2, dp1, rc); r.u2[c] = buf_ld_f32(ru2, dp2, rc);
        }
        return r;
    };
    auto u2h = [&](const U2Raw& r) __attribute__((always_inline)) {
        U2H o;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // (scalar on purpose: as packed operations these need the overlapping pairs (um, u0), (u0, u1), (u1, u2), which the
            // compiler built through a stack slot; 10 plain operations against 5 packed ones + 10 register moves)
            const float um = r.um[c], u0 = r.u0[c], u1 = r.u1[c], u2 = r.u2[c];
            float odd[2] = {(um + u0) * 4.f, (u0 + u1) * 4.f};    // columns i0 - 1 (source k0 - 1) and i0 + 1 (source k0)
            float even[2] = {(um + u0 * 6.f) + u1, (u0 + u1 * 6.f) + u2};   // columns i0 (source k0) and i0 + 2 (source k0 + 1)
            if (BORDER) {
                even[0] = sel(g0, u0 * 6.f + u1 * 2.f, sel(gl0, um + u0 * 7.f, even[0]));
                odd[1] = sel(gl0, u0 * 8.f, odd[1]);
                even[1] = sel(gl1, u0 + u1 * 7.f, even[1]);
            }
            o.a[c] = mk2(odd[0], even[0]); o.b[c] = mk2(odd[1], even[1]);
        }
        return o;
    };
    // ring[(slot * 6 + k) * 256 + thread]: consecutive lanes, consecutive 8-byte words (conflict-free ds_read_b64 / ds_write_b64)
    f2* const mine = ring + threadIdx.x;
    auto ring_put = [&](int r, const U2H& h) __attribute__((always_inline)) {      // logical U2 row r (may be -1 or h2: border-mapped data)
        f2* q = mine + ((r + 3) % 3) * (6 * 256);
#pragma unroll
        for (int c = 0; c < 3; ++c) { q[c * 256] = h.a[c]; q[(3 + c) * 256] = h.b[c]; }
    };
    auto ring_get = [&](int r) __attribute__((always_inline)) {
        U2H h;
        const f2* q = mine + ((r + 3) % 3) * (6 * 256);
#pragma unroll
        for (int c = 0; c < 3; ++c) { h.a[c] = q[c * 256]; h.b[c] = q[(3 + c) * 256]; }
        return h;
    };
    int jcur = -1000;                                              // the ring holds the logical U2 rows jcur - 1, jcur, jcur + 1
    // horizontal pass of V row vy (border-mapped, made from the U2 window, which must stand at vy >> 1)
    auto hrow = [&](int vy) __attribute__((always_inline)) {
        HRowP o;
        const U2H Q = ring_get(jcur), R = ring_get(jcur + 1);
        U2H P{};
        if ((vy & 1) == 0) P = ring_get(jcur - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f2 va, vb;                                             // V columns (i0 - 1, i0) and (i0 + 1, i0 + 2)
            if ((vy & 1) == 0) { va = ((P.a[c] + Q.a[c] * bc2(6.f)) + R.a[c]) * bc2(1.f / 64.f); vb = ((P.b[c] + Q.b[c] * bc2(6.f)) + R.b[c]) * bc2(1.f / 64.f); }
            else { va = (Q.a[c] + R.a[c]) * bc2(1.f / 16.f); vb = (Q.b[c] + R.b[c]) * bc2(1.f / 16.f); }   // ((x * 4) * (1/64): two exact power-of-two scalings = one)
            const f2 c01 = mk2(va[1], vb[0]);
            o.xz[c] = (va + c01 * bc2(6.f)) + vb;                   // x = s[-1] + s0*6 + s1, z = s0 + s1*6 + s2
            o.yw[c] = (c01 + vb) * bc2(4.f);                        // y = (s0 + s1)*4,       w = (s1 + s2)*4
            if (BORDER) {
                const float s0 = va[1], s1 = vb[0];
                o.xz[c][0] = sel(f0, s0 * 6.f + s1 * 2.f, o.xz[c][0]);
                o.xz[c][1] = sel(l1, s0 + s1 * 7.f, o.xz[c][1]);
                o.yw[c][1] = sel(l1, s1 * 8.f, o.yw[c][1]);
            }
        }
        return o;
    };
    auto hrow_init = [&](int vy) __attribute__((always_inline)) { // any row order (strip start): positions the U2 window with its own loads
        vy = vy < 0 ? 1 : (vy >= a.vh ? a.vh - 1 : vy);
        const int jj = vy >> 1;
        if (jj == jcur + 1) ring_put(jj + 1, u2h(u2load(jj + 1)));
        else if (jj != jcur) { ring_put(jj - 1, u2h(u2load(jj - 1))); ring_put(jj, u2h(u2load(jj))); ring_put(jj + 1, u2h(u2load(jj + 1))); }
        jcur = jj;
        return hrow(vy);
    };
    float osc = 0.f, osh = 0.f;
    if (WRITE) {   // convertTo(CV_8U, 255/(max-min), -min*255/(max-min)) (MagnifyCore.hpp:202)
        const double mn = (double)mm_min(a.mm[b].mn2), mx = (double)mm_max(a.mm[b].mx2);
        osc = (float)(255.0 / (mx - mn)); osh = (float)(-mn * 255.0 / (mx - mn));
    }
    const int yend = y0 + rows < a.h ? y0 + rows : a.h;
    int js = (yofs[y0] - 1) >> 1;
    const int jlast = (yofs[yend - 1] - 1) >> 1;
    // The vertical pass keeps, instead of the three H rows (A, B, C) of a window position, the partial sum T = A + B*6 of the even row
    // E = ((A + B*6) + C) / 64 (same operations in the same order) and the rows B, C: the next position's T is B + C*6 and its
    // (B, C) are (C, new row) -- with two positions written out per iteration the two row variables just swap roles and nothing
    // is moved (rotating A, B, C cost 24 of the ~180 vector instructions per output row).
    HRowP X, Y;                                                    // the H rows B, C of the position (roles alternate)
    URowP T;
    URowP O0, O1;
    {
        const HRowP A0 = hrow_init(js);
        X = hrow_init(js + 1); Y = hrow_init(js + 2);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            T.xz[c] = A0.xz[c] + X.xz[c] * bc2(6.f); T.yw[c] = A0.yw[c] + X.yw[c] * bc2(6.f);
            O0.xz[c] = (A0.xz[c] + X.xz[c]) * bc2(1.f / 16.f); O0.yw[c] = (A0.yw[c] + X.yw[c]) * bc2(1.f / 16.f);
        }
    }
    auto odd_row = [&](const HRowP& p, const HRowP& q) __attribute__((always_inline)) {   // row 2 j + 1 from H rows j, j + 1
        URowP u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            u.xz[c] = (p.xz[c] + q.xz[c]) * bc2(1.f / 16.f);      // = ((p + q) * 4) * (1/64): both scalings are exact
            u.yw[c] = (p.yw[c] + q.yw[c]) * bc2(1.f / 16.f);
        }
        return u;
    };
    // output row gy = input + (u0 * b0 + u1 * b1): resize INTER_LINEAR with horizontal taps (1, 0), vertical D = S0*b0 + S1*b1, then
    // the unscaled input (:169, :197)
    auto emit = [&](int gy, float b1, const URowP& u0, const URowP& u1, const B96 pin) __attribute__((always_inline)) {
        const float b0 = 1.f - b1;
        const int Bv[4] = {(int)(pin.a & 255), (int)(pin.a >> 24), (int)((pin.b >> 16) & 255), (int)((pin.c >> 8) & 255)};
        const int Gv[4] = {(int)((pin.a >> 8) & 255), (int)(pin.b & 255), (int)(pin.b >> 24), (int)((pin.c >> 16) & 255)};
        const int Rv[4] = {(int)((pin.a >> 16) & 255), (int)((pin.b >> 8) & 255), (int)(pin.c & 255), (int)(pin.c >> 24)};
        const f2 vb0 = bc2(b0), vb1 = bc2(b1);
        f2 oxz[3], oyw[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int* iv = c == 0 ? Bv : (c == 1 ? Gv : Rv);
            oxz[c] = mk2((float)iv[0], (float)iv[2]) + (u0.xz[c] * vb0 + u1.xz[c] * vb1);
            oyw[c] = mk2((float)iv[1], (float)iv[3]) + (u0.yw[c] * vb0 + u1.yw[c] * vb1);
        }
        if (WRITE) {
            if (DBG && a.dbg && b == 0) {
                float* d = a.dbg + ((size_t)gy * a.w + gx) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) { d[c] = oxz[c][0]; d[3 + c] = oyw[c][0]; d[6 + c] = oxz[c][1]; d[9 + c] = oyw[c][1]; }
            }
            const f2 vsc = bc2(osc), vsh = bc2(osh);
            f2 sxz[3], syw[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { sxz[c] = oxz[c] * vsc + vsh; syw[c] = oyw[c] * vsc + vsh; }
            // pixel k, channel c = byte 3 k + c: k = 0 -> xz[.][0], 1 -> yw[.][0], 2 -> xz[.][1], 3 -> yw[.][1]
            B96 q;
            q.a = pack_u8x4(sxz[0][0], sxz[1][0], sxz[2][0], syw[0][0]);
            q.b = pack_u8x4(syw[1][0], syw[2][0], sxz[0][1], sxz[1][1]);
            q.c = pack_u8x4(sxz[2][1], syw[0][1], syw[1][1], syw[2][1]);
            buf_sts_b96(q, rout, xoff, (unsigned)gy * out_stride);      // streaming store: the output frame is not read again on the device
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                vmin = fminf(vmin, fminf(fminf(oxz[c][0], oxz[c][1]), fminf(oyw[c][0], oyw[c][1])));
                vmax = fmaxf(vmax, fmaxf(fmaxf(oxz[c][0], oxz[c][1]), fmaxf(oyw[c][0], oyw[c][1])));
            }
        }
    };
    // the four row slots 2 j + 1 .. 2 j + 4 of the iteration at window position j (rows outside this strip: -1)
    auto slots_at = [&](int j) __attribute__((always_inline)) {
        YSlot4 q;
        const const_tab<YSlot> p = yslot + (2 * j + 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = p[k].row; q.s[k].row = r >= y0 && r < yend ? r : -1; q.s[k].b1 = p[k].b1;
        }
        return q;
    };
    auto row_load = [&](int r) __attribute__((always_inline)) {    // (a slot without a row re-reads row y0)
        return buf_ld_b96(rin, xoff, (unsigned)(r >= 0 ? r : y0) * in_stride);
    };
    U2Raw u2c{};
    // one window position: its two row slots, then the window advances by V row j + 3 (clamped to vh - 1: the U2 window stands still)
    // (the input row of a slot is re-loaded for the NEXT iteration's slot right after its use: one register set per slot, no copies)
    auto step = [&](int j, const YSlot sO, const YSlot sE, int nO, int nE, B96& pO, B96& pE, const URowP& Oin, URowP& Oout, HRowP& Bm, const HRowP& Cm)
        __attribute__((always_inline)) {
        URowP Ecur;
#pragma unroll
        for (int c = 0; c < 3; ++c) { Ecur.xz[c] = (T.xz[c] + Cm.xz[c]) * bc2(1.f / 64.f); Ecur.yw[c] = (T.yw[c] + Cm.yw[c]) * bc2(1.f / 64.f); }
        Oout = odd_row(Bm, Cm);
        // sy1 = min(sy0 + 1, uh - 1): the last row of the up-sampled image blends with itself (no later row can need the original)
        if (2 * j + 2 > uh - 1) Ecur = Oin;
        if (2 * j + 3 > uh - 1) Oout = Ecur;
        if (sO.row >= 0) emit(sO.row, sO.b1, Oin, Ecur, pO);
        pO = row_load(nO);
        if (sE.row >= 0) emit(sE.row, sE.b1, Ecur, Oout, pE);
        pE = row_load(nE);
#pragma unroll
        for (int c = 0; c < 3; ++c) { T.xz[c] = Bm.xz[c] + Cm.xz[c] * bc2(6.f); T.yw[c] = Bm.yw[c] + Cm.yw[c] * bc2(6.f); }
        const int vy = j + 3 < a.vh ? j + 3 : a.vh - 1;            // (j + 3 >= 2 in the loop)
        if ((vy >> 1) == jcur + 1) { jcur = vy >> 1; ring_put(jcur + 1, u2h(u2c)); }
        Bm = hrow(vy);                                             // the new row takes the place of the oldest one
    };
    // the U2 window shifts once per iteration (V rows js + 3, js + 4 -> js + 2 and js + 4 differ by one U2 row), to U2 row
    // (js + 4) >> 1, and takes in row ((js + 4) >> 1) + 1 = (js + 6) >> 1.  The slot table entries are read one iteration ahead
    // (scalar loads: the rows of the next iteration's slots are needed for the input loads of this one).
    YSlot4 S0, S1 = slots_at(js);
    B96 p0 = row_load(S1.s[0].row), p1 = row_load(S1.s[1].row), p2 = row_load(S1.s[2].row), p3 = row_load(S1.s[3].row);
    U2Raw u2n = u2load((js + 6) >> 1);
    for (; js <= jlast; js += 2) {                                 // (the second position of the last pair may be past the strip: no row, no output)
        u2c = u2n;
        S0 = S1; S1 = slots_at(js + 2);
        u2n = u2load((js + 8) >> 1);
        step(js, S0.s[0], S0.s[1], S1.s[0].row, S1.s[1].row, p0, p1, O0, O1, X, Y);
        step(js + 1, S0.s[2], S0.s[3], S1.s[2].row, S1.s[3].row, p2, p3, O1, O0, Y, X);
    }
}

template <bool WRITE, bool DBG>
__global__ __launch_bounds__(256) void k_col_out_strips(OutArgs a, int strips_x, int strips_y, int ntasks, int rows) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + wave;
    __shared__ __attribute__((aligned(16))) float s_ring[kU2RingFloats];
    f2* ring = reinterpret_cast<f2*>(s_ring);
    float vmin = INFINITY, vmax = -INFINITY;
    int b = 0;
    if (task < ntasks) {
        b = task / (strips_x * strips_y);
        const int r = task - b * (strips_x * strips_y);
        const int ty = r / strips_x, tx = r - ty * strips_x;
        // strips holding a lane with a border formula: the first, the last, and (U2 column vw - 4) the one before a last strip of one group
        if (tx == 0 || 256 * (tx + 1) + 4 >= a.w) col_out_strip<WRITE, DBG, true>(a, b, tx, ty, rows, lane, vmin, vmax, ring);
        else col_out_strip<WRITE, DBG, false>(a, b, tx, ty, rows, lane, vmin, vmax, ring);
    }
    if (!WRITE) wave_minmax(vmin, vmax, a.mm, b, task < ntasks);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct ColorState : ModeState {
    int levels = 0, planes = 0, channels = 0;
    LevelGeom g[kMaxLevels + 1];
    float* arena = nullptr;
    float* G[kMaxLevels + 1] = {};
    float* up[kMaxLevels + 1] = {};
    float* col1 = nullptr;
    int rows = 0, rows_ps = 0;       // padded rows (all streams) and per stream
    float* win = nullptr; float* Y = nullptr; int cap = 0;
    int n = 0, slot0 = 0;            // logical window: n columns starting at ring slot slot0
    int max_images = 0;
    bool yofs_strict = false;        // the vertical resize map yofs[] is strictly increasing (always, for sizes the up chain produces)
    bool co_rows_ok = false;         // the vectorised output kernel's LDS tile covers every output tile
    bool thin_dft = true;            // thread-per-row DFT for narrow bands (LVM_COL_THIN_DFT=0: wave-per-row kernel)
    bool norm_in_up = true;          // normalisation inside the first pyrUp launch (LVM_COL_NORM_IN_UP=0: k_col_norm as its own launch)
    float amp = 0.f;                 // amplification of the current call (col_filter -> col_up_out)
    bool thin8_dft = true;           // ... eight lanes per row for at most four entries and windows up to 256 frames (LVM_COL_THIN8_DFT=0)
    long out_min_tasks = 2048;       // strips are shortened until a launch has this many (LVM_COL_OUT_MIN_TASKS)
    long out_min_tasks_lean = 900;   // ... k_col_out_strips: ~1 wave per SIMD is enough (per-frame 1080p: 960 strips of 9 rows, min/max pass 31 -> 21 us:
                                     // fewer strip start-ups and fewer atomics on the frame's 64 min/max cells)
    bool up_rows = true;             // barrier-free pyrUp of the up chain (LVM_COL_UP_ROWS=0: tiled k_pyr_up)
    bool d0_rows_on = true;          // wave-strip first kernel (LVM_D0_ROWS=0: LDS-tiled k_down0_v4 always)
    bool d01_on = true;              // first TWO pyramid levels in one pass for large launches (LVM_COL_DOWN01=0: level 1 through HBM)
    int d01_rows_forced = 0;         // LVM_COL_DOWN01_ROWS: strip height of k_down01_rows (tests)
    long d0_min_tasks = 4096;        // ... for launches with at least this many strips (LVM_D0_MIN_TASKS)
    int out_rows = 16;               // rows per wave strip of k_col_out_rows (LVM_COL_OUT_ROWS; 0 = tiled k_col_out_v4)
    int out_rows_lean = 36;          // ... of k_col_out_strips (strip start-up = 3 V rows + 3 U2 rows: longer strips; 1080 = 30 x 36)
    bool out_lean = true;            // k_col_out_strips (both pyrUps inside, packed FP32, loads an iteration ahead); LVM_COL_OUT_LEAN=0: k_col_out_rows
    int thin_min_frames = 1;         // ... from this many frames per launch (LVM_COL_THIN_MIN_FRAMES).  Round 5: 4 -> 1 -- the eight-lanes-per-row kernel of round 4 also wins
                                     // for ONE frame per launch (24 -> 16 us per 1080p frame; the wave-per-row kernel remains for wide bands)
    long rows_min_elems = 1 << 20;   // planes x pixels from which pyrDown uses k_pyr_down_rows (LVM_ROWS_MIN_ELEMS)
    double* tw = nullptr; int tw_n = 0;          // table of the current window length (points into tw_all or at tw_own)
    double* tw_all = nullptr; int tw_all_max = 0; std::vector<size_t> tw_off;   // tables of every length 2 .. tw_all_max
    double* tw_own = nullptr;
    MinMax* mm = nullptr;
    int *xofs = nullptr, *yofs = nullptr; float *xa = nullptr, *ya = nullptr; YSlot* yslot = nullptr;
    // temporal batching
    int tcap = 0; float* tarena = nullptr; float* Gt[kMaxLevels + 1] = {}; float* upt[kMaxLevels + 1] = {}; float* col1t = nullptr; MinMax* mmt = nullptr;
    ~ColorState() override {
        void* p[] = {arena, win, Y, tw_all, tw_own, mm, xofs, yofs, yslot, xa, ya, tarena, mmt};
        for (void* q : p) if (q) (void)hipFree(q);
    }
};

static void resize_tab(int d, int s, std::vector<int>& ofs, std::vector<float>& al) {   // cv::resize INTER_LINEAR tables
    const double scale = 1. / ((double)d / s);
    ofs.resize(d); al.resize(d);
    for (int i = 0; i < d; ++i) {
        float f = (float)((i + 0.5) * scale - 0.5);
        int si = (int)std::floor(f);
        f -= (float)si;
        if (si < 0) { f = 0; si = 0; }
        if (si >= s - 1) { f = 0; si = s - 1; }
        ofs[i] = si; al[i] = f;
    }
}

static int color_reserve_frames(Ctx* c, ColorState* st, int nt, hipStream_t s);
static int color_alloc(Ctx* c, ColorState* st, int w, int h, int channels, int levels) {
    st->levels = levels; st->channels = channels; st->planes = c->nstreams * channels;
    st->g[0] = {w, h, (size_t)w * h};
    for (int l = 1; l <= levels; ++l) {
        const int lw = (st->g[l - 1].w + 1) / 2, lh = (st->g[l - 1].h + 1) / 2;
        st->g[l] = {lw, lh, (size_t)lw * lh};
    }
    auto pad = [](size_t n) { return (n + 63) & ~(size_t)63; };
    const size_t nL = st->g[levels].n;
    st->rows_ps = (int)((nL * channels + 255) / 256 * 256);
    st->rows = st->rows_ps * c->nstreams;
    size_t total = 0;
    for (int l = 1; l <= levels; ++l) total += pad(st->g[l].n * st->planes);
    for (int k = 0; k < levels; ++k) total += pad((nL << (2 * k)) * st->planes);
    total += pad((size_t)st->rows);
    if (hipMalloc((void**)&st->arena, total * sizeof(float)) != hipSuccess) { st->arena = nullptr; c->err = "color: hipMalloc failed"; return LVM_ERR_OOM; }
    float* p = st->arena;
    for (int l = 1; l <= levels; ++l) { st->G[l] = p; p += pad(st->g[l].n * st->planes); }
    for (int k = 0; k < levels; ++k) { st->up[k] = p; p += pad((nL << (2 * k)) * st->planes); }
    st->col1 = p;
    // resize tables from (w_L 2^L, h_L 2^L) to (w, h); equal sizes => identity taps (cv::resize copies)
    const int UW = st->g[levels].w << levels, UH = st->g[levels].h << levels;
    std::vector<int> xo, yo; std::vector<float> xa, ya;
    resize_tab(w, UW, xo, xa); resize_tab(h, UH, yo, ya);
    if (UW == w && UH == h) { for (int i = 0; i < w; ++i) { xo[i] = i; xa[i] = 0; } for (int i = 0; i < h; ++i) { yo[i] = i; ya[i] = 0; } }
    LVM_HIP_TRY(c, hipMalloc((void**)&st->xofs, w * sizeof(int))); LVM_HIP_TRY(c, hipMalloc((void**)&st->xa, w * sizeof(float)));
    LVM_HIP_TRY(c, hipMalloc((void**)&st->yofs, h * sizeof(int))); LVM_HIP_TRY(c, hipMalloc((void**)&st->ya, h * sizeof(float)));
    LVM_HIP_TRY(c, hipMemcpy(st->xofs, xo.data(), w * sizeof(int), hipMemcpyHostToDevice));
    LVM_HIP_TRY(c, hipMemcpy(st->xa, xa.data(), w * sizeof(float), hipMemcpyHostToDevice));
    LVM_HIP_TRY(c, hipMemcpy(st->yofs, yo.data(), h * sizeof(int), hipMemcpyHostToDevice));
    LVM_HIP_TRY(c, hipMemcpy(st->ya, ya.data(), h * sizeof(float), hipMemcpyHostToDevice));
    {   // inverse of the row map over the rows of the up chain's last image, with the blend weights (k_col_out_strips)
        std::vector<YSlot> ys((size_t)kYSlotPad + UH + kYSlotTail, YSlot{-1, 0.f});
        for (int i = h - 1; i >= 0; --i) if (yo[i] >= 0 && yo[i] < UH) ys[(size_t)kYSlotPad + yo[i]] = YSlot{i, ya[i]};
        LVM_HIP_TRY(c, hipMalloc((void**)&st->yslot, ys.size() * sizeof(YSlot)));
        LVM_HIP_TRY(c, hipMemcpy(st->yslot, ys.data(), ys.size() * sizeof(YSlot), hipMemcpyHostToDevice));
    }
    LVM_HIP_TRY(c, hipMalloc((void**)&st->mm, sizeof(MinMax) * c->nstreams));
    // the kernel's LDS tiles assume the resize never shrinks by more than 1.5 (true for every size
    // calculateMaxLevels admits); verify the per-tile extents once
    for (int x0 = 0; x0 < w; x0 += CT_W) { const int xe = (x0 + CT_W < w ? x0 + CT_W : w) - 1; if (xo[xe] + 1 - xo[x0] + 1 > CU_W) { c->err = "color: resize tile too wide"; return LVM_ERR_INVALID; } }
    st->yofs_strict = true;                         // k_col_out_strips walks window positions: needs a strictly increasing row map
    for (int i = 1; i < h; ++i) if (yo[i] <= yo[i - 1]) st->yofs_strict = false;
    st->co_rows_ok = true;
    for (int y0 = 0; y0 < h; y0 += CT_H) {
        const int ye = (y0 + CT_H < h ? y0 + CT_H : h) - 1;
        if (yo[ye] + 1 - yo[y0] + 1 > CU_H) { c->err = "color: resize tile too tall"; return LVM_ERR_INVALID; }
        int u1 = yo[ye] + 1; u1 = u1 < UH ? u1 : UH - 1;
        if ((u1 >> 1) + 1 - ((yo[y0] >> 1) - 1) + 1 > CO_ROWS) st->co_rows_ok = false;
    }
    return LVM_OK;
}

// ring capacity >= needed columns; linearises the live columns on growth
static int color_reserve(Ctx* c, ColorState* st, int need, hipStream_t s) {
    if (need <= st->cap) return LVM_OK;
    int ncap = 16;
    while (ncap < need) ncap *= 2;
    float *nw = nullptr, *ny = nullptr;
    LVM_HIP_TRY(c, hipMalloc((void**)&nw, (size_t)ncap * st->rows * sizeof(float)));
    LVM_HIP_TRY(c, hipMalloc((void**)&ny, (size_t)ncap * st->rows * sizeof(float)));
    if (st->n > 0) hipLaunchKernelGGL(k_col_regrow, dim3(st->rows), dim3(256), 0, s, (const float*)st->win, st->cap, st->slot0, st->n, nw, ncap, st->rows);
    LVM_HIP_TRY(c, hipStreamSynchronize(s));
    if (st->win) (void)hipFree(st->win);
    if (st->Y) (void)hipFree(st->Y);
    st->win = nw; st->Y = ny; st->cap = ncap; st->slot0 = 0;
    return LVM_OK;
}

struct ColBufs { float** G; float** up; float* col1; MinMax* mm; int nt; };   // nt frames laid out [frame][stream]

// Gaussian pyramid of the unscaled frames (MagnifyCore.hpp:169-172)
static void col_down(Ctx* c, ColorState* st, const FrameIO& io, const ColBufs& B, hipStream_t s) {
    const int C = io.channels, NZ = c->nstreams * B.nt, planes = st->planes * B.nt, w = io.w, h = io.h, levels = st->levels;
    const dim3 blk(256);
    const LevelGeom& g1 = st->g[1];
    const dim3 grid0((g1.w + 31) / 32, (g1.h + 15) / 16, NZ);
    const bool vec4 = C == 3 && w % 4 == 0 && io.in_stride % 4 == 0 && io.in_sstride % 4 == 0 && ((uintptr_t)io.d_in % 4) == 0;
    const int d0_sx = (g1.w + D0R_OUT - 1) / D0R_OUT;
    long d0_tasks = 0;
    const int d0_rows = down0_rows_choice(g1.w, g1.h, NZ, st->d0_min_tasks, &d0_tasks);
    // two levels in one pass (k_down01_rows): G_1 is never written.  Large launches only (temporal batches / many streams)
    long d01_tasks = 0;
    int d01_rows = (levels >= 2 && st->g[2].w >= 2) ? down01_rows_choice(st->g[2].w, st->g[2].h, NZ, st->d0_min_tasks, &d01_tasks) : 0;
    if (st->d01_rows_forced > 0 && d01_tasks > 0) {     // tests: a given strip height
        d01_rows = st->d01_rows_forced;
        d01_tasks = (long)((st->g[2].w + D01_OUT - 1) / D01_OUT) * ((st->g[2].h + d01_rows - 1) / d01_rows) * NZ;
    }
    const bool d01 = st->d01_on && vec4 && w % 8 == 0 && levels >= 2 && d01_tasks > 0 && st->g[1].w * 2 == w && st->g[2].w * 4 == w &&
                     io.in_stride > 0 && (long)io.in_stride * h < (1L << 31);
    int l = 1;
    if (d01) {
        D01Args q;
        q.in = io.d_in; q.in_stride = (long)io.in_stride; q.in_sstride = (long)io.in_sstride; q.w = w; q.h = h;
        q.G2 = B.G[2]; q.w2 = st->g[2].w; q.h2 = st->g[2].h; q.h1 = st->g[1].h;
        q.strips_x = (st->g[2].w + D01_OUT - 1) / D01_OUT; q.rows = d01_rows; q.strips_y = (st->g[2].h + d01_rows - 1) / d01_rows;
        q.ntasks = (int)d01_tasks;
        LVM_LAUNCH(c, "col_down01", k_down01_rows<1>, dim3((unsigned)((d01_tasks + D01_THREADS / 64 - 1) / (D01_THREADS / 64))), dim3(D01_THREADS), s, q);
        l = 2;
    } else if (vec4 && st->d0_rows_on && d0_tasks > 0) {   // wave strips with DPP halo exchange (pyramid.h)
        const dim3 gridr((unsigned)((d0_tasks + D0R_THREADS / 64 - 1) / (D0R_THREADS / 64)));
        LVM_LAUNCH(c, "col_down0", (k_down0_rows<false, FL_LUT_EXACT>), gridr, dim3(D0R_THREADS), s, io.d_in, (long)io.in_stride, (long)io.in_sstride,
                   w, h, B.G[1], g1.w, g1.h, c->lab, d0_sx, (g1.h + d0_rows - 1) / d0_rows, (int)d0_tasks, d0_rows, LabPlanes{nullptr, nullptr});
    } else if (vec4) {
        auto kv = k_down0_v4<false, FL_LUT_EXACT>;
        LVM_LAUNCH(c, "col_down0", kv, grid0, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, w, h, B.G[1], g1.w, g1.h, c->lab, LabPlanes{nullptr, nullptr});
    } else {
        auto kd0 = (C == 3) ? k_down0<3, false, FL_LUT_EXACT> : k_down0<1, false, FL_LUT_EXACT>;
        LVM_LAUNCH(c, "col_down0", kd0, grid0, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, w, h, B.G[1], g1.w, g1.h, c->lab, 1.0f, LabPlanes{nullptr, nullptr});
    }
    while (l < levels) {            // two pyramid levels per launch while possible
        if (st->g[l].w % 4 == 0 && (long)st->g[l].n * planes >= st->rows_min_elems) {
            // large planes (temporal batches / many streams): barrier-free wave strips (pyramid.h)
            const LevelGeom &a = st->g[l], &b = st->g[l + 1];
            const int sx = (b.w + 127) / 128;
            int rows = 16;
            while (rows > 4 && (long)sx * ((b.h + rows - 1) / rows) * planes < 8192) rows >>= 1;
            const int sy = (b.h + rows - 1) / rows;
            const long ntasks = (long)sx * sy * planes;
            const dim3 grid((unsigned)((ntasks + PD_THREADS / 64 - 1) / (PD_THREADS / 64)));
            LVM_LAUNCH(c, LName("pyr_down_rows", l), k_pyr_down_rows<1>, grid, dim3(PD_THREADS), s, (const float*)B.G[l], a.w, a.h, B.G[l + 1], b.w, b.h,
                       sx, sy, (int)ntasks, rows);
            l += 1;
        } else if (levels - l >= 2) {
            const LevelGeom &a = st->g[l], &b1 = st->g[l + 1], &b2 = st->g[l + 2];
            const dim3 grid((b2.w + ML_T - 1) / ML_T, (b2.h + ML_T - 1) / ML_T, planes);
            LVM_LAUNCH(c, LName("pyr_down2", l), k_pyr_down_multi<2>, grid, blk, s, (const float*)B.G[l], a.w, a.h, B.G[l + 1], b1.w, b1.h, B.G[l + 2],
                       b2.w, b2.h, (float*)nullptr, 0, 0);
            l += 2;
        } else {
            const LevelGeom &a = st->g[l], &b = st->g[l + 1];
            const dim3 grid((b.w + 31) / 32, (b.h + 15) / 16, planes);
            LVM_LAUNCH(c, LName("pyr_down", l), k_pyr_down<1>, grid, blk, s, (const float*)B.G[l], a.w, a.h, B.G[l + 1], b.w, b.h);
            l += 1;
        }
    }
}

// window append (nt columns) + ideal band-pass + normalisation of column 1 for every frame of the batch;
// slot = ring slot of the first new column, slot0 = window start seen by the first frame, n = window length
static int col_filter(Ctx* c, ColorState* st, const lvm_params& p, const ColBufs& B, int slot, int slot0, int n, hipStream_t s) {
    const int C = st->channels, NS = c->nstreams, levels = st->levels;
    const int nL = (int)st->g[levels].n, live = nL * C;
    const dim3 blk(256);
    LVM_LAUNCH(c, "col_append", k_col_append, dim3((live + 255) / 256, NS, B.nt), blk, s, (const float*)B.G[levels], st->win, live,
               st->rows_ps, st->cap, slot, B.mm, NS);
    if (n < 2) return LVM_OK;
    double lo = p.coLow, hi = p.coHigh;
    if (lo == 0.00) lo += 0.01;                                                       // TemporalFilter.cpp:26-27
    const float width = (float)n;
    const double fl = 2 * lo * width / p.framerate, fh = 2 * hi * width / p.framerate; // :65-66
    // entries of the packed spectrum the 0/1 mask lets through (TemporalFilter.cpp:65-77)
    ThinBins eb{};
    int ne = 0;
    {
        auto m = [&](int x) { return (x >= fl && x <= fh) ? 1.0f : 0.0f; };
        const int half = (n - 1) / 2;
        auto push = [&](int bin, int kind, float ma, float mb) { if (ne < kThinBins) { eb.bin[ne] = bin; eb.kind[ne] = kind; eb.ma[ne] = ma; eb.mb[ne] = mb; } ++ne; };
        if (m(0) != 0.f) push(0, 1, m(0), 0.f);
        for (int k = 1; k <= half; ++k) if (m(2 * k - 1) != 0.f || m(2 * k) != 0.f) push(k, 0, m(2 * k - 1), m(2 * k));
        if (n % 2 == 0 && m(n - 1) != 0.f) push(n / 2, 2, m(n - 1), 0.f);
        eb.ne = ne;
    }
    if (ne <= kThin8Bins && n <= kThin8MaxN && st->thin_dft && st->thin8_dft && B.nt >= st->thin_min_frames) {
        LVM_LAUNCH(c, "col_dft", n <= 128 ? k_col_dft_thin8<128> : k_col_dft_thin8<kThin8MaxN>, dim3((live + kThin8Rows - 1) / kThin8Rows, NS, B.nt), blk, s, (const float*)st->win, slot0, n, st->cap,
                   st->rows_ps, live, (const double*)st->tw, B.col1, B.mm, eb);
    } else if (ne <= kThinBins && n <= kDftMaxN && st->thin_dft && B.nt >= st->thin_min_frames) {
        LVM_LAUNCH(c, "col_dft", k_col_dft_thin, dim3((live + 255) / 256, NS, B.nt), blk, s, (const float*)st->win, slot0, n, st->cap,
                   st->rows_ps, live, (const double*)st->tw, B.col1, B.mm, eb);
    } else if (n <= kDftMaxN) {
        int gx = (live + kDftRows - 1) / kDftRows;
        gx = gx < 1024 ? gx : 1024;
        LVM_LAUNCH(c, "col_dft", k_col_dft, dim3(gx, NS, B.nt), blk, s, (const float*)st->win, slot0, n, st->cap, st->rows_ps, live, fl, fh,
                   (const double*)st->tw, B.col1, B.mm);
    } else {   // very long windows: serial kernel, one frame per launch
        LVM_LAUNCH(c, "col_dft", k_col_dft_serial, dim3((live + 255) / 256, NS), blk, s, (const float*)st->win, slot0, n, st->cap,
                   st->rows_ps, live, fl, fh, (const double*)st->tw, st->Y, B.col1, B.mm);
    }
    st->amp = (float)p.amplification;          // the normalisation runs at the head of col_up_out (on its own or inside the first pyrUp)
    return LVM_OK;
}

// up chain (SpatialFilter.cpp:40-50) + output (MagnifyCore.hpp:197-203)
static void col_up_out(Ctx* c, ColorState* st, const FrameIO& io, const ColBufs& B, hipStream_t s) {
    const int C = io.channels, NZ = c->nstreams * B.nt, planes = st->planes * B.nt, levels = st->levels;
    const dim3 blk(256);
    int uw = st->g[levels].w, uh = st->g[levels].h;
    // (a fused LDS-resident launch for the first three pyrUps measured 23.5 us against 3 x 7 us: not kept)
    // vectorised strip output kernels: decided up front, because with them the LAST generic pyrUp is not launched either
    // (the up chain doubles the TOP level's size every step: its images are not the pyramid's sizes when a level was odd)
    const int vw_f = levels >= 1 ? st->g[levels].w << (levels - 1) : 0;
    const bool vec4_f = C == 3 && io.w == 2 * vw_f && io.w % 4 == 0 && io.in_stride % 4 == 0 && io.in_sstride % 4 == 0 && io.out_stride % 4 == 0 &&
                        io.out_sstride % 4 == 0 && ((uintptr_t)io.d_in % 4) == 0 && ((uintptr_t)io.d_out % 4) == 0 && st->co_rows_ok;
    // k_col_out_strips: both pyrUps inside (strictly increasing row map; a level-2 image of at least 2 x 2 for the border maps)
    const bool lean = st->out_lean && st->yofs_strict && vec4_f && st->out_rows > 0 && levels >= 2 &&
                      (st->g[levels].w << (levels - 2)) >= 2 && (st->g[levels].h << (levels - 2)) >= 2 &&
                      io.in_stride > 0 && io.out_stride > 0 && (long)io.in_stride * io.h < (1L << 31) && (long)io.out_stride * io.h < (1L << 31);
    const bool fuse2 = lean;                      // k_col_out_strips makes the last TWO pyrUps itself
    const int n_generic = levels - (fuse2 ? 1 : 0) - 1;            // generic pyrUp launches of the chain
    // normalize(0, 1, NORM_MINMAX) x amplification (TemporalFilter.cpp:55, MagnifyCore.hpp:185): inside the first pyrUp when that is a
    // k_pyr_up_rows launch, a kernel of its own otherwise (no generic pyrUp left, or the tiled pyrUp)
    const int nL = (int)st->g[levels].n, live = nL * C;
    const bool norm_in_up = st->norm_in_up && n_generic >= 1 && st->up_rows && (2 * uw) % 4 == 0;
    if (!norm_in_up)
        LVM_LAUNCH(c, "col_norm", k_col_norm, dim3((live + 255) / 256, c->nstreams * B.nt), blk, s, (const float*)B.col1, B.up[0], live, st->rows_ps,
                   (const MinMax*)B.mm, st->amp);
    for (int k = 0; k + 1 < levels - (fuse2 ? 1 : 0); ++k) {      // L-1 generic pyrUps, the last one is fused into k_col_out (k_col_out_strips: the last two)
        if (k == 0 && norm_in_up) {
            const long ngroups = (long)(2 * uw / 4) * uh;
            LVM_LAUNCH(c, "pyr_up_l0", k_pyr_up_rows_norm, dim3((unsigned)((ngroups + 255) / 256), planes), blk, s, (const float*)B.col1, st->rows_ps, C, uw, uh,
                       B.up[1], (int)ngroups, (const MinMax*)B.mm, st->amp);
        } else if (st->up_rows && (2 * uw) % 4 == 0) {      // barrier-free blocks of 4 x 2 outputs per lane (pyramid.h)
            const long ngroups = (long)(2 * uw / 4) * uh;
            LVM_LAUNCH(c, LName("pyr_up", k), k_pyr_up_rows<1>, dim3((unsigned)((ngroups + 255) / 256), planes), blk, s, (const float*)B.up[k], uw, uh,
                       B.up[k + 1], (int)ngroups);
        } else {
            const dim3 grid((2 * uw + 63) / 64, (2 * uh + 15) / 16, planes);
            LVM_LAUNCH(c, LName("pyr_up", k), k_pyr_up<1>, grid, blk, s, (const float*)B.up[k], uw, uh, B.up[k + 1], 2 * uw, 2 * uh);
        }
        uw *= 2; uh *= 2;
    }
    OutArgs a;
    a.in = io.d_in; a.in_stride = (long)io.in_stride; a.in_sstride = (long)io.in_sstride;
    a.out = io.d_out; a.out_stride = (long)io.out_stride; a.out_sstride = (long)io.out_sstride;
    a.U2 = nullptr; a.w2 = a.h2 = 0;
    if (fuse2) { a.U2 = B.up[levels - 2]; a.w2 = uw; a.h2 = uh; uw *= 2; uh *= 2; }
    a.w = io.w; a.h = io.h; a.V = B.up[levels - 1]; a.vw = uw; a.vh = uh;
    a.xofs = st->xofs; a.xa = st->xa; a.yofs = st->yofs; a.ya = st->ya; a.yslot = st->yslot; a.mm = B.mm;
    a.tiles_x = (io.w + CT_W - 1) / CT_W; a.tiles_y = (io.h + CT_H - 1) / CT_H;
    a.dbg = c->keep_float ? c->d_float : nullptr;
    const dim3 grid(a.tiles_x, a.tiles_y, NZ);
    const bool vec4 = C == 3 && io.w == 2 * uw && io.w % 4 == 0 && io.in_stride % 4 == 0 && io.in_sstride % 4 == 0 &&
                      io.out_stride % 4 == 0 && io.out_sstride % 4 == 0 && ((uintptr_t)io.d_in % 4) == 0 && ((uintptr_t)io.d_out % 4) == 0 &&
                      st->co_rows_ok;
    if (vec4 && st->out_rows > 0) {
        const int sx = (io.w + 255) / 256;
        int rows = lean ? st->out_rows_lean : st->out_rows;
        const long min_tasks = lean ? st->out_min_tasks_lean : st->out_min_tasks;
        while (rows > 2 && (long)sx * ((io.h + rows - 1) / rows) * NZ < min_tasks) rows >>= 1;
        const int sy = (io.h + rows - 1) / rows;
        const long ntasks = (long)sx * sy * NZ;
        const dim3 g2((unsigned)((ntasks + 3) / 4));
        if (lean) {
            LVM_LAUNCH(c, "col_minmax_u2", (k_col_out_strips<false, false>), g2, blk, s, a, sx, sy, (int)ntasks, rows);
            if (a.dbg) LVM_LAUNCH(c, "col_out_u2", (k_col_out_strips<true, true>), g2, blk, s, a, sx, sy, (int)ntasks, rows);
            else LVM_LAUNCH(c, "col_out_u2", (k_col_out_strips<true, false>), g2, blk, s, a, sx, sy, (int)ntasks, rows);
        } else {
            LVM_LAUNCH(c, "col_minmax", (k_col_out_rows<false, false>), g2, blk, s, a, sx, sy, (int)ntasks, rows);
            if (a.dbg) LVM_LAUNCH(c, "col_out", (k_col_out_rows<true, true>), g2, blk, s, a, sx, sy, (int)ntasks, rows);
            else LVM_LAUNCH(c, "col_out", (k_col_out_rows<true, false>), g2, blk, s, a, sx, sy, (int)ntasks, rows);
        }
    } else if (vec4) {
        LVM_LAUNCH(c, "col_minmax", k_col_out_v4<false>, grid, blk, s, a);
        LVM_LAUNCH(c, "col_out", k_col_out_v4<true>, grid, blk, s, a);
    } else {
        auto k1 = (C == 3) ? k_col_out<3, false> : k_col_out<1, false>;
        auto k2 = (C == 3) ? k_col_out<3, true> : k_col_out<1, true>;
        LVM_LAUNCH(c, "col_minmax", k1, grid, blk, s, a);
        LVM_LAUNCH(c, "col_out", k2, grid, blk, s, a);
    }
}

int color_process(Ctx* c, const lvm_params& p, int levels, const FrameIO& io, hipStream_t s, int* produced) {
    *produced = 0;
    ColorState* st = static_cast<ColorState*>(c->state);
    if (!st) {
        st = new ColorState();
        c->state = st;
        if (const char* e = std::getenv("LVM_COL_THIN_DFT")) st->thin_dft = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_COL_THIN8_DFT")) st->thin8_dft = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_COL_NORM_IN_UP")) st->norm_in_up = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_COL_UP_ROWS")) st->up_rows = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_D0_ROWS")) st->d0_rows_on = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_D0_MIN_TASKS")) st->d0_min_tasks = std::atol(e);
        if (const char* e = std::getenv("LVM_COL_OUT_ROWS")) st->out_rows = st->out_rows_lean = std::atoi(e);
        if (const char* e = std::getenv("LVM_COL_OUT_LEAN")) st->out_lean = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_COL_DOWN01")) st->d01_on = std::atoi(e) != 0;
        if (const char* e = std::getenv("LVM_COL_DOWN01_ROWS")) st->d01_rows_forced = std::atoi(e);
        if (const char* e = std::getenv("LVM_COL_OUT_MIN_TASKS")) st->out_min_tasks = st->out_min_tasks_lean = std::atol(e);
        if (const char* e = std::getenv("LVM_COL_THIN_MIN_FRAMES")) st->thin_min_frames = std::atoi(e);
        if (const char* e = std::getenv("LVM_ROWS_MIN_ELEMS")) st->rows_min_elems = std::atol(e);
        int rc = color_alloc(c, st, io.w, io.h, io.channels, levels);
        if (rc == LVM_OK && c->max_frames > 1) rc = color_reserve_frames(c, st, c->max_frames, s);
        if (rc != LVM_OK) return rc;
    }
    const ColBufs B{st->G, st->up, st->col1, st->mm, 1};
    col_down(c, st, io, B, s);
    // ---- rolling window (:175-176, SpatialFilter.cpp:63-84) ----
    const int maxImages = optimal_buffer_size((int)p.framerate);
    st->max_images = maxImages;
    {
        const int want = maxImages > 0 && maxImages <= 4096 ? maxImages + kColorBatchMax + 1 : 0;      // room for temporal batches
        const int rc = color_reserve(c, st, st->n + 1 > want ? st->n + 1 : want, s);
        if (rc != LVM_OK) return rc;
    }
    const int slot = (st->slot0 + st->n) % st->cap;
    int n = st->n + 1, slot0 = st->slot0;
    if (n > maxImages && maxImages > 0) { slot0 = (slot0 + 1) % st->cap; n -= 1; }
    if (n >= 2 && st->tw_n != n) {   // twiddles cos/sin(2 pi k / n), float64, computed on the host like the oracle's
        // The window length grows by one per frame during warm-up (2, 3, .. maxImages): the tables of every length up
        // to maxImages are built and uploaded ONCE (sum 2n doubles: 132 KB for T = 128), so a warm-up frame costs no
        // host synchronisation, allocation or copy.  Lengths beyond kTwAllMax (fps > 256) keep the per-length upload.
        const double two_pi = 2.0 * 3.1415926535897932384626433832795;
        if (maxImages <= kTwAllMax && n <= maxImages) {
            if (st->tw_all_max < maxImages) {
                std::vector<double> t; std::vector<size_t> off((size_t)maxImages + 1, 0);
                for (int m = 2; m <= maxImages; ++m) {
                    off[(size_t)m] = t.size();
                    for (int k = 0; k < m; ++k) t.push_back(std::cos(two_pi * k / m));
                    for (int k = 0; k < m; ++k) t.push_back(std::sin(two_pi * k / m));
                }
                LVM_HIP_TRY(c, hipStreamSynchronize(s));
                sync_streams(c);
                if (st->tw_all) (void)hipFree(st->tw_all);
                st->tw_all = nullptr; st->tw_all_max = 0;
                LVM_HIP_TRY(c, hipMalloc((void**)&st->tw_all, t.size() * sizeof(double)));
                LVM_HIP_TRY(c, hipMemcpy(st->tw_all, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
                st->tw_off = off; st->tw_all_max = maxImages;
            }
            st->tw = st->tw_all + st->tw_off[(size_t)n];
        } else {
            std::vector<double> t(2 * (size_t)n);
            for (int k = 0; k < n; ++k) { t[k] = std::cos(two_pi * k / n); t[n + k] = std::sin(two_pi * k / n); }
            LVM_HIP_TRY(c, hipStreamSynchronize(s));
            sync_streams(c);
            if (st->tw_own) (void)hipFree(st->tw_own);
            st->tw_own = nullptr;
            LVM_HIP_TRY(c, hipMalloc((void**)&st->tw_own, t.size() * sizeof(double)));
            LVM_HIP_TRY(c, hipMemcpy(st->tw_own, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
            st->tw = st->tw_own;
        }
        st->tw_n = n;
    }
    const int rc = col_filter(c, st, p, B, slot, slot0, n, s);
    if (rc != LVM_OK) return rc;
    st->n = n; st->slot0 = slot0;
    if (st->n < 2) { LVM_HIP_TRY(c, hipGetLastError()); return LVM_OK; }            // :180
    col_up_out(c, st, io, B, s);
    LVM_HIP_TRY(c, hipGetLastError());
    *produced = 1;
    return LVM_OK;
}

// Buffers of a temporal batch; sized for max(nt, lvm_set_max_frames hint capped at kColorBatchMax) so steady-state
// calls never allocate.
static int color_reserve_frames(Ctx* c, ColorState* st, int nt, hipStream_t s) {
    const int hint = c->max_frames < kColorBatchMax ? c->max_frames : kColorBatchMax;
    if (nt < hint) nt = hint;
    if (nt <= st->tcap) return LVM_OK;
    const int levels = st->levels, NS = c->nstreams;
    LVM_HIP_TRY(c, hipStreamSynchronize(s));
    sync_streams(c);
    if (st->tarena) (void)hipFree(st->tarena);
    if (st->mmt) (void)hipFree(st->mmt);
    st->tarena = nullptr; st->mmt = nullptr; st->tcap = 0;
    auto pad = [](size_t n) { return (n + 63) & ~(size_t)63; };
    const size_t nL = st->g[levels].n;
    size_t total = 64;
    for (int l = 1; l <= levels; ++l) total += pad(st->g[l].n * st->planes * nt);
    for (int k = 0; k < levels; ++k) total += pad((nL << (2 * k)) * st->planes * nt);
    total += pad((size_t)st->rows * nt);
    if (hipMalloc((void**)&st->tarena, total * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); st->tarena = nullptr; c->err = "color: hipMalloc (frames) failed"; return LVM_ERR_OOM; }
    LVM_HIP_TRY(c, hipMalloc((void**)&st->mmt, sizeof(MinMax) * NS * nt));
    float* q = st->tarena;
    for (int l = 1; l <= levels; ++l) { st->Gt[l] = q; q += pad(st->g[l].n * st->planes * nt); }
    for (int k = 0; k < levels; ++k) { st->upt[k] = q; q += pad((nL << (2 * k)) * st->planes * nt); }
    st->col1t = q;
    st->tcap = nt;
    return LVM_OK;
}

// Temporal batch: nt consecutive frames; only in the steady state of the rolling window (full window, so
// its length -- and with it the twiddle table and the mask -- is the same for every frame of the batch).
bool color_can_batch(const Ctx* c, const lvm_params& p, int nt) {
    const ColorState* st = dynamic_cast<const ColorState*>(c->state);
    if (!st) return false;
    const int maxImages = optimal_buffer_size((int)p.framerate);
    return maxImages > 0 && st->max_images == maxImages && st->n == maxImages && st->tw_n == st->n && st->n <= kDftMaxN &&
           st->cap >= st->n + nt;
}

int color_process_frames(Ctx* c, const lvm_params& p, const FrameIO& io, int nt, hipStream_t s) {
    ColorState* st = static_cast<ColorState*>(c->state);
    if (nt > st->tcap) { const int rc = color_reserve_frames(c, st, nt, s); if (rc != LVM_OK) return rc; }
    const ColBufs B{st->Gt, st->upt, st->col1t, st->mmt, nt};
    col_down(c, st, io, B, s);
    // frame f appends at slot (slot0 + n + f) and, the window being full, sees it start at slot0 + f + 1
    const int slot = (st->slot0 + st->n) % st->cap, slot0 = (st->slot0 + 1) % st->cap;
    const int rc = col_filter(c, st, p, B, slot, slot0, st->n, s);
    if (rc != LVM_OK) return rc;
    st->slot0 = (st->slot0 + nt) % st->cap;
    col_up_out(c, st, io, B, s);
    LVM_HIP_TRY(c, hipGetLastError());
    return LVM_OK;
}

}  // namespace lvm
