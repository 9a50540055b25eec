// laplace.hip -- Laplacian-pyramid motion magnification on gfx950.
//
// Replaces magcore::magnifyMotion (reference: processing/magnification/MagnifyCore.hpp:83-160)
// and the helpers it calls: buildLaplacePyrFromImg / buildImgFromLaplacePyr
// (SpatialFilter.cpp:25-38, 52-61) and iirFilter (TemporalFilter.cpp:9-22).
//
// Data layout in HBM (per context, all streams): planar float32, one plane per
// (stream, channel): G_l[plane][h_l][w_l] for l = 1..L (Gaussian pyramid of the Lab frame),
// the two IIR low-pass states hi_l/lo_l[plane][n_l] for the live bands l = 1..L-1 and the
// collapse accumulators cur_l[plane][n_l] for l = 1..L-1.  Level 0 never exists as float in
// HBM: the u8 frame is converted to Lab on the fly by the first and the last kernel.
// Band 0 and the residual are multiplied by 0 in the reference (MagnifyCore.hpp:129-131), so
// their bands and IIR states are never computed (dead state).
//
// Launch sequence per frame (L levels):
//   k_lap_down0           u8 BGR -> Lab -> pyrDown           -> G_1
//   k_pyr_down  x (L-1)   G_l -> G_{l+1}
//   k_lap_up    x (L-1)   l = L-1..1, fused: band_l = G_l - pyrUp(G_{l+1}); IIR x2 (state R/W);
//                         x gain_l; cur_l = pyrUp(cur_{l+1}) + motion_l
//   k_lap_final           out = u8(Lab2BGR(Lab(u8 in) + [1,ca,ca] * pyrUp(cur_1)))
// All stencils are LDS-staged tiles; pyrDown/pyrUp follow OpenCV's border rules and operation
// order exactly (see the per-kernel comments) so the result is order-faithful to the oracle.
#include "pyramid.h"

namespace lvm {

struct UpArgs {
    const float* Gl;    // G_l            [planes][h][w]
    const float* Gn;    // G_{l+1}        [planes][hn][wn]
    const float* curn;  // cur_{l+1} or nullptr (top live level: pyrUp of the zeroed residual)
    float* hi; float* lo;  // IIR states  [planes][h*w]
    float* cur;         // cur_l
    int w, h, wn, hn;
    float aHi, bHi, aLo, bLo, gain;
};

// Fused band + temporal IIR + gain + collapse step of one level.  SEED = first frame: both
// low-pass states are seeded with the band (MagnifyCore.hpp:100-103) and nothing else happens.
template <bool SEED>
__global__ __launch_bounds__(256) void k_lap_up(UpArgs a) {
    __shared__ float s_g[US_H][US_W + 1], s_c[US_H][US_W + 1];
    __shared__ float h_g[US_H][UT_W + 1], h_c[US_H][UT_W + 1];
    const int x0 = blockIdx.x * UT_W, y0 = blockIdx.y * UT_H;
    const int sx0 = x0 / 2 - 1, sy0 = y0 / 2 - 1;
    const size_t pn = (size_t)blockIdx.z * a.wn * a.hn, pl = (size_t)blockIdx.z * a.w * a.h;
    const bool has_cur = !SEED && a.curn != nullptr;
    pyrup_stage(s_g, a.Gn + pn, a.wn, a.hn, sx0, sy0);
    if (has_cur) pyrup_stage(s_c, a.curn + pn, a.wn, a.hn, sx0, sy0);
    __syncthreads();
    pyrup_hpass(h_g, s_g, x0, sx0, a.wn, a.w);
    if (has_cur) pyrup_hpass(h_c, s_c, x0, sx0, a.wn, a.w);
    __syncthreads();
    for (int i = threadIdx.x; i < UT_H * UT_W; i += 256) {
        const int y = i / UT_W, x = i - y * UT_W;
        const int gx = x0 + x, gy = y0 + y;
        if (gx >= a.w || gy >= a.h) continue;
        const size_t idx = pl + (size_t)gy * a.w + gx;
        const float band = a.Gl[idx] - pyrup_v(h_g, x, gy, sy0);        // SpatialFilter.cpp:33
        if (SEED) {
            a.hi[idx] = band; a.lo[idx] = band;
        } else {
            const float t1 = a.hi[idx] * a.aHi + band * a.bHi;          // TemporalFilter.cpp:16
            const float t2 = a.lo[idx] * a.aLo + band * a.bLo;          // :17
            a.hi[idx] = t1; a.lo[idx] = t2;
            const float m = (t1 - t2) * a.gain;                         // :21, MagnifyCore.hpp:129-132
            const float up = has_cur ? pyrup_v(h_c, x, gy, sy0) : 0.f;
            a.cur[idx] = up + m;                                        // SpatialFilter.cpp:58
        }
    }
}

// Final level: out = u8(Lab2BGR(Lab(in) + [1, ca, ca] * pyrUp(cur_1))).  MOTION = false is
// the first frame / L == 1 case (motion image is identically zero).  Persistent workgroups
// walk over (stream, tile); the inverse-gamma spline table lives in LDS.
template <int C, bool MOTION, bool EXACT>
__global__ __launch_bounds__(256) void k_lap_final(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                                   uint8_t* __restrict__ out, long out_stride, long out_sstride,
                                                   int w, int h, const float* __restrict__ cur1, int w1, int h1,
                                                   LabCoef lab, float ca, int tiles_x, int tiles_y, int nstreams,
                                                   float* __restrict__ dbg) {
    __shared__ __attribute__((aligned(16))) float s_igt[C == 3 ? 4096 : 4];
    __shared__ float s_gam[C == 3 ? 256 : 1];
    __shared__ float s_c[C][US_H][US_W + 1];
    __shared__ float h_c[C][US_H][UT_W + 1];
    if (C == 3) { load_invgamma(s_igt, lab.invgamma); load_gamma_u8(s_gam, lab.gamma_u8); }
    __syncthreads();
    const int ntiles = tiles_x * tiles_y * nstreams;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (tiles_x * tiles_y);
        const int r = t - b * (tiles_x * tiles_y);
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int x0 = tx * UT_W, y0 = ty * UT_H;
        const int sx0 = x0 / 2 - 1, sy0 = y0 / 2 - 1;
        if (MOTION) {
#pragma unroll
            for (int c = 0; c < C; ++c)
                pyrup_stage(s_c[c], cur1 + ((size_t)b * C + c) * w1 * h1, w1, h1, sx0, sy0);
            __syncthreads();
#pragma unroll
            for (int c = 0; c < C; ++c) pyrup_hpass(h_c[c], s_c[c], x0, sx0, w1, w);
            __syncthreads();
        }
        const uint8_t* src = in + (size_t)b * in_sstride;
        uint8_t* dst = out + (size_t)b * out_sstride;
        for (int i = threadIdx.x; i < UT_H * UT_W; i += 256) {
            const int y = i / UT_W, x = i - y * UT_W;
            const int gx = x0 + x, gy = y0 + y;
            if (gx >= w || gy >= h) continue;
            const uint8_t* p = src + (size_t)gy * in_stride + (size_t)gx * C;
            uint8_t* q = dst + (size_t)gy * out_stride + (size_t)gx * C;
            if (C == 3) {
                float L, a, bb;
                lin_bgr_to_lab<EXACT>(s_gam[p[0]], s_gam[C == 3 ? p[1] : 0], s_gam[C == 3 ? p[2] : 0], lab.fwd, L, a, bb);
                if (MOTION) {
                    const float m0 = pyrup_v(h_c[0], x, gy, sy0);
                    const float m1 = pyrup_v(h_c[C > 1 ? 1 : 0], x, gy, sy0) * ca;   // MagnifyCore.hpp:143-144
                    const float m2 = pyrup_v(h_c[C > 2 ? 2 : 0], x, gy, sy0) * ca;
                    L = L + m0; a = a + m1; bb = bb + m2;                            // :148
                }
                float o0, o1, o2;
                lab_to_bgr<EXACT>(L, a, bb, lab.inv, s_igt, o0, o1, o2);                    // :152
                if (dbg && b == 0) {
                    float* d = dbg + ((size_t)gy * w + gx) * 3;
                    d[0] = o0; d[1] = o1; d[2] = o2;
                }
                q[0] = sat_u8(o0 * 255.0f + lab.a255);                               // :153
                q[1] = sat_u8(o1 * 255.0f + lab.a255);
                q[2] = sat_u8(o2 * 255.0f + lab.a255);
            } else {
                float v = (float)p[0] * lab.a255;                                    // :92
                if (MOTION) v = v + pyrup_v(h_c[0], x, gy, sy0);
                if (dbg && b == 0) dbg[(size_t)gy * w + gx] = v;
                q[0] = sat_u8(v * 255.0f + lab.a255);                                // :156
            }
        }
        if (MOTION) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct LaplaceState : ModeState {
    int levels = 0, planes = 0;
    LevelGeom g[kMaxLevels + 1];
    float* arena = nullptr;
    float* G[kMaxLevels + 1] = {};
    float *hi[kMaxLevels + 1] = {}, *lo[kMaxLevels + 1] = {}, *cur[kMaxLevels + 1] = {};
    bool seeded = false;
    ~LaplaceState() override { if (arena) (void)hipFree(arena); }
};

static int laplace_alloc(Ctx* c, LaplaceState* st, int w, int h, int channels, int levels) {
    st->levels = levels;
    st->planes = c->nstreams * channels;
    st->g[0] = {w, h, (size_t)w * h};
    for (int l = 1; l <= levels; ++l) {
        const int lw = (st->g[l - 1].w + 1) / 2, lh = (st->g[l - 1].h + 1) / 2;
        st->g[l] = {lw, lh, (size_t)lw * lh};
    }
    size_t total = 0;
    auto pad = [](size_t n) { return (n + 63) & ~(size_t)63; };
    for (int l = 1; l <= levels; ++l) total += pad(st->g[l].n * st->planes);
    for (int l = 1; l < levels; ++l) total += 3 * pad(st->g[l].n * st->planes);
    if (total == 0) total = 64;
    if (hipMalloc((void**)&st->arena, total * sizeof(float)) != hipSuccess) {
        c->err = "laplace: hipMalloc failed";
        st->arena = nullptr;
        return LVM_ERR_OOM;
    }
    float* p = st->arena;
    for (int l = 1; l <= levels; ++l) { st->G[l] = p; p += pad(st->g[l].n * st->planes); }
    for (int l = 1; l < levels; ++l) {
        st->hi[l] = p; p += pad(st->g[l].n * st->planes);
        st->lo[l] = p; p += pad(st->g[l].n * st->planes);
        st->cur[l] = p; p += pad(st->g[l].n * st->planes);
    }
    return LVM_OK;
}

// MagnifyCore.hpp:114-134 (all float/double conversions as in the reference)
static void laplace_gains(int w, int h, int levels, double amplification, double coWavelength, float* gains) {
    const float delta = (float)(coWavelength / (8.0 * (1.0 + amplification)));
    const float exaggeration = 2.0f;
    float lambda = (float)(std::sqrt((double)(w * w + h * h)) / 3.0);
    for (int l = levels; l >= 0; --l) {
        const float currAlpha = (float)((lambda / (delta * 8.0) - 1.0) * exaggeration);
        const float amp = (float)amplification;
        gains[l] = (l == levels || l == 0) ? 0.0f : (amp < currAlpha ? amp : currAlpha);
        lambda = (float)(lambda / 2.0);
    }
}

int laplace_process(Ctx* c, const lvm_params& p, int levels, const FrameIO& io, hipStream_t s, int* produced) {
    LaplaceState* st = static_cast<LaplaceState*>(c->state);
    if (!st) {
        st = new LaplaceState();
        c->state = st;
        const int rc = laplace_alloc(c, st, io.w, io.h, io.channels, levels);
        if (rc != LVM_OK) return rc;
    }
    const int C = io.channels, NS = c->nstreams;
    const dim3 blk(256);
    const bool first = !st->seeded;
    float* dbg = c->keep_float ? c->d_float : nullptr;

    // ---- down sweep: Gaussian pyramid G_1..G_L (needed when any live band exists) ----
    if (levels >= 2) {
        const LevelGeom& g1 = st->g[1];
        const dim3 grid0((g1.w + DT_W - 1) / DT_W, (g1.h + DT_H - 1) / DT_H, NS);
        auto kd0 = (C == 3) ? (c->exact_lab ? k_down0<3, true, true> : k_down0<3, true, false>) : k_down0<1, false, true>;
        LVM_LAUNCH(c, "lap_down0", kd0, grid0, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.w, io.h,
                   st->G[1], g1.w, g1.h, c->lab, c->lab.a255);
        for (int l = 1; l < levels; ++l) {
            const LevelGeom &a = st->g[l], &b = st->g[l + 1];
            const dim3 grid((b.w + DT_W - 1) / DT_W, (b.h + DT_H - 1) / DT_H, st->planes);
            LVM_LAUNCH(c, "pyr_down", k_pyr_down<0>, grid, blk, s, (const float*)st->G[l], a.w, a.h, st->G[l + 1], b.w, b.h);
        }
    }
    // ---- up sweep ----
    float gains[kMaxLevels + 2];
    laplace_gains(io.w, io.h, levels, p.amplification, p.coWavelength, gains);
    double cLo = p.coLow, cHi = p.coHigh;
    if (cLo == 0) cLo = 0.01;                                            // TemporalFilter.cpp:11-12
    for (int l = levels - 1; l >= 1; --l) {
        UpArgs a;
        a.Gl = st->G[l]; a.Gn = st->G[l + 1];
        a.curn = (l + 1 <= levels - 1) ? st->cur[l + 1] : nullptr;
        a.hi = st->hi[l]; a.lo = st->lo[l]; a.cur = st->cur[l];
        a.w = st->g[l].w; a.h = st->g[l].h; a.wn = st->g[l + 1].w; a.hn = st->g[l + 1].h;
        a.aHi = (float)(1 - cHi); a.bHi = (float)cHi; a.aLo = (float)(1 - cLo); a.bLo = (float)cLo;
        a.gain = gains[l];
        const dim3 grid((a.w + UT_W - 1) / UT_W, (a.h + UT_H - 1) / UT_H, st->planes);
        if (first) LVM_LAUNCH(c, "lap_seed", k_lap_up<true>, grid, blk, s, a);
        else LVM_LAUNCH(c, "lap_up", k_lap_up<false>, grid, blk, s, a);
    }
    // ---- final ----
    {
        const int tx = (io.w + UT_W - 1) / UT_W, ty = (io.h + UT_H - 1) / UT_H;
        const int ntiles = tx * ty * NS;
        const dim3 grid(ntiles < 2048 ? ntiles : 2048);
        const bool motion = !first && levels >= 2;
        const float ca = (float)p.chromAttenuation;
        const float* cur1 = motion ? st->cur[1] : nullptr;
        const int w1 = st->g[1].w, h1 = st->g[1].h;
        auto kf = (C == 3) ? (motion ? (c->exact_lab ? k_lap_final<3, true, true> : k_lap_final<3, true, false>)
                                     : (c->exact_lab ? k_lap_final<3, false, true> : k_lap_final<3, false, false>))
                           : (motion ? k_lap_final<1, true, true> : k_lap_final<1, false, true>);
        LVM_LAUNCH(c, "lap_final", kf, grid, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.d_out,
                   (long)io.out_stride, (long)io.out_sstride, io.w, io.h, cur1, w1, h1, c->lab, ca, tx, ty, NS, dbg);
    }
    LVM_HIP_TRY(c, hipGetLastError());
    st->seeded = true;
    *produced = 1;                                                       // MagnifyCore.hpp:159
    return LVM_OK;
}

}  // namespace lvm
