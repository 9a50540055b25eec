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
// Launch sequence (L levels; a launch covers one frame or, in temporal batches, T frames of every stream):
//   k_down0_rows | k_down0_v4        u8 BGR -> Lab -> pyrDown -> G_1   (wave strips + DPP halo | LDS tile)
//   k_pyr_down_rows | _multi | plain G_l -> G_{l+1}                      (wave strips for large planes)
//   k_lap_up | k_lap_up_rows         l = L-1..1, fused: band_l = G_l - pyrUp(G_{l+1}); IIR x2; x gain_l;
//                                    cur_l = pyrUp(cur_{l+1}) + motion_l; in a batch the frame loop runs inside
//                                    the kernel with the IIR states in registers (LDS tiles | 2x2 blocks per lane)
//   k_lap_tail                       per-frame calls: all levels with <= 8192 pixels in one LDS-resident launch
//   k_lap_final_v4 | k_lap_final     out = u8(Lab2BGR(Lab(u8 in) + [1,ca,ca] * pyrUp(cur_1)))   (wave strips | tile)
// pyrDown/pyrUp follow OpenCV's border rules and operation order exactly (see the per-kernel comments) so
// every variant is order-faithful to the oracle; which variant runs is decided per launch from its size.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pyramid.h"

namespace lvm {

struct UpArgs {
    const float* Gl;    // G_l            [frame][planes][h][w]
    const float* Gn;    // G_{l+1}        [frame][planes][hn][wn]
    const float* curn;  // cur_{l+1} or nullptr (top live level: pyrUp of the zeroed residual)
    float* hi; float* lo;  // IIR states  [planes][h*w]
    float* cur;         // cur_l          [frame][planes][h*w]
    int w, h, wn, hn;
    float aHi, bHi, aLo, bLo, gain;
    int nt;             // frames handled by this launch, in temporal order (1 = per-frame mode)
    long fsl, fsn;      // frame strides (floats) of the level-l and level-(l+1) arrays
};

// Fused band + temporal IIR + gain + collapse step of one level.  SEED = first frame: both
// low-pass states are seeded with the band (MagnifyCore.hpp:100-103) and nothing else happens.
// With nt > 1 the workgroup walks over nt consecutive frames of the stream: every thread owns the
// same 4 pixels for all frames and keeps their two low-pass states in registers, so the state is
// read and written once per launch instead of once per frame.
template <bool SEED, int D>
__global__ __launch_bounds__(256) void k_lap_up(UpArgs a) {
    __shared__ float s_g[US_H][US_W + 1], s_c[US_H][US_W + 1];
    __shared__ float h_g[US_H][UT_W + 1], h_c[US_H][UT_W + 1];
    const Bid3 bq = xcd_swizzle3();
    const int x0 = bq.x * UT_W, y0 = bq.y * UT_H;
    const int sx0 = x0 / 2 - 1, sy0 = y0 / 2 - 1;
    const size_t pn = (size_t)bq.z * a.wn * a.hn, pl = (size_t)bq.z * a.w * a.h;
    const bool has_cur = !SEED && a.curn != nullptr;
    constexpr int NP = UT_H * UT_W / 256;                 // pixels per thread
    constexpr int NS = (US_H * US_W + 255) / 256;         // staged source elements per thread
    float hi_r[NP], lo_r[NP];
    unsigned idx_r[NP];                                   // offsets inside the plane (uniform plane/frame bases stay scalar)
    bool ok[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int i = threadIdx.x + k * 256;
        const int y = i / UT_W, x = i - y * UT_W;
        const int gx = x0 + x, gy = y0 + y;
        ok[k] = gx < a.w && gy < a.h;
        idx_r[k] = (unsigned)((ok[k] ? gy : 0) * a.w + (ok[k] ? gx : 0));
        hi_r[k] = lo_r[k] = 0.f;
        if (!SEED && ok[k]) { hi_r[k] = (a.hi + pl)[idx_r[k]]; lo_r[k] = (a.lo + pl)[idx_r[k]]; }
    }
    // source-tile elements this thread stages (same for every frame): vertical border map row -1 -> 1,
    // row >= hn -> hn-1; columns clamped (the border columns are handled by pyrup_h's formulas)
    unsigned soff[NS]; int sdst[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int i = threadIdx.x + k * 256;
        sdst[k] = -1; soff[k] = 0;
        if (i < US_H * US_W) {
            const int ly = i / US_W, lx = i - ly * US_W;
            int gy = sy0 + ly; gy = gy < 0 ? 1 : (gy >= a.hn ? a.hn - 1 : gy);
            int gx = sx0 + lx; gx = gx < 0 ? 0 : (gx >= a.wn ? a.wn - 1 : gx);
            soff[k] = (unsigned)(gy * a.wn + gx);
            sdst[k] = ly * (US_W + 1) + lx;
        }
    }
    // software pipeline over the frames, D deep: the global loads of frames t+1..t+D are in flight while
    // frame t is filtered (register slot t % D holds a frame until it is staged into LDS).  The coarse
    // levels have too few workgroups to hide a ~2 us load behind other waves, so their launches use a
    // deeper ring; the arithmetic is the same for every D.
    // (the launch code picks a D that divides nt; refills past the last frame re-read the last frame, so
    //  every slot is refilled unconditionally and the ring stays in fixed registers)
    float rg[D][NS], rc[D][NS], gl[D][NP];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float* Gn = a.Gn + pn + (size_t)d * a.fsn;
        const float* Cn = has_cur ? a.curn + pn + (size_t)d * a.fsn : nullptr;
        const float* Gl = a.Gl + pl + (size_t)d * a.fsl;
#pragma unroll
        for (int k = 0; k < NS; ++k) { rg[d][k] = Gn[soff[k]]; rc[d][k] = has_cur ? Cn[soff[k]] : 0.f; }
#pragma unroll
        for (int k = 0; k < NP; ++k) gl[d][k] = ld_stream_f32(Gl + idx_r[k]);   // (last reader of G_l: streaming load)
    }
    for (int t0 = 0; t0 < a.nt; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int t = t0 + d;
            // (no barrier needed here: the staging arrays were last read before the mid barrier of frame
            //  t-1, and the h arrays are rewritten only after the barrier below)
#pragma unroll
            for (int k = 0; k < NS; ++k)
                if (sdst[k] >= 0) { (&s_g[0][0])[sdst[k]] = rg[d][k]; if (has_cur) (&s_c[0][0])[sdst[k]] = rc[d][k]; }
            float glc[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) glc[k] = gl[d][k];
            __syncthreads();
            {
                const int tn = t + D < a.nt ? t + D : a.nt - 1;
                const float* Gn = a.Gn + pn + (size_t)tn * a.fsn;
                const float* Cn = has_cur ? a.curn + pn + (size_t)tn * a.fsn : nullptr;
                const float* Gl = a.Gl + pl + (size_t)tn * a.fsl;
#pragma unroll
                for (int k = 0; k < NS; ++k) { rg[d][k] = Gn[soff[k]]; rc[d][k] = has_cur ? Cn[soff[k]] : 0.f; }
#pragma unroll
                for (int k = 0; k < NP; ++k) gl[d][k] = ld_stream_f32(Gl + idx_r[k]);   // (last reader of G_l: streaming load)
            }
            pyrup_hpass(h_g, s_g, x0, sx0, a.wn, a.w);
            if (has_cur) pyrup_hpass(h_c, s_c, x0, sx0, a.wn, a.w);
            __syncthreads();
            float* cur = a.cur + pl + (size_t)t * a.fsl;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (!ok[k]) continue;
                const int i = threadIdx.x + k * 256;
                const int y = i / UT_W, x = i - y * UT_W;
                const int gy = y0 + y;
                const float band = glc[k] - pyrup_v(h_g, x, gy, sy0);           // SpatialFilter.cpp:33
                if (SEED) {
                    hi_r[k] = band; lo_r[k] = band;
                } else {
                    const float t1 = hi_r[k] * a.aHi + band * a.bHi;            // TemporalFilter.cpp:16
                    const float t2 = lo_r[k] * a.aLo + band * a.bLo;            // :17
                    hi_r[k] = t1; lo_r[k] = t2;
                    const float m = (t1 - t2) * a.gain;                         // :21, MagnifyCore.hpp:129-132
                    const float up = has_cur ? pyrup_v(h_c, x, gy, sy0) : 0.f;
                    cur[idx_r[k]] = up + m;                                     // SpatialFilter.cpp:58
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k)
        if (ok[k]) { (a.hi + pl)[idx_r[k]] = hi_r[k]; (a.lo + pl)[idx_r[k]] = lo_r[k]; }
}

// pyrUp horizontal pass for 4 consecutive destination columns gx0..gx0+3 (gx0 even) from the source
// values s[i0-1..i0+2], i0 = gx0/2 (indices outside the plane are never used by the border rules)
__device__ __forceinline__ float4 pyrup_h4(float sm1, float s0, float s1, float s2, int i0, int sw) {
    // every border variant is evaluated and the result selected (v_cndmask): divergent branches around
    // two or three float operations cost far more than the operations
    const bool f0 = i0 == 0, l0 = i0 == sw - 1, l1 = i0 + 1 == sw - 1;
    const float p6 = s0 * 6.f, q6 = s1 * 6.f;
    const float e_gen = sm1 + p6 + s1, e_first = p6 + s1 * 2.f, e_last = sm1 + s0 * 7.f;
    const float o_gen = (s0 + s1) * 4.f, o_last = s0 * 8.f;
    const float z_gen = s0 + q6 + s2, z_last = s0 + s1 * 7.f;      // i0 + 1 >= 1 always
    const float w_gen = (s1 + s2) * 4.f, w_last = s1 * 8.f;
    float4 o;
    o.x = sel(f0, e_first, sel(l0, e_last, e_gen));
    o.y = sel(l0, o_last, o_gen);
    o.z = sel(l1, z_last, z_gen);
    o.w = sel(l1, w_last, w_gen);
    return o;
}

// The same fused step without LDS and without barriers (levels whose width is a multiple of W, steady
// state).  A lane owns a block of W x 2 pixels of one plane for all nt frames: it keeps their two
// low-pass states in registers, computes the horizontal pyrUp pass of the three rows of G_{l+1} and
// cur_{l+1} it needs straight from global memory and reads / writes level l as 4W-byte vectors.  The raw
// loads of the next D frames are kept in flight in a register ring (D divides nt; refills past the
// last frame re-read the last frame so that every slot is refilled unconditionally).
// Used with W = 2 for the coarse levels, where a launch is only a few hundred waves and the length of
// the dependent instruction stream per frame, not throughput, sets the time (W = 4 measured slower
// everywhere: 167 registers at ring depth 2).
template <int W> struct UpRaw { float g[3][W / 2 + 2]; float c[3][W / 2 + 2]; float gl[2][W]; };
template <int W, int D, bool HC>                        // HC: cur_{l+1} exists (every level but the top live one)
__global__ __launch_bounds__(256) void k_lap_up_rows(UpArgs a, int gw, int ngroups) {
    constexpr int NT = W / 2 + 2;                        // source taps per row: columns i0-1 .. i0+W/2
    const int gi = blockIdx.x * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    const int plane = blockIdx.y;
    const int gy = gi / gw, gxg = gi - gy * gw;
    const int gx = gxg * W, y0 = gy * 2;
    constexpr bool has_cur = HC;
    const bool row1 = y0 + 1 < a.h;                      // the second row of the block exists
    const size_t pn = (size_t)plane * a.wn * a.hn, pl = (size_t)plane * a.w * a.h;
    const int i0 = gx >> 1, j0 = y0 >> 1;
    unsigned soff[3][NT];                                // byte offsets of the taps inside the level-(l+1) plane
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        int sy = j0 - 1 + q; sy = sy < 0 ? 1 : (sy >= a.hn ? a.hn - 1 : sy);       // vertical border map
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            int sx = i0 - 1 + k; sx = sx < 0 ? 0 : (sx >= a.wn ? a.wn - 1 : sx);
            soff[q][k] = 4u * (unsigned)(sy * a.wn + sx);
        }
    }
    const unsigned loff0 = 4u * (unsigned)(y0 * a.w + gx), loff1 = row1 ? loff0 + 4u * (unsigned)a.w : loff0;
    float hi_r[2][W], lo_r[2][W];
    {
        const char* H = reinterpret_cast<const char*>(a.hi + pl); const char* Lo = reinterpret_cast<const char*>(a.lo + pl);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            hi_r[0][k] = *reinterpret_cast<const float*>(H + loff0 + 4 * k); lo_r[0][k] = *reinterpret_cast<const float*>(Lo + loff0 + 4 * k);
            hi_r[1][k] = *reinterpret_cast<const float*>(H + loff1 + 4 * k); lo_r[1][k] = *reinterpret_cast<const float*>(Lo + loff1 + 4 * k);
        }
    }
    // frame cursors of the ring refills (uniform pointers, advanced by one frame per refill and parked on
    // the last frame)
    const char* Gn = reinterpret_cast<const char*>(a.Gn + pn);
    const char* Cn = reinterpret_cast<const char*>((has_cur ? a.curn : a.Gn) + pn);
    const char* Gl = reinterpret_cast<const char*>(a.Gl + pl);
    int tl = 0;
    auto load = [&](UpRaw<W>& r) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                r.g[q][k] = *reinterpret_cast<const float*>(Gn + soff[q][k]);
                r.c[q][k] = has_cur ? *reinterpret_cast<const float*>(Cn + soff[q][k]) : 0.f;
            }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            r.gl[0][k] = *reinterpret_cast<const float*>(Gl + loff0 + 4 * k);
            r.gl[1][k] = *reinterpret_cast<const float*>(Gl + loff1 + 4 * k);
        }
        if (tl + 1 < a.nt) { ++tl; Gn += a.fsn * sizeof(float); Cn += a.fsn * sizeof(float); Gl += a.fsl * sizeof(float); }
    };
    // horizontal pyrUp pass of one source row for the W destination columns gx .. gx+W-1 (all border
    // variants evaluated, result selected: no divergent branches)
    auto hpass = [&](const float (&sv)[NT], float (&o)[W]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < W / 2; ++k) {
            const int i = i0 + k;                        // source column of the destination pair (2i, 2i+1)
            const bool fi = i == 0, la = i == a.wn - 1;
            const float sm1 = sv[k], s0 = sv[k + 1], s1 = sv[k + 2];
            const float p6 = s0 * 6.f;
            const float e_gen = sm1 + p6 + s1, e_first = p6 + s1 * 2.f, e_last = sm1 + s0 * 7.f;
            const float o_gen = (s0 + s1) * 4.f, o_last = s0 * 8.f;
            o[2 * k] = sel(fi, e_first, sel(la, e_last, e_gen));
            o[2 * k + 1] = sel(la, o_last, o_gen);
        }
    };
    char* cur = reinterpret_cast<char*>(a.cur + pl);
    auto filter = [&](const UpRaw<W>& r) __attribute__((always_inline)) {
        float hg[3][W], hc[3][W];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            hpass(r.g[q], hg[q]);
            if (has_cur) hpass(r.c[q], hc[q]);
        }
        float o[2][W];
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const float upg = y == 0 ? (hg[0][k] + hg[1][k] * 6.f + hg[2][k]) * (1.f / 64.f) : ((hg[1][k] + hg[2][k]) * 4.f) * (1.f / 64.f);
                const float band = r.gl[y][k] - upg;                             // SpatialFilter.cpp:33
                const float t1 = hi_r[y][k] * a.aHi + band * a.bHi;              // TemporalFilter.cpp:16
                const float t2 = lo_r[y][k] * a.aLo + band * a.bLo;              // :17
                hi_r[y][k] = t1; lo_r[y][k] = t2;
                const float m = (t1 - t2) * a.gain;                              // :21, MagnifyCore.hpp:129-132
                const float up = has_cur ? (y == 0 ? (hc[0][k] + hc[1][k] * 6.f + hc[2][k]) * (1.f / 64.f) : ((hc[1][k] + hc[2][k]) * 4.f) * (1.f / 64.f)) : 0.f;
                o[y][k] = up + m;                                                // SpatialFilter.cpp:58
            }
#pragma unroll
        for (int k = 0; k < W; ++k) *reinterpret_cast<float*>(cur + loff0 + 4 * k) = o[0][k];
        if (row1) {
#pragma unroll
            for (int k = 0; k < W; ++k) *reinterpret_cast<float*>(cur + loff1 + 4 * k) = o[1][k];
        }
        cur += a.fsl * sizeof(float);
    };
    UpRaw<W> ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) load(ring[d]);
    for (int t0 = 0; t0 < a.nt; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const UpRaw<W> now = ring[d];
            load(ring[d]);
            filter(now);
        }
    }
    {
        char* H = reinterpret_cast<char*>(a.hi + pl); char* Lo = reinterpret_cast<char*>(a.lo + pl);
#pragma unroll
        for (int k = 0; k < W; ++k) { *reinterpret_cast<float*>(H + loff0 + 4 * k) = hi_r[0][k]; *reinterpret_cast<float*>(Lo + loff0 + 4 * k) = lo_r[0][k]; }
        if (row1) {
#pragma unroll
            for (int k = 0; k < W; ++k) { *reinterpret_cast<float*>(H + loff1 + 4 * k) = hi_r[1][k]; *reinterpret_cast<float*>(Lo + loff1 + 4 * k) = lo_r[1][k]; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Temporal batches, levels 2 .. L-1: the level-by-level chain of fused launches (each a few hundred waves walking
// over the frames of the batch, i.e. the length of the dependent instruction stream times the number of levels) is
// cut in two:
//   k_lap_iir_levels   ONE launch for all those levels.  m_l(t) = gain_l * (hi_l(t) - lo_l(t)) depends on G_l and
//                      G_{l+1} only, not on the other levels' results, so every level runs side by side: a lane owns
//                      a 2 x 2 block of one (level, plane) for all frames, states in registers, raw loads of the next
//                      D frames in flight (as k_lap_up_rows).  m_l is stored where cur_l will be.
//   k_lap_collapse     cur_l = pyrUp(cur_{l+1}) + m_l for l = L-2 .. 2 is stateless: ONE launch, a workgroup per
//                      (frame, plane, 64 x 32 tile of level 2), the few coarse pixels a tile depends on recomputed in
//                      LDS (34 x 18, 19 x 11, 12 x 8 .. per level).
// Same arithmetic and operation order as k_lap_up / k_lap_tail, hence the same frames.
// ------------------------------------------------------------------------------------------
constexpr int kIirLevels = 10;
struct IirLevel {
    const float* Gl; const float* Gn; float* hi; float* lo; float* out;
    int w, h, wn, hn; float gain; long fsl, fsn; int block0, gw, ngroups, top;
};
struct IirArgs { IirLevel lv[kIirLevels]; int nlv, nt; float aHi, bHi, aLo, bLo; };

template <int D>
__global__ __launch_bounds__(256) void k_lap_iir_levels(IirArgs aa) {
    int k = 0;
    while (k + 1 < aa.nlv && (int)blockIdx.x >= aa.lv[k + 1].block0) ++k;
    const IirLevel& a = aa.lv[k];
    const int gi = ((int)blockIdx.x - a.block0) * 256 + threadIdx.x;
    if (gi >= a.ngroups) return;
    const int plane = blockIdx.y;
    const int gy = gi / a.gw, gxg = gi - gy * a.gw;
    const int gx = gxg * 2, y0 = gy * 2;
    const bool row1 = y0 + 1 < a.h;
    const size_t pn = (size_t)plane * a.wn * a.hn, pl = (size_t)plane * a.w * a.h;
    const int i0 = gx >> 1, j0 = y0 >> 1;
    unsigned soff[3][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        int sy = j0 - 1 + q; sy = sy < 0 ? 1 : (sy >= a.hn ? a.hn - 1 : sy);       // vertical border map
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int sx = i0 - 1 + c; sx = sx < 0 ? 0 : (sx >= a.wn ? a.wn - 1 : sx);
            soff[q][c] = 4u * (unsigned)(sy * a.wn + sx);
        }
    }
    const unsigned loff0 = 4u * (unsigned)(y0 * a.w + gx), loff1 = row1 ? loff0 + 4u * (unsigned)a.w : loff0;
    float hi_r[2][2], lo_r[2][2];
    {
        const char* H = reinterpret_cast<const char*>(a.hi + pl); const char* Lo = reinterpret_cast<const char*>(a.lo + pl);
        const float2 h0 = *reinterpret_cast<const float2*>(H + loff0), h1 = *reinterpret_cast<const float2*>(H + loff1);
        const float2 l0 = *reinterpret_cast<const float2*>(Lo + loff0), l1 = *reinterpret_cast<const float2*>(Lo + loff1);
        hi_r[0][0] = h0.x; hi_r[0][1] = h0.y; hi_r[1][0] = h1.x; hi_r[1][1] = h1.y;
        lo_r[0][0] = l0.x; lo_r[0][1] = l0.y; lo_r[1][0] = l1.x; lo_r[1][1] = l1.y;
    }
    struct Raw { float g[3][3]; float2 gl[2]; };
    const char* Gn = reinterpret_cast<const char*>(a.Gn + pn);
    const char* Gl = reinterpret_cast<const char*>(a.Gl + pl);
    int tl = 0;
    auto load = [&](Raw& r) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) r.g[q][c] = *reinterpret_cast<const float*>(Gn + soff[q][c]);
        r.gl[0] = *reinterpret_cast<const float2*>(Gl + loff0);
        r.gl[1] = *reinterpret_cast<const float2*>(Gl + loff1);
        if (tl + 1 < aa.nt) { ++tl; Gn += a.fsn * sizeof(float); Gl += a.fsl * sizeof(float); }
    };
    const bool fi = i0 == 0, la = i0 == a.wn - 1;
    char* out = reinterpret_cast<char*>(a.out + pl);
    auto filter = [&](const Raw& r) __attribute__((always_inline)) {
        float hg[3][2];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float sm1 = r.g[q][0], s0 = r.g[q][1], s1 = r.g[q][2];
            const float p6 = s0 * 6.f;
            hg[q][0] = sel(fi, p6 + s1 * 2.f, sel(la, sm1 + s0 * 7.f, sm1 + p6 + s1));
            hg[q][1] = sel(la, s0 * 8.f, (s0 + s1) * 4.f);
        }
        float o[2][2];
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float upg = y == 0 ? (hg[0][c] + hg[1][c] * 6.f + hg[2][c]) * (1.f / 64.f) : ((hg[1][c] + hg[2][c]) * 4.f) * (1.f / 64.f);
                const float glv = y == 0 ? (c == 0 ? r.gl[0].x : r.gl[0].y) : (c == 0 ? r.gl[1].x : r.gl[1].y);
                const float band = glv - upg;                                    // SpatialFilter.cpp:33
                const float t1 = hi_r[y][c] * aa.aHi + band * aa.bHi;            // TemporalFilter.cpp:16
                const float t2 = lo_r[y][c] * aa.aLo + band * aa.bLo;            // :17
                hi_r[y][c] = t1; lo_r[y][c] = t2;
                const float m = (t1 - t2) * a.gain;                              // :21, MagnifyCore.hpp:129-132
                o[y][c] = a.top ? 0.f + m : m;                                   // top live level: pyrUp of the zeroed residual + m
            }
        *reinterpret_cast<float2*>(out + loff0) = make_float2(o[0][0], o[0][1]);
        if (row1) *reinterpret_cast<float2*>(out + loff1) = make_float2(o[1][0], o[1][1]);
        out += a.fsl * sizeof(float);
    };
    Raw ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) load(ring[d]);
    for (int t0 = 0; t0 < aa.nt; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const Raw now = ring[d];
            load(ring[d]);
            filter(now);
        }
    }
    {
        char* H = reinterpret_cast<char*>(a.hi + pl); char* Lo = reinterpret_cast<char*>(a.lo + pl);
        *reinterpret_cast<float2*>(H + loff0) = make_float2(hi_r[0][0], hi_r[0][1]);
        *reinterpret_cast<float2*>(Lo + loff0) = make_float2(lo_r[0][0], lo_r[0][1]);
        if (row1) {
            *reinterpret_cast<float2*>(H + loff1) = make_float2(hi_r[1][0], hi_r[1][1]);
            *reinterpret_cast<float2*>(Lo + loff1) = make_float2(lo_r[1][0], lo_r[1][1]);
        }
    }
}

constexpr int CT_W = 64, CT_H = 32, CP_R = 36, kCollapsePool = 4096;      // CP_R: row pitch of the coarser levels' regions (<= 34 columns)
struct CollapseArgs { float* cur[kIirLevels]; int w[kIirLevels], h[kIirLevels]; int nlv; };   // index 0 = level 2, nlv - 1 = top live level
// Worst-case LDS footprint of k_lap_collapse<NLV>: a region of r rows depends on at most r / 2 + 3 rows of the level above
// (rows [y0 / 2 - 1, y1 / 2 + 1], one more when y0 is odd); the regions of levels 1 .. NLV - 1 at pitch CP_R, then the
// horizontal-pass rows of the largest source region at pitch CT_W.
constexpr int collapse_pool_floats(int nlv) {
    int r = CT_H, total = 0, first = 0;
    for (int k = 1; k < nlv; ++k) { r = r / 2 + 3; total += CP_R * r; if (k == 1) first = r; }
    return total + first * CT_W;
}
static_assert(collapse_pool_floats(kIirLevels) <= kCollapsePool, "k_lap_collapse: LDS pool too small for CT_W / CT_H / CP_R / kIirLevels");
static_assert(CT_W / 2 + 2 <= CP_R, "k_lap_collapse: a region row of the level above does not fit the row pitch");
template <int NLV>                                      // number of levels (compile time: the per-level region scalars stay in SGPRs)
__global__ __launch_bounds__(256) void k_lap_collapse(CollapseArgs a) {
    __shared__ float pool[kCollapsePool];
    const int tid = threadIdx.x;
    const size_t z = blockIdx.z;                         // frame * planes + plane
    // regions of the levels a tile of level 2 depends on: columns / rows [x0/2 - 1, x1/2 + 1] of the level above, clipped
    // (uniform values: scalar registers)
    int rx0[NLV], ry0[NLV], rw[NLV], rh[NLV], roff[NLV];
    int toff;
    {
        int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
        int x1 = x0 + CT_W - 1 < a.w[0] - 1 ? x0 + CT_W - 1 : a.w[0] - 1, y1 = y0 + CT_H - 1 < a.h[0] - 1 ? y0 + CT_H - 1 : a.h[0] - 1;
        rx0[0] = x0; ry0[0] = y0; rw[0] = x1 - x0 + 1; rh[0] = y1 - y0 + 1; roff[0] = 0;
        int off = 0;
#pragma unroll
        for (int k = 1; k < NLV; ++k) {
            {
                int nx0 = x0 / 2 - 1, nx1 = x1 / 2 + 1, ny0 = y0 / 2 - 1, ny1 = y1 / 2 + 1;
                nx0 = nx0 < 0 ? 0 : nx0; ny0 = ny0 < 0 ? 0 : ny0;
                nx1 = nx1 > a.w[k] - 1 ? a.w[k] - 1 : nx1; ny1 = ny1 > a.h[k] - 1 ? a.h[k] - 1 : ny1;
                rx0[k] = nx0; ry0[k] = ny0; rw[k] = nx1 - nx0 + 1; rh[k] = ny1 - ny0 + 1; roff[k] = off;
                off += CP_R * rh[k];
                x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1;
            }
        }
        toff = off;                                      // temporaries (horizontal passes) start here
    }
    // ONE round of global loads: the m regions of every coarser level into LDS, the tile's own m_2 into registers.
    // Thread (tx, ty) = (tid & 63, tid >> 6) owns column tx of every region and the rows ty, ty + 4, ..: no index
    // division anywhere, the column's border case and (rows advancing by 4) the row parity are fixed per thread.
    // LDS rows have constant pitches (CP_R for the regions, CT_W for the horizontal-pass rows) and the global
    // accesses walk row pointers: no per-element integer multiply.
    const int tx = tid & 63, ty = tid >> 6;
    constexpr int NP = CT_H / 4;
    float m2[NP];
    float* g2 = a.cur[0] + z * ((size_t)a.w[0] * a.h[0]);
    const bool in0 = tx < rw[0];
    const int w0 = a.w[0];
    float* gp = g2 + ((ry0[0] + ty * NP) * w0 + rx0[0] + tx);          // this thread's first level-2 pixel
#pragma unroll
    for (int q = 0; q < NP; ++q) {                       // level 2: this thread's rows are the NP consecutive ones ty * NP ..
        m2[q] = (in0 && ty * NP + q < rh[0]) ? gp[q * w0] : 0.f;
    }
#pragma unroll
    for (int k = 1; k < NLV; ++k) {
        const int wk = a.w[k];
        const float* src = a.cur[k] + z * ((size_t)wk * a.h[k]) + ((ry0[k] + ty) * wk + rx0[k] + tx);
        float* dst = pool + roff[k] + ty * CP_R + tx;
        if (tx < rw[k])
            for (int y = ty; y < rh[k]; y += 4) { *dst = *src; dst += 4 * CP_R; src += 4 * wk; }
    }
    __syncthreads();
    float* tmp = pool + toff;
#pragma unroll
    for (int k = NLV - 2; k >= 0; --k) {
        const int sw = a.w[k + 1], sh = a.h[k + 1];
        const float* srcn = pool + roff[k + 1];
        const int nh = rh[k + 1], nx0 = rx0[k + 1], ny0 = ry0[k + 1];
        const int cw = rw[k], chh = rh[k], cx0 = rx0[k], cy0 = ry0[k];
        const bool inx = tx < cw;
        {   // horizontal pass of the region rows of level k+1 for this thread's column (pyrup_h with the column's case hoisted)
            const int gx = cx0 + tx, i = gx >> 1, li = i - nx0;
            const bool fi = i == 0, la = i == sw - 1, even = (gx & 1) == 0;
            const int lm = fi ? li : li - 1, lp = la ? li : li + 1;
            if (inx) {
                const float* sr = srcn + ty * CP_R;
                float* tr = tmp + ty * CT_W + tx;
                for (int r = ty; r < nh; r += 4) {
                    const float sm1 = sr[lm], s0 = sr[li], s1 = sr[lp];
                    const float p6 = s0 * 6.f;
                    const float ev = sel(fi, p6 + s1 * 2.f, sel(la, sm1 + s0 * 7.f, sm1 + p6 + s1));
                    const float od = sel(la, s0 * 8.f, (s0 + s1) * 4.f);
                    *tr = sel(even, ev, od);
                    sr += 4 * CP_R; tr += 4 * CT_W;
                }
            }
        }
        __syncthreads();
        // vertical pass + add: rows ty, ty + 4, .. of the region (cy0 is even or the region starts at an odd row: parity per row)
        if (k > 0) {
            float* dst = pool + roff[k];                 // holds m_k, becomes cur_k
            if (inx)
                for (int y = ty; y < chh; y += 4) {
                    const int gy = cy0 + y, j = gy >> 1;
                    const int jm = (j == 0 ? 1 : j - 1) - ny0, jj = j - ny0, jp = (j == sh - 1 ? sh - 1 : j + 1) - ny0;
                    const float up = ((gy & 1) == 0) ? (tmp[jm * CT_W + tx] + tmp[jj * CT_W + tx] * 6.f + tmp[jp * CT_W + tx]) * (1.f / 64.f)
                                                     : ((tmp[jj * CT_W + tx] + tmp[jp * CT_W + tx]) * 4.f) * (1.f / 64.f);
                    dst[y * CP_R + tx] = up + dst[y * CP_R + tx];                // SpatialFilter.cpp:58
                }
            __syncthreads();
        } else {
            // NP consecutive rows per thread (the tile starts on an even row): the NP / 2 + 2 source rows they need are
            // read once; a row index beyond the last source row is that row (pyrUp's bottom rule), row -1 is row 1
            if (inx) {
                const int jb = (cy0 >> 1) + ty * (NP / 2);
                float R[NP / 2 + 2];
                R[0] = tmp[((jb == 0 ? 1 : jb - 1) - ny0) * CT_W + tx];
#pragma unroll
                for (int i = 0; i <= NP / 2; ++i) {
                    const int jr = jb + i < sh - 1 ? jb + i : sh - 1;
                    R[i + 1] = tmp[(jr - ny0) * CT_W + tx];
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int i = q >> 1;                // source row jb + i is R[i + 1]
                    const float up = (q & 1) == 0 ? (R[i] + R[i + 1] * 6.f + R[i + 2]) * (1.f / 64.f) : ((R[i + 1] + R[i + 2]) * 4.f) * (1.f / 64.f);
                    if (ty * NP + q < chh) gp[q * w0] = up + m2[q];
                }
            }
        }
    }
}

// Final level: out = u8(Lab2BGR(Lab(in) + [1, ca, ca] * pyrUp(cur_1))).  MOTION = false is
// the first frame / L == 1 case (motion image is identically zero).  Persistent workgroups
// walk over (stream, tile); the inverse-gamma spline table lives in LDS.
template <int C, bool MOTION, int FL>
__global__ __launch_bounds__(256) void k_lap_final(const uint8_t* __restrict__ in, long in_stride, long in_sstride,
                                                   uint8_t* __restrict__ out, long out_stride, long out_sstride,
                                                   int w, int h, const float* __restrict__ cur1, int w1, int h1,
                                                   LabCoef lab, float ca, int tiles_x, int tiles_y, int nstreams,
                                                   float* __restrict__ dbg, LabPlanes lp) {
    constexpr bool EXACT = fl_exact(FL);
    __shared__ __attribute__((aligned(16))) float s_igt[C == 3 ? 4096 : 4];
    __shared__ float s_gam[C == 3 && !fl_lut(FL) ? 256 : 1];
    __shared__ float s_c[C][US_H][US_W + 1];
    __shared__ float h_c[C][US_H][UT_W + 1];
    if (C == 3) { load_invgamma(s_igt, lab.invgamma); if (!fl_lut(FL)) load_gamma_u8(s_gam, lab.gamma_u8); }
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
                fetch_lab_px<FL>(src, in_stride, lp, (size_t)b * w * h, w, gy, gx, s_gam, lab, L, a, bb);
                if (MOTION) {
                    const float m0 = pyrup_v(h_c[0], x, gy, sy0);
                    const float m1 = pyrup_v(h_c[C > 1 ? 1 : 0], x, gy, sy0) * ca;   // MagnifyCore.hpp:143-144
                    const float m2 = pyrup_v(h_c[C > 2 ? 2 : 0], x, gy, sy0) * ca;
                    L = L + m0; a = a + m1; bb = bb + m2;                            // :148
                }
                float o0, o1, o2;
                lab_to_bgr<EXACT>(L, a, bb, EXACT ? lab.inv : lab.inv1024, s_igt, o0, o1, o2);                    // :152
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
// Vectorised first/last kernels (3-channel frames whose width, strides and base pointers are
// multiples of 4: every group of 4 pixels is then three aligned dwords).  Same arithmetic and
// operation order as k_down0 / k_lap_final; only the data movement differs: one 12-byte load or
// store per 4 pixels instead of 12 byte accesses, 128-bit LDS reads, no per-pixel index math.
// ------------------------------------------------------------------------------------------
// Persistent 512-thread workgroups; the two Lab tables are loaded into LDS once per workgroup and
// nothing else is shared: every WAVE walks over its own strips of 256 x `rows` output pixels, one
// lane per group of 4 pixels, with no barrier inside the loop.  A lane computes the horizontal pyrUp
// pass of the cur_1 rows it needs straight from global memory (cur_1 was just written by k_lap_up and
// sits in L2) and slides a three-row register window down the strip for the vertical pass; the row
// parity is uniform across the wave, so only the formula of that parity is executed.
#ifndef LVM_FIN_THREADS
#define LVM_FIN_THREADS 512
#endif
constexpr int FIN_THREADS = LVM_FIN_THREADS;
struct Row3 { float4 c[3]; };            // horizontal-pass results of one source row, 3 channels x 4 columns
#ifndef LVM_FIN_WAVES
#define LVM_FIN_WAVES 0          // minimum waves per SIMD asked of the register allocator (0: none; 121 VGPRs = 4 waves)
#endif
#if LVM_FIN_WAVES
#define LVM_FIN_BOUNDS __launch_bounds__(FIN_THREADS, LVM_FIN_WAVES)
#else
#define LVM_FIN_BOUNDS __launch_bounds__(FIN_THREADS)
#endif
struct FinArgs {
    const uint8_t* in; long in_stride, in_sstride;
    uint8_t* out; long out_stride, out_sstride;
    int w, h; const float* cur1; int w1, h1;
    LabCoef lab; float ca; int strips_x, strips_y, nstreams, rows;
    float* dbg; LabPlanes lp;
};
// One output row of 4 pixels: Lab(in) + [1, ca, ca] * motion -> Lab2BGR -> u8 (MagnifyCore.hpp:143-153).  m = the motion image of the
// row (EXACT: scaled by 1/64 as pyrUp does; otherwise the unscaled vertical sum, whose power-of-two scale `msc` is folded into the
// add -- fma(m, 2^-k, L) rounds exactly like L + m * 2^-k).  dbg_px: where the float pixels go (lvm_debug_keep_float) or null.
// STEPS (the default flavour without the float frame): s_igt points at the u8 step table instead of the spline (lab_to_u8)
template <bool MOTION, bool DBG, int FL>
__device__ __forceinline__ B96 lap_emit_row(const Raw4 pin, const float (&m)[3][4], const float msc, const float ca, const LabCoef& lab,
                                            const float* s_igt, const float* s_gam, float* dbg_px) {
    constexpr bool EXACT = fl_exact(FL);
    constexpr bool STEPS = fin_steps(FL, DBG);
    float L4[4], a4[4], b4[4];
    raw4_to_lab<FL>(pin, s_gam, lab, L4, a4, b4);
    if (STEPS) {
        uint32_t u[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float L = L4[k], a = a4[k], bb = b4[k];
            if (MOTION) {
                L = __builtin_fmaf(m[0][k], msc, L);
                a = __builtin_fmaf(m[1][k], msc * ca, a); bb = __builtin_fmaf(m[2][k], msc * ca, bb);
            }
            lab_to_u8(L, a, bb, lab.inv4096, reinterpret_cast<const uint2*>(s_igt), u[3 * k], u[3 * k + 1], u[3 * k + 2]);
        }
        return B96{lvm_pack_b4(u[0], u[1], u[2], u[3]), lvm_pack_b4(u[4], u[5], u[6], u[7]), lvm_pack_b4(u[8], u[9], u[10], u[11])};
    }
    float ov[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float o0, o1, o2;
        float L = L4[k], a = a4[k], bb = b4[k];
        if (MOTION) {
            if (EXACT) { L = L + m[0][k]; a = a + m[1][k] * ca; bb = bb + m[2][k] * ca; }
            else {
                L = __builtin_fmaf(m[0][k], msc, L);
                a = __builtin_fmaf(m[1][k], msc * ca, a); bb = __builtin_fmaf(m[2][k], msc * ca, bb);
            }
        }
        lab_to_bgr<EXACT>(L, a, bb, EXACT ? lab.inv : lab.inv1024, s_igt, o0, o1, o2);
        if (DBG && dbg_px) { float* d = dbg_px + k * 3; d[0] = o0; d[1] = o1; d[2] = o2; }
        if (EXACT) {
            ov[3 * k] = o0 * 255.0f + lab.a255; ov[3 * k + 1] = o1 * 255.0f + lab.a255; ov[3 * k + 2] = o2 * 255.0f + lab.a255;
        } else {   // fma(o, 255, 1/255) differs from mul + add only far below the rounding step
            ov[3 * k] = __builtin_fmaf(o0, 255.0f, lab.a255); ov[3 * k + 1] = __builtin_fmaf(o1, 255.0f, lab.a255);
            ov[3 * k + 2] = __builtin_fmaf(o2, 255.0f, lab.a255);
        }
    }
    return B96{pack_u8x4(ov[0], ov[1], ov[2], ov[3]), pack_u8x4(ov[4], ov[5], ov[6], ov[7]), pack_u8x4(ov[8], ov[9], ov[10], ov[11])};
}
// one wave strip (task) of the last kernel; s_igt = inverse-gamma spline in LDS, s_gam = gamma table (analytic flavour)
template <bool MOTION, bool DBG, int FL>     // DBG: also store the float frame (lvm_debug_keep_float); a per-pixel branch
__device__ __forceinline__ void lap_final_strip(const FinArgs& q, int task, int lane, const float* s_igt, const float* s_gam) {
    constexpr bool EXACT = fl_exact(FL);
    constexpr bool PLANES = fl_lut(FL);
    const uint8_t* __restrict__ in = q.in; const long in_stride = q.in_stride, in_sstride = q.in_sstride;
    uint8_t* __restrict__ out = q.out; const long out_stride = q.out_stride, out_sstride = q.out_sstride;
    const int w = q.w, h = q.h, w1 = q.w1, h1 = q.h1, strips_x = q.strips_x, strips_y = q.strips_y, rows = q.rows;
    const float* __restrict__ cur1 = q.cur1; const LabCoef& lab = q.lab; const float ca = q.ca;
    float* __restrict__ dbg = q.dbg; const LabPlanes lp = q.lp;
    const int b = task / (strips_x * strips_y);
    const int r = task - b * (strips_x * strips_y);
    const int ty = r / strips_x, tx = r - ty * strips_x;
    const int gx = tx * 256 + 4 * lane, y0 = ty * rows;        // rows is even: y0 is even
    if (gx >= w) return;
    // uniform (scalar) bases + 32-bit lane offsets
    const uint8_t* src = in + (size_t)b * in_sstride;
    const size_t poff = (size_t)b * w * h;
    uint8_t* dst = out + (size_t)b * out_sstride;
    const unsigned xoff = (unsigned)gx * 3u;
    const float* pl = cur1 + (size_t)b * 3 * ((size_t)w1 * h1);
    const int i0 = gx >> 1;
    // byte offsets of the four taps.  Default flavour: the border rule is applied by the LOADS (column -1 reads column 1,
    // column w1 reads column w1 - 1) and every lane runs the interior formulas -- s1 + 6 s0 + s1 = 6 s0 + 2 s1,
    // s0 + 6 s1 + s1 = s0 + 7 s1, (s1 + s1) 4 = 8 s1: equal up to the rounding of one addition at the two border
    // columns, 16 selects / border variants less per lane and source row.
    const unsigned cm1 = 4u * (i0 > 0 ? i0 - 1 : (EXACT ? 0 : 1)), c00 = 4u * i0, cp1 = 4u * (i0 + 1 < w1 ? i0 + 1 : w1 - 1), cp2 = 4u * (i0 + 2 < w1 ? i0 + 2 : w1 - 1);
    const size_t pstride = (size_t)w1 * h1;
    // Round 5: every access of the strip goes through a buffer resource (base + size in SGPRs, a 32-bit lane offset, a wave-uniform row
    // offset): no 64-bit address arithmetic in vector registers (the generic form cost ~55 v_lshl_add_u64 per three steps and 24
    // VGPRs of address pairs).  The three cur_1 planes of this frame are one resource, the integer planes and the output frame another each.
    const BufRsrc rcur = buf_rsrc(MOTION ? (const void*)pl : (const void*)in, MOTION ? (uint32_t)(3 * pstride * sizeof(float)) : 0u);
    // horizontal pass of source row sy (vertical border map: row -1 -> 1, row h1 -> h1 - 1); the row
    // base is uniform, the four column offsets are per-lane byte offsets
    auto hrow = [&](int sy) __attribute__((always_inline)) {
        Row3 o;
        sy = sy < 0 ? 1 : (sy >= h1 ? h1 - 1 : sy);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t rb = (uint32_t)(((size_t)sy * w1 + c * pstride) * sizeof(float));
            const float sm1 = buf_ld_f32(rcur, cm1, rb), s0 = buf_ld_f32(rcur, c00, rb), s1 = buf_ld_f32(rcur, cp1, rb), s2 = buf_ld_f32(rcur, cp2, rb);
            if (EXACT) o.c[c] = pyrup_h4(sm1, s0, s1, s2, i0, w1);
            else if (LVM_FAST_FMA) {   // one rounding less per even column (fma), a few 1e-8 of the motion image
                o.c[c].x = __builtin_fmaf(s0, 6.f, sm1 + s1); o.c[c].y = (s0 + s1) * 4.f; o.c[c].z = __builtin_fmaf(s1, 6.f, s0 + s2); o.c[c].w = (s1 + s2) * 4.f;
            } else { o.c[c].x = sm1 + s0 * 6.f + s1; o.c[c].y = (s0 + s1) * 4.f; o.c[c].z = s0 + s1 * 6.f + s2; o.c[c].w = (s1 + s2) * 4.f; }
        }
        return o;
    };
    const int yend = (y0 + rows < h) ? y0 + rows : h;
    int gy = y0, j = y0 >> 1;
    // the frame's integer planes (read once by this launch: streaming loads) / the u8 frame of the analytic flavour, and the output frame
    const BufRsrc rL = buf_rsrc(PLANES ? (const void*)(lp.iL + poff) : (const void*)src, PLANES ? (uint32_t)((size_t)w * h * 2) : 0u);
    const BufRsrc rAB = buf_rsrc(PLANES ? (const void*)(lp.iab + poff) : (const void*)src, PLANES ? (uint32_t)((size_t)w * h * 4) : 0u);
    const BufRsrc rout = buf_rsrc(dst, (uint32_t)((size_t)out_stride * (h - 1) + (size_t)w * 3));
    auto ld_in = [&](int y) __attribute__((always_inline)) {
        if (!PLANES) return load_raw4<false>(src, in_stride, lp, poff, w, y, (unsigned)gx);
        Raw4 r{};
        const uint2 l = buf_lds_u32x2(rL, (uint32_t)gx * 2u, (uint32_t)y * (uint32_t)w * 2u);
        const uint4 ab = buf_lds_u32x4(rAB, (uint32_t)gx * 4u, (uint32_t)y * (uint32_t)w * 4u);
        r.d[0] = l.x; r.d[1] = l.y; r.d[2] = ab.x; r.d[3] = ab.y; r.d[4] = ab.z; r.d[5] = ab.w;
        return r;
    };
    auto emit = [&](const Raw4 pin, const float (&m)[3][4], const float msc) __attribute__((always_inline)) {
        const B96 o = lap_emit_row<MOTION, DBG, FL>(pin, m, msc, ca, lab, s_igt, s_gam, (DBG && dbg && b == 0) ? dbg + ((size_t)gy * w + gx) * 3 : nullptr);
        buf_sts_b96(o, rout, xoff, (uint32_t)gy * (uint32_t)out_stride);      // (the output frame is not read again on the device: streaming store)
    };
    // two output rows (2j, 2j+1) from the window rows A = j-1, B = j, C = j+1; afterwards A holds row
    // j+2, i.e. the window has rotated to (B, C, A).  Returns false when the strip is finished.
    // (round 6, measured and NOT taken: the input rows of a step loaded one step ahead -- 12 more VGPRs, 3.5 % more instructions for the
    //  register rotation -- made the kernel SLOWER, 188-191 -> 199-202 us per 32 frames: it is not waiting for its own loads)
    auto step = [&](Row3& A, const Row3& B, const Row3& C) __attribute__((always_inline)) {
        const Raw4 pe = ld_in(gy);
        const bool has_odd = gy + 1 < yend;
        Raw4 po = pe;
        if (has_odd) po = ld_in(gy + 1);
        float m[3][4] = {};
        if (MOTION) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float sc = EXACT ? (1.f / 64.f) : 1.f;           // (x * 1.f folds away)
                if (!EXACT && LVM_FAST_FMA) {
                    m[c][0] = __builtin_fmaf(B.c[c].x, 6.f, A.c[c].x + C.c[c].x); m[c][1] = __builtin_fmaf(B.c[c].y, 6.f, A.c[c].y + C.c[c].y);
                    m[c][2] = __builtin_fmaf(B.c[c].z, 6.f, A.c[c].z + C.c[c].z); m[c][3] = __builtin_fmaf(B.c[c].w, 6.f, A.c[c].w + C.c[c].w);
                    continue;
                }
                m[c][0] = (A.c[c].x + B.c[c].x * 6.f + C.c[c].x) * sc; m[c][1] = (A.c[c].y + B.c[c].y * 6.f + C.c[c].y) * sc;
                m[c][2] = (A.c[c].z + B.c[c].z * 6.f + C.c[c].z) * sc; m[c][3] = (A.c[c].w + B.c[c].w * 6.f + C.c[c].w) * sc;
            }
        }
        const bool more = gy + 2 < yend;
        if (MOTION && more) A = hrow(j + 2);                 // in flight during the colour math of both rows
        emit(pe, m, 1.f / 64.f);
        ++gy;
        if (!has_odd) return false;
        if (MOTION) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (EXACT) {
                    m[c][0] = ((B.c[c].x + C.c[c].x) * 4.f) * (1.f / 64.f); m[c][1] = ((B.c[c].y + C.c[c].y) * 4.f) * (1.f / 64.f);
                    m[c][2] = ((B.c[c].z + C.c[c].z) * 4.f) * (1.f / 64.f); m[c][3] = ((B.c[c].w + C.c[c].w) * 4.f) * (1.f / 64.f);
                } else {
                    m[c][0] = B.c[c].x + C.c[c].x; m[c][1] = B.c[c].y + C.c[c].y; m[c][2] = B.c[c].z + C.c[c].z; m[c][3] = B.c[c].w + C.c[c].w;
                }
            }
        }
        emit(po, m, 1.f / 16.f);
        ++gy; ++j;
        return more;
    };
    Row3 r0{}, r1{}, r2{};
    if (MOTION) { r0 = hrow(j - 1); r1 = hrow(j); r2 = hrow(j + 1); }
    while (true) {
        if (!step(r0, r1, r2)) break;
        if (!step(r1, r2, r0)) break;
        if (!step(r2, r0, r1)) break;
    }
}
template <bool MOTION, bool DBG, int FL>
__global__ LVM_FIN_BOUNDS void k_lap_final_v4(FinArgs q) {
    constexpr bool STEPS = fin_steps(FL, DBG);
    __shared__ __attribute__((aligned(16))) float s_igt[STEPS ? 2 * kU8StepSlices : 4096];      // the inverse-gamma spline | the u8 step table
    __shared__ float s_gam[fl_lut(FL) ? 1 : 256];
    if (STEPS) load_u8steps(reinterpret_cast<uint2*>(s_igt), q.lab.u8steps);
    else {
        const float4* src = reinterpret_cast<const float4*>(q.lab.invgamma);
        for (int i = threadIdx.x; i < 1024; i += FIN_THREADS) reinterpret_cast<float4*>(s_igt)[i] = src[i];
        if (!fl_lut(FL) && threadIdx.x < 256) s_gam[threadIdx.x] = q.lab.gamma_u8[threadIdx.x];
    }
    __syncthreads();
    constexpr int WAVES = FIN_THREADS / 64;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int ntasks = q.strips_x * q.strips_y * q.nstreams;
    // (strip groups handed out round-robin: a contiguous run per workgroup in XCD-aware order -- the next group is the row of
    // strips below, whose first cur_1 rows the CU has just read -- measured 194-203 -> 222-226 us per 32 frames)
    for (int task = blockIdx.x * WAVES + wave; task < ntasks; task += gridDim.x * WAVES)
        lap_final_strip<MOTION, DBG, FL>(q, task, lane, s_igt, s_gam);
}


// ------------------------------------------------------------------------------------------
// Tail kernel: every pyramid level from T upwards (G_T has at most kTailMax pixels) is handled by
// ONE workgroup per plane, entirely in LDS: pyrDown T->T+1->..->L, then the fused band / IIR /
// gain / collapse steps for l = L-1..T.  Replaces 2(L-T) tiny launches, each of which is pure
// launch + latency (5-6 us on MI355X for a few hundred pixels).  Same arithmetic as k_pyr_down /
// k_lap_up.  LDS pool (floats): G_{T+1..L} | cur_{T+1..L-1} | tmpA | tmpB.
// ------------------------------------------------------------------------------------------
constexpr int kTailMax = 8192;          // pixels of G_T
constexpr int kTailLevels = 8;
constexpr int kTailPool = (kTailMax / 4 + kTailMax / 16 + kTailMax / 32 + 64) * 2 + kTailMax;   // see laplace_tail_plan
struct TailArgs {
    const float* GT;                    // G_T planes (global)
    float* curT;                        // cur_T planes (global, output)
    float* hi[kTailLevels]; float* lo[kTailLevels];   // states of levels T..L-1 (global planes)
    int w[kTailLevels + 1], h[kTailLevels + 1];       // geometry of levels T..L
    int offG[kTailLevels + 1], offC[kTailLevels + 1]; // LDS offsets of G_{T+k}, cur_{T+k} (k >= 1)
    int offA, offB;                     // two temporaries
    float gain[kTailLevels];
    float aHi, bHi, aLo, bLo;
    int n;                              // number of down steps = L - T (>= 1)
    int nt;                             // frames handled by this launch, in temporal order
    long fsT;                           // frame stride (floats) of the level-T arrays (G_T, cur_T)
};

__device__ __forceinline__ float tail_src(const float* s, int w, int h, int y, int x) { return s[(size_t)y * w + x]; }

// i = y * w + x for 0 <= i < 2^20, w < 2^12: (i + 0.5) / w is at least 0.5 / w away from an integer,
// far more than the float error of the product, so the truncation is exact
__device__ __forceinline__ int row_of(int i, float inv_w) { return (int)(((float)i + 0.5f) * inv_w); }

constexpr int TAIL_THREADS = 1024;
template <bool SEED>
__global__ __launch_bounds__(TAIL_THREADS) void k_lap_tail(TailArgs a) {
    __shared__ float pool[kTailPool];
    const int tid = threadIdx.x;
    const size_t plane = blockIdx.x;
    for (int t = 0; t < a.nt; ++t) {     // frames in temporal order; the states of frame t-1 were written by this workgroup
    const float* GTt = a.GT + (size_t)t * a.fsT;
    float* curTt = a.curT + (size_t)t * a.fsT;
    // ---- down sweep ----
    for (int k = 0; k < a.n; ++k) {
        const int w = a.w[k], h = a.h[k], dw = a.w[k + 1], dh = a.h[k + 1];
        const float inv_dw = 1.0f / (float)dw;
        const float* __restrict__ src = (k == 0) ? GTt + plane * ((size_t)w * h) : pool + a.offG[k];
        float* __restrict__ tmp = pool + a.offA;                       // h x dw horizontal results
#pragma unroll 4
        for (int i = tid; i < h * dw; i += TAIL_THREADS) {
            const int y = row_of(i, inv_dw), x = i - y * dw;
            const float* s = src + (size_t)y * w;
            const int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = 2 * x,
                      x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
            tmp[i] = s[x2] * 6.f + (s[x1] + s[x3]) * 4.f + s[x0] + s[x4];
        }
        __syncthreads();
        float* __restrict__ dst = pool + a.offG[k + 1];
        for (int i = tid; i < dh * dw; i += TAIL_THREADS) {
            const int y = row_of(i, inv_dw), x = i - y * dw;
            const float r0 = tmp[reflect101(2 * y - 2, h) * dw + x], r1 = tmp[reflect101(2 * y - 1, h) * dw + x],
                        r2 = tmp[(2 * y) * dw + x], r3 = tmp[reflect101(2 * y + 1, h) * dw + x],
                        r4 = tmp[reflect101(2 * y + 2, h) * dw + x];
            dst[i] = (r2 * 6.f + (r1 + r3) * 4.f + r0 + r4) * (1.f / 256.f);
        }
        __syncthreads();
    }
    // ---- up sweep: level index k = n-1..0 (pyramid level T+k) ----
    for (int k = a.n - 1; k >= 0; --k) {
        const int w = a.w[k], h = a.h[k], sw = a.w[k + 1], sh = a.h[k + 1];
        const float inv_w = 1.0f / (float)w;
        const float* __restrict__ Gn = pool + a.offG[k + 1];
        const bool has_cur = !SEED && (k + 1 <= a.n - 1);
        const float* __restrict__ Cn = pool + a.offC[k + 1];
        float* __restrict__ tA = pool + a.offA;                        // sh x w horizontal pyrUp of G_{k+1}
        float* __restrict__ tB = pool + a.offB;                        // same for cur_{k+1}
        for (int i = tid; i < sh * w; i += TAIL_THREADS) {
            const int y = row_of(i, inv_w), x = i - y * w;
            tA[i] = pyrup_h(Gn + (size_t)y * sw, x, 0, sw);
            if (has_cur) tB[i] = pyrup_h(Cn + (size_t)y * sw, x, 0, sw);
        }
        __syncthreads();
        const float* __restrict__ Gl = (k == 0) ? GTt + plane * ((size_t)w * h) : pool + a.offG[k];
        float* __restrict__ hi = a.hi[k] + plane * ((size_t)w * h);
        float* __restrict__ lo = a.lo[k] + plane * ((size_t)w * h);
        float* __restrict__ cur = (k == 0) ? curTt + plane * ((size_t)w * h) : pool + a.offC[k];
#pragma unroll 4
        for (int i = tid; i < h * w; i += TAIL_THREADS) {
            const int y = row_of(i, inv_w), x = i - y * w;
            const int j = y >> 1;
            const int jm = j == 0 ? 1 : j - 1, jp = j == sh - 1 ? sh - 1 : j + 1;
            const float upG = ((y & 1) == 0) ? (tA[jm * w + x] + tA[j * w + x] * 6.f + tA[jp * w + x]) * (1.f / 64.f)
                                             : ((tA[j * w + x] + tA[jp * w + x]) * 4.f) * (1.f / 64.f);
            const float band = Gl[i] - upG;
            if (SEED) {
                hi[i] = band; lo[i] = band;
            } else {
                const float t1 = hi[i] * a.aHi + band * a.bHi;
                const float t2 = lo[i] * a.aLo + band * a.bLo;
                hi[i] = t1; lo[i] = t2;
                const float m = (t1 - t2) * a.gain[k];
                float up = 0.f;
                if (has_cur)
                    up = ((y & 1) == 0) ? (tB[jm * w + x] + tB[j * w + x] * 6.f + tB[jp * w + x]) * (1.f / 64.f)
                                        : ((tB[j * w + x] + tB[jp * w + x]) * 4.f) * (1.f / 64.f);
                cur[i] = up + m;
            }
        }
        __syncthreads();
    }
    }   // frames
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
    float* Gp[2][kMaxLevels + 1] = {};    // Gaussian pyramid, double-buffered by frame parity (pipelined mode)
    float* curT[2] = {};                  // cur_T written by the tail kernel, double-buffered likewise
    struct Pending { bool valid = false; FrameIO io{}; lvm_params p{}; int par = 0; } pending;
    int par = 0, depth = 0;
    // temporal batching (lvm_process_device_frames): pyramids / accumulators of up to tcap frames
    int tcap = 0; float* tarena = nullptr;
    float* Gt[kMaxLevels + 1] = {}; float* curt[kMaxLevels + 1] = {};
    // integer Lab planes of the input frames (labconv.hip; 3-channel frames): per frame parity and for temporal batches
    uint16_t* iLp[2] = {}; uint32_t* iabp[2] = {};
    uint16_t* iLt = nullptr; uint32_t* iabt = nullptr;
    int split_min_nt = 1;                 // frames per launch from which levels >= 2 run as IIR + collapse launches (LVM_LAP_SPLIT_MIN_NT)
    long fin_min_tasks = 4096;            // strips are shortened until a launch has this many of them (LVM_FIN_MIN_TASKS)
    long rows_min_elems = 2000000;        // planes x pixels from which pyrDown uses k_pyr_down_rows (LVM_ROWS_MIN_ELEMS): 32-frame batches keep the strips on
                                          // levels 1-3 (3.1 M at level 3), four streams per per-frame call take the two-level kernel from level 2 (one launch less)
    int split_from = 3;                   // first level of the IIR + collapse launches: 3 (round 6; with >= 5 levels) = level 2 as a fused step like level 1, its m_2 neither
                                          // written nor read back (34 + 41 -> 20 + 14 + 34 us per 32 frames at 1080p); LVM_LAP_SPLIT_FROM=2: levels 2 .. L-1 decoupled
    int split_levels = 1;                 // temporal batches: levels >= 2 as one IIR launch + one collapse launch (LVM_LAP_SPLIT=0: level-by-level chain)
    int up_rows = 1;                      // barrier-free k_lap_up_rows for the steady state (LVM_UP_ROWS=0: tiled k_lap_up)
    long up_rows_max_blocks = 1024;       // launches with fewer tiled workgroups than this use k_lap_up_rows (LVM_UP_ROWS_MAX_BLOCKS)
    bool d0_rows_on = true;               // wave-strip first kernel (LVM_D0_ROWS=0: LDS-tiled k_down0_v4 always)
    bool d0_fused = true;                 // table conversion fused into the first kernel on large launches (LVM_D0_FUSED=0: separate kernels)
    int d0_fused_rows = 0;                // force the fused first kernel with strips of this many rows (LVM_D0_FUSED_ROWS; measurement only)
    long d0_fused_groups = 0;             // persistent workgroups of the fused first kernel (LVM_D0_FUSED_GROUPS; 0 = one per CU)
    long d0_fused_waves = 0;              // ... = launches with at least this many strips (LVM_D0_FUSED_WAVES; 0 = one per resident wave)
    long d0_min_tasks = 4096;             // ... for launches with at least this many strips (LVM_D0_MIN_TASKS)
    long fin_groups = 0;                  // workgroups of the persistent last kernel (LVM_FIN_GROUPS; 0 = 1024)
    int fin_rows = 8;                     // rows per wave strip of k_lap_final_v4 (LVM_FIN_ROWS, power of two)
    int up_depth_big = 1;                 // ... at the levels with >= 1024 workgroups (LVM_UP_DEPTH_BIG)
    int up_depth = 8;                     // frame-loop prefetch depth of k_lap_up at the coarse levels (LVM_UP_DEPTH=1|2|4|8)
    int pd_rows = 16;                     // output rows per wave strip of k_pyr_down_rows (LVM_PD_ROWS)
    int fuse_down = 2;                    // pyramid levels per pyrDown launch (LVM_FUSE_DOWN=2|3 selects the fused kernels)
    int tailT = 0;                       // first level handled by k_lap_tail (0 = tail disabled)
    TailArgs tail{};
    ~LaplaceState() override {
        if (arena) (void)hipFree(arena);
        if (tarena) (void)hipFree(tarena);
    }
};

static void laplace_tail_plan(LaplaceState* st);
static int laplace_reserve_frames(Ctx* c, LaplaceState* st, int nt, hipStream_t s);

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
    for (int l = 1; l <= levels; ++l) total += 2 * pad(st->g[l].n * st->planes);
    for (int l = 1; l < levels; ++l) total += 5 * pad(st->g[l].n * st->planes);
    const size_t npx = st->g[0].n * c->nstreams;                 // pixels of one frame set
    if (channels == 3) total += 2 * (pad(npx) + pad((npx + 1) / 2));
    if (total == 0) total = 64;
    if (hipMalloc((void**)&st->arena, total * sizeof(float)) != hipSuccess) {
        c->err = "laplace: hipMalloc failed";
        st->arena = nullptr;
        return LVM_ERR_OOM;
    }
    float* p = st->arena;
    for (int q = 0; q < 2; ++q)
        for (int l = 1; l <= levels; ++l) { st->Gp[q][l] = p; p += pad(st->g[l].n * st->planes); }
    for (int l = 1; l <= levels; ++l) st->G[l] = st->Gp[0][l];
    if (channels == 3)
        for (int q = 0; q < 2; ++q) {
            st->iabp[q] = reinterpret_cast<uint32_t*>(p); p += pad(npx);
            st->iLp[q] = reinterpret_cast<uint16_t*>(p); p += pad((npx + 1) / 2);
        }
    for (int l = 1; l < levels; ++l) {
        st->hi[l] = p; p += pad(st->g[l].n * st->planes);
        st->lo[l] = p; p += pad(st->g[l].n * st->planes);
        st->cur[l] = p; p += pad(st->g[l].n * st->planes);
    }
    laplace_tail_plan(st);
    if (const char* e = std::getenv("LVM_FUSE_DOWN")) st->fuse_down = std::atoi(e);
    if (const char* e = std::getenv("LVM_UP_DEPTH")) st->up_depth = std::atoi(e);
    if (const char* e = std::getenv("LVM_UP_DEPTH_BIG")) st->up_depth_big = std::atoi(e);
    if (const char* e = std::getenv("LVM_PD_ROWS")) st->pd_rows = std::atoi(e);
    if (const char* e = std::getenv("LVM_D0_ROWS")) st->d0_rows_on = std::atoi(e) != 0;
    if (const char* e = std::getenv("LVM_D0_FUSED")) st->d0_fused = std::atoi(e) != 0;
    if (const char* e = std::getenv("LVM_D0_FUSED_WAVES")) st->d0_fused_waves = std::atol(e);
    if (const char* e = std::getenv("LVM_D0_FUSED_GROUPS")) st->d0_fused_groups = std::atol(e);
    if (const char* e = std::getenv("LVM_D0_FUSED_ROWS")) st->d0_fused_rows = std::atoi(e);
    if (const char* e = std::getenv("LVM_D0_MIN_TASKS")) st->d0_min_tasks = std::atol(e);
    if (const char* e = std::getenv("LVM_LAP_SPLIT")) st->split_levels = std::atoi(e);
    if (const char* e = std::getenv("LVM_LAP_SPLIT_MIN_NT")) st->split_min_nt = std::atoi(e);
    if (const char* e = std::getenv("LVM_UP_ROWS")) st->up_rows = std::atoi(e);
    if (const char* e = std::getenv("LVM_UP_ROWS_MAX_BLOCKS")) st->up_rows_max_blocks = std::atol(e);
    if (const char* e = std::getenv("LVM_ROWS_MIN_ELEMS")) st->rows_min_elems = std::atol(e);
    if (const char* e = std::getenv("LVM_LAP_SPLIT_FROM")) { const int v = std::atoi(e); if (v == 2 || v == 3) st->split_from = v; }
    if (const char* e = std::getenv("LVM_FIN_MIN_TASKS")) st->fin_min_tasks = std::atol(e);
    if (const char* e = std::getenv("LVM_FIN_GROUPS")) st->fin_groups = std::atol(e);
    if (const char* e = std::getenv("LVM_FIN_ROWS")) { const int v = std::atoi(e); if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32) st->fin_rows = v; }
    if (st->tailT) {
        st->curT[0] = p; p += pad(st->g[st->tailT].n * st->planes);
        st->curT[1] = p; p += pad(st->g[st->tailT].n * st->planes);
    }
    return LVM_OK;
}

// Decide which levels the tail kernel covers and lay out its LDS pool.
static void laplace_tail_plan(LaplaceState* st) {
    const int L = st->levels;
    st->tailT = 0;
    int T = 1;
    while (T <= L - 1 && st->g[T].n > (size_t)kTailMax) ++T;
    if (T > L - 1 || L - T > kTailLevels || L - T < 1) return;      // nothing small enough / too deep
    TailArgs& a = st->tail;
    a.n = L - T;
    int off = 0;
    for (int k = 0; k <= a.n; ++k) { a.w[k] = st->g[T + k].w; a.h[k] = st->g[T + k].h; }
    for (int k = 1; k <= a.n; ++k) { a.offG[k] = off; off += (int)st->g[T + k].n; }
    for (int k = 1; k <= a.n - 1; ++k) { a.offC[k] = off; off += (int)st->g[T + k].n; }
    a.offC[a.n] = 0; a.offG[0] = 0; a.offC[0] = 0;
    // temporaries: max over steps of h_k*w_{k+1} (down) and h_{k+1}*w_k (up); both <= n_T/2 + slack
    int tmp = 0;
    for (int k = 0; k < a.n; ++k) {
        const int d = a.h[k] * a.w[k + 1], u = a.h[k + 1] * a.w[k];
        tmp = d > tmp ? d : tmp; tmp = u > tmp ? u : tmp;
    }
    a.offA = off; off += tmp;
    a.offB = off; off += tmp;
    if (off > kTailPool) return;
    st->tailT = T;
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

static bool lap_vec4(const FrameIO& io) {   // 4-pixel (12-byte) vector I/O needs dword-aligned pixel groups
    return io.channels == 3 && io.w % 4 == 0 && io.in_stride % 4 == 0 && io.in_sstride % 4 == 0 &&
           io.out_stride % 4 == 0 && io.out_sstride % 4 == 0 && ((uintptr_t)io.d_in % 4) == 0 &&
           ((uintptr_t)io.d_out % 4) == 0;
}

// Stage B of a frame: u8 -> Lab -> Gaussian pyramid G_1..G_T (parity buffer `par`) and, when the tail
// kernel is enabled, everything that happens at the levels >= T (their IIR states, cur_T[par]).
struct LapBufs { float** G; float** cur; float* curT; int nt; bool no_tail = false; uint16_t* iL = nullptr; uint32_t* iab = nullptr; bool dbg_frame = true; };   // nt frames laid out [frame][stream][channel]
static LapBufs lap_bufs_frame(LaplaceState* st, int par) { return LapBufs{st->Gp[par], st->cur, st->curT[par], 1, false, st->iLp[par], st->iabp[par]}; }
// planes the kernels of a launch read (null: analytic flavour / gray frames)
static LabPlanes lap_planes(const Ctx* c, const FrameIO& io, const LapBufs& B) {
    return (io.channels == 3 && fl_lut(lab_flavour(c))) ? LabPlanes{B.iL, B.iab} : LabPlanes{nullptr, nullptr};
}

// Levels 2 .. L-1 decoupled into one IIR launch + one stateless collapse launch (k_lap_iir_levels / k_lap_collapse) instead
// of the level-by-level chain / the LDS-resident tail kernel?  Temporal batches of >= 4 frames always; single frames too
// (LVM_LAP_SPLIT_MIN_NT, default 1): per-frame calls at 1080p L6 then run pyrDown 4->5->6, the IIR launch and the collapse
// launch (6 + 7 + 8 us) in place of the tail kernel and two level launches (21 + 7 + 7 us), 13.2 k -> see profiles/README.md.
static bool lap_split_now(const LaplaceState* st, const LapBufs& B, bool first) {
    const int levels = st->levels;
    bool split = !first && st->split_levels && B.nt >= st->split_min_nt && levels >= 3 && levels - 2 <= kIirLevels;
    for (int l = 2; l <= levels - 1 && split; ++l) split = st->g[l].w % 2 == 0;
    return split;
}

// persistent workgroups of k_down0_lut_rows: one per CU (its table takes the CU's LDS), or fewer (LVM_D0_FUSED_GROUPS / the pipelined
// batch schedule: the CUs left free take the other stream's launches)
static int lap_d0l_groups(const Ctx* c, const LaplaceState* st) {
    return (st->d0_fused_groups > 0 && st->d0_fused_groups < c->num_cus) ? (int)st->d0_fused_groups : c->num_cus;
}
// arguments of k_down0_lut_rows for the frames of B (false: this launch does not take the fused first kernel)
static bool lap_d0l_args(Ctx* c, LaplaceState* st, const FrameIO& io, const LapBufs& B, D0LArgs* out) {
    const int NS = c->nstreams * B.nt, levels = st->levels;
    const LabPlanes lp = lap_planes(c, io, B);
    if (!(lp.iab && levels >= 2 && st->d0_fused && lap_vec4(io) && st->g[1].w % 2 == 0)) return false;
    long dl_tasks = 0;
    const long dl_waves = st->d0_fused_waves > 0 ? st->d0_fused_waves : (long)lap_d0l_groups(c, st) * (D0L_THREADS / 64);
    int dl_rows = down0_lut_rows_choice(st->g[1].w, st->g[1].h, NS, dl_waves, &dl_tasks);
    if (st->d0_fused_rows > 0) {            // measurement switch (LVM_D0_FUSED_ROWS): strips of this many output rows whatever the launch size
        dl_rows = st->d0_fused_rows;
        dl_tasks = (long)((st->g[1].w + D0R_OUT - 1) / D0R_OUT) * ((st->g[1].h + dl_rows - 1) / dl_rows) * NS;
    }
    if (dl_tasks <= 0) return false;
    const LevelGeom& g1 = st->g[1];
    const int sx = (g1.w + D0R_OUT - 1) / D0R_OUT, sy = (g1.h + dl_rows - 1) / dl_rows;
    *out = D0LArgs{io.d_in, (long)io.in_stride, (long)io.in_sstride, io.w, io.h, B.G[1], g1.w, g1.h, c->lab_lut, sx, sy, (int)dl_tasks, dl_rows, B.iL, B.iab};
    return true;
}


static void lap_stage_b(Ctx* c, LaplaceState* st, const lvm_params& p, const FrameIO& io, const LapBufs& B, bool first, hipStream_t s) {
    const int C = io.channels, levels = st->levels;
    const int NS = c->nstreams * B.nt;        // stateless kernels: a frame of the batch is just one more stream
    const int planes = st->planes * B.nt;
    const dim3 blk(256);
    // every frame goes through OpenCV's forward table exactly once: here (this stage comes first for every frame)
    const LabPlanes lp = lap_planes(c, io, B);
    // large launches: conversion and first pyramid kernel in one pass (k_down0_lut_rows); LVM_D0_FUSED=0 keeps them apart
    D0LArgs da{};
    const bool fused = lap_d0l_args(c, st, io, B, &da);
    if (lp.iab && !fused) lab_lut_planes(c, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.w, io.h, NS, B.iL, nullptr, B.iab, s);
    if (levels < 2) return;
    float** G = B.G;
    const LevelGeom& g1 = st->g[1];
    const dim3 grid0((g1.w + DT_W - 1) / DT_W, (g1.h + DT_H - 1) / DT_H, NS);
    // wave strips (DPP halo exchange, no LDS tile) when the launch has enough of them, else the LDS-tiled kernel
    const int d0_sx = (g1.w + D0R_OUT - 1) / D0R_OUT;
    long d0_tasks = 0;
    const int d0_rows = down0_rows_choice(g1.w, g1.h, NS, st->d0_min_tasks, &d0_tasks);
    const int fl = lab_flavour(c);
    if (fused) {
        auto kdl = fl == FL_LUT_EXACT ? k_down0_lut_rows<FL_LUT_EXACT> : k_down0_lut_rows<FL_LUT_FAST>;
        LVM_LAUNCH(c, "lap_down0_lut", kdl, dim3((unsigned)lap_d0l_groups(c, st)), dim3(D0L_THREADS), s, da);
    } else if (lap_vec4(io) && st->d0_rows_on && d0_tasks > 0) {
        auto kd0 = LVM_FL_PICK(fl, k_down0_rows, true);
        const dim3 gridr((unsigned)((d0_tasks + D0R_THREADS / 64 - 1) / (D0R_THREADS / 64)));
        LVM_LAUNCH(c, "lap_down0", kd0, gridr, dim3(D0R_THREADS), s, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.w, io.h,
                   G[1], g1.w, g1.h, c->lab, d0_sx, (g1.h + d0_rows - 1) / d0_rows, (int)d0_tasks, d0_rows, lp);
    } else if (lap_vec4(io)) {
        auto kd0 = LVM_FL_PICK(fl, k_down0_v4, true);
        LVM_LAUNCH(c, "lap_down0", kd0, grid0, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.w, io.h,
                   G[1], g1.w, g1.h, c->lab, lp);
    } else {
        auto kd0 = (C == 3) ? LVM_FL_PICK(fl, k_down0, 3, true) : k_down0<1, false, FL_LUT_EXACT>;
        LVM_LAUNCH(c, "lap_down0", kd0, grid0, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.w, io.h,
                   G[1], g1.w, g1.h, c->lab, c->lab.a255, lp);
    }
    const bool use_tail = st->tailT && B.nt == 1 && !B.no_tail && !lap_split_now(st, B, first);   // batched frames: every level gets many workgroups anyway
    const int down_end = use_tail ? st->tailT : levels;            // the tail builds G_{T+1..L} itself
    int l = 1;
    while (l < down_end) {          // G_l -> next levels, three (or two) per launch when possible
        const int left = down_end - l;
        // (level 1 keeps the strips from half the size: a single 1080p stream per call, 1.55 M, was measured better with them)
        if (st->g[l].w % 4 == 0 && (long)st->g[l].n * planes >= (l == 1 ? st->rows_min_elems / 2 : st->rows_min_elems)) {
            // large planes (temporal batches / many streams): barrier-free wave strips, one level per launch
            const LevelGeom &a = st->g[l], &b = st->g[l + 1];
            const int sx = (b.w + 127) / 128;
            int rows = st->pd_rows;
            while (rows > 4 && (long)sx * ((b.h + rows - 1) / rows) * planes < 8192) rows >>= 1;
            const int sy = (b.h + rows - 1) / rows;
            const long ntasks = (long)sx * sy * planes;
            const dim3 grid((unsigned)((ntasks + PD_THREADS / 64 - 1) / (PD_THREADS / 64)));
            LVM_LAUNCH(c, LName("pyr_down_rows", l), k_pyr_down_rows<0>, grid, dim3(PD_THREADS), s, (const float*)G[l], a.w, a.h, G[l + 1], b.w, b.h,
                       sx, sy, (int)ntasks, rows);
            l += 1;
        } else if (left >= 3 && st->fuse_down >= 3) {
            const LevelGeom &a = st->g[l], &b1 = st->g[l + 1], &b2 = st->g[l + 2], &b3 = st->g[l + 3];
            const dim3 grid((b3.w + ML_T - 1) / ML_T, (b3.h + ML_T - 1) / ML_T, planes);
            LVM_LAUNCH(c, LName("pyr_down3", l), k_pyr_down_multi<3>, grid, blk, s, (const float*)G[l], a.w, a.h, G[l + 1], b1.w, b1.h,
                       G[l + 2], b2.w, b2.h, G[l + 3], b3.w, b3.h);
            l += 3;
        } else if (left >= 2 && st->fuse_down >= 2) {
            const LevelGeom &a = st->g[l], &b1 = st->g[l + 1], &b2 = st->g[l + 2];
            const dim3 grid((b2.w + ML_T - 1) / ML_T, (b2.h + ML_T - 1) / ML_T, planes);
            LVM_LAUNCH(c, LName("pyr_down2", l), k_pyr_down_multi<2>, grid, blk, s, (const float*)G[l], a.w, a.h, G[l + 1], b1.w, b1.h,
                       G[l + 2], b2.w, b2.h, (float*)nullptr, 0, 0);
            l += 2;
        } else {
            const LevelGeom &a = st->g[l], &b = st->g[l + 1];
            const dim3 grid((b.w + DT_W - 1) / DT_W, (b.h + DT_H - 1) / DT_H, planes);
            LVM_LAUNCH(c, LName("pyr_down", l), k_pyr_down<0>, grid, blk, s, (const float*)G[l], a.w, a.h, G[l + 1], b.w, b.h);
            l += 1;
        }
    }
    if (use_tail) {
        float gains[kMaxLevels + 2];
        laplace_gains(io.w, io.h, levels, p.amplification, p.coWavelength, gains);
        double cLo = p.coLow, cHi = p.coHigh;
        if (cLo == 0) cLo = 0.01;                                        // TemporalFilter.cpp:11-12
        TailArgs& t = st->tail;
        const int T = st->tailT;
        t.GT = G[T]; t.curT = B.curT;
        t.nt = B.nt; t.fsT = (long)st->planes * (long)st->g[T].n;
        for (int k = 0; k < t.n; ++k) { t.hi[k] = st->hi[T + k]; t.lo[k] = st->lo[T + k]; t.gain[k] = gains[T + k]; }
        t.aHi = (float)(1 - cHi); t.bHi = (float)cHi; t.aLo = (float)(1 - cLo); t.bLo = (float)cLo;
        if (first) LVM_LAUNCH(c, "lap_tail_seed", k_lap_tail<true>, dim3(st->planes), dim3(TAIL_THREADS), s, t);
        else LVM_LAUNCH(c, "lap_tail", k_lap_tail<false>, dim3(st->planes), dim3(TAIL_THREADS), s, t);
    }
}

// Stage A of a frame: the fused band/IIR/collapse steps of the levels below T and the final kernel.
static void lap_stage_a(Ctx* c, LaplaceState* st, const lvm_params& p, const FrameIO& io, const LapBufs& B, bool first, hipStream_t s) {
    const int C = io.channels, levels = st->levels;
    const int NS = c->nstreams * B.nt;
    const dim3 blk(256);
    const LabPlanes lp = lap_planes(c, io, B);
    float** G = B.G;
    float gains[kMaxLevels + 2];
    laplace_gains(io.w, io.h, levels, p.amplification, p.coWavelength, gains);
    double cLo = p.coLow, cHi = p.coHigh;
    if (cLo == 0) cLo = 0.01;                                            // TemporalFilter.cpp:11-12
    const bool split = lap_split_now(st, B, first);
    const bool use_tail = st->tailT && B.nt == 1 && !B.no_tail && !split;
    int up_start = use_tail ? st->tailT - 1 : levels - 1;
    // levels F .. L-1 decoupled (one IIR launch for all of them + one stateless collapse launch); F = 2, or 3 (round 6, LVM_LAP_SPLIT_FROM)
    // where level 2 then takes the fused band / IIR / collapse step like level 1: its m_2 is neither written nor read back
    if (split) {
        const int F = (st->split_from == 3 && levels >= 5 && B.nt >= 4) ? 3 : 2;      // (per-frame calls: one launch more would cost more than m_2's round trip)
        IirArgs ia;
        ia.nlv = levels - F; ia.nt = B.nt;
        ia.aHi = (float)(1 - cHi); ia.bHi = (float)cHi; ia.aLo = (float)(1 - cLo); ia.bLo = (float)cLo;
        int blocks = 0;
        for (int l = F; l <= levels - 1; ++l) {          // finest level first: its blocks are the long ones
            IirLevel& v = ia.lv[l - F];
            v.Gl = G[l]; v.Gn = G[l + 1]; v.hi = st->hi[l]; v.lo = st->lo[l]; v.out = B.cur[l];
            v.w = st->g[l].w; v.h = st->g[l].h; v.wn = st->g[l + 1].w; v.hn = st->g[l + 1].h;
            v.gain = gains[l]; v.fsl = (long)st->planes * (long)st->g[l].n; v.fsn = (long)st->planes * (long)st->g[l + 1].n;
            v.gw = v.w / 2; v.ngroups = v.gw * ((v.h + 1) / 2); v.block0 = blocks; v.top = l == levels - 1;
            blocks += (v.ngroups + 255) / 256;
        }
        int depth = 8;
        while (depth > 1 && B.nt % depth != 0) depth >>= 1;
        auto ki = depth == 8 ? k_lap_iir_levels<8> : (depth == 4 ? k_lap_iir_levels<4> : (depth == 2 ? k_lap_iir_levels<2> : k_lap_iir_levels<1>));
        LVM_LAUNCH(c, "lap_iir", ki, dim3((unsigned)blocks, (unsigned)st->planes), blk, s, ia);
        if (levels >= F + 2) {
            CollapseArgs ca;
            ca.nlv = levels - F;
            for (int l = F; l <= levels - 1; ++l) { ca.cur[l - F] = B.cur[l]; ca.w[l - F] = st->g[l].w; ca.h[l - F] = st->g[l].h; }
            const dim3 gridc((st->g[F].w + CT_W - 1) / CT_W, (st->g[F].h + CT_H - 1) / CT_H, (unsigned)(st->planes * B.nt));
            void (*kc)(CollapseArgs) = nullptr;
            switch (ca.nlv) {
            case 2: kc = k_lap_collapse<2>; break; case 3: kc = k_lap_collapse<3>; break; case 4: kc = k_lap_collapse<4>; break;
            case 5: kc = k_lap_collapse<5>; break; case 6: kc = k_lap_collapse<6>; break; case 7: kc = k_lap_collapse<7>; break;
            case 8: kc = k_lap_collapse<8>; break; case 9: kc = k_lap_collapse<9>; break; default: kc = k_lap_collapse<10>; break;
            }
            LVM_LAUNCH(c, "lap_collapse", kc, gridc, blk, s, ca);
        }
        up_start = F - 1;
    }
    for (int l = up_start; l >= 1; --l) {
        UpArgs a;
        a.Gl = G[l]; a.Gn = G[l + 1];
        a.curn = (l + 1 <= levels - 1) ? ((use_tail && l + 1 == st->tailT) ? B.curT : B.cur[l + 1]) : nullptr;
        a.hi = st->hi[l]; a.lo = st->lo[l]; a.cur = B.cur[l];
        a.nt = B.nt; a.fsl = (long)st->planes * (long)st->g[l].n; a.fsn = (long)st->planes * (long)st->g[l + 1].n;
        a.w = st->g[l].w; a.h = st->g[l].h; a.wn = st->g[l + 1].w; a.hn = st->g[l + 1].h;
        a.aHi = (float)(1 - cHi); a.bHi = (float)cHi; a.aLo = (float)(1 - cLo); a.bLo = (float)cLo;
        a.gain = gains[l];
        const dim3 grid((a.w + UT_W - 1) / UT_W, (a.h + UT_H - 1) / UT_H, st->planes);
        const long blocks = (long)grid.x * grid.y * grid.z;
        if (!first && st->up_rows && a.w % 2 == 0 && blocks < st->up_rows_max_blocks) {
            // small launches (coarse levels of few streams): barrier-free 2 x 2 blocks per lane with a
            // 4-frame load ring; measured 14-25 us per 16 frames against 27-43 us for the tiled kernel.
            // Large launches keep the LDS-tiled kernel (bandwidth-bound there, and it needs fewer registers).
            const int gw = a.w / 2;
            const long ngroups = (long)gw * ((a.h + 1) / 2);
            const dim3 g2((unsigned)((ngroups + 255) / 256), (unsigned)st->planes);
            int depth = 4;
            while (depth > 1 && a.nt % depth != 0) depth >>= 1;
            const bool hc = a.curn != nullptr;
            auto kr = depth == 4 ? (hc ? k_lap_up_rows<2, 4, true> : k_lap_up_rows<2, 4, false>)
                                 : (depth == 2 ? (hc ? k_lap_up_rows<2, 2, true> : k_lap_up_rows<2, 2, false>)
                                               : (hc ? k_lap_up_rows<2, 1, true> : k_lap_up_rows<2, 1, false>));
            LVM_LAUNCH(c, LName("lap_up", l), kr, g2, blk, s, a, gw, (int)ngroups);
            continue;
        }
        int depth = (blocks >= 1024) ? st->up_depth_big : st->up_depth;   // frame ring of the tiled kernel
        while (depth > 1 && a.nt % depth != 0) depth >>= 1;           // the ring depth must divide the frame count
        if (first) LVM_LAUNCH(c, LName("lap_seed", l), (k_lap_up<true, 1>), grid, blk, s, a);
        else if (depth == 1) LVM_LAUNCH(c, LName("lap_up", l), (k_lap_up<false, 1>), grid, blk, s, a);
        else if (depth == 2) LVM_LAUNCH(c, LName("lap_up", l), (k_lap_up<false, 2>), grid, blk, s, a);
        else if (depth <= 4) LVM_LAUNCH(c, LName("lap_up", l), (k_lap_up<false, 4>), grid, blk, s, a);
        else LVM_LAUNCH(c, LName("lap_up", l), (k_lap_up<false, 8>), grid, blk, s, a);
    }
    const int fl = lab_flavour(c);
    float* dbg = (c->keep_float && B.dbg_frame) ? c->d_float : nullptr;   // (the float frame kept is the first one of the batch)
    const int tx = (io.w + UT_W - 1) / UT_W, ty = (io.h + UT_H - 1) / UT_H;
    const int ntiles = tx * ty * NS;
    const dim3 grid(ntiles < 2048 ? ntiles : 2048);
    const bool motion = !first && levels >= 2;
    const float ca = (float)p.chromAttenuation;
    const float* cur1 = motion ? ((use_tail && st->tailT == 1) ? B.curT : B.cur[1]) : nullptr;
    const int w1 = st->g[1].w, h1 = st->g[1].h;
    auto kf4 = dbg ? (motion ? LVM_FL_PICK(fl, k_lap_final_v4, true, true) : LVM_FL_PICK(fl, k_lap_final_v4, false, true))
                   : (motion ? LVM_FL_PICK(fl, k_lap_final_v4, true, false) : LVM_FL_PICK(fl, k_lap_final_v4, false, false));
    auto kf = (C == 3) ? (motion ? LVM_FL_PICK(fl, k_lap_final, 3, true) : LVM_FL_PICK(fl, k_lap_final, 3, false))
                       : (motion ? k_lap_final<1, true, FL_LUT_EXACT> : k_lap_final<1, false, FL_LUT_EXACT>);
    if (lap_vec4(io)) {
        // wave strips of 256 x rows pixels; shorter strips when there are few of them (per-frame calls)
        const int sx = (io.w + 255) / 256;
        int rows = st->fin_rows;
        while (rows > 2 && (long)sx * ((io.h + rows - 1) / rows) * NS < st->fin_min_tasks) rows >>= 1;
        const int sy = (io.h + rows - 1) / rows;
        const int waves = FIN_THREADS / 64;
        const long groups = ((long)sx * sy * NS + waves - 1) / waves;
        const long cap = st->fin_groups > 0 ? st->fin_groups : 1024;
        const dim3 grid4((unsigned)(groups < cap ? groups : cap)), blk4(FIN_THREADS);
        const FinArgs fa{io.d_in, (long)io.in_stride, (long)io.in_sstride, io.d_out, (long)io.out_stride, (long)io.out_sstride, io.w, io.h, cur1, w1, h1,
                         c->lab, ca, sx, sy, NS, rows, dbg, lp};
        LVM_LAUNCH(c, "lap_final", kf4, grid4, blk4, s, fa);
    } else {
        LVM_LAUNCH(c, "lap_final", kf, grid, blk, s, io.d_in, (long)io.in_stride, (long)io.in_sstride, io.d_out,
                   (long)io.out_stride, (long)io.out_sstride, io.w, io.h, cur1, w1, h1, c->lab, ca, tx, ty, NS, dbg, lp);
    }
}

// Emits stage A of the pending frame (pipelined mode): its output lands in the d_out it was given.
int laplace_flush(Ctx* c, hipStream_t s) {
    LaplaceState* st = dynamic_cast<LaplaceState*>(c->state);
    if (!st || !st->pending.valid) return LVM_OK;
    lap_stage_a(c, st, st->pending.p, st->pending.io, lap_bufs_frame(st, st->pending.par), false, s);
    st->pending.valid = false;
    LVM_HIP_TRY(c, hipGetLastError());
    return LVM_OK;
}

// Buffers of a temporal batch (pyramids / collapse accumulators of nt frames).  Sized for the larger of nt and the
// caller's lvm_set_max_frames hint, so a steady-state call never allocates; called when the state is created
// (hint given) or when a batch exceeds what is there.
static int laplace_reserve_frames(Ctx* c, LaplaceState* st, int nt, hipStream_t s) {
    if (nt < c->max_frames) nt = c->max_frames;
    if (nt <= st->tcap) return LVM_OK;
    const int levels = st->levels;
    LVM_HIP_TRY(c, hipStreamSynchronize(s));
    sync_streams(c);
    if (st->tarena) (void)hipFree(st->tarena);
    st->tarena = nullptr; st->tcap = 0;
    auto pad = [](size_t n) { return (n + 63) & ~(size_t)63; };
    size_t total = 64;
    for (int l = 1; l <= levels; ++l) total += pad(st->g[l].n * st->planes * nt);
    for (int l = 1; l < levels; ++l) total += pad(st->g[l].n * st->planes * nt);
    const size_t npx = st->g[0].n * c->nstreams * nt;
    const bool lab = st->planes == 3 * c->nstreams;
    if (lab) total += pad(npx) + pad((npx + 1) / 2);
    if (hipMalloc((void**)&st->tarena, total * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); st->tarena = nullptr; c->err = "laplace: hipMalloc (frames) failed"; return LVM_ERR_OOM; }
    float* q = st->tarena;
    for (int l = 1; l <= levels; ++l) { st->Gt[l] = q; q += pad(st->g[l].n * st->planes * nt); }
    for (int l = 1; l < levels; ++l) { st->curt[l] = q; q += pad(st->g[l].n * st->planes * nt); }
    if (lab) { st->iabt = reinterpret_cast<uint32_t*>(q); q += pad(npx); st->iLt = reinterpret_cast<uint16_t*>(q); q += pad((npx + 1) / 2); }
    st->tcap = nt;
    return LVM_OK;
}

// Temporal batch: nt consecutive frames of every stream in one pass (the reference's export loop,
// export/Exporter.cpp:216-259, sees its frames in exactly this order).  Frame f of stream b lives at
// d_in + (f * n_streams + b) * in_sstride.  Preconditions (checked by the caller): state seeded,
// pipeline depth 0.  Kernels and arithmetic are those of the per-frame path.
int laplace_process_frames(Ctx* c, const lvm_params& p, const FrameIO& io, int nt, hipStream_t s) {
    LaplaceState* st = static_cast<LaplaceState*>(c->state);
    const int levels = st->levels;
    if (nt > st->tcap) { const int rc = laplace_reserve_frames(c, st, nt, s); if (rc != LVM_OK) return rc; }
    if (st->pending.valid) { const int rc = laplace_flush(c, s); if (rc != LVM_OK) return rc; }
    // One pass over the batch: the stateless kernels take the frames as one more batch dimension, the IIR kernels walk over them
    // inside the launch.  (Rounds 1-3 also carried two chunked schedules -- the next chunk's table conversion on a second stream, and the
    // last kernel of a chunk sharing a launch with the next chunk's first kernel; both measured slower, profiles/README.md, and were removed.)
    float* G[kMaxLevels + 1]; float* cur[kMaxLevels + 1];
    for (int l = 1; l <= levels; ++l) G[l] = st->Gt[l];
    for (int l = 1; l < levels; ++l) cur[l] = st->curt[l];
    LapBufs B{G, cur, nullptr, nt, true};            // (the tail kernel filters inside stage B: per-frame calls only)
    if (st->iabt) { B.iL = st->iLt; B.iab = st->iabt; }
    lap_stage_b(c, st, p, io, B, false, s);
    lap_stage_a(c, st, p, io, B, false, s);
    LVM_HIP_TRY(c, hipGetLastError());
    return LVM_OK;
}

bool laplace_can_batch(const Ctx* c) {
    const LaplaceState* st = dynamic_cast<const LaplaceState*>(c->state);
    return st && st->seeded && c->pipeline_depth == 0;
}

int laplace_process(Ctx* c, const lvm_params& p, int levels, const FrameIO& io, hipStream_t s, int* produced) {
    LaplaceState* st = static_cast<LaplaceState*>(c->state);
    if (!st) {
        st = new LaplaceState();
        c->state = st;
        int rc = laplace_alloc(c, st, io.w, io.h, io.channels, levels);
        if (rc == LVM_OK && c->max_frames > 1) rc = laplace_reserve_frames(c, st, c->max_frames, s);
        if (rc != LVM_OK) return rc;
    }
    const bool first = !st->seeded;
    st->depth = c->pipeline_depth;
    if (first || c->pipeline_depth == 0 || !c->aux_stream) {
        // plain schedule: both stages of this frame back to back on the caller's stream
        if (st->pending.valid) { const int rc = laplace_flush(c, s); if (rc != LVM_OK) return rc; }
        lap_stage_b(c, st, p, io, lap_bufs_frame(st, 0), first, s);
        lap_stage_a(c, st, p, io, lap_bufs_frame(st, 0), first, s);
        st->par = 1;
    } else {
        // depth-1 software pipeline across frames: stage B of THIS frame runs on the auxiliary stream
        // concurrently with stage A of the PREVIOUS frame.  The two touch disjoint buffers: G_1..G_T and
        // cur_T are double-buffered by frame parity, the IIR states are split at level T.
        const int par = st->par;
        if (st->pending.valid) {
            LVM_HIP_TRY(c, hipEventRecord(c->ev_fork, s));
            LVM_HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
            lap_stage_b(c, st, p, io, lap_bufs_frame(st, par), false, c->aux_stream);
            LVM_HIP_TRY(c, hipEventRecord(c->ev_join, c->aux_stream));
            lap_stage_a(c, st, st->pending.p, st->pending.io, lap_bufs_frame(st, st->pending.par), false, s);
            LVM_HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
        } else {
            lap_stage_b(c, st, p, io, lap_bufs_frame(st, par), false, s);
        }
        st->pending.valid = true; st->pending.io = io; st->pending.p = p; st->pending.par = par;
        st->par = par ^ 1;
    }
    LVM_HIP_TRY(c, hipGetLastError());
    st->seeded = true;
    *produced = 1;                                                       // MagnifyCore.hpp:159
    return LVM_OK;
}

}  // namespace lvm
