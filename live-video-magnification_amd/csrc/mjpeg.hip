// mjpeg.hip -- Motion-JPEG encode of device-resident BGR canvases (the encode half).
//
// Replaces what `cv::VideoWriter::write(canvas)` does for ExportFormat::AviMjpg -- the reference's AVI export format and the fallback of
// every other one (reference: export/Exporter.cpp:107-117 openWriter, :259 writer.write(canvas)): the composed canvas never leaves the
// device, only the compressed frame crosses PCIe (an order of magnitude fewer bytes than the 12.4 MB canvas that bounds
// lvm_export_frames), and the host's software codec is gone from the export loop.
//
// Format: ITU-T T.81 baseline sequential DCT, 8 bit, YCbCr 4:2:0 (JFIF, BT.601 full range), the Huffman tables of Annex K, restart
// intervals of 8 MCUs.  JPEG leaves colour rounding, the DCT and the quantiser rounding to the encoder; this one is all-integer and is
// restated line by line by oracle/mjpeg_oracle.py (the arithmetic is in its header): the bitstream is BYTE-IDENTICAL to the oracle's and
// is decoded by libjpeg (Pillow) in the tests.
//
// Why restart intervals: entropy coding is a serial bit stream, but T.81's restart markers cut it into byte-aligned pieces whose DC
// predictors start at zero (F.1.1.5.1): 8 MCUs per piece (the default; lvm_mjpeg_set_restart_interval) make a 1080p side-by-side canvas
// 2 040 independent streams per frame.
// Inside a piece the 64 lanes of a wave ARE the 64 coefficients of a block in zigzag order: zero runs, code lengths and bit positions
// are three wave scans, every lane ORs its own code word into the bit buffer.
//
//   k_mj_transform   BGR -> YCbCr 4:2:0 -> FDCT -> quantise -> zigzag, bits per block   one wave per 16 x 16 MCU
//   k_mj_scan        bit position of every block, bit buffer zeroed               one workgroup per restart interval
//   k_mj_pack        the code words, ORed into the bit buffer                     one wave per run of 8 blocks (lane = coefficient)
//   k_mj_size        stuffed size of every interval (FF -> FF 00)                one workgroup per restart interval
//   k_mj_frame_offsets / k_mj_offsets   byte offsets of the intervals in their frame, of the frames in the output
//   k_mj_write       header, stuffed intervals, RSTn / EOI markers               one workgroup per restart interval
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lvm_internal.h"
#include "mjpeg_tables.h"

namespace lvm {
namespace {

// device copy of everything the kernels look up
struct MjTables {
    uint32_t ac[2][256];      // (code << 8) | length, 0 where a symbol has no code; [0] luminance, [1] chrominance
    uint32_t dc[2][12];
    uint32_t qr[2][64];       // ceil(2^24 / Q) in ZIGZAG order: n / Q == (n * qr) >> 24 for every n * Q < 2^24
    uint16_t q[2][64];        // Q in zigzag order
    uint8_t zz[64];
    uint8_t aclen[2][256];    // the lengths of ac[][] alone (k_mj_transform counts bits, it does not build code words)
};
constexpr int MJ_MAX_MW = 512;                  // MCUs per restart interval k_mj_scan's LDS holds; also the widest frame in MCUs (8192 pixels)
constexpr int MJ_BLOCK_WORDS = 54;              // worst case of one block: 20 + 63 * 26 bits = 1658 -> 52 words, + 2 for the straddling word and the pad
constexpr int MJ_MAX_HEADER = 1024;
constexpr int MJ_DEFAULT_RESTART = 8;

struct MjGeom {
    int w, h, mw, mh;
    long stride, fstride;
    int ri, nint;               // MCUs per restart interval (raster order, T.81 E.1.4), intervals per frame
};
// interval i of frame f: its first MCU in the [frame][MCU] arrays, its number of blocks
__device__ __forceinline__ void mj_interval(const MjGeom& g, int f, int i, size_t& mcu0, int& nblk) {
    const int nmcu = g.mw * g.mh, m0 = i * g.ri;
    mcu0 = (size_t)f * nmcu + m0;
    nblk = (nmcu - m0 < g.ri ? nmcu - m0 : g.ri) * 6;
}

struct MjState {
    int w = 0, h = 0, quality = -1, frames_cap = 0;
    int restart = 0;            // wanted MCUs per restart interval (lvm_mjpeg_set_restart_interval; 0 = MJ_DEFAULT_RESTART)
    int ri = 0, nint = 0;       // in effect for (w, h): MCUs per interval, intervals per frame
    MjTables* d_tab = nullptr;
    uint8_t* d_header = nullptr; int header_bytes = 0;
    std::vector<uint8_t> header;
    int16_t* d_coef = nullptr;
    uint32_t* d_raw = nullptr;
    uint32_t *d_ibits = nullptr, *d_isize = nullptr, *d_blk = nullptr;
    unsigned long long *d_ioff = nullptr, *d_foff = nullptr, *d_run = nullptr;     // d_run[0] running byte count, d_run[1] overflow flag
    unsigned long long* d_fsz = nullptr;                                            // per frame of a call: its size, then its start
    uint8_t* d_jpeg = nullptr; size_t jpeg_cap = 0;
    size_t foff_cap = 0;
    // incremental download: a page-locked mirror of d_foff, one event per encode call, the copy queue, what has been queued so far
    unsigned long long* h_foff = nullptr;
    std::vector<hipEvent_t> ev;
    struct Call { int frame0, nframes; };
    std::vector<Call> calls;
    size_t drained = 0, copied = 0;
    hipStream_t dl = nullptr;
};

void build_huff(const uint8_t* bits, const uint8_t* vals, int nvals, uint32_t* table, int table_len) {      // T.81 Annex C
    for (int i = 0; i < table_len; ++i) table[i] = 0;
    uint32_t code = 0;
    int k = 0;
    for (int len = 1; len <= 16; ++len) {
        for (int i = 0; i < bits[len - 1] && k < nvals; ++i, ++k) { table[vals[k]] = (code << 8) | (uint32_t)len; ++code; }
        code <<= 1;
    }
}

// libjpeg's jpeg_quality_scaling + jpeg_add_quant_table (baseline: entries 1..255)
void scaled_quant(int quality, const uint8_t* base, int* out) {
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) {
        int t = (base[i] * scale + 50) / 100;
        out[i] = t < 1 ? 1 : (t > 255 ? 255 : t);
    }
}

void put16(std::vector<uint8_t>& v, int x) { v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }
void segment(std::vector<uint8_t>& v, int marker, const std::vector<uint8_t>& payload) {
    v.push_back(0xFF); v.push_back((uint8_t)marker); put16(v, (int)payload.size() + 2);
    v.insert(v.end(), payload.begin(), payload.end());
}

// SOI, APP0 (JFIF 1.01), DQT x 2, SOF0 (Y 2x2, Cb 1x1, Cr 1x1), DHT x 4, DRI, SOS
void build_header(std::vector<uint8_t>& v, int w, int h, const int* ql, const int* qc, int restart_interval) {
    v.clear();
    v.push_back(0xFF); v.push_back(0xD8);
    segment(v, 0xE0, {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});
    for (int t = 0; t < 2; ++t) {
        std::vector<uint8_t> p(1, (uint8_t)t);
        for (int i = 0; i < 64; ++i) p.push_back((uint8_t)(t ? qc : ql)[kZigzag[i]]);
        segment(v, 0xDB, p);
    }
    segment(v, 0xC0, {8, (uint8_t)(h >> 8), (uint8_t)h, (uint8_t)(w >> 8), (uint8_t)w, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1});
    const struct { int id; const uint8_t* bits; const uint8_t* vals; int n; } hts[4] = {
        {0x00, kDcLumaBits, kDcVals, 12}, {0x10, kAcLumaBits, kAcLumaVals, 162}, {0x01, kDcChromaBits, kDcVals, 12}, {0x11, kAcChromaBits, kAcChromaVals, 162}};
    for (const auto& t : hts) {
        std::vector<uint8_t> p(1, (uint8_t)t.id);
        p.insert(p.end(), t.bits, t.bits + 16);
        p.insert(p.end(), t.vals, t.vals + t.n);
        segment(v, 0xC4, p);
    }
    segment(v, 0xDD, {(uint8_t)(restart_interval >> 8), (uint8_t)restart_interval});
    segment(v, 0xDA, {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});
}

// ---- kernels ----------------------------------------------------------------------------------------------------------------------------

// One wave per MCU (four per workgroup).  Lane = one 2 x 2 pixel quad = one chroma sample: the 64 lanes of a wave are the 8 x 8 chroma
// block and the four 8 x 8 luminance blocks of the MCU.  Pixels outside the frame repeat the last row / column.
__global__ __launch_bounds__(256) void k_mj_transform(const uint8_t* __restrict__ src, MjGeom g, const MjTables* __restrict__ tb, int16_t* __restrict__ coef,
                                                      uint32_t* __restrict__ acbits) {
    __shared__ int s_a[4][6][64], s_b[4][6][64];
    __shared__ uint8_t s_len[2][256];
    __shared__ int s_bits[4][8];
    __shared__ uint32_t s_qr[2][64];
    __shared__ uint16_t s_q[2][64];
    __shared__ uint8_t s_zz[64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < 128) { s_qr[tid >> 6][tid & 63] = tb->qr[tid >> 6][tid & 63]; s_q[tid >> 6][tid & 63] = tb->q[tid >> 6][tid & 63]; }
    if (tid < 64) s_zz[tid] = tb->zz[tid];
    if (tid < 128) reinterpret_cast<uint32_t*>(&s_len[0][0])[tid] = reinterpret_cast<const uint32_t*>(&tb->aclen[0][0])[tid];
    if (lane < 8) s_bits[wave][lane] = 0;
    const int mx = blockIdx.x * 4 + wave, my = blockIdx.y, f = blockIdx.z;
    const bool act = mx < g.mw;
    if (act) {
        const int qx = lane & 7, qy = lane >> 3;
        const uint8_t* fr = src + (size_t)f * g.fstride;
        int cbs = 0, crs = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            int py = my * 16 + 2 * qy + dy; py = py < g.h ? py : g.h - 1;
            const uint8_t* row = fr + (size_t)py * g.stride;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int px = mx * 16 + 2 * qx + dx; px = px < g.w ? px : g.w - 1;
                const int b = row[px * 3], gg = row[px * 3 + 1], r = row[px * 3 + 2];
                const int y = (19595 * r + 38470 * gg + 7471 * b + 32768) >> 16;
                cbs += (-11059 * r - 21709 * gg + 32768 * b + 8421375) >> 16;
                crs += (32768 * r - 27439 * gg - 5329 * b + 8421375) >> 16;
                const int ly = 2 * qy + dy, lx = 2 * qx + dx;
                s_a[wave][(ly >> 3) * 2 + (lx >> 3)][(ly & 7) * 8 + (lx & 7)] = y - 128;
            }
        }
        s_a[wave][4][lane] = ((cbs + 2) >> 2) - 128;
        s_a[wave][5][lane] = ((crs + 2) >> 2) - 128;
    }
    __syncthreads();
    if (act && lane < 48) {               // rows: t[y][u] = (sum_x M[u][x] d[y][x] + 512) >> 10
        const int blk = lane >> 3, r = lane & 7;
        int d[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) d[x] = s_a[wave][blk][r * 8 + x];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int acc = 0;
#pragma unroll
            for (int x = 0; x < 8; ++x) acc += kDct[u][x] * d[x];
            s_b[wave][blk][r * 8 + u] = (acc + 512) >> 10;
        }
    }
    __syncthreads();
    if (act && lane < 48) {               // columns: S[v][u] = (sum_y M[v][y] t[y][u] + 32768) >> 16
        const int blk = lane >> 3, u = lane & 7;
        int d[8];
#pragma unroll
        for (int y = 0; y < 8; ++y) d[y] = s_b[wave][blk][y * 8 + u];
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            int acc = 0;
#pragma unroll
            for (int y = 0; y < 8; ++y) acc += kDct[v][y] * d[y];
            s_a[wave][blk][v * 8 + u] = (acc + 32768) >> 16;
        }
    }
    __syncthreads();
    // quantise, zigzag: lane = zigzag index.  The bits the AC coefficients of each block will take (T.81 F.1.2.2: run / size symbols, ZRL,
    // EOB) are counted on the way: one ballot of the non-zero lanes gives every lane its zero run.  (Every wave runs this -- the ballot
    // needs all lanes -- and only the existing MCUs store.)
    {
        int16_t* out = coef + ((size_t)(f * g.mh + my) * g.mw + (act ? mx : 0)) * 384;
        const int nat = s_zz[lane];
#pragma unroll
        for (int blk = 0; blk < 6; ++blk) {
            const int t = blk < 4 ? 0 : 1;
            const int s = act ? s_a[wave][blk][nat] : 0;
            const uint32_t n = (uint32_t)(s < 0 ? -s : s) + (uint32_t)(s_q[t][lane] >> 1);
            const int a = act ? (int)(((unsigned long long)n * s_qr[t][lane]) >> 24) : 0;         // = n / Q
            if (act) out[blk * 64 + lane] = (int16_t)(s < 0 ? -a : a);
            const unsigned long long nz = lvm_ballot64(lane >= 1 && a != 0);
            const unsigned long long below = nz & ((1ull << lane) - 1ull);
            const int prev = below ? 63 - __builtin_clzll(below) : 0;
            const int last = nz ? 63 - __builtin_clzll(nz) : 0;
            int bits = 0;
            if (lane >= 1 && a != 0) {
                const int run = lane - prev - 1, cat = 32 - __builtin_clz((unsigned)a);
                bits = (run >> 4) * s_len[t][0xF0] + s_len[t][((run & 15) << 4) | cat] + cat;
            } else if (lane == last + 1) {
                bits = s_len[t][0];
            }
            if (bits) atomicAdd(&s_bits[wave][blk], bits);
        }
    }
    lvm_wave_lds_sync();
    if (act && lane < 6) acbits[((size_t)(f * g.mh + my) * g.mw + mx) * 6 + lane] = (uint32_t)s_bits[wave][lane];
}

struct LaneCode { uint32_t code; int len, nzrl, bits; };
// The code word of zigzag position `lane` of block b of a restart interval (T.81 F.1.2): lane 0 the DC difference, a non-zero lane its
// (run, size) symbol + amplitude bits behind run / 16 ZRL symbols, the lane behind the last non-zero coefficient the end-of-block symbol.
// The zero runs come from ONE ballot of the non-zero lanes: the previous non-zero position is the highest set bit below the lane.
// Every lane of the wave calls it; `act` = the block exists; v = the lane's coefficient, the DC DIFFERENCE in lane 0 (mj_fetch).
__device__ __forceinline__ LaneCode lane_code(int v, int t, bool act, int lane, const uint32_t (*s_ac)[256], const uint32_t (*s_dc)[12]) {
    const unsigned long long nz = lvm_ballot64(lane >= 1 && v != 0);
    const unsigned long long below = nz & ((1ull << lane) - 1ull);
    const int prev = below ? 63 - __builtin_clzll(below) : 0;              // position of the previous non-zero AC coefficient (0: none)
    const int last = nz ? 63 - __builtin_clzll(nz) : 0;
    const int a = v < 0 ? -v : v;
    const int s = a ? 32 - __builtin_clz((unsigned)a) : 0;
    const uint32_t amp = (uint32_t)(v + (v >> 31)) & ((1u << s) - 1u);      // v, or v - 1 for v < 0: the low s bits
    LaneCode lc;
    lc.code = 0; lc.len = 0; lc.nzrl = 0;
    if (lane == 0) {
        const uint32_t e = s_dc[t][s];
        lc.code = ((e >> 8) << s) | amp; lc.len = (int)(e & 255u) + s;
    } else if (v != 0) {
        const int run = lane - prev - 1;
        lc.nzrl = run >> 4;
        const uint32_t e = s_ac[t][((run & 15) << 4) | s];
        lc.code = ((e >> 8) << s) | amp; lc.len = (int)(e & 255u) + s;
    } else if (lane == last + 1) {
        const uint32_t e = s_ac[t][0];
        lc.code = e >> 8; lc.len = (int)(e & 255u);
    }
    lc.bits = act ? lc.nzrl * (int)(s_ac[t][0xF0] & 255u) + lc.len : 0;
    return lc;
}

// `len` bits of `code` at bit position `off` of a big-endian bit buffer of zeroed 32-bit words
__device__ __forceinline__ void put_bits(uint32_t* __restrict__ out, uint32_t off, uint32_t code, int len) {
    const uint32_t w = off >> 5;
    const unsigned long long v = (unsigned long long)code << (64 - (int)(off & 31u) - len);      // len <= 27: never shifted out
    atomicOr(out + w, (uint32_t)(v >> 32));
    if ((uint32_t)v) atomicOr(out + w + 1, (uint32_t)v);
}

// exclusive scan of s[0 .. n) in place by the 256 threads of a workgroup (part: 256 words of LDS); returns the total
__device__ __forceinline__ uint32_t wg_exclusive_scan(uint32_t* s, int n, uint32_t* part, int tid) {
    const int per = (n + 255) / 256, lo = tid * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += s[i];
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t t = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0u;
    for (int i = lo; i < hi; ++i) { const uint32_t t = s[i]; s[i] = run; run += t; }
    const uint32_t total = part[255];
    __syncthreads();
    return total;
}

__device__ __forceinline__ void load_huff(uint32_t (*s_ac)[256], uint32_t (*s_dc)[12], const MjTables* __restrict__ tb, int tid) {
    for (int i = tid; i < 512; i += 256) s_ac[i >> 8][i & 255] = tb->ac[i >> 8][i & 255];
    if (tid < 24) s_dc[tid / 12][tid % 12] = tb->dc[tid / 12][tid % 12];
}

// After k_mj_scan turned the bit counts into bit positions: the code words.  A wave owns a run of MJ_NB consecutive blocks of one interval and
// assembles it in LDS at the bit phase the run has in the interval's stream (every lane ORs its own code word in: LDS atomics), so that LDS
// word j IS word (position / 32) + j of the stream.  (One block per wave with two global atomics each measured 283 us per 8 canvases, a
// run of 8 with two atomics per run 150 us; what remains is the loads and the code construction.)
constexpr int MJ_NB = 8;
struct MjFetch { int v; uint32_t base; int t; bool act; };
// the loads of block b of an interval (cf, bp: its coefficients and bit positions): the lane's coefficient (lane 0: minus the DC of the previous
// block of the same component -- the predictor restarts at 0 with the interval), the block's bit position
__device__ __forceinline__ MjFetch mj_fetch(const int16_t* __restrict__ cf, const uint32_t* __restrict__ bp, int b, bool act, int lane) {
    MjFetch f;
    f.act = act;
    const int m = b / 6, k = b - m * 6;
    f.t = k < 4 ? 0 : 1;
    f.v = act ? cf[(size_t)b * 64 + lane] : 0;
    if (lane == 0 && act && ((k >= 1 && k <= 3) || m > 0)) f.v -= cf[(size_t)(k == 0 ? b - 3 : (k < 4 ? b - 1 : b - 6)) * 64];
    f.base = act ? bp[b] : 0u;
    return f;
}
constexpr int MJ_RUN_WORDS = MJ_NB * (MJ_BLOCK_WORDS - 2) + 2;       // LDS words of a run of MJ_NB blocks at any bit phase
__global__ __launch_bounds__(256) void k_mj_pack(const int16_t* __restrict__ coef, MjGeom g, uint32_t nintervals, const MjTables* __restrict__ tb,
                                                  const uint32_t* __restrict__ blk, uint32_t* __restrict__ raw) {
    __shared__ uint32_t s_ac[2][256], s_dc[2][12];
    __shared__ uint32_t s_stage[4][MJ_RUN_WORDS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    load_huff(s_ac, s_dc, tb, tid);
    // A wave owns a run of MJ_NB consecutive blocks of ONE interval (the last run of an interval is shorter) and assembles it in LDS at the
    // bit phase the run has in the interval's stream.  All loads of the run are issued before the first block is coded.
    const uint32_t rpi = (uint32_t)(g.ri * 6 + MJ_NB - 1) / MJ_NB, run = (uint32_t)blockIdx.x * 4u + (uint32_t)wave;
    const uint32_t gi = run / rpi;
    const int b0 = (int)(run - gi * rpi) * MJ_NB;
    const bool wact = gi < nintervals;
    const int fr = wact ? (int)(gi / (uint32_t)g.nint) : 0, iv = wact ? (int)(gi - (uint32_t)fr * (uint32_t)g.nint) : 0;
    size_t mcu0; int nblk;
    mj_interval(g, fr, iv, mcu0, nblk);
    const int16_t* cf = coef + mcu0 * 384;
    const uint32_t* bp = blk + mcu0 * 6;
    MjFetch fs[MJ_NB];
#pragma unroll
    for (int it = 0; it < MJ_NB; ++it) fs[it] = mj_fetch(cf, bp, wact && b0 + it < nblk ? b0 + it : 0, wact && b0 + it < nblk, lane);
    for (int j = lane; j < MJ_RUN_WORDS; j += 64) s_stage[wave][j] = 0u;
    __syncthreads();
    const uint32_t word0 = fs[0].base >> 5;                 // (block b0 of an existing run always exists)
    uint32_t end = fs[0].base;
#pragma unroll
    for (int it = 0; it < MJ_NB; ++it) {
        const MjFetch f = fs[it];
        const LaneCode lc = lane_code(f.v, f.t, f.act, lane, s_ac, s_dc);
        const int incl = lvm_wave_prefix_add(lc.bits, lane);
        const int nbits = lvm_wave_last(incl);
        if (f.act) end = f.base + (uint32_t)nbits;
        if (f.act && lc.bits) {
            uint32_t off = f.base - (word0 << 5) + (uint32_t)(incl - lc.bits);
            const uint32_t zrl = s_ac[f.t][0xF0];
            for (int k = 0; k < lc.nzrl; ++k) { put_bits(s_stage[wave], off, zrl >> 8, (int)(zrl & 255u)); off += zrl & 255u; }
            put_bits(s_stage[wave], off, lc.code, lc.len);
        }
    }
    lvm_wave_lds_sync();
    // LDS word j IS word word0 + j of the stream: the inner words are plain stores, only the first and the last one -- shared with the
    // neighbouring runs -- are atomic ORs into the zeroed buffer
    const int nwords = (int)(((end + 31u) >> 5) - word0);
    if (wact && b0 < nblk) {
        uint32_t* out = raw + mcu0 * 6 * MJ_BLOCK_WORDS + word0;
        for (int j = lane; j < nwords; j += 64) {
            const uint32_t v = s_stage[wave][j];
            if (j == 0 || j == nwords - 1) { if (v) atomicOr(out + j, v); }
            else out[j] = v;
        }
    }
}

// One workgroup per restart interval: AC bits of its blocks (k_mj_transform) + the bits of their DC differences (F.1.2.1) -> the bit
// position of every block, the interval's length; its bit buffer zeroed
__global__ __launch_bounds__(256) void k_mj_scan(const int16_t* __restrict__ coef, MjGeom g, const MjTables* __restrict__ tb, uint32_t* __restrict__ blk,
                                                 uint32_t* __restrict__ raw, uint32_t* __restrict__ ibits) {
    __shared__ uint32_t s_off[MJ_MAX_MW * 6];
    __shared__ uint32_t s_part[256];
    __shared__ uint32_t s_dc[2][12];
    const int tid = threadIdx.x;
    const size_t interval = (size_t)blockIdx.y * g.nint + blockIdx.x;
    size_t mcu0; int nblk;
    mj_interval(g, blockIdx.y, blockIdx.x, mcu0, nblk);
    if (tid < 24) s_dc[tid / 12][tid % 12] = tb->dc[tid / 12][tid % 12];
    __syncthreads();
    uint32_t* mine = blk + mcu0 * 6;
    const int16_t* cf = coef + mcu0 * 384;
    for (int b = tid; b < nblk; b += 256) {
        const int m = b / 6, k = b - m * 6;
        const bool has_pred = (k >= 1 && k <= 3) || m > 0;
        const int pb = k == 0 ? b - 3 : (k < 4 ? b - 1 : b - 6);
        const int diff = cf[(size_t)b * 64] - (has_pred ? cf[(size_t)pb * 64] : 0);
        const int a = diff < 0 ? -diff : diff, cat = a ? 32 - __builtin_clz((unsigned)a) : 0;
        s_off[b] = mine[b] + (s_dc[k < 4 ? 0 : 1][cat] & 255u) + (uint32_t)cat;
    }
    __syncthreads();
    const uint32_t total_bits = wg_exclusive_scan(s_off, nblk, s_part, tid);
    for (int i = tid; i < nblk; i += 256) mine[i] = s_off[i];
    if (tid == 0) ibits[interval] = total_bits;
    uint32_t* out = raw + mcu0 * 6 * MJ_BLOCK_WORDS;
    const uint32_t nwords = (total_bits + 31u) / 32u + 1u;
    for (uint32_t i = tid; i < nwords; i += 256) out[i] = 0u;
}

// word i (four stream bytes, first byte in the top bits) of an interval's bit buffer; the last byte is filled up with 1-bits (T.81 F.1.2.3)
__device__ __forceinline__ uint32_t raw_word(const uint32_t* __restrict__ raw, uint32_t i, uint32_t nbytes, uint32_t bits) {
    uint32_t w = raw[i];
    if ((bits & 7u) && i == (nbytes - 1u) >> 2) w |= ((1u << (8u - (bits & 7u))) - 1u) << (24u - 8u * ((nbytes - 1u) & 3u));
    return w;
}
// number of FF bytes among the first n (1..4) bytes of a word
__device__ __forceinline__ uint32_t ff_bytes(uint32_t w, uint32_t n) {
    uint32_t c = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) c += (j < n && ((w >> (24u - 8u * j)) & 255u) == 255u) ? 1u : 0u;
    return c;
}

__global__ __launch_bounds__(256) void k_mj_size(const uint32_t* __restrict__ raw, MjGeom g, const uint32_t* __restrict__ ibits, uint32_t* __restrict__ isize) {
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const size_t interval = (size_t)blockIdx.y * g.nint + blockIdx.x;
    size_t mcu0; int nblk_unused;
    mj_interval(g, blockIdx.y, blockIdx.x, mcu0, nblk_unused);
    const uint32_t* in = raw + mcu0 * 6 * MJ_BLOCK_WORDS;
    const uint32_t bits = ibits[interval], nbytes = (bits + 7u) >> 3, nwords = (nbytes + 3u) >> 2;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int cnt = 0;
    for (uint32_t i = tid; i < nwords; i += 256) cnt += (int)ff_bytes(raw_word(in, i, nbytes, bits), nbytes - 4u * i < 4u ? nbytes - 4u * i : 4u);
    if (cnt) atomicAdd(&s_cnt, cnt);
    __syncthreads();
    if (tid == 0) isize[interval] = nbytes + (uint32_t)s_cnt;
}

// One workgroup per frame: the byte offset of every interval from the start of ITS frame (header, then the intervals, each followed by its
// two marker bytes) and the frame's size
__global__ __launch_bounds__(256) void k_mj_frame_offsets(MjGeom g, int header_bytes, const uint32_t* __restrict__ isize, unsigned long long* __restrict__ ioff,
                                                          unsigned long long* __restrict__ fsize) {
    __shared__ uint32_t s_sz[1024];                      // the intervals of the frame, 1024 at a time
    __shared__ uint32_t s_part[256];
    __shared__ unsigned long long s_carry;
    const int tid = threadIdx.x, f = blockIdx.x;
    const uint32_t* sz = isize + (size_t)f * g.nint;
    if (tid == 0) s_carry = (unsigned long long)header_bytes;
    __syncthreads();
    for (int c0 = 0; c0 < g.nint; c0 += 1024) {
        const int n = g.nint - c0 < 1024 ? g.nint - c0 : 1024;
        for (int i = tid; i < n; i += 256) s_sz[i] = sz[c0 + i] + 2u;
        __syncthreads();
        const uint32_t body = wg_exclusive_scan(s_sz, n, s_part, tid);
        const unsigned long long carry = s_carry;
        for (int i = tid; i < n; i += 256) ioff[(size_t)f * g.nint + c0 + i] = carry + s_sz[i];
        __syncthreads();
        if (tid == 0) s_carry = carry + body;
        __syncthreads();
    }
    if (tid == 0) fsize[f] = s_carry;
}

// One thread: frame f of this launch starts where frame f - 1 ended (run[0] carries the position from launch to launch).  A frame that does
// not fit `cap` any more sets run[1] and is given the start ~0 (k_mj_write skips it).
__global__ void k_mj_offsets(int nframes, const unsigned long long* __restrict__ fsize, unsigned long long* __restrict__ fbase, unsigned long long* __restrict__ foff,
                             unsigned long long* __restrict__ run, unsigned long long cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long base = run[0];
    for (int f = 0; f < nframes; ++f) {
        const bool fits = base + fsize[f] <= cap;
        foff[f] = base;
        fbase[f] = fits ? base : ~0ull;
        if (fits) base += fsize[f]; else run[1] = 1ull;
    }
    foff[nframes] = base;
    run[0] = base;
}

__global__ __launch_bounds__(256) void k_mj_write(const uint32_t* __restrict__ raw, MjGeom g, const uint32_t* __restrict__ ibits, const uint32_t* __restrict__ isize,
                                                  const unsigned long long* __restrict__ ioff, const unsigned long long* __restrict__ fbase,
                                                  const uint8_t* __restrict__ header, int header_bytes, uint8_t* __restrict__ jpeg) {
    __shared__ uint32_t s_part[256];
    const int tid = threadIdx.x;
    const int my = blockIdx.x;                            // (the interval's index in its frame)
    const size_t interval = (size_t)blockIdx.y * g.nint + my;
    const unsigned long long fb = fbase[blockIdx.y];
    if (fb == ~0ull) return;                              // (uniform: the frame did not fit)
    const unsigned long long o = fb + ioff[interval];
    size_t mcu0; int nblk_unused;
    mj_interval(g, blockIdx.y, my, mcu0, nblk_unused);
    const uint32_t* in = raw + mcu0 * 6 * MJ_BLOCK_WORDS;
    uint8_t* dst = jpeg + o;
    if (my == 0) for (int i = tid; i < header_bytes; i += 256) (dst - header_bytes)[i] = header[i];
    const uint32_t bits = ibits[interval], nbytes = (bits + 7u) >> 3, nwords = (nbytes + 3u) >> 2;
    const uint32_t per = (nwords + 255u) / 256u, lo = tid * per, hi = lo + per < nwords ? lo + per : nwords;     // a run of words per thread
    uint32_t ff = 0;
    for (uint32_t i = lo; i < hi; ++i) ff += ff_bytes(raw_word(in, i, nbytes, bits), nbytes - 4u * i < 4u ? nbytes - 4u * i : 4u);
    s_part[tid] = ff;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t t = tid >= d ? s_part[tid - d] : 0u;
        __syncthreads();
        s_part[tid] += t;
        __syncthreads();
    }
    uint32_t pos = 4u * lo + (tid ? s_part[tid - 1] : 0u);
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t w = raw_word(in, i, nbytes, bits), n = nbytes - 4u * i < 4u ? nbytes - 4u * i : 4u;
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t b = (w >> (24u - 8u * j)) & 255u;
            dst[pos++] = (uint8_t)b;
            if (b == 255u) dst[pos++] = 0;
        }
    }
    if (tid == 0) {
        const uint32_t n = isize[interval];
        dst[n] = 0xFF;
        dst[n + 1] = (uint8_t)(my + 1 < g.nint ? 0xD0 + (my & 7) : 0xD9);      // RSTm between the intervals, EOI behind the last one
    }
}

template <class T>
int mj_reserve(Ctx* c, T*& p, size_t count) {
    if (p) (void)hipFree(p);
    p = nullptr;
    LVM_HIP_TRY(c, hipMalloc((void**)&p, count * sizeof(T)));
    return LVM_OK;
}

}  // namespace

size_t mjpeg_bound(int w, int h) {
    const size_t mw = (size_t)(w + 15) / 16, mh = (size_t)(h + 15) / 16;
    return (size_t)MJ_MAX_HEADER + mw * mh * 6 * MJ_BLOCK_WORDS * 4 * 2 + mw * mh * 2;      // every entropy byte stuffed, a marker per MCU: a bound, not an estimate
}

void mjpeg_set_restart(Ctx* c, int mcus) {
    MjState* st = static_cast<MjState*>(c->mjpeg);
    if (!st) { st = new MjState; c->mjpeg = st; }
    st->restart = mcus > 0 ? mcus : 0;
}

void mjpeg_release(Ctx* c) {
    MjState* st = static_cast<MjState*>(c->mjpeg);
    if (!st) return;
    void* ptrs[] = {st->d_tab, st->d_header, st->d_coef, st->d_raw, st->d_blk, st->d_fsz, st->d_ibits, st->d_isize, st->d_ioff, st->d_foff, st->d_run, st->d_jpeg};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (st->h_foff) (void)hipHostFree(st->h_foff);
    for (hipEvent_t e : st->ev) (void)hipEventDestroy(e);
    if (st->dl) (void)hipStreamDestroy(st->dl);
    delete st;
    c->mjpeg = nullptr;
}

// Begins a sequence of mjpeg_encode_device calls whose frames are appended to one output buffer of `capacity` bytes on the device.
int mjpeg_begin(Ctx* c, int w, int h, int quality, int max_frames_per_call, size_t total_frames, size_t capacity, hipStream_t s) {
    if (w < 1 || h < 1 || w > MJ_MAX_MW * 16 || h > 16384) { c->err = "lvm_mjpeg: frame size out of range (1..8192 x 1..16384)"; return LVM_ERR_INVALID; }
    if (quality < 1 || quality > 100) { c->err = "lvm_mjpeg: quality must be 1..100"; return LVM_ERR_INVALID; }
    MjState* st = static_cast<MjState*>(c->mjpeg);
    if (!st) { st = new MjState; c->mjpeg = st; }
    const int mw = (w + 15) / 16, mh = (h + 15) / 16;
    // MCUs per restart interval: what was asked for, at most what k_mj_scan's LDS holds, at least what keeps the interval count inside the
    // DRI / grid range.  Default 8: measured against one MCU row (240 MCUs of a 3840-pixel canvas) the encoder is 3 % FASTER (the stuffing
    // kernels work on short pieces), the frames 0.4 % larger, and a decoder that gives every interval a lane (mjpeg_decode.hip) 9 x faster
    int ri = st->restart > 0 ? st->restart : MJ_DEFAULT_RESTART;
    ri = ri > MJ_MAX_MW ? MJ_MAX_MW : ri;
    if ((mw * mh + ri - 1) / ri > 65535) ri = (mw * mh + 65534) / 65535;
    const int nintf = (mw * mh + ri - 1) / ri;
    int rc;
    if (!st->d_tab) { rc = mj_reserve(c, st->d_tab, 1); if (rc != LVM_OK) return rc; }
    if (!st->d_header) { rc = mj_reserve(c, st->d_header, (size_t)MJ_MAX_HEADER); if (rc != LVM_OK) return rc; }
    if (!st->d_run) { rc = mj_reserve(c, st->d_run, 2); if (rc != LVM_OK) return rc; }
    if (st->quality != quality || st->w != w || st->h != h || st->ri != ri) {
        LVM_HIP_TRY(c, hipStreamSynchronize(s));                   // (earlier launches may still read the tables)
        MjTables t;
        std::memset(&t, 0, sizeof t);
        build_huff(kDcLumaBits, kDcVals, 12, t.dc[0], 12);
        build_huff(kDcChromaBits, kDcVals, 12, t.dc[1], 12);
        build_huff(kAcLumaBits, kAcLumaVals, 162, t.ac[0], 256);
        build_huff(kAcChromaBits, kAcChromaVals, 162, t.ac[1], 256);
        int ql[64], qc[64];
        scaled_quant(quality, kQLuma, ql);
        scaled_quant(quality, kQChroma, qc);
        for (int i = 0; i < 64; ++i) {
            const int a = ql[kZigzag[i]], b = qc[kZigzag[i]];
            t.q[0][i] = (uint16_t)a; t.q[1][i] = (uint16_t)b;
            t.qr[0][i] = (uint32_t)(((1u << 24) + a - 1) / a); t.qr[1][i] = (uint32_t)(((1u << 24) + b - 1) / b);
            t.zz[i] = kZigzag[i];
        }
        for (int i = 0; i < 512; ++i) t.aclen[i >> 8][i & 255] = (uint8_t)(t.ac[i >> 8][i & 255] & 255u);
        build_header(st->header, w, h, ql, qc, ri);
        if ((int)st->header.size() > MJ_MAX_HEADER) { c->err = "lvm_mjpeg: header too long"; return LVM_ERR_INVALID; }
        LVM_HIP_TRY(c, hipMemcpy(st->d_tab, &t, sizeof t, hipMemcpyHostToDevice));
        LVM_HIP_TRY(c, hipMemcpy(st->d_header, st->header.data(), st->header.size(), hipMemcpyHostToDevice));
        st->header_bytes = (int)st->header.size();
        if (st->w != w || st->h != h || st->ri != ri) st->frames_cap = 0;          // the scratch buffers are sized per geometry
        st->quality = quality; st->w = w; st->h = h; st->ri = ri; st->nint = nintf;
    }
    if (st->frames_cap < max_frames_per_call) {
        LVM_HIP_TRY(c, hipStreamSynchronize(s));
        const size_t nint = (size_t)nintf * max_frames_per_call, nmcu = (size_t)mh * mw * max_frames_per_call;
        st->frames_cap = 0;
        if ((rc = mj_reserve(c, st->d_coef, nmcu * 384)) != LVM_OK) return rc;
        if ((rc = mj_reserve(c, st->d_raw, nmcu * 6 * MJ_BLOCK_WORDS)) != LVM_OK) return rc;
        if ((rc = mj_reserve(c, st->d_blk, nmcu * 6)) != LVM_OK) return rc;
        if ((rc = mj_reserve(c, st->d_ibits, nint)) != LVM_OK) return rc;
        if ((rc = mj_reserve(c, st->d_isize, nint)) != LVM_OK) return rc;
        if ((rc = mj_reserve(c, st->d_ioff, nint)) != LVM_OK) return rc;
        if ((rc = mj_reserve(c, st->d_fsz, (size_t)max_frames_per_call * 2)) != LVM_OK) return rc;
        st->frames_cap = max_frames_per_call;
    }
    if (st->foff_cap < total_frames + 1) {
        LVM_HIP_TRY(c, hipStreamSynchronize(s));
        st->foff_cap = 0;
        if ((rc = mj_reserve(c, st->d_foff, total_frames + 1)) != LVM_OK) return rc;
        if (st->h_foff) (void)hipHostFree(st->h_foff);
        st->h_foff = nullptr;
        LVM_HIP_TRY(c, hipHostMalloc((void**)&st->h_foff, (total_frames + 1) * sizeof(unsigned long long), 0));
        st->foff_cap = total_frames + 1;
    }
    if (st->jpeg_cap < capacity) {
        LVM_HIP_TRY(c, hipStreamSynchronize(s));
        st->jpeg_cap = 0;
        if ((rc = mj_reserve(c, st->d_jpeg, capacity)) != LVM_OK) return rc;
        st->jpeg_cap = capacity;
    }
    LVM_HIP_TRY(c, hipMemsetAsync(st->d_run, 0, 2 * sizeof(unsigned long long), s));
    if (!st->dl) LVM_HIP_TRY(c, hipStreamCreateWithFlags(&st->dl, hipStreamNonBlocking));
    st->calls.clear(); st->drained = 0; st->copied = 0;
    return LVM_OK;
}

// Encodes `nframes` BGR frames (device) and appends them behind the frames of the earlier calls since mjpeg_begin; frame0 = index of the
// first one in the offsets array.  Everything is enqueued on `s`.
int mjpeg_encode_device(Ctx* c, const uint8_t* d_bgr, ptrdiff_t stride, ptrdiff_t fstride, int nframes, int frame0, size_t capacity, hipStream_t s) {
    MjState* st = static_cast<MjState*>(c->mjpeg);
    if (!st || nframes < 1 || nframes > st->frames_cap || (size_t)(frame0 + nframes + 1) > st->foff_cap || capacity > st->jpeg_cap) { c->err = "lvm_mjpeg: encode without begin"; return LVM_ERR_INVALID; }
    MjGeom g;
    g.w = st->w; g.h = st->h; g.mw = (st->w + 15) / 16; g.mh = (st->h + 15) / 16; g.stride = (long)stride; g.fstride = (long)fstride;
    g.ri = st->ri; g.nint = st->nint;
    const dim3 blk(256);
    LVM_LAUNCH(c, "mj_transform", k_mj_transform, dim3((g.mw + 3) / 4, g.mh, nframes), blk, s, d_bgr, g, (const MjTables*)st->d_tab, st->d_coef, st->d_blk);
    const uint32_t nint = (uint32_t)nframes * (uint32_t)g.nint, rpi = (uint32_t)(g.ri * 6 + MJ_NB - 1) / MJ_NB;
    LVM_LAUNCH(c, "mj_scan", k_mj_scan, dim3(g.nint, nframes), blk, s, (const int16_t*)st->d_coef, g, (const MjTables*)st->d_tab, st->d_blk, st->d_raw, st->d_ibits);
    LVM_LAUNCH(c, "mj_pack", k_mj_pack, dim3((nint * rpi + 3) / 4), blk, s, (const int16_t*)st->d_coef, g, nint, (const MjTables*)st->d_tab, (const uint32_t*)st->d_blk, st->d_raw);
    LVM_LAUNCH(c, "mj_size", k_mj_size, dim3(g.nint, nframes), blk, s, (const uint32_t*)st->d_raw, g, (const uint32_t*)st->d_ibits, st->d_isize);
    LVM_LAUNCH(c, "mj_frame_offsets", k_mj_frame_offsets, dim3(nframes), blk, s, g, st->header_bytes, (const uint32_t*)st->d_isize, st->d_ioff, st->d_fsz);
    LVM_LAUNCH(c, "mj_offsets", k_mj_offsets, dim3(1), dim3(64), s, nframes, (const unsigned long long*)st->d_fsz, st->d_fsz + nframes, st->d_foff + frame0, st->d_run,
               (unsigned long long)capacity);
    LVM_LAUNCH(c, "mj_write", k_mj_write, dim3(g.nint, nframes), blk, s, (const uint32_t*)st->d_raw, g, (const uint32_t*)st->d_ibits, (const uint32_t*)st->d_isize,
               (const unsigned long long*)st->d_ioff, (const unsigned long long*)(st->d_fsz + nframes), (const uint8_t*)st->d_header, st->header_bytes, st->d_jpeg);
    // where these frames ended up: to the page-locked mirror, then an event -- mjpeg_drain downloads finished frames while later calls run
    LVM_HIP_TRY(c, hipMemcpyAsync(st->h_foff + frame0, st->d_foff + frame0, (size_t)(nframes + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    const size_t k = st->calls.size();
    while (st->ev.size() <= k) { hipEvent_t e = nullptr; LVM_HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); st->ev.push_back(e); }
    LVM_HIP_TRY(c, hipEventRecord(st->ev[k], s));
    st->calls.push_back({frame0, nframes});
    return LVM_OK;
}

// Queues the download of the frames of the first `upto` encode calls since mjpeg_begin (those not queued yet): waits for each call's event
// on the HOST (later calls keep the device busy meanwhile), then copies its bytes on the copy queue.
// error paths of the callers: nothing may still be copying into the caller's host buffer when the call returns (ADVICE round 5:
// mjpeg_drain queues its downloads on the encoder's private stream, which the callers' own drains do not know)
void mjpeg_abort(Ctx* c) {
    MjState* st = static_cast<MjState*>(c->mjpeg);
    if (st && st->dl) (void)hipStreamSynchronize(st->dl);
}

int mjpeg_drain(Ctx* c, uint8_t* out_host, size_t upto) {
    MjState* st = static_cast<MjState*>(c->mjpeg);
    if (!st) return LVM_OK;
    for (; st->drained < upto && st->drained < st->calls.size(); ++st->drained) {
        const MjState::Call& k = st->calls[st->drained];
        LVM_HIP_TRY(c, hipEventSynchronize(st->ev[st->drained]));
        const size_t end = (size_t)st->h_foff[k.frame0 + k.nframes];
        if (end > st->copied) {
            LVM_HIP_TRY(c, hipMemcpyAsync(out_host + st->copied, st->d_jpeg + st->copied, end - st->copied, hipMemcpyDeviceToHost, st->dl));
            st->copied = end;
        }
    }
    return LVM_OK;
}

// Completes the sequence: the remaining downloads, the offsets.  offsets[0 .. total_frames] (bytes)
int mjpeg_finish(Ctx* c, size_t total_frames, uint8_t* out_host, size_t* offsets, hipStream_t s) {
    MjState* st = static_cast<MjState*>(c->mjpeg);
    if (!st) { c->err = "lvm_mjpeg: finish without begin"; return LVM_ERR_INVALID; }
    unsigned long long run[2] = {0, 0};
    int rc = mjpeg_drain(c, out_host, st->calls.size());
    if (rc != LVM_OK) { (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(st->dl); return rc; }
    LVM_HIP_TRY(c, hipMemcpyAsync(run, st->d_run, sizeof run, hipMemcpyDeviceToHost, s));
    LVM_HIP_TRY(c, hipStreamSynchronize(s));
    LVM_HIP_TRY(c, hipStreamSynchronize(st->dl));
    if (run[1]) { c->err = "lvm_mjpeg: output buffer too small (lvm_mjpeg_bound() bytes per frame always suffice)"; return LVM_ERR_INVALID; }
    for (size_t i = 0; i <= total_frames; ++i) offsets[i] = (size_t)st->h_foff[i];
    return LVM_OK;
}

}  // namespace lvm
