// mjpeg_decode.hip -- Motion-JPEG decode onto the device (the decode half).
//
// Replaces what `cv::VideoCapture::read(frame)` does for an AVI / Motion-JPEG file (reference: source/FileSource.cpp:99): the compressed
// frames go up (a tenth of the pixels), the decoded BGR frames appear where the chain wants them -- in device memory.
// Accepted: ITU-T T.81 baseline sequential, 8 bit, three components sampled 2x2 / 1x1 / 1x1 (YCbCr 4:2:0, JFIF) in one scan, any quantiser and
// Huffman tables (frames without DHT get the Annex K tables, as AVI MJPEG implies), with or without restart intervals -- what this
// repository's encoder, libjpeg and FFmpeg's mjpeg encoder write.  Everything else is refused (LVM_ERR_INVALID), nothing is guessed.
// The entropy layer is exact by the standard; the arithmetic behind it is the decoder's choice and is restated line by line by
// oracle/mjpeg_oracle.py::decode_frame (all integer; its header has the formulas): the frames are BIT-identical to the oracle's, and within
// the bars of tests/test_mjpeg_decode.py of libjpeg's own decoder.
//
// Huffman decoding is serial inside a restart interval and nowhere else: a LANE per interval (k_mjd_huffman; 68 lanes per frame for this
// repository's streams, one for a stream without restart markers -- then the frames of the batch are the parallelism).
//   host              headers -> per-frame tables (quantisers, 9-bit look-up + canonical tables of the four Huffman codes)
//   k_mjd_intervals   RSTn markers of the entropy-coded segment -> start / end of every interval            one workgroup per frame
//   k_mjd_huffman     bits -> quantised coefficients (zigzag order, int16, zero-filled beforehand)           one lane per interval
//   k_mjd_pixels      dequantise, IDCT, chroma replication, YCbCr -> BGR                                      one wave per 16 x 16 MCU
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lvm_internal.h"
#include "mjpeg_tables.h"

namespace lvm {
namespace {

constexpr int MJD_LUT_BITS = 9;
struct MjdHuff {
    uint16_t lut[1 << MJD_LUT_BITS];    // (length << 8) | symbol for codes of up to 9 bits, 0 = longer
    int32_t maxcode[18];                // T.81 F.2.2.3, -1 where a length has no code; [17] = sentinel
    int32_t mincode[17];
    int32_t valptr[17];
    uint8_t vals[256];
};
struct MjdFrame {
    uint32_t data_off, data_len;        // the entropy-coded segment inside the uploaded bytes
    uint32_t restart;                   // MCUs per restart interval (0: the whole scan)
    uint32_t nintervals;                // expected
    uint32_t par;                       // 1: no restart markers -> decoded by the self-synchronising kernels (k_mjp_*), not a lane per interval
    uint32_t usub;                      // upper bound of its subsequences (from the stuffed length)
    uint32_t tq[3], td[3], ta[3];
    uint16_t q[4][64];                  // quantisers in ZIGZAG order
    uint8_t zz[64];                     // zigzag index -> natural position
    MjdHuff huff[2][2];                 // [class: 0 DC, 1 AC][id 0, 1]
};

struct MjdState {
    int frames_cap = 0, w = 0, h = 0;
    size_t bytes_cap = 0;
    MjdFrame* d_frames = nullptr;
    uint8_t* d_bytes = nullptr;
    int16_t* d_coef = nullptr;
    uint32_t *d_ivstart = nullptr, *d_ivend = nullptr, *d_err = nullptr;
    int iv_cap = 0;
    int n = 0, max_iv = 1;               // the frames of the current begin .. finish sequence
    // self-synchronising path (frames without restart markers)
    uint8_t* d_ubytes = nullptr; size_t ubytes_cap = 0;
    uint32_t *d_ulen = nullptr, *d_before = nullptr, *d_changed = nullptr;
    unsigned long long* d_exit[2] = {nullptr, nullptr};
    int sub_cap = 0, par_frames_cap = 0, npar = 0;
    std::vector<MjdFrame> frames;
    std::vector<uint32_t> foff;
};

int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// false: the code counts do not fit the code space (T.81 Annex C: at every length the codes assigned so far must leave room,
// code + bits[len] <= 2^len) -- an over-subscribed DHT would index past lut[]
bool build_decode_table(const uint8_t* bits, const uint8_t* vals, int nvals, MjdHuff& t) {
    std::memset(&t, 0, sizeof t);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
        t.maxcode[len] = -1;
        if (code + (int)bits[len - 1] > (1 << len)) return false;
        if (bits[len - 1]) {
            t.valptr[len] = k; t.mincode[len] = code;
            for (int i = 0; i < bits[len - 1] && k < nvals; ++i, ++k, ++code) {
                if (len <= MJD_LUT_BITS)
                    for (int f = 0; f < (1 << (MJD_LUT_BITS - len)); ++f) t.lut[(code << (MJD_LUT_BITS - len)) | f] = (uint16_t)((len << 8) | vals[k]);
            }
            t.maxcode[len] = code - 1;
        }
        code <<= 1;
    }
    t.maxcode[17] = 0x7FFFFFFF;
    for (int i = 0; i < nvals && i < 256; ++i) t.vals[i] = vals[i];
    return true;
}

// SOI .. SOS of one frame (oracle: parse_header).  Returns nullptr, or what is unsupported / malformed.
const char* parse_frame(const uint8_t* j, size_t n, int w, int h, MjdFrame& f) {
    if (n < 4 || j[0] != 0xFF || j[1] != 0xD8) return "no SOI";
    bool have_q[4] = {false, false, false, false}, have_h[2][2] = {{false, false}, {false, false}}, sof = false;
    uint32_t cid[3] = {0, 0, 0};
    f.restart = 0;
    for (int z = 0; z < 64; ++z) f.zz[z] = kZigzag[z];
    size_t i = 2;
    for (;;) {
        if (i + 4 > n || j[i] != 0xFF) return "marker expected";
        const int m = j[i + 1];
        if (m == 0xFF) { ++i; continue; }
        const size_t len = (size_t)be16(j + i + 2);
        if (len < 2 || i + 2 + len > n) return "segment runs past the frame";
        const uint8_t* p = j + i + 4;
        const size_t pn = len - 2;
        if (m == 0xDB) {
            for (size_t k = 0; k < pn; k += 65) {
                if (k + 65 > pn || (p[k] >> 4) || (p[k] & 15) > 3) return "quantiser table (16 bit or bad id)";
                for (int z = 0; z < 64; ++z) f.q[p[k] & 15][z] = p[k + 1 + z];
                have_q[p[k] & 15] = true;
            }
        } else if (m == 0xC4) {
            for (size_t k = 0; k < pn;) {
                if (k + 17 > pn) return "Huffman table";
                int nv = 0;
                for (int b = 0; b < 16; ++b) nv += p[k + 1 + b];
                const int cls = p[k] >> 4, id = p[k] & 15;
                if (cls > 1 || id > 1 || nv > 256 || k + 17 + (size_t)nv > pn) return "Huffman table (class / id / size)";
                if (!build_decode_table(p + k + 1, p + k + 17, nv, f.huff[cls][id])) return "Huffman table (code space)";
                have_h[cls][id] = true;
                k += 17 + (size_t)nv;
            }
        } else if (m == 0xC0) {
            if (pn < 15 || p[0] != 8 || p[5] != 3) return "not 8-bit three-component";
            if (be16(p + 1) != h || be16(p + 3) != w) return "frame size differs from the call's";
            for (int c = 0; c < 3; ++c) { cid[c] = p[6 + 3 * c]; f.tq[c] = p[8 + 3 * c]; if (f.tq[c] > 3) return "quantiser id"; }
            if (p[7] != 0x22 || p[10] != 0x11 || p[13] != 0x11) return "sampling is not 4:2:0";
            sof = true;
        } else if (m == 0xC1 || m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            return "not baseline sequential";
        } else if (m == 0xDD) {
            if (pn < 2) return "DRI";
            f.restart = (uint32_t)be16(p);
        } else if (m == 0xDA) {
            if (!sof || pn < 10 || p[0] != 3) return "scan without frame header / not three components";
            for (int c = 0; c < 3; ++c) {
                if (p[1 + 2 * c] != cid[c]) return "scan component order";
                f.td[c] = p[2 + 2 * c] >> 4; f.ta[c] = p[2 + 2 * c] & 15;
                if (f.td[c] > 1 || f.ta[c] > 1) return "Huffman table id";
            }
            if (p[7] != 0 || p[8] != 63 || p[9] != 0) return "not a full baseline scan";        // Ss, Se, Ah | Al
            for (int c = 0; c < 3; ++c) if (!have_q[f.tq[c]]) return "quantiser table missing";
            // frames of an AVI may leave the Huffman tables out: the Annex K tables are implied
            if (!have_h[0][0]) build_decode_table(kDcLumaBits, kDcVals, 12, f.huff[0][0]);
            if (!have_h[1][0]) build_decode_table(kAcLumaBits, kAcLumaVals, 162, f.huff[1][0]);
            if (!have_h[0][1]) build_decode_table(kDcChromaBits, kDcVals, 12, f.huff[0][1]);
            if (!have_h[1][1]) build_decode_table(kAcChromaBits, kAcChromaVals, 162, f.huff[1][1]);
            size_t start = i + 2 + len, end = n;
            if (end >= start + 2 && j[end - 2] == 0xFF && j[end - 1] == 0xD9) end -= 2;      // EOI
            // (bit positions of a frame are 32-bit in the kernels: data_len * 8 must not wrap)
            if (end - start >= ((size_t)1 << 28)) return "entropy-coded segment of 256 MiB or more";
            f.data_off = (uint32_t)start; f.data_len = (uint32_t)(end - start);
            return nullptr;
        }
        i += 2 + len;
    }
}

// ---- kernels ----------------------------------------------------------------------------------------------------------------------------

// One workgroup per frame: the k-th RSTn marker of the entropy-coded segment ends interval k and starts interval k + 1 (FF D0..D7 cannot
// occur inside the data: an FF byte of the data is followed by 00).
__global__ __launch_bounds__(256) void k_mjd_intervals(const uint8_t* __restrict__ bytes, const MjdFrame* __restrict__ frames, const uint32_t* __restrict__ foff,
                                                       uint32_t* __restrict__ ivstart, uint32_t* __restrict__ ivend, int iv_cap, uint32_t* __restrict__ err) {
    __shared__ uint32_t s_part[256];
    const int tid = threadIdx.x, f = blockIdx.x;
    const MjdFrame& fr = frames[f];
    const uint8_t* d = bytes + foff[f] + fr.data_off;
    const uint32_t n = fr.data_len;
    // a run of aligned 32-bit words per thread (+ the first byte of the next one: a marker may straddle two words)
    const uint32_t mis = (uint32_t)((uintptr_t)d & 3u);
    const uint32_t* wbase = reinterpret_cast<const uint32_t*>(d - mis);
    const uint32_t nwords = (n + mis + 3u) >> 2, per = (nwords + 255u) / 256u, lo = tid * per, hi = lo + per < nwords ? lo + per : nwords;
    auto scan = [&](auto&& hit) {
        if (lo >= hi) return;
        uint32_t w = wbase[lo];
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t wn = wbase[i + 1];                                   // (the buffer is padded: always readable)
            const unsigned long long v = (unsigned long long)w | ((unsigned long long)wn << 32);
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const long pos = (long)(4u * i + j) - (long)mis;                // byte position inside the segment
                const uint32_t b0 = (uint32_t)(v >> (8u * j)) & 0xFFu, b1 = (uint32_t)(v >> (8u * j + 8u)) & 0xFFu;
                if (pos >= 0 && pos + 1 < (long)n && b0 == 0xFFu && (b1 & 0xF8u) == 0xD0u) hit((uint32_t)pos);
            }
            w = wn;
        }
    };
    uint32_t cnt = 0;
    scan([&](uint32_t) { ++cnt; });
    s_part[tid] = cnt;
    __syncthreads();
    for (int dd = 1; dd < 256; dd <<= 1) {
        const uint32_t t = tid >= dd ? s_part[tid - dd] : 0u;
        __syncthreads();
        s_part[tid] += t;
        __syncthreads();
    }
    uint32_t k = tid ? s_part[tid - 1] : 0u;
    uint32_t* st = ivstart + (size_t)f * iv_cap;
    uint32_t* en = ivend + (size_t)f * iv_cap;
    scan([&](uint32_t pos) {
        if (k + 1 < (uint32_t)iv_cap) { en[k] = pos; st[k + 1] = pos + 2; }
        ++k;
    });
    if (tid == 255) {
        const uint32_t total = s_part[255] + 1u;
        if (total != fr.nintervals) atomicOr(err + f, 1u);            // the markers do not match the restart interval of the header
        else { st[0] = 0; en[total - 1] = n; }
    }
}

// bit reader over `left` bytes from p: big-endian bits, FF 00 -> FF; behind the end the stream continues with 1-bits (a truncated interval
// decodes to something and reads nothing outside).  The bytes come in aligned 64-bit words, the next word always in flight: a lane's reads are
// a serial chain, and a byte-wise chain costs a memory latency per byte (1.85 ms per 1080p frame measured, against 0.1 with words).
struct MjdBits {
    const unsigned long long* wp;       // the word after `nxt`
    unsigned long long cur, nxt;        // bytes are consumed from the low end of cur
    int nb;                             // bytes left in cur
    long left;                          // bytes left in the interval
    uint32_t acc; int cnt;              // the next bits are the top `cnt` bits of acc
    __device__ __forceinline__ void init(const uint8_t* base, uint32_t start, uint32_t end) {
        const uint8_t* a = base + start;
        const unsigned mis = (unsigned)((uintptr_t)a & 7u);
        wp = reinterpret_cast<const unsigned long long*>(a - mis);
        cur = *wp++ >> (8u * mis); nb = 8 - (int)mis;
        nxt = *wp++;
        left = (long)end - (long)start;
        acc = 0; cnt = 0;
    }
    __device__ __forceinline__ void advance() { cur = nxt; nxt = *wp++; nb = 8; }
    __device__ __forceinline__ void fill() {
        while (cnt <= 24) {
            uint32_t b = 0xFFu;
            if (left > 0) {
                if (nb == 0) advance();
                b = (uint32_t)cur & 0xFFu; cur >>= 8; --nb; --left;
                if (b == 0xFFu && left > 0) {               // a stuffed zero behind it is dropped
                    if (nb == 0) advance();
                    if (((uint32_t)cur & 0xFFu) == 0u) { cur >>= 8; --nb; --left; }
                }
            }
            acc |= b << (24 - cnt);
            cnt += 8;
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return acc >> (32 - n); }
    __device__ __forceinline__ void skip(int n) { acc <<= n; cnt -= n; }
    __device__ __forceinline__ uint32_t get(int n) { if (n == 0) return 0u; fill(); const uint32_t v = peek(n); skip(n); return v; }
};

// one Huffman symbol (T.81 F.2.2.3); -1 = no such code
__device__ __forceinline__ int mjd_symbol(MjdBits& br, const uint16_t* lut, const MjdHuff& t) {
    br.fill();
    const uint32_t e = lut[br.peek(MJD_LUT_BITS)];
    if (e) { br.skip((int)(e >> 8)); return (int)(e & 255u); }
    int code = (int)br.peek(MJD_LUT_BITS);
    for (int len = MJD_LUT_BITS + 1; len <= 16; ++len) {
        code = (int)br.peek(len);
        if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len]) { br.skip(len); return t.vals[t.valptr[len] + code - t.mincode[len]]; }
    }
    return -1;
}
__device__ __forceinline__ int mjd_extend(uint32_t v, int s) { return (s == 0 || v >= (1u << (s - 1))) ? (int)v : (int)v - (1 << s) + 1; }

// A lane per restart interval; the 64 lanes of a workgroup are consecutive intervals of ONE frame, whose look-up tables sit in LDS.
__global__ __launch_bounds__(64) void k_mjd_huffman(const uint8_t* __restrict__ bytes, const MjdFrame* __restrict__ frames, const uint32_t* __restrict__ foff,
                                                    const uint32_t* __restrict__ ivstart, const uint32_t* __restrict__ ivend, int iv_cap, int nmcu,
                                                    int16_t* __restrict__ coef, uint32_t* __restrict__ err) {
    __shared__ uint16_t s_lut[2][2][1 << MJD_LUT_BITS];
    const int f = blockIdx.y, tid = threadIdx.x;
    const MjdFrame& fr = frames[f];
    for (int i = tid; i < 4 << MJD_LUT_BITS; i += 64) s_lut[i >> (MJD_LUT_BITS + 1)][(i >> MJD_LUT_BITS) & 1][i & ((1 << MJD_LUT_BITS) - 1)] =
        fr.huff[i >> (MJD_LUT_BITS + 1)][(i >> MJD_LUT_BITS) & 1].lut[i & ((1 << MJD_LUT_BITS) - 1)];
    __syncthreads();
    const uint32_t k = blockIdx.x * 64u + (uint32_t)tid;
    if (k >= fr.nintervals || err[f] || fr.par) return;
    const uint8_t* d = bytes + foff[f] + fr.data_off;
    MjdBits br;
    br.init(d, ivstart[(size_t)f * iv_cap + k], ivend[(size_t)f * iv_cap + k]);
    const uint32_t ri = fr.restart ? fr.restart : (uint32_t)nmcu;
    const uint32_t m0 = k * ri, m1 = m0 + ri < (uint32_t)nmcu ? m0 + ri : (uint32_t)nmcu;
    int pred[3] = {0, 0, 0};
    bool bad = false;
    for (uint32_t m = m0; m < m1 && !bad; ++m) {
        int16_t* out = coef + ((size_t)f * nmcu + m) * 384;
        for (int bi = 0; bi < 6 && !bad; ++bi) {
            const int comp = bi < 4 ? 0 : bi - 3;
            const uint32_t td = fr.td[comp], ta = fr.ta[comp];
            int s = mjd_symbol(br, s_lut[0][td], fr.huff[0][td]);
            if (s < 0 || s > 11) { bad = true; break; }
            pred[comp] += mjd_extend(br.get(s), s);
            out[bi * 64] = (int16_t)pred[comp];
            for (int kk = 1; kk < 64;) {
                const int rs = mjd_symbol(br, s_lut[1][ta], fr.huff[1][ta]);
                if (rs < 0) { bad = true; break; }
                const int r = rs >> 4;
                s = rs & 15;
                if (s == 0) { if (r != 15) break; kk += 16; continue; }
                kk += r;
                if (kk > 63) { bad = true; break; }
                out[bi * 64 + kk] = (int16_t)mjd_extend(br.get(s), s);
                ++kk;
            }
        }
    }
    if (bad) atomicOr(err + f, 2u);
}

// ---- streams WITHOUT restart markers: self-synchronising parallel decoding ------------------------------------------------------------------
// A Huffman decoder that starts inside a stream at a wrong position almost always falls into step with the true code word boundaries after a
// few symbols (the codes are not fixed-length and the run / size symbols keep re-aligning it).  So (Klein & Wiseman 2003; Weissenberger &
// Schmidt 2018 for JPEG on GPUs): cut the unstuffed entropy segment into subsequences of 1024 bits, a lane each;
//   k_mjp_unstuff   FF 00 -> FF, so that bit positions are plain arithmetic
//   k_mjp_sync      lane i decodes subsequence i from where -- and in the state (block of the MCU, coefficient index) in which -- lane i - 1
//                   last left ITS subsequence, and publishes its own exit (position, state, blocks completed).  First pass: every lane
//                   guesses (start of its subsequence, start of an MCU).  Repeated until no exit changes: lane 0's entry is exact, hence by
//                   induction every lane's.  Code word boundaries are found again within a few symbols, the position inside the MCU (the four
//                   luminance blocks share their tables) takes longer -- hence the look-back of the first pass (8 subsequences): 6 / 16 / 29 changing passes for 102 / 581 / 1 116 subsequences
//                   without it, 0 / 2 / 7 with it.
//   k_mjp_scan      blocks completed before every lane (prefix sum)
//   k_mjp_write     the same decoding once more from the exact entries, now storing coefficients (DC as differences)
//   k_mjp_dc        DC prediction: prefix sums over the blocks of each component
// The result is what the one-lane decoder produces (same tables, same rules), the tests compare both with the oracle.
constexpr uint32_t MJP_SUB_BITS = 1024;
#ifndef MJP_LOOKBACK_N
#define MJP_LOOKBACK_N 8
#endif
constexpr uint32_t MJP_LOOKBACK = MJP_LOOKBACK_N;
struct MjpBits {
    const uint32_t* w; uint32_t nwords, next;
    unsigned long long acc; int cnt;
    uint32_t pos;
    __device__ __forceinline__ uint32_t word(uint32_t i) const { return i < nwords ? __builtin_bswap32(w[i]) : 0xFFFFFFFFu; }     // behind the end: 1-bits
    __device__ __forceinline__ void init(const uint32_t* words, uint32_t n, uint32_t p0) {
        w = words; nwords = n; pos = p0;
        const uint32_t i = p0 >> 5, sh = p0 & 31u;
        acc = (((unsigned long long)word(i) << 32) | word(i + 1)) << sh;
        cnt = 64 - (int)sh; next = i + 2;
    }
    __device__ __forceinline__ void fill() { if (cnt <= 32) { acc |= (unsigned long long)word(next++) << (32 - cnt); cnt += 32; } }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(acc >> (64 - n)); }
    __device__ __forceinline__ void skip(int n) { acc <<= n; cnt -= n; pos += (uint32_t)n; }
    __device__ __forceinline__ uint32_t get(int n) { if (n == 0) return 0u; fill(); const uint32_t v = peek(n); skip(n); return v; }
};
__device__ __forceinline__ int mjp_symbol(MjpBits& br, const uint16_t* lut, const MjdHuff& t) {
    br.fill();
    const uint32_t e = lut[br.peek(MJD_LUT_BITS)];
    if (e) { br.skip((int)(e >> 8)); return (int)(e & 255u); }
    for (int len = MJD_LUT_BITS + 1; len <= 16; ++len) {
        const int code = (int)br.peek(len);
        if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len]) { br.skip(len); return t.vals[t.valptr[len] + code - t.mincode[len]]; }
    }
    return -1;
}
// One symbol in state (bi = block of the MCU, kk = next coefficient index; kk == 0: a DC code comes next).  Returns the zigzag index of the
// coefficient it produced (`value`), -1 for a symbol without one, -2 for what the one-lane decoder calls an error.  `done` = the block ended.
__device__ __forceinline__ int mjp_step(MjpBits& br, int& bi, int& kk, const MjdFrame& fr, const uint16_t (*s_lut)[2][1 << MJD_LUT_BITS], int& value, bool& done) {
    const int comp = bi < 4 ? 0 : bi - 3;
    int ci = -1;
    done = false;
    if (kk == 0) {
        const uint32_t td = fr.td[comp];
        const int s = mjp_symbol(br, s_lut[0][td], fr.huff[0][td]);
        if (s < 0 || s > 11) return -2;
        value = mjd_extend(br.get(s), s);
        ci = 0; kk = 1;
    } else {
        const uint32_t ta = fr.ta[comp];
        const int rs = mjp_symbol(br, s_lut[1][ta], fr.huff[1][ta]);
        if (rs < 0) return -2;
        const int r = rs >> 4, s = rs & 15;
        if (s == 0) {
            if (r == 15) kk += 16; else kk = 64;                 // ZRL / EOB
        } else {
            if (kk + r > 63) return -2;                          // (the state stays what it was: the caller may go on from here)
            kk += r;
            value = mjd_extend(br.get(s), s);
            ci = kk; ++kk;
        }
    }
    if (kk >= 64) { kk = 0; bi = bi == 5 ? 0 : bi + 1; done = true; }
    return ci;
}

// one workgroup per frame: the entropy-coded segment without its stuffed zeros, from a 4-byte aligned address, + its length
__global__ __launch_bounds__(256) void k_mjp_unstuff(const uint8_t* __restrict__ bytes, const MjdFrame* __restrict__ frames, const uint32_t* __restrict__ foff,
                                                     uint8_t* __restrict__ ubytes, uint32_t* __restrict__ ulen) {
    __shared__ uint32_t s_part[256];
    const int tid = threadIdx.x, f = blockIdx.x;
    const MjdFrame& fr = frames[f];
    if (!fr.par) return;
    const uint32_t o = foff[f] + fr.data_off, n = fr.data_len;
    const uint8_t* d = bytes + o;
    uint8_t* u = ubytes + ((o + 3u) & ~3u);
    const uint32_t per = (n + 255u) / 256u, lo = tid * per, hi = lo + per < n ? lo + per : n;
    uint32_t keep = 0;
    for (uint32_t i = lo; i < hi; ++i) keep += (d[i] == 0 && i > 0 && d[i - 1] == 0xFF) ? 0u : 1u;
    s_part[tid] = keep;
    __syncthreads();
    for (int dd = 1; dd < 256; dd <<= 1) {
        const uint32_t t = tid >= dd ? s_part[tid - dd] : 0u;
        __syncthreads();
        s_part[tid] += t;
        __syncthreads();
    }
    uint32_t k = tid ? s_part[tid - 1] : 0u;
    for (uint32_t i = lo; i < hi; ++i) if (!(d[i] == 0 && i > 0 && d[i - 1] == 0xFF)) u[k++] = d[i];
    const uint32_t total = s_part[255];
    if (tid < 8) u[total + tid] = 0xFF;                          // (the word readers look a little ahead)
    if (tid == 0) ulen[f] = total;
}

// exit of a subsequence: bit position in the low half, state and the blocks completed on the way in the high half
__device__ __forceinline__ unsigned long long mjp_pack(uint32_t pos, int bi, int kk, uint32_t blocks) { return (unsigned long long)pos | ((unsigned long long)(bi * 64 + kk) << 32) | ((unsigned long long)blocks << 41); }

__global__ __launch_bounds__(64) void k_mjp_sync(const uint8_t* __restrict__ ubytes, const MjdFrame* __restrict__ frames, const uint32_t* __restrict__ foff,
                                                 const uint32_t* __restrict__ ulen, int sub_cap, int first, const unsigned long long* __restrict__ ein,
                                                 unsigned long long* __restrict__ eout, uint32_t* __restrict__ changed) {
    __shared__ uint16_t s_lut[2][2][1 << MJD_LUT_BITS];
    const int f = blockIdx.y, tid = threadIdx.x;
    const MjdFrame& fr = frames[f];
    for (int i = tid; i < 4 << MJD_LUT_BITS; i += 64) s_lut[i >> (MJD_LUT_BITS + 1)][(i >> MJD_LUT_BITS) & 1][i & ((1 << MJD_LUT_BITS) - 1)] =
        fr.huff[i >> (MJD_LUT_BITS + 1)][(i >> MJD_LUT_BITS) & 1].lut[i & ((1 << MJD_LUT_BITS) - 1)];
    __syncthreads();
    if (!fr.par) return;
    const uint32_t total_bits = ulen[f] * 8u, nsub = (total_bits + MJP_SUB_BITS - 1) / MJP_SUB_BITS, i = blockIdx.x * 64u + (uint32_t)tid;
    if (i >= nsub) return;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(ubytes + ((foff[f] + fr.data_off + 3u) & ~3u));
    // first pass: the guess starts MJP_LOOKBACK subsequences EARLIER (start of an MCU assumed there) and runs up to the lane's own
    // subsequence: by then it has usually fallen into step with the code words and with the blocks of the MCU, so that most lanes are
    // exact after this pass already and the later passes only repair the chains of lanes that were not
    uint32_t p0 = first ? (i > MJP_LOOKBACK ? (i - MJP_LOOKBACK) * MJP_SUB_BITS : 0u) : i * MJP_SUB_BITS;
    int bi = 0, kk = 0;
    if (!first && i > 0) { const unsigned long long e = ein[(size_t)f * sub_cap + i - 1]; p0 = (uint32_t)e; const int st = (int)((e >> 32) & 511u); bi = st >> 6; kk = st & 63; }
    const uint32_t end = (i + 1) * MJP_SUB_BITS < total_bits ? (i + 1) * MJP_SUB_BITS : total_bits;
    MjpBits br;
    br.init(words, (ulen[f] + 3u) >> 2, p0);
    if (first) {
        const uint32_t own = i * MJP_SUB_BITS;
        while (br.pos < own) {
            int value; bool done;
            if (mjp_step(br, bi, kk, fr, s_lut, value, done) == -2) br.skip(1);
        }
    }
    uint32_t blocks = 0;
    while (br.pos < end) {
        int value; bool done;
        if (mjp_step(br, bi, kk, fr, s_lut, value, done) == -2) { br.skip(1); continue; }     // (a wrong guess runs into impossible codes: move on)
        blocks += done ? 1u : 0u;
    }
    const unsigned long long e = mjp_pack(br.pos, bi, kk, blocks);
    if (first || e != ein[(size_t)f * sub_cap + i]) { if (!first) atomicOr(changed, 1u); }
    eout[(size_t)f * sub_cap + i] = e;
}

// one workgroup per frame: blocks completed before every subsequence
__global__ __launch_bounds__(256) void k_mjp_scan(const MjdFrame* __restrict__ frames, const uint32_t* __restrict__ ulen, int sub_cap, const unsigned long long* __restrict__ e,
                                                  uint32_t* __restrict__ before) {
    __shared__ uint32_t s_v[1024];
    __shared__ uint32_t s_part[256];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, f = blockIdx.x;
    if (!frames[f].par) return;
    const int nsub = (int)((ulen[f] * 8u + MJP_SUB_BITS - 1) / MJP_SUB_BITS);
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nsub; c0 += 1024) {
        const int n = nsub - c0 < 1024 ? nsub - c0 : 1024;
        for (int i = tid; i < n; i += 256) s_v[i] = (uint32_t)(e[(size_t)f * sub_cap + c0 + i] >> 41);
        __syncthreads();
        // exclusive scan of s_v[0 .. n)
        const int per = (n + 255) / 256, lo = tid * per, hi = lo + per < n ? lo + per : n;
        uint32_t sum = 0;
        for (int i = lo; i < hi; ++i) sum += s_v[i];
        s_part[tid] = sum;
        __syncthreads();
        for (int dd = 1; dd < 256; dd <<= 1) {
            const uint32_t t = tid >= dd ? s_part[tid - dd] : 0u;
            __syncthreads();
            s_part[tid] += t;
            __syncthreads();
        }
        const uint32_t carry = s_carry;
        uint32_t run = carry + (tid ? s_part[tid - 1] : 0u);
        for (int i = lo; i < hi; ++i) { before[(size_t)f * sub_cap + c0 + i] = run; run += s_v[i]; }
        __syncthreads();
        if (tid == 0) s_carry = carry + s_part[255];
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k_mjp_write(const uint8_t* __restrict__ ubytes, const MjdFrame* __restrict__ frames, const uint32_t* __restrict__ foff,
                                                  const uint32_t* __restrict__ ulen, int sub_cap, const unsigned long long* __restrict__ e, const uint32_t* __restrict__ before,
                                                  int nmcu, int16_t* __restrict__ coef, uint32_t* __restrict__ err) {
    __shared__ uint16_t s_lut[2][2][1 << MJD_LUT_BITS];
    const int f = blockIdx.y, tid = threadIdx.x;
    const MjdFrame& fr = frames[f];
    for (int i = tid; i < 4 << MJD_LUT_BITS; i += 64) s_lut[i >> (MJD_LUT_BITS + 1)][(i >> MJD_LUT_BITS) & 1][i & ((1 << MJD_LUT_BITS) - 1)] =
        fr.huff[i >> (MJD_LUT_BITS + 1)][(i >> MJD_LUT_BITS) & 1].lut[i & ((1 << MJD_LUT_BITS) - 1)];
    __syncthreads();
    if (!fr.par) return;
    const uint32_t total_bits = ulen[f] * 8u, nsub = (total_bits + MJP_SUB_BITS - 1) / MJP_SUB_BITS, i = blockIdx.x * 64u + (uint32_t)tid;
    if (i >= nsub) return;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(ubytes + ((foff[f] + fr.data_off + 3u) & ~3u));
    uint32_t p0 = 0;
    int bi = 0, kk = 0;
    if (i > 0) { const unsigned long long x = e[(size_t)f * sub_cap + i - 1]; p0 = (uint32_t)x; const int st = (int)((x >> 32) & 511u); bi = st >> 6; kk = st & 63; }
    const bool last = i + 1 == nsub;
    const uint32_t end = last ? total_bits + 64u : (uint32_t)e[(size_t)f * sub_cap + i];
    const uint32_t nblk = (uint32_t)nmcu * 6u;
    uint32_t b = before[(size_t)f * sub_cap + i];
    bool bad = (uint32_t)bi != b % 6u;                                    // (the state and the count must tell the same story)
    MjpBits br;
    br.init(words, (ulen[f] + 3u) >> 2, p0);
    int16_t* out = coef + (size_t)f * nmcu * 384;
    while (!bad && b < nblk && br.pos < end) {
        int value = 0; bool done;
        const int ci = mjp_step(br, bi, kk, fr, s_lut, value, done);
        if (ci == -2) { bad = true; break; }
        if (ci >= 0) out[(size_t)b * 64 + ci] = (int16_t)value;
        b += done ? 1u : 0u;
    }
    if (last && b < nblk) bad = true;                                     // the stream ended before the frame did
    if (bad) atomicOr(err + f, 2u);
}

// one workgroup per (component, frame): DC coefficients = prefix sums of the differences over the blocks of the component, in scan order
__global__ __launch_bounds__(256) void k_mjp_dc(const MjdFrame* __restrict__ frames, int nmcu, int16_t* __restrict__ coef) {
    __shared__ int s_v[1024];
    __shared__ int s_part[256];
    __shared__ int s_carry;
    const int tid = threadIdx.x, comp = blockIdx.x, f = blockIdx.y;
    if (!frames[f].par) return;
    int16_t* c = coef + (size_t)f * nmcu * 384;
    const int n_all = comp == 0 ? nmcu * 4 : nmcu;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_all; c0 += 1024) {
        const int n = n_all - c0 < 1024 ? n_all - c0 : 1024;
        for (int i = tid; i < n; i += 256) { const int el = c0 + i; const int b = comp == 0 ? (el >> 2) * 6 + (el & 3) : el * 6 + 3 + comp; s_v[i] = c[(size_t)b * 64]; }
        __syncthreads();
        const int per = (n + 255) / 256, lo = tid * per, hi = lo + per < n ? lo + per : n;
        int sum = 0;
        for (int i = lo; i < hi; ++i) sum += s_v[i];
        s_part[tid] = sum;
        __syncthreads();
        for (int dd = 1; dd < 256; dd <<= 1) {
            const int t = tid >= dd ? s_part[tid - dd] : 0;
            __syncthreads();
            s_part[tid] += t;
            __syncthreads();
        }
        const int carry = s_carry;
        int run = carry + (tid ? s_part[tid - 1] : 0);
        for (int i = lo; i < hi; ++i) { run += s_v[i]; const int el = c0 + i; const int b = comp == 0 ? (el >> 2) * 6 + (el & 3) : el * 6 + 3 + comp; c[(size_t)b * 64] = (int16_t)run; }
        __syncthreads();
        if (tid == 0) s_carry = carry + s_part[255];
        __syncthreads();
    }
}

struct MjdGeom { int w, h, mw, mh; long stride, fstride; };

// One wave per MCU (four per workgroup): dequantise + un-zigzag (lane = zigzag index), IDCT (48 of the 64 lanes: a column / row of one of
// the six blocks each), then lane = one 2 x 2 pixel quad with its replicated chroma sample -> four BGR pixels.
__global__ __launch_bounds__(256) void k_mjd_pixels(const int16_t* __restrict__ coef, const MjdFrame* __restrict__ frames, MjdGeom g, uint8_t* __restrict__ dst) {
    __shared__ int s_a[4][6][64], s_b[4][6][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int mx = blockIdx.x * 4 + wave, my = blockIdx.y, f = blockIdx.z;
    const bool act = mx < g.mw;
    const MjdFrame& fr = frames[f];
    if (act) {
        const int16_t* in = coef + (((size_t)f * g.mh + my) * g.mw + mx) * 384;
        const int nat = fr.zz[lane];
#pragma unroll
        for (int blk = 0; blk < 6; ++blk) {
            const int q = fr.q[fr.tq[blk < 4 ? 0 : blk - 3]][lane];
            int v = in[blk * 64 + lane] * q;
            v = v < -4096 ? -4096 : (v > 4095 ? 4095 : v);
            s_a[wave][blk][nat] = v;
        }
    }
    __syncthreads();
    if (act && lane < 48) {               // columns: t[y][u] = (sum_v M[v][y] S[v][u] + 512) >> 10
        const int blk = lane >> 3, u = lane & 7;
        int d[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) d[v] = s_a[wave][blk][v * 8 + u];
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            int acc = 0;
#pragma unroll
            for (int v = 0; v < 8; ++v) acc += kDct[v][y] * d[v];
            s_b[wave][blk][y * 8 + u] = (acc + 512) >> 10;
        }
    }
    __syncthreads();
    if (act && lane < 48) {               // rows: s[y][x] = (sum_u M[u][x] t[y][u] + 32768) >> 16; + 128; clamp
        const int blk = lane >> 3, y = lane & 7;
        int d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = s_b[wave][blk][y * 8 + u];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            int acc = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += kDct[u][x] * d[u];
            int p = ((acc + 32768) >> 16) + 128;
            s_a[wave][blk][y * 8 + x] = p < 0 ? 0 : (p > 255 ? 255 : p);
        }
    }
    __syncthreads();
    if (act) {
        const int qx = lane & 7, qy = lane >> 3;
        const int cb = s_a[wave][4][lane] - 128, cr = s_a[wave][5][lane] - 128;
        const int dr = (91881 * cr + 32768) >> 16, dg = (-22554 * cb - 46802 * cr + 32768) >> 16, db = (116130 * cb + 32768) >> 16;
        uint8_t* fr_out = dst + (size_t)f * g.fstride;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int ly = 2 * qy + dy, py = my * 16 + ly;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int lx = 2 * qx + dx, px = mx * 16 + lx;
                if (py < g.h && px < g.w) {
                    const int y = s_a[wave][(ly >> 3) * 2 + (lx >> 3)][(ly & 7) * 8 + (lx & 7)];
                    int r = y + dr, gg = y + dg, b = y + db;
                    r = r < 0 ? 0 : (r > 255 ? 255 : r); gg = gg < 0 ? 0 : (gg > 255 ? 255 : gg); b = b < 0 ? 0 : (b > 255 ? 255 : b);
                    uint8_t* o = fr_out + (size_t)py * g.stride + (size_t)px * 3;
                    o[0] = (uint8_t)b; o[1] = (uint8_t)gg; o[2] = (uint8_t)r;
                }
            }
        }
    }
}

template <class T>
int mjd_reserve(Ctx* c, T*& p, size_t count) {
    if (p) (void)hipFree(p);
    p = nullptr;
    LVM_HIP_TRY(c, hipMalloc((void**)&p, count * sizeof(T)));
    return LVM_OK;
}

}  // namespace

void mjpeg_decode_release(Ctx* c) {
    MjdState* st = static_cast<MjdState*>(c->mjpeg_dec);
    if (!st) return;
    void* ptrs[] = {st->d_frames, st->d_bytes, st->d_coef, st->d_ivstart, st->d_ivend, st->d_err, st->d_ubytes, st->d_ulen, st->d_before, st->d_changed, st->d_exit[0], st->d_exit[1]};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete st;
    c->mjpeg_dec = nullptr;
}

// Three steps, so that a caller can decode sub-batch by sub-batch between other work on the same queue (lvm_export_mjpeg_frames):
//   begin    parses every frame, uploads all bytes and tables (asynchronously on s)
//   enqueue  the three kernels for frames [f0, f0 + nf) -> BGR frames at d_bgr
//   finish   waits for s and turns the error flags into the status of the call (a malformed stream is an error, not a picture)
int mjpeg_decode_begin(Ctx* c, const uint8_t* jpegs, const size_t* offsets, int n, int w, int h, hipStream_t s) {
    if (w < 1 || h < 1 || w > 8192 || h > 16384) { c->err = "lvm_mjpeg_decode: frame size out of range (1..8192 x 1..16384)"; return LVM_ERR_INVALID; }
    MjdState* st = static_cast<MjdState*>(c->mjpeg_dec);
    if (!st) { st = new MjdState; c->mjpeg_dec = st; }
    const int mw = (w + 15) / 16, mh = (h + 15) / 16, nmcu = mw * mh;
    st->frames.resize((size_t)n);
    std::vector<uint32_t> foff((size_t)n + 1);
    int max_iv = 1, max_sub = 1, npar = 0;
    for (int i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[0] > 0xFFFFFFF0ull) { c->err = "lvm_mjpeg_decode: bad offsets"; return LVM_ERR_INVALID; }
        MjdFrame& f = st->frames[(size_t)i];
        const char* why = parse_frame(jpegs + offsets[i], offsets[i + 1] - offsets[i], w, h, f);
        if (why) { c->err = std::string("lvm_mjpeg_decode: frame ") + std::to_string(i) + ": " + why; return LVM_ERR_INVALID; }
        f.nintervals = f.restart ? (uint32_t)((nmcu + (int)f.restart - 1) / (int)f.restart) : 1u;
        if ((int)f.nintervals > max_iv) max_iv = (int)f.nintervals;
        // no restart markers: one lane would decode the whole frame -- the self-synchronising kernels take it (LVM_MJD_PARALLEL=0: never, =2: also tiny frames)
        static const int par_mode = [] { const char* e = std::getenv("LVM_MJD_PARALLEL"); return e ? std::atoi(e) : 1; }();
        f.par = (f.restart == 0 && par_mode != 0 && (f.data_len >= 2048u || par_mode == 2)) ? 1u : 0u;
        f.usub = (f.data_len * 8u + MJP_SUB_BITS - 1) / MJP_SUB_BITS + 1u;
        if (f.par) { ++npar; if ((int)f.usub > max_sub) max_sub = (int)f.usub; }
        foff[(size_t)i] = (uint32_t)(offsets[i] - offsets[0]);
    }
    foff[(size_t)n] = (uint32_t)(offsets[n] - offsets[0]);
    const size_t nbytes = offsets[n] - offsets[0];
    int rc;
    if (st->frames_cap < n || st->w != w || st->h != h) {
        LVM_HIP_TRY(c, hipStreamSynchronize(s));
        st->frames_cap = 0;
        if ((rc = mjd_reserve(c, st->d_frames, (size_t)n)) != LVM_OK) return rc;
        if ((rc = mjd_reserve(c, st->d_coef, (size_t)n * nmcu * 384)) != LVM_OK) return rc;
        if ((rc = mjd_reserve(c, st->d_err, (size_t)n * 2 + 2)) != LVM_OK) return rc;       // error flags, then the frame offsets (n + 1)
        st->frames_cap = n; st->w = w; st->h = h; st->iv_cap = 0;
    }
    if (st->iv_cap < max_iv) {
        LVM_HIP_TRY(c, hipStreamSynchronize(s));
        st->iv_cap = 0;
        if ((rc = mjd_reserve(c, st->d_ivstart, (size_t)st->frames_cap * max_iv)) != LVM_OK) return rc;
        if ((rc = mjd_reserve(c, st->d_ivend, (size_t)st->frames_cap * max_iv)) != LVM_OK) return rc;
        st->iv_cap = max_iv;
    }
    if (st->bytes_cap < nbytes + 32) {
        LVM_HIP_TRY(c, hipStreamSynchronize(s));
        st->bytes_cap = 0;
        if ((rc = mjd_reserve(c, st->d_bytes, nbytes + 32)) != LVM_OK) return rc;      // (+ 32: the word readers look one or two words ahead)
        st->bytes_cap = nbytes + 32;
    }
    st->npar = npar;
    if (npar) {
        if (st->ubytes_cap < nbytes + 64) {
            LVM_HIP_TRY(c, hipStreamSynchronize(s));
            st->ubytes_cap = 0;
            if ((rc = mjd_reserve(c, st->d_ubytes, nbytes + 64)) != LVM_OK) return rc;
            st->ubytes_cap = nbytes + 64;
        }
        if (st->par_frames_cap < n || st->sub_cap < max_sub) {
            LVM_HIP_TRY(c, hipStreamSynchronize(s));
            st->par_frames_cap = 0;
            const int cap = max_sub > st->sub_cap ? max_sub : st->sub_cap;
            if ((rc = mjd_reserve(c, st->d_ulen, (size_t)n)) != LVM_OK) return rc;
            if ((rc = mjd_reserve(c, st->d_before, (size_t)n * cap)) != LVM_OK) return rc;
            if ((rc = mjd_reserve(c, st->d_exit[0], (size_t)n * cap)) != LVM_OK) return rc;
            if ((rc = mjd_reserve(c, st->d_exit[1], (size_t)n * cap)) != LVM_OK) return rc;
            if (!st->d_changed && (rc = mjd_reserve(c, st->d_changed, (size_t)1)) != LVM_OK) return rc;
            st->par_frames_cap = n; st->sub_cap = cap;
        }
    }
    st->n = n; st->max_iv = max_iv;
    st->foff = foff;                                             // (stays alive until the copy below has run)
    LVM_HIP_TRY(c, hipMemcpyAsync(st->d_bytes, jpegs + offsets[0], nbytes, hipMemcpyHostToDevice, s));
    LVM_HIP_TRY(c, hipMemcpyAsync(st->d_frames, st->frames.data(), (size_t)n * sizeof(MjdFrame), hipMemcpyHostToDevice, s));
    LVM_HIP_TRY(c, hipMemsetAsync(st->d_err, 0, (size_t)n * sizeof(uint32_t), s));
    LVM_HIP_TRY(c, hipMemcpyAsync(st->d_err + n, st->foff.data(), ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    return LVM_OK;
}

int mjpeg_decode_enqueue(Ctx* c, int f0, int nf, uint8_t* d_bgr, ptrdiff_t stride, ptrdiff_t fstride, hipStream_t s) {
    MjdState* st = static_cast<MjdState*>(c->mjpeg_dec);
    if (!st || f0 < 0 || nf < 1 || f0 + nf > st->n) { c->err = "lvm_mjpeg_decode: enqueue without begin"; return LVM_ERR_INVALID; }
    const int w = st->w, h = st->h, mw = (w + 15) / 16, mh = (h + 15) / 16, nmcu = mw * mh;
    const MjdFrame* fr = st->d_frames + f0;
    const uint32_t* d_foff = st->d_err + st->n + f0;
    int16_t* coef = st->d_coef + (size_t)f0 * nmcu * 384;
    uint32_t *ivs = st->d_ivstart + (size_t)f0 * st->iv_cap, *ive = st->d_ivend + (size_t)f0 * st->iv_cap, *err = st->d_err + f0;
    LVM_HIP_TRY(c, hipMemsetAsync(coef, 0, (size_t)nf * nmcu * 384 * sizeof(int16_t), s));
    LVM_LAUNCH(c, "mjd_intervals", k_mjd_intervals, dim3(nf), dim3(256), s, (const uint8_t*)st->d_bytes, fr, d_foff, ivs, ive, st->iv_cap, err);
    LVM_LAUNCH(c, "mjd_huffman", k_mjd_huffman, dim3((st->max_iv + 63) / 64, nf), dim3(64), s, (const uint8_t*)st->d_bytes, fr, d_foff, (const uint32_t*)ivs,
               (const uint32_t*)ive, st->iv_cap, nmcu, coef, err);
    bool any_par = false;
    for (int i = f0; i < f0 + nf; ++i) any_par = any_par || st->frames[(size_t)i].par;
    if (any_par) {
        const uint32_t* ulen = st->d_ulen + f0;
        unsigned long long* ex[2] = {st->d_exit[0] + (size_t)f0 * st->sub_cap, st->d_exit[1] + (size_t)f0 * st->sub_cap};
        uint32_t* before = st->d_before + (size_t)f0 * st->sub_cap;
        const dim3 gsub((unsigned)((st->sub_cap + 63) / 64), (unsigned)nf);
        LVM_LAUNCH(c, "mjp_unstuff", k_mjp_unstuff, dim3(nf), dim3(256), s, (const uint8_t*)st->d_bytes, fr, d_foff, st->d_ubytes, st->d_ulen + f0);
        LVM_LAUNCH(c, "mjp_sync", k_mjp_sync, gsub, dim3(64), s, (const uint8_t*)st->d_ubytes, fr, d_foff, ulen, st->sub_cap, 1, (const unsigned long long*)ex[1], ex[0], st->d_changed);
        int cur = 0;                                               // ex[cur] holds the latest exits
        for (int it = 0; it <= st->sub_cap; ++it) {                // (every pass makes at least one more lane exact: sub_cap passes always suffice)
            uint32_t changed = 0;
            LVM_HIP_TRY(c, hipMemsetAsync(st->d_changed, 0, sizeof(uint32_t), s));
            LVM_LAUNCH(c, "mjp_sync", k_mjp_sync, gsub, dim3(64), s, (const uint8_t*)st->d_ubytes, fr, d_foff, ulen, st->sub_cap, 0, (const unsigned long long*)ex[cur], ex[cur ^ 1],
                       st->d_changed);
            LVM_HIP_TRY(c, hipMemcpyAsync(&changed, st->d_changed, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            LVM_HIP_TRY(c, hipStreamSynchronize(s));
            cur ^= 1;
            if (!changed) break;
        }
        LVM_LAUNCH(c, "mjp_scan", k_mjp_scan, dim3(nf), dim3(256), s, fr, ulen, st->sub_cap, (const unsigned long long*)ex[cur], before);
        LVM_LAUNCH(c, "mjp_write", k_mjp_write, gsub, dim3(64), s, (const uint8_t*)st->d_ubytes, fr, d_foff, ulen, st->sub_cap, (const unsigned long long*)ex[cur],
                   (const uint32_t*)before, nmcu, coef, err);
        LVM_LAUNCH(c, "mjp_dc", k_mjp_dc, dim3(3, nf), dim3(256), s, fr, nmcu, coef);
    }
    MjdGeom g;
    g.w = w; g.h = h; g.mw = mw; g.mh = mh; g.stride = (long)stride; g.fstride = (long)fstride;
    LVM_LAUNCH(c, "mjd_pixels", k_mjd_pixels, dim3((mw + 3) / 4, mh, nf), dim3(256), s, (const int16_t*)coef, fr, g, d_bgr);
    return LVM_OK;
}

int mjpeg_decode_finish(Ctx* c, hipStream_t s) {
    MjdState* st = static_cast<MjdState*>(c->mjpeg_dec);
    if (!st) { c->err = "lvm_mjpeg_decode: finish without begin"; return LVM_ERR_INVALID; }
    std::vector<uint32_t> err((size_t)st->n);
    LVM_HIP_TRY(c, hipMemcpyAsync(err.data(), st->d_err, (size_t)st->n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LVM_HIP_TRY(c, hipStreamSynchronize(s));
    for (int i = 0; i < st->n; ++i)
        if (err[(size_t)i]) {
            c->err = std::string("lvm_mjpeg_decode: frame ") + std::to_string(i) + (err[(size_t)i] & 1u ? ": restart markers do not match the restart interval" : ": invalid Huffman code / coefficient index");
            return LVM_ERR_INVALID;
        }
    return LVM_OK;
}

int mjpeg_decode_device(Ctx* c, const uint8_t* jpegs, const size_t* offsets, int n, int w, int h, uint8_t* d_bgr, ptrdiff_t stride, ptrdiff_t fstride, hipStream_t s) {
    int rc = mjpeg_decode_begin(c, jpegs, offsets, n, w, h, s);
    if (rc == LVM_OK) rc = mjpeg_decode_enqueue(c, 0, n, d_bgr, stride, fstride, s);
    if (rc != LVM_OK) { (void)hipStreamSynchronize(s); return rc; }
    return mjpeg_decode_finish(c, s);
}

}  // namespace lvm
