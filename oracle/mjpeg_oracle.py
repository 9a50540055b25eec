"""CPU ORACLE of the Motion-JPEG encode step (the encode half): what `cv::VideoWriter::write(canvas)` does for
`ExportFormat::AviMjpg` -- the reference's AVI export format and its universal fallback (src/export/Exporter.cpp:107-117, :259).

TEST INFRASTRUCTURE ONLY: imported by tests/ and bench.py's verification leg, never by the product package.

The arithmetic lives in a third-party dependency that is absent here (OpenCV 4's videoio: its own `cap_mjpeg_encoder.cpp` or FFmpeg's
`mjpeg` encoder, depending on the build; vcpkg `opencv4[ffmpeg]`, not vendored).  JPEG is lossy and every encoder picks its own colour
rounding, DCT and quantiser rounding: there is no bit-exact target at that boundary.  What the boundary does fix is the FORMAT -- ITU-T T.81
baseline sequential DCT, 8-bit, YCbCr 4:2:0 (JFIF / BT.601 full range), Huffman tables of Annex K -- so this file restates T.81 with one
fully specified integer arithmetic, and parity means:
  (1) the HIP encoder's bitstream is BYTE-IDENTICAL to `encode_frame()` here (tests/test_mjpeg*.py);
  (2) the stream is pinned against an independent decoder that IS in this image: Pillow 12 (libjpeg-turbo) decodes it, the decoded frame is
      within the quantisation error of the input (PSNR bars in the tests), and the DQT / DHT segments equal the ones libjpeg itself writes.
The arithmetic (every step integer, `>>` is the arithmetic shift):
  colour     Y  = (19595 R + 38470 G +  7471 B + 32768) >> 16
             Cb = (-11059 R - 21709 G + 32768 B + 8421375) >> 16          (8421375 = (128 << 16) + 32767)
             Cr = ( 32768 R - 27439 G -  5329 B + 8421375) >> 16
  4:2:0      chroma sample = (sum of the 2 x 2 Cb (Cr) values + 2) >> 2; frames are padded to 16 x 16 MCUs by edge replication
  FDCT       samples - 128; rows: t[u] = (sum_x M[u][x] d[x] + 512) >> 10; columns: S[v] = (sum_y M[v][y] t[y] + 32768) >> 16,
             M[u][x] = round(8192 * C(u) / 2 * cos((2 x + 1) u pi / 16)), C(0) = 1 / sqrt 2
  quantiser  libjpeg's quality scaling of the Annex K tables; q(c) = sign(c) * ((|c| + Q / 2) // Q)
  entropy    Annex K Huffman tables; restart intervals of 8 MCUs (or as asked), DC predictors reset there (T.81 F.1.1.5.1 / E.1.4);
             byte stuffing FF -> FF 00, intervals padded with 1-bits, RST0..7 between them
"""
import math

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42,
                   49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], np.int32)
# T.81 Annex K.1 / K.2 (natural order)
Q_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                   18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100,
                   103, 99], np.int32)
Q_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99]
                    + [99] * 32, np.int32)
# T.81 Annex K.3: (number of codes of length 1..16, values)
DC_LUMA = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_CHROMA = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_LUMA = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
           [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
            0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
            0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
            0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
            0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
            0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
            0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
AC_CHROMA = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
             [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
              0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
              0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
              0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
              0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
              0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
              0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])


def dct_matrix():
    """M[u][x] = round(8192 * C(u) / 2 * cos((2 x + 1) u pi / 16))   (T.81 A.3.3, one dimension)"""
    m = np.zeros((8, 8), np.int64)
    for u in range(8):
        cu = 1.0 / math.sqrt(2.0) if u == 0 else 1.0
        for x in range(8):
            m[u, x] = int(round(8192.0 * cu / 2.0 * math.cos((2 * x + 1) * u * math.pi / 16.0)))
    return m


def quant_tables(quality):
    """libjpeg's jpeg_quality_scaling + jpeg_add_quant_table (baseline: entries clamped to 1..255); natural order"""
    quality = max(1, min(100, int(quality)))
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    out = []
    for base in (Q_LUMA, Q_CHROMA):
        t = (base.astype(np.int64) * scale + 50) // 100
        out.append(np.clip(t, 1, 255).astype(np.int32))
    return out


def huff_codes(spec):
    """T.81 Annex C: code and length of every symbol"""
    bits, vals = spec
    code, k, table = 0, 0, {}
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            table[vals[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return table


def _seg(marker, payload):
    return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload


def header(w, h, quality, restart_interval):
    """SOI APP0(JFIF) DQT DQT SOF0 DHT x 4 DRI SOS"""
    ql, qc = quant_tables(quality)
    out = b"\xff\xd8"
    out += _seg(0xE0, b"JFIF\x00" + bytes([1, 1, 0, 0, 1, 0, 1, 0, 0]))
    out += _seg(0xDB, bytes([0]) + bytes(int(ql[z]) for z in ZIGZAG))
    out += _seg(0xDB, bytes([1]) + bytes(int(qc[z]) for z in ZIGZAG))
    out += _seg(0xC0, bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1]))
    for cls_id, spec in ((0x00, DC_LUMA), (0x10, AC_LUMA), (0x01, DC_CHROMA), (0x11, AC_CHROMA)):
        out += _seg(0xC4, bytes([cls_id]) + bytes(spec[0]) + bytes(spec[1]))
    out += _seg(0xDD, restart_interval.to_bytes(2, "big"))
    out += _seg(0xDA, bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0]))
    return out


def coefficients(bgr, quality):
    """BGR u8 [h][w][3] -> quantised coefficients int32 [mcu_rows][mcu_cols][6][64] in zigzag order (blocks: Y00 Y01 Y10 Y11 Cb Cr)"""
    h, w, _ = bgr.shape
    mw, mh = (w + 15) // 16, (h + 15) // 16
    p = np.pad(bgr, ((0, mh * 16 - h), (0, mw * 16 - w), (0, 0)), mode="edge").astype(np.int64)
    b, g, r = p[..., 0], p[..., 1], p[..., 2]
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + 8421375) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + 8421375) >> 16

    def sub(c):
        return (c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2] + 2) >> 2
    cb, cr = sub(cb), sub(cr)
    m = dct_matrix()
    ql, qc = quant_tables(quality)

    def blocks(plane, n):        # [H][W] -> [H/8][W/8][8][8]
        hh, ww = plane.shape
        return plane.reshape(hh // 8, 8, ww // 8, 8).transpose(0, 2, 1, 3)

    def fdct_q(blk, q):
        d = blk - 128
        t = (np.einsum("ux,...yx->...yu", m, d) + 512) >> 10          # rows
        s = (np.einsum("vy,...yu->...vu", m, t) + 32768) >> 16        # columns
        s = s.reshape(s.shape[:-2] + (64,))
        qq = q.astype(np.int64)
        a = (np.abs(s) + qq // 2) // qq
        c = np.where(s < 0, -a, a)
        return c[..., ZIGZAG].astype(np.int32)
    yb = fdct_q(blocks(y, 8), ql)            # [2 mh][2 mw][64]
    cbb, crb = fdct_q(blocks(cb, 8), qc), fdct_q(blocks(cr, 8), qc)
    out = np.zeros((mh, mw, 6, 64), np.int32)
    out[:, :, 0] = yb[0::2, 0::2]
    out[:, :, 1] = yb[0::2, 1::2]
    out[:, :, 2] = yb[1::2, 0::2]
    out[:, :, 3] = yb[1::2, 1::2]
    out[:, :, 4] = cbb
    out[:, :, 5] = crb
    return out


class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, code, length):
        self.acc = (self.acc << length) | (code & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _category(v):
    return int(abs(int(v))).bit_length()


def entropy_interval(mcus, tabs):
    """One restart interval (a row of MCUs): [n][6][64] coefficients -> stuffed, byte-aligned bytes (T.81 F.1.2)"""
    dcl, acl, dcc, acc_ = tabs
    bw = _Bits()
    pred = [0, 0, 0]
    for mcu in mcus:
        for bi in range(6):
            comp = 0 if bi < 4 else bi - 3
            dc_t, ac_t = (dcl, acl) if comp == 0 else (dcc, acc_)
            blk = mcu[bi]
            diff = int(blk[0]) - pred[comp]
            pred[comp] = int(blk[0])
            s = _category(diff)
            bw.put(*dc_t[s])
            if s:
                bw.put(diff if diff > 0 else diff - 1, s)
            run = 0
            last = int(np.max(np.nonzero(blk[1:])[0])) + 1 if np.any(blk[1:]) else 0
            for k in range(1, last + 1):
                v = int(blk[k])
                if v == 0:
                    run += 1
                    continue
                while run >= 16:
                    bw.put(*ac_t[0xF0])
                    run -= 16
                s = _category(v)
                bw.put(*ac_t[(run << 4) | s])
                bw.put(v if v > 0 else v - 1, s)
                run = 0
            if last < 63:
                bw.put(*ac_t[0x00])
    bw.flush()
    return bytes(bw.out)


def encode_frame(bgr, quality=75, restart=None):
    """BGR u8 frame -> one JPEG (bytes); restart = MCUs per restart interval (raster order, T.81 E.1.4), default 8"""
    h, w, _ = bgr.shape
    c = coefficients(bgr, quality)
    mh, mw = c.shape[:2]
    ri = 8 if not restart else min(int(restart), 512)
    if (mh * mw + ri - 1) // ri > 65535:
        ri = (mh * mw + 65534) // 65535
    flat = c.reshape(mh * mw, 6, 64)
    nint = (mh * mw + ri - 1) // ri
    tabs = tuple(huff_codes(s) for s in (DC_LUMA, AC_LUMA, DC_CHROMA, AC_CHROMA))
    out = bytearray(header(w, h, quality, ri))
    for r in range(nint):
        out += entropy_interval(flat[r * ri:(r + 1) * ri], tabs)
        if r + 1 < nint:
            out += bytes([0xFF, 0xD0 + (r & 7)])
    out += b"\xff\xd9"
    return bytes(out)


def psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return 99.0 if mse == 0 else 10.0 * math.log10(255.0 * 255.0 / mse)


# ---- decoder (the decode half: cv::VideoCapture::read on an AVI / Motion-JPEG file, source/FileSource.cpp:99) ----------
# Baseline sequential, 8 bit, three components sampled 2x2 / 1x1 / 1x1, any quantiser and Huffman tables, with or without restart intervals
# -- what this repo's encoder, libjpeg (4:2:0) and FFmpeg's mjpeg encoder (yuvj420p) write.  Again the standard fixes the entropy layer
# exactly and leaves the arithmetic after it to the decoder; this one (all integer):
#   dequantise   S = clamp(c * Q, -4096, 4095)
#   IDCT         columns: t[y][u] = (sum_v M[v][y] S[v][u] + 512) >> 10; rows: s[y][x] = (sum_u M[u][x] t[y][u] + 32768) >> 16; + 128, clamp 0..255
#   chroma       replicated 2 x 2 (no smoothing)
#   colour       R = Y + ((91881 Cr' + 32768) >> 16), G = Y + ((-22554 Cb' - 46802 Cr' + 32768) >> 16), B = Y + ((116130 Cb' + 32768) >> 16),
#                Cb' = Cb - 128, Cr' = Cr - 128; clamp 0..255
# The HIP decoder is BIT-identical to decode_frame(); libjpeg's own decoder (slow integer IDCT, "fancy" chroma upsampling) agrees with it to
# within the bars in tests/test_mjpeg_decode.py.

class JpegError(ValueError):
    pass


def parse_header(j):
    """-> dict(w, h, q[4] (natural order), huff {(class, id): (bits, vals)}, restart, comps [(id, h, v, tq)], scan [(cid, td, ta)], data_start)"""
    if j[:2] != b"\xff\xd8":
        raise JpegError("no SOI")
    i, out = 2, {"q": {}, "huff": {}, "restart": 0}
    while True:
        if i + 4 > len(j) or j[i] != 0xFF:
            raise JpegError("marker expected at %d" % i)
        m = j[i + 1]
        if m == 0xFF:
            i += 1
            continue
        n = int.from_bytes(j[i + 2:i + 4], "big")
        p = j[i + 4:i + 2 + n]
        if m == 0xDB:
            k = 0
            while k < len(p):
                if p[k] >> 4:
                    raise JpegError("16-bit quantiser")
                t = np.zeros(64, np.int32)
                t[ZIGZAG] = np.frombuffer(p[k + 1:k + 65], np.uint8)
                out["q"][p[k] & 15] = t
                k += 65
        elif m == 0xC4:
            k = 0
            while k < len(p):
                bits = list(p[k + 1:k + 17])
                nv = sum(bits)
                out["huff"][(p[k] >> 4, p[k] & 15)] = (bits, list(p[k + 17:k + 17 + nv]))
                k += 17 + nv
        elif m == 0xC0:
            if p[0] != 8:
                raise JpegError("precision")
            out["h"], out["w"] = int.from_bytes(p[1:3], "big"), int.from_bytes(p[3:5], "big")
            out["comps"] = [(p[6 + 3 * c], p[7 + 3 * c] >> 4, p[7 + 3 * c] & 15, p[8 + 3 * c]) for c in range(p[5])]
        elif m in (0xC1, 0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise JpegError("not baseline")
        elif m == 0xDD:
            out["restart"] = int.from_bytes(p[:2], "big")
        elif m == 0xDA:
            out["scan"] = [(p[1 + 2 * c], p[2 + 2 * c] >> 4, p[2 + 2 * c] & 15) for c in range(p[0])]
            out["data_start"] = i + 2 + n
            return out
        i += 2 + n


def _decode_tables(spec):
    """T.81 F.2.2.3: mincode / maxcode / valptr per code length"""
    bits, vals = spec
    mincode, maxcode, valptr, code, k = [0] * 17, [-1] * 17, [0] * 17, 0, 0
    for length in range(1, 17):
        if bits[length - 1]:
            valptr[length], mincode[length] = k, code
            code += bits[length - 1]
            k += bits[length - 1]
            maxcode[length] = code - 1
        code <<= 1
    return mincode, maxcode, valptr, vals


class _BitReader:
    def __init__(self, data):
        self.d, self.i, self.acc, self.n = data, 0, 0, 0

    def bit(self):
        if self.n == 0:
            if self.i >= len(self.d):
                self.acc, self.n = 0xFF, 8            # past the end: 1-bits (a truncated stream decodes to something, never reads outside)
            else:
                b = self.d[self.i]
                self.i += 1
                if b == 0xFF and self.i < len(self.d) and self.d[self.i] == 0:
                    self.i += 1
                self.acc, self.n = b, 8
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v


def _decode_symbol(br, tab):
    mincode, maxcode, valptr, vals = tab
    code = 0
    for length in range(1, 17):
        code = (code << 1) | br.bit()
        if maxcode[length] >= 0 and code <= maxcode[length] and code >= mincode[length]:
            return vals[valptr[length] + code - mincode[length]]
    raise JpegError("invalid Huffman code")


def _extend(v, s):
    return v if s == 0 or v >= (1 << (s - 1)) else v - (1 << s) + 1


def split_intervals(data):
    """entropy-coded segment -> list of intervals (bytes between RSTn markers), the segment ends at the first other marker"""
    out, start, i = [], 0, 0
    while i + 1 < len(data):
        if data[i] == 0xFF:
            nx = data[i + 1]
            if 0xD0 <= nx <= 0xD7:
                out.append(data[start:i])
                start = i + 2
                i += 2
                continue
            if nx != 0:
                break
            i += 1
        i += 1
    else:
        i = len(data)
    out.append(data[start:i])
    return out


def decode_coefficients(j):
    """-> (header, int32 [mh][mw][6][64] quantised coefficients in ZIGZAG order)"""
    hd = parse_header(j)
    if [c[1:3] for c in hd.get("comps", [])] != [(2, 2), (1, 1), (1, 1)] or len(hd["scan"]) != 3:
        raise JpegError("only YCbCr 4:2:0 in one scan")
    tabs = {}
    for key, std in (((0, 0), DC_LUMA), ((1, 0), AC_LUMA), ((0, 1), DC_CHROMA), ((1, 1), AC_CHROMA)):
        tabs[key] = _decode_tables(hd["huff"].get(key, std))          # AVI MJPEG frames may omit DHT: the Annex K tables are implied
    for key, spec in hd["huff"].items():
        tabs[key] = _decode_tables(spec)
    w, h = hd["w"], hd["h"]
    mw, mh = (w + 15) // 16, (h + 15) // 16
    coef = np.zeros((mh * mw, 6, 64), np.int32)
    ri = hd["restart"] if hd["restart"] else mh * mw
    ivs = split_intervals(j[hd["data_start"]:])
    sel = [hd["scan"][0]] * 4 + [hd["scan"][1], hd["scan"][2]]
    for k, seg in enumerate(ivs):
        br, pred = _BitReader(seg), [0, 0, 0]
        for m in range(k * ri, min((k + 1) * ri, mh * mw)):
            for bi in range(6):
                comp = 0 if bi < 4 else bi - 3
                _, td, ta = sel[bi]
                s = _decode_symbol(br, tabs[(0, td)])
                pred[comp] += _extend(br.bits(s), s)
                coef[m, bi, 0] = pred[comp]
                kk = 1
                while kk < 64:
                    rs = _decode_symbol(br, tabs[(1, ta)])
                    r, s = rs >> 4, rs & 15
                    if s == 0:
                        if r != 15:
                            break
                        kk += 16
                        continue
                    kk += r
                    if kk > 63:
                        raise JpegError("coefficient index out of range")
                    coef[m, bi, kk] = _extend(br.bits(s), s)
                    kk += 1
    return hd, coef.reshape(mh, mw, 6, 64)


def upsample_fancy(c):
    """libjpeg's h2v2_fancy_upsample (jdsample.c; do_fancy_upsampling, its default): triangle filter -- the nearer chroma sample weighs 3/4,
    the farther 1/4 in each direction (9 : 3 : 3 : 1), rounding constants 8 / 7 alternating by column, the edge samples replicated.
    c [H][W] -> [2 H][2 W].  What Pillow's decoder (and most software decoders) do where `reconstruct` replicates."""
    c = c.astype(np.int64)
    H, W = c.shape
    above = np.vstack([c[:1], c[:-1]])
    below = np.vstack([c[1:], c[-1:]])
    out = np.empty((2 * H, 2 * W), np.int64)
    for v, nb in ((0, above), (1, below)):
        cs = 3 * c + nb
        last = np.hstack([cs[:, :1], cs[:, :-1]])
        nxt = np.hstack([cs[:, 1:], cs[:, -1:]])
        even = (3 * cs + last + 8) >> 4
        odd = (3 * cs + nxt + 7) >> 4
        even[:, 0] = (cs[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (cs[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def reconstruct(hd, coef, fancy=False):
    """quantised coefficients -> BGR u8 [h][w][3].  fancy: chroma through libjpeg's triangle filter (upsample_fancy) instead of replication --
    a VARIANT for tests (what the decode step in front of the reference, FFmpeg / libjpeg, would feed the chain); the kernels replicate."""
    mh, mw = coef.shape[:2]
    m = dct_matrix()
    tq = [hd["comps"][0][3]] * 4 + [hd["comps"][1][3], hd["comps"][2][3]]
    planes = []
    for bi in range(6):
        q = hd["q"][tq[bi]].astype(np.int64)
        nat = np.zeros(coef.shape[:2] + (64,), np.int64)
        nat[..., ZIGZAG] = coef[:, :, bi, :]
        s = np.clip(nat * q, -4096, 4095).reshape(mh, mw, 8, 8)                       # [v][u]
        t = (np.einsum("vy,...vu->...yu", m, s) + 512) >> 10                          # columns
        p = (np.einsum("ux,...yu->...yx", m, t) + 32768) >> 16                        # rows
        planes.append(np.clip(p + 128, 0, 255))
    y = np.zeros((mh * 16, mw * 16), np.int64)
    for bi in range(4):
        blk = planes[bi]                                                              # [mh][mw][8][8]
        oy, ox = (bi >> 1) * 8, (bi & 1) * 8
        for r in range(8):
            y[oy + r::16, :].reshape(mh, mw, 16)[:, :, ox:ox + 8] = blk[:, :, r, :]
    def up(pl):
        full = pl.transpose(0, 2, 1, 3).reshape(mh * 8, mw * 8)
        if fancy:
            ch, cw = (hd["h"] + 1) // 2, (hd["w"] + 1) // 2        # the component's own size: the filter replicates ITS edge samples
            o = np.zeros((mh * 16, mw * 16), np.int64)
            o[:2 * ch, :2 * cw] = upsample_fancy(full[:ch, :cw])
            return o
        return np.repeat(np.repeat(full, 2, axis=0), 2, axis=1)
    cb, cr = up(planes[4]) - 128, up(planes[5]) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb - 46802 * cr + 32768) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    out = np.clip(np.stack([b, g, r], -1), 0, 255).astype(np.uint8)
    return out[:hd["h"], :hd["w"]]


def decode_frame(j, fancy=False):
    hd, coef = decode_coefficients(j)
    return reconstruct(hd, coef, fancy)
