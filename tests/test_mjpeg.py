"""Motion-JPEG encode on the device (the encode half): `cv::VideoWriter::write(canvas)` for ExportFormat::AviMjpg
(export/Exporter.cpp:107-117, :259).

The boundary fixes the FORMAT (ITU-T T.81 baseline, YCbCr 4:2:0, Annex K tables), not the bits: every JPEG encoder rounds its own way.
So (1) oracle/mjpeg_oracle.py restates T.81 with one fully specified integer arithmetic and is PINNED here against the independent
codec in this image, Pillow's libjpeg: libjpeg decodes the oracle's stream, the decoded frame is as close to the input as libjpeg's own
encoder gets at the same quality (PSNR within 0.25 dB, size within 12 %), and the quantiser / Huffman segments are byte-identical to the
ones libjpeg writes; (2) the HIP encoder is BYTE-IDENTICAL to the oracle -- on the CPU through the emulation build, on the GPU through
the C ABI."""
import ctypes
import io
import math

import numpy as np
import pytest

from helpers import c_params
from oracle import mjpeg_oracle as mo

PIL_Image = pytest.importorskip("PIL.Image")


def texture(w, h, seed=7, noise=12.0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = 96 + 48 * np.sin(2 * np.pi * (x / 37 + y / 53)) + 32 * np.sin(2 * np.pi * (x / 11 - y / 7))
    f = np.stack([base + rng.uniform(-noise, noise, (h, w)) for _ in range(3)], -1)
    f[..., 1] += 20 * np.sin(x / 19.0)
    f[..., 2] -= 30 * np.cos(y / 23.0)
    return np.clip(f, 0, 255).astype(np.uint8)


def decode(jpeg):
    im = PIL_Image.open(io.BytesIO(jpeg))
    im.load()
    assert im.mode == "RGB"
    return np.array(im)[..., ::-1]


def segments(jpeg):
    """marker -> list of payloads, up to SOS"""
    out, i = {}, 2
    assert jpeg[:2] == b"\xff\xd8"
    while i < len(jpeg):
        assert jpeg[i] == 0xFF
        m, n = jpeg[i + 1], int.from_bytes(jpeg[i + 2:i + 4], "big")
        out.setdefault(m, []).append(bytes(jpeg[i + 4:i + 2 + n]))
        i += 2 + n
        if m == 0xDA:
            break
    return out


# ---- the oracle against libjpeg (CPU) ----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("w,h,q", [(64, 48, 75), (100, 70, 95), (333, 211, 50), (16, 16, 100), (17, 9, 30), (1, 1, 75), (640, 360, 85)])
def test_oracle_stream_is_decoded_by_libjpeg_as_well_as_libjpegs_own(w, h, q):
    f = texture(w, h, seed=w + h)
    j = mo.encode_frame(f, q)
    dec = decode(j)
    assert dec.shape == f.shape
    buf = io.BytesIO()
    PIL_Image.fromarray(f[..., ::-1]).save(buf, "JPEG", quality=q, subsampling=2)       # libjpeg, 4:2:0, the same quality scale
    theirs = buf.getvalue()
    p_ours, p_theirs = mo.psnr(f, dec), mo.psnr(f, decode(theirs))
    assert p_ours >= p_theirs - 0.25, (p_ours, p_theirs)
    if w * h >= 64 * 48:
        assert len(j) <= 1.12 * len(theirs) + 64, (len(j), len(theirs))                  # restart markers + DC resets cost a few percent


def test_oracle_tables_are_the_ones_libjpeg_writes():
    f = texture(64, 48)
    for q in (30, 75, 95):
        buf = io.BytesIO()
        PIL_Image.fromarray(f[..., ::-1]).save(buf, "JPEG", quality=q, subsampling=2, optimize=False)
        theirs, ours = segments(buf.getvalue()), segments(mo.encode_frame(f, q))
        dqt = lambda s: sorted(b"".join(s[0xDB])[k:k + 65] for k in range(0, len(b"".join(s[0xDB])), 65))
        assert dqt(theirs) == dqt(ours), "quantiser tables at quality %d" % q

        def dht(s):
            blob, out, i = b"".join(s[0xC4]), {}, 0
            while i < len(blob):
                n = sum(blob[i + 1:i + 17])
                out[blob[i]] = blob[i + 1:i + 17 + n]
                i += 17 + n
            return out
        assert dht(theirs) == dht(ours), "Huffman tables"
        assert theirs[0xC0][0][5:] == ours[0xC0][0][5:]                                   # three components, 2x2 / 1x1 / 1x1 sampling, table ids


def test_oracle_dct_matrix_and_reciprocal_quantiser():
    m = mo.dct_matrix()
    for u in range(8):
        for x in range(8):
            cu = 1 / math.sqrt(2) if u == 0 else 1.0
            assert abs(m[u, x] - 8192 * cu / 2 * math.cos((2 * x + 1) * u * math.pi / 16)) <= 0.5
    # the device divides by multiplying: n // Q == (n * ceil(2^24 / Q)) >> 24 for every n the quantiser can see
    n = np.arange(0, 8192, dtype=np.int64)
    for qv in range(1, 256):
        r = -(-(1 << 24) // qv)
        assert np.array_equal(n // qv, (n * r) >> 24)
    # a flat block: only the DC term, 8 x (value - 128)
    c = mo.coefficients(np.full((16, 16, 3), 200, np.uint8), 100)
    assert c[0, 0, 0, 0] == 8 * (200 - 128) and not c[0, 0, 0, 1:].any() and not c[0, 0, 4:, :].any()


def test_oracle_restart_intervals_and_stuffing():
    f = texture(48, 40, noise=60.0)
    j = mo.encode_frame(f, 100, restart=3)
    seg = segments(j)
    assert int.from_bytes(seg[0xDD][0], "big") == 3                                       # one MCU row of a 48-pixel frame
    body = j[j.index(b"\xff\xda"):]
    rst = [body[i + 1] for i in range(len(body) - 1) if body[i] == 0xFF and 0xD0 <= body[i + 1] <= 0xD7]
    assert rst == [0xD0, 0xD1]                                                            # three MCU rows: two markers, in order
    assert j[-2:] == b"\xff\xd9"
    i = body.index(b"\xff\xda") + 2 + int.from_bytes(body[2:4], "big")
    while i < len(body) - 2:                                                              # every FF of the entropy data is stuffed or a marker
        if body[i] == 0xFF:
            assert body[i + 1] == 0 or 0xD0 <= body[i + 1] <= 0xD7, hex(body[i + 1])
            i += 1
        i += 1


# ---- the HIP encoder against the oracle --------------------------------------------------------------------------------------------------

CASES = [(64, 48, 75, 1), (100, 70, 95, 2), (333, 211, 50, 1), (16, 16, 100, 3), (17, 9, 30, 1), (1, 1, 75, 1), (130, 34, 100, 2), (72, 18, 1, 1)]


def _encode_and_compare(lvm, lib, to_dev, cases, noise=12.0, restart=0):
    ctx = lvm.Context(0, 1, lib)
    try:
        ctx.mjpeg_set_restart_interval(restart)
        for (w, h, q, n) in cases:
            frames = np.stack([texture(w, h, seed=11 * w + k, noise=noise) for k in range(n)])
            pad = 7 if w % 2 else 0                                                       # ragged rows on the odd widths
            staged = np.full((n, h, w * 3 + pad), 0xEE, np.uint8)
            staged[:, :, :w * 3] = frames.reshape(n, h, w * 3)
            d = to_dev(staged)
            got = ctx.mjpeg_encode_device(d[0], w, h, n, quality=q, stride=w * 3 + pad, frame_stride=(w * 3 + pad) * h)
            for k in range(n):
                want = mo.encode_frame(frames[k], q, restart=restart)
                assert got[k] == want, "frame %d of %dx%d q%d: %d vs %d bytes, first difference at %d" % (
                    k, w, h, q, len(got[k]), len(want), next((i for i, (a, b) in enumerate(zip(got[k], want)) if a != b), -1))
                assert decode(got[k]).shape == (h, w, 3)
    finally:
        ctx.close()


def _numpy_dev():
    keep = []

    def to_dev(a):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return (ctypes.c_void_p(a.ctypes.data), a)
    return to_dev


def test_mjpeg_emu_byte_identical_to_the_oracle(lvm, emu):
    _encode_and_compare(lvm, emu, _numpy_dev(), CASES)


@pytest.mark.parametrize("restart", [1, 3, 8, 1000])
def test_mjpeg_emu_restart_intervals_of_any_length(lvm, emu, restart):
    """lvm_mjpeg_set_restart_interval: intervals of 1, 3, 8 MCUs (crossing the MCU rows in raster order; the last one shorter) and one longer than
    the frame (a single interval, no marker at all)"""
    _encode_and_compare(lvm, emu, _numpy_dev(), [(64, 48, 75, 1), (100, 70, 95, 2), (17, 9, 30, 1), (130, 34, 100, 2)], restart=restart)


def test_mjpeg_emu_noise_at_quality_100_long_codes_and_stuffing(lvm, emu):
    """white noise at quality 100: 16-bit codes with 8-10 amplitude bits, words straddled by almost every code, FF bytes to stuff"""
    _encode_and_compare(lvm, emu, _numpy_dev(), [(96, 32, 100, 1), (40, 40, 97, 2)], noise=200.0)


def test_mjpeg_emu_zero_runs(lvm, emu):
    """a frame that is flat except for a few pixels: blocks that are only DC + EOB, and isolated high-frequency coefficients behind
    zero runs of more than 15 (ZRL symbols)"""
    ctx = lvm.Context(0, 1, emu)
    try:
        f = np.full((32, 48, 3), 120, np.uint8)
        f[5, 7] = (255, 0, 0)
        f[20, 40] = (0, 255, 0)
        f[31, 47] = (0, 0, 255)
        got = ctx.mjpeg_encode_device(ctypes.c_void_p(f.ctypes.data), 48, 32, 1, quality=90)
        assert got[0] == mo.encode_frame(f, 90)
    finally:
        ctx.close()


def test_mjpeg_emu_capacity_and_arguments(lvm, emu):
    ctx = lvm.Context(0, 1, emu)
    try:
        f = texture(64, 48)
        p = ctypes.c_void_p(f.ctypes.data)
        full = ctx.mjpeg_encode_device(p, 64, 48, 1)
        with pytest.raises(lvm.LvmError, match="too small"):
            ctx.mjpeg_encode_device(p, 64, 48, 1, capacity=len(full[0]) - 1)
        assert ctx.mjpeg_encode_device(p, 64, 48, 1, capacity=len(full[0])) == full          # exactly enough
        with pytest.raises(lvm.LvmError):
            ctx.mjpeg_encode_device(p, 64, 48, 1, quality=0)
        with pytest.raises(lvm.LvmError):
            ctx.mjpeg_encode_device(p, 64, 48, 1, stride=64 * 3 - 1)
        assert ctx.mjpeg_encode_device(p, 64, 48, 1) == full                                  # the context still works
        assert int(ctx.lib.lvm_mjpeg_bound(64, 48)) >= len(full[0]) and int(ctx.lib.lvm_mjpeg_bound(0, 48)) == 0
    finally:
        ctx.close()


def _export_case(lvm, po, lib, w, h, n, split, gray, quality):
    """lvm_export_frames_mjpeg == the oracle's encoder on the canvases lvm_export_frames returns"""
    ck, pk = lvm.synth.config(0)
    ck = dict(ck, w=w, h=h)
    clip = lvm.synth.Clip(seed=99, **ck)
    frames = [clip.frame(t) for t in range(n)]
    pre = lvm.LvmPreprocessParams()
    pre.downscale = 1
    pre.grayscale = 1 if gray else 0
    cp = c_params(lvm, pk)
    a, b = lvm.Context(0, 1, lib), lvm.Context(0, 1, lib)
    try:
        canvases, prod_a = a.export_frames(frames, pre, cp, split)
        jpegs, prod_b = b.export_frames_mjpeg(frames, pre, cp, split, quality=quality)
        assert prod_a == prod_b
        for k in range(n):
            assert jpegs[k] == mo.encode_frame(canvases[k], quality), "frame %d" % k
            assert mo.psnr(decode(jpegs[k]), canvases[k]) > 28.0
    finally:
        a.close()
        b.close()


def test_export_mjpeg_emu(lvm, po, emu):
    _export_case(lvm, po, emu, 66, 38, 5, 1, False, 80)
    _export_case(lvm, po, emu, 40, 30, 3, 2, True, 95)


@pytest.mark.gpu
def test_mjpeg_gpu_byte_identical_to_the_oracle(lvm, hip):
    import torch

    def to_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return (ctypes.c_void_p(t.data_ptr()), t)
    _encode_and_compare(lvm, hip, to_dev, CASES + [(640, 360, 85, 3)])
    _encode_and_compare(lvm, hip, to_dev, [(96, 32, 100, 1), (200, 120, 98, 2)], noise=200.0)
    for restart in (1, 8, 1000):
        _encode_and_compare(lvm, hip, to_dev, [(100, 70, 95, 2), (333, 211, 50, 1), (640, 360, 85, 2)], restart=restart)


@pytest.mark.gpu
def test_mjpeg_gpu_1080p_canvas_decodes_and_matches_the_oracle_rows(lvm, hip):
    """a 3840 x 1080 side-by-side canvas (the export's shape): libjpeg decodes it; the first and last restart intervals and the header are
    the oracle's (the whole frame through the Python oracle would take minutes)"""
    import torch
    w, h, q = 3840, 1080, 85
    f = np.concatenate([texture(1920, h, seed=1), texture(1920, h, seed=2)], axis=1)
    ctx = lvm.Context(0, 1, hip)
    try:
        t = torch.from_numpy(f).cuda()
        ctx.mjpeg_set_restart_interval(240)                                           # one interval per MCU row of this canvas
        j = ctx.mjpeg_encode_device(ctypes.c_void_p(t.data_ptr()), w, h, 1, quality=q)[0]
    finally:
        ctx.close()
    dec = decode(j)
    assert dec.shape == f.shape and mo.psnr(f, dec) > 30.0
    hdr = mo.header(w, h, q, 240)
    assert j[:len(hdr)] == hdr
    tabs = tuple(mo.huff_codes(s) for s in (mo.DC_LUMA, mo.AC_LUMA, mo.DC_CHROMA, mo.AC_CHROMA))
    first = mo.entropy_interval(mo.coefficients(f[:16], q)[0], tabs)
    assert j[len(hdr):len(hdr) + len(first)] == first and j[len(hdr) + len(first):len(hdr) + len(first) + 2] == b"\xff\xd0"
    last = mo.entropy_interval(mo.coefficients(f[1072:], q)[0], tabs)                      # rows 1072..1079 + 8 replicated rows
    assert j[-2 - len(last):-2] == last and j[-2:] == b"\xff\xd9"


@pytest.mark.gpu
def test_export_mjpeg_gpu(lvm, po, hip):
    _export_case(lvm, po, hip, 640, 360, 9, 1, False, 85)
    _export_case(lvm, po, hip, 322, 182, 4, 2, True, 75)
