"""Motion-JPEG decode onto the device (the decode half): `cv::VideoCapture::read` on an AVI / Motion-JPEG file
(source/FileSource.cpp:99).

The entropy layer of JPEG is exact by the standard, the arithmetic behind it (IDCT, chroma upsampling, colour conversion) is the decoder's:
oracle/mjpeg_oracle.py::decode_frame restates ONE such decoder in integers and is pinned here against Pillow's libjpeg -- on this
repository's streams AND on libjpeg's own (no restart markers, per-image optimised Huffman tables): the two decoders agree to within the
difference between libjpeg's smoothed and the oracle's replicated chroma (PSNR and largest-difference bars below), and both are equally
close to the source.  The HIP decoder is BIT-identical to the oracle (emulation build here, the GPU through the C ABI)."""
import ctypes
import io

import numpy as np
import pytest

from oracle import mjpeg_oracle as mo
from test_mjpeg import decode as pil_decode, texture

PIL_Image = pytest.importorskip("PIL.Image")


def pil_encode(f, q, **kw):
    buf = io.BytesIO()
    PIL_Image.fromarray(f[..., ::-1]).save(buf, "JPEG", quality=q, **kw)
    return buf.getvalue()


STREAMS = [(64, 48, 75), (100, 70, 95), (33, 21, 50), (16, 16, 100), (17, 9, 30), (1, 1, 75), (130, 34, 100)]


def streams_of(w, h, q, seed=0):
    f = texture(w, h, seed=w + h + seed)
    return f, [mo.encode_frame(f, q), pil_encode(f, q, subsampling=2), pil_encode(f, q, subsampling=2, optimize=True), mo.encode_frame(f, q, restart=3)]


@pytest.mark.parametrize("w,h,q", STREAMS)
def test_oracle_decoder_agrees_with_libjpeg(w, h, q):
    f, js = streams_of(w, h, q)
    for j in js:
        mine, theirs = mo.decode_frame(j), pil_decode(j)
        assert mine.shape == theirs.shape == f.shape
        if w * h >= 256:
            assert mo.psnr(mine, theirs) > 40.0 and np.abs(mine.astype(int) - theirs).max() <= 16
            assert mo.psnr(mine, f) > mo.psnr(theirs, f) - 0.3
    assert np.array_equal(mo.decode_coefficients(js[1])[1], mo.decode_coefficients(js[2])[1])      # the same coefficients under both Huffman codes


def test_oracle_round_trip_is_the_quantiser_and_nothing_else():
    f = texture(48, 32)
    c = mo.coefficients(f, 90)
    hd, got = mo.decode_coefficients(mo.encode_frame(f, 90, restart=3))
    assert np.array_equal(got, c) and hd["restart"] == 3 and (hd["w"], hd["h"]) == (48, 32)
    assert mo.parse_header(mo.encode_frame(f, 90))["restart"] == 8


def _strip_dht(j):
    """the frame without its Huffman tables (as AVI MJPEG frames may be stored)"""
    out, i = bytearray(j[:2]), 2
    while True:
        m, n = j[i + 1], int.from_bytes(j[i + 2:i + 4], "big")
        if m != 0xC4:
            out += j[i:i + 2 + n]
        i += 2 + n
        if m == 0xDA:
            return bytes(out + j[i:])


def _decode_and_compare(lvm, lib, alloc, read):
    ctx = lvm.Context(0, 1, lib)
    try:
        for (w, h, q) in STREAMS:
            f, js = streams_of(w, h, q)
            js.append(_strip_dht(js[0]))
            f2, more = streams_of(w, h, max(1, q - 20), seed=5)                       # other tables in the same batch
            js += more[:2]
            pad = 5 if w % 2 else 0
            buf = alloc(len(js), h, w * 3 + pad)
            ctx.mjpeg_decode_device(js, w, h, buf[0], stride=w * 3 + pad, frame_stride=(w * 3 + pad) * h)
            got = read(buf)
            for k, j in enumerate(js):
                want = mo.decode_frame(j)
                assert np.array_equal(got[k, :, :w * 3].reshape(h, w, 3), want), "%dx%d q%d stream %d" % (w, h, q, k)
                assert (got[k, :, w * 3:] == 0xEE).all()
    finally:
        ctx.close()


def _numpy_alloc():
    def alloc(n, h, row):
        a = np.full((n, h, row), 0xEE, np.uint8)
        return (ctypes.c_void_p(a.ctypes.data), a)
    return alloc, (lambda b: b[1])


def test_mjpeg_decode_emu_bit_identical_to_the_oracle(lvm, emu):
    _decode_and_compare(lvm, emu, *_numpy_alloc())


def test_mjpeg_decode_emu_refuses_what_it_does_not_decode(lvm, emu):
    ctx = lvm.Context(0, 1, emu)
    try:
        f = texture(64, 48)
        out = np.zeros((1, 48, 64 * 3), np.uint8)
        p = ctypes.c_void_p(out.ctypes.data)
        good = mo.encode_frame(f, 80)
        for bad, why in [(pil_encode(f, 80, subsampling=0), "4:2:0"), (pil_encode(f, 80, subsampling=2, progressive=True), "baseline"),
                         (mo.encode_frame(texture(32, 48), 80), "size"), (good[:200], "marker|segment"), (b"\x00\x01" + good, "SOI")]:
            with pytest.raises(lvm.LvmError, match=why):
                ctx.mjpeg_decode_device([bad], 64, 48, p)
        # a DHT whose code counts do not fit the code space (255 codes of length 1: ADVICE round 5 -- the look-up table build
        # would have written ~130 KB past the table; tools/emu_asan.sh runs this file under AddressSanitizer), and scans with Ah | Al set
        i = good.index(b"\xff\xc4")
        n = (good[i + 2] << 8) | good[i + 3]
        evil = b"\xff\xc4" + (2 + 17 + 255).to_bytes(2, "big") + b"\x00" + bytes([255] + [0] * 15) + bytes(range(255))
        with pytest.raises(lvm.LvmError, match="code space"):
            ctx.mjpeg_decode_device([good[:i] + evil + good[i + 2 + n:]], 64, 48, p)
        for bits in ([3] + [0] * 15, [1, 3] + [0] * 14, [0, 0, 0, 0, 0, 0, 0, 0, 255, 2] + [0] * 6):        # over-subscribed at length 1, 2, 10
            evil = b"\xff\xc4" + (2 + 17 + sum(bits)).to_bytes(2, "big") + b"\x10" + bytes(bits) + bytes(range(sum(bits) % 256)) + bytes(max(0, sum(bits) - 256 + 1))
            evil = evil[:4 + 17 + sum(bits)]
            with pytest.raises(lvm.LvmError, match="code space|Huffman"):
                ctx.mjpeg_decode_device([good[:i] + evil + good[i + 2 + n:]], 64, 48, p)
        hd0 = mo.parse_header(good)
        sos = bytearray(good)
        sos[hd0["data_start"] - 1] = 0x01
        with pytest.raises(lvm.LvmError, match="baseline scan"):
            ctx.mjpeg_decode_device([bytes(sos)], 64, 48, p)
        # wrong restart-marker count
        cut = good.replace(b"\xff\xd0", b"\xff\x00", 1)
        with pytest.raises(lvm.LvmError, match="restart"):
            ctx.mjpeg_decode_device([cut], 64, 48, p)
        # entropy data overwritten with zeros: decodes (zeros are valid codes) or is refused, but never reads outside; then the context still works
        hd = mo.parse_header(good)
        junk = good[:hd["data_start"]] + bytes(len(good) - hd["data_start"] - 2) + good[-2:]
        try:
            ctx.mjpeg_decode_device([junk], 64, 48, p)
        except lvm.LvmError:
            pass
        ctx.mjpeg_decode_device([good], 64, 48, p)
        assert np.array_equal(out[0].reshape(48, 64, 3), mo.decode_frame(good))
    finally:
        ctx.close()


def test_mjpeg_decode_emu_survives_corrupted_streams(lvm, emu):
    """random damage to valid streams (flipped bytes in the entropy data and in the headers, truncation, inserted markers): the call decodes
    something or fails with a message -- and the context decodes the intact stream afterwards.  (tools/emu_asan.sh runs this file under
    AddressSanitizer: a read or write outside a buffer would stop it.)"""
    rng = np.random.default_rng(3)
    f = texture(96, 64)
    streams = [mo.encode_frame(f, 85), mo.encode_frame(f, 85, restart=2), pil_encode(f, 85, subsampling=2)]
    ctx = lvm.Context(0, 1, emu)
    try:
        out = np.zeros((1, 64, 96 * 3), np.uint8)
        p = ctypes.c_void_p(out.ctypes.data)
        failed = 0
        for t in range(90):
            j = bytearray(streams[t % 3])
            kind = t % 5
            if kind == 0:                                            # a few flipped bytes in the entropy-coded segment
                for _ in range(1 + t % 4):
                    j[int(rng.integers(len(j) // 2, len(j) - 2))] = int(rng.integers(0, 256))
            elif kind == 1:                                          # ... anywhere behind SOI
                j[int(rng.integers(2, len(j)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 2:                                          # truncated
                del j[int(rng.integers(20, len(j))):]
            elif kind == 3:                                          # a restart marker where none belongs
                k = int(rng.integers(len(j) // 2, len(j) - 4))
                j[k:k + 2] = b"\xff" + bytes([0xD0 + t % 8])
            else:                                                    # entropy data replaced by noise
                hd = mo.parse_header(bytes(j))
                j[hd["data_start"]:-2] = rng.integers(0, 256, len(j) - 2 - hd["data_start"], dtype=np.uint8).tobytes()
            try:
                ctx.mjpeg_decode_device([bytes(j)], 96, 64, p)
            except lvm.LvmError as e:
                failed += 1
                assert "lvm_mjpeg_decode" in str(e)
        assert failed > 10
        for j in streams:
            ctx.mjpeg_decode_device([j], 96, 64, p)
            assert np.array_equal(out[0].reshape(64, 96, 3), mo.decode_frame(j))
    finally:
        ctx.close()


def test_mjpeg_decode_emu_self_synchronising_path_on_every_stream_without_restart_markers():
    """Frames without restart markers of 2 KB and more are decoded by the self-synchronising kernels (k_mjp_*: a lane per 1024 bits, iterated
    until every lane starts where its predecessor ended); LVM_MJD_PARALLEL=2 sends the small ones there as well, =0 none: this file again
    under both settings -- the same bits either way (the switch is read once per process, hence the subprocess)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("2", "0"):
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "tests/test_mjpeg_decode.py", "-m", "not gpu", "-k", "emu and not self_synchronising"],
                           capture_output=True, text=True, env=dict(os.environ, LVM_MJD_PARALLEL=mode), cwd=root, timeout=900)
        assert r.returncode == 0 and "passed" in r.stdout, (mode, (r.stdout + r.stderr)[-3000:])


def test_mjpeg_round_trip_emu(lvm, emu):
    """encode on the device, decode on the device: the frame comes back within the quantisation error"""
    ctx = lvm.Context(0, 1, emu)
    try:
        f = np.stack([texture(80, 48, seed=k) for k in range(3)])
        ctx.mjpeg_set_restart_interval(4)
        js = ctx.mjpeg_encode_device(ctypes.c_void_p(f.ctypes.data), 80, 48, 3, quality=92)
        assert js[0] == mo.encode_frame(f[0], 92, restart=4)
        out = np.zeros_like(f)
        ctx.mjpeg_decode_device(js, 80, 48, ctypes.c_void_p(out.ctypes.data))
        for k in range(3):
            assert np.array_equal(out[k], mo.decode_frame(js[k])) and mo.psnr(out[k], f[k]) > 30.0
    finally:
        ctx.close()


def _transcode_case(lvm, lib, w, h, n, split, pre_kw, q_in, q_out):
    """lvm_export_mjpeg_frames == decode (oracle) -> lvm_export_frames -> encode (oracle)"""
    from helpers import c_params
    ck, pk = lvm.synth.config(0)
    clip = lvm.synth.Clip(seed=7, **dict(ck, w=w, h=h))
    jin = [mo.encode_frame(clip.frame(t), q_in) if t % 2 else pil_encode(clip.frame(t), q_in, subsampling=2) for t in range(n)]     # both kinds of stream
    decoded = [mo.decode_frame(j) for j in jin]
    pre = lvm.LvmPreprocessParams(1, 0, 0.0, 0.0, 1.0, 1.0, 0)
    for k, v in pre_kw.items():
        setattr(pre, k, v)
    cp = c_params(lvm, pk)
    a, b = lvm.Context(0, 1, lib), lvm.Context(0, 1, lib)
    try:
        canvases, prod_a = a.export_frames(decoded, pre, cp, split)
        jout, prod_b = b.export_mjpeg_frames(jin, w, h, pre, cp, split, quality=q_out)
        assert prod_a == prod_b
        for k in range(n):
            assert jout[k] == mo.encode_frame(canvases[k], q_out), "frame %d" % k
    finally:
        a.close()
        b.close()


def test_export_mjpeg_to_mjpeg_emu(lvm, emu):
    _transcode_case(lvm, emu, 66, 38, 6, 1, {}, 90, 80)
    _transcode_case(lvm, emu, 80, 60, 5, 2, dict(downscale=2, roi_enabled=1, roiX=0.1, roiY=0.2, roiW=0.7, roiH=0.6, grayscale=1), 85, 95)
    _transcode_case(lvm, emu, 64, 48, 3, 0, dict(roi_enabled=1, roiX=0.25, roiY=0.25, roiW=0.5, roiH=0.5), 85, 75)     # ROI only: the magnifier reads a view of the decoded frame
    _transcode_case(lvm, emu, 160, 96, 4, 1, {}, 97, 85)                # libjpeg's frames of > 2 KB: the self-synchronising kernels in front of the chain


@pytest.mark.gpu
def test_export_mjpeg_to_mjpeg_gpu(lvm, hip):
    _transcode_case(lvm, hip, 640, 360, 9, 1, {}, 90, 85)
    _transcode_case(lvm, hip, 322, 182, 5, 2, dict(downscale=2, roi_enabled=1, roiX=0.1, roiY=0.2, roiW=0.7, roiH=0.6, grayscale=1), 85, 90)


@pytest.mark.gpu
def test_mjpeg_decode_gpu_bit_identical_to_the_oracle(lvm, hip):
    import torch

    def alloc(n, h, row):
        t = torch.full((n, h, row), 0xEE, dtype=torch.uint8, device="cuda")
        return (ctypes.c_void_p(t.data_ptr()), t)
    _decode_and_compare(lvm, hip, alloc, lambda b: b[1].cpu().numpy())


@pytest.mark.gpu
def test_mjpeg_round_trip_gpu_1080p(lvm, hip):
    """a batch of 1080p frames: GPU encode -> GPU decode, and libjpeg's stream of the same frame (one interval per frame) -> GPU decode"""
    import torch
    w, h, n = 1920, 1080, 4
    f = np.stack([texture(w, h, seed=k) for k in range(n)])
    ctx = lvm.Context(0, 1, hip)
    try:
        d = torch.from_numpy(f).cuda()
        js = ctx.mjpeg_encode_device(ctypes.c_void_p(d.data_ptr()), w, h, n, quality=90)
        out = torch.zeros_like(d)
        ctx.mjpeg_decode_device(js, w, h, ctypes.c_void_p(out.data_ptr()))
        got = out.cpu().numpy()
        for k in range(n):
            assert mo.psnr(got[k], f[k]) > 30.0 and mo.psnr(got[k], pil_decode(js[k])) > 40.0
        lj = [pil_encode(f[k], 90, subsampling=2) for k in range(2)]
        ctx.mjpeg_decode_device(lj, w, h, ctypes.c_void_p(out.data_ptr()))
        got = out.cpu().numpy()
        for k in range(2):
            assert mo.psnr(got[k], pil_decode(lj[k])) > 40.0
        assert np.array_equal(got[0][:32], mo.reconstruct(*_first_rows(lj[0], 2))[:32])
    finally:
        ctx.close()


def _first_rows(j, mcu_rows):
    """header + coefficients of the first MCU rows of a stream (the whole 1080p frame through the Python oracle would take minutes)"""
    hd, coef = mo.decode_coefficients(j) if False else (mo.parse_header(j), None)
    hd2 = dict(hd)
    full = _decode_prefix(j, hd, mcu_rows)
    hd2["h"] = mcu_rows * 16
    return hd2, full


def _decode_prefix(j, hd, mcu_rows):
    tabs = {}
    for key, std in (((0, 0), mo.DC_LUMA), ((1, 0), mo.AC_LUMA), ((0, 1), mo.DC_CHROMA), ((1, 1), mo.AC_CHROMA)):
        tabs[key] = mo._decode_tables(hd["huff"].get(key, std))
    mw = (hd["w"] + 15) // 16
    coef = np.zeros((mcu_rows * mw, 6, 64), np.int32)
    br, pred = mo._BitReader(mo.split_intervals(j[hd["data_start"]:])[0]), [0, 0, 0]
    sel = [hd["scan"][0]] * 4 + [hd["scan"][1], hd["scan"][2]]
    for m in range(mcu_rows * mw):
        for bi in range(6):
            comp = 0 if bi < 4 else bi - 3
            _, td, ta = sel[bi]
            s = mo._decode_symbol(br, tabs[(0, td)])
            pred[comp] += mo._extend(br.bits(s), s)
            coef[m, bi, 0] = pred[comp]
            kk = 1
            while kk < 64:
                rs = mo._decode_symbol(br, tabs[(1, ta)])
                r, s = rs >> 4, rs & 15
                if s == 0:
                    if r != 15:
                        break
                    kk += 16
                    continue
                kk += r
                coef[m, bi, kk] = mo._extend(br.bits(s), s)
                kk += 1
    return coef.reshape(mcu_rows, mw, 6, 64)


# ---- what the decoder's chroma upsampling does to a MAGNIFIED frame (Missing 5 / Next 4b) -----------------------------------
# The decode step in front of the reference is cv::VideoCapture (FileSource.cpp:99): FFmpeg's mjpeg decoder or libjpeg, both of which INTERPOLATE
# the 4:2:0 chroma planes; k_mjd_pixels replicates them (bit-identical to mjpeg_oracle.reconstruct).  mjpeg_oracle.reconstruct(fancy=True) is the
# variant with libjpeg's triangle filter.  The magnifier then amplifies whatever difference the two inputs have; this measures how much.
def test_oracle_fancy_upsampling_is_what_libjpeg_decodes():
    """the variant against the independent decoder in the image (Pillow = libjpeg-turbo, fancy upsampling on): closer to it than the
    replicating decoder by ~6 dB, what is left is the two inverse DCTs' rounding"""
    for (w, h, q) in [(96, 64, 90), (160, 90, 75), (37, 29, 95)]:
        f = texture(w, h)
        j = pil_encode(f, q, subsampling=2)
        pil = pil_decode(j).astype(int)
        rep, fan = mo.decode_frame(j).astype(int), mo.decode_frame(j, fancy=True).astype(int)
        e_rep, e_fan = float(np.mean((rep - pil) ** 2)), float(np.mean((fan - pil) ** 2))
        assert np.abs(fan - pil).max() <= 4 and e_fan < 0.5 * e_rep, (w, h, q, np.abs(fan - pil).max(), e_fan, e_rep)


@pytest.mark.parametrize("cfg", [0, 2])
def test_chroma_upsampling_variant_through_the_magnifier(lvm, po, cfg):
    """BASELINE config 0 (Laplace) / 2 (Riesz) at 320 x 180 for 48 frames, every frame JPEG-coded (libjpeg, quality 90, 4:2:0) and decoded twice:
    replicated chroma (the kernels' decoder) and triangle-filtered chroma (libjpeg's / FFmpeg's).  Both sequences go through the CPU oracle of the
    magnifier; reported: how far the INPUTS are apart and how far the MAGNIFIED frames are (`-s` prints the line DESIGN.md quotes)."""
    size = (320, 180, 4 if cfg == 0 else 5)
    ck, pk = lvm.synth.config(cfg, size)
    clip = lvm.synth.Clip(**ck)
    P = po.make_params(**pk)
    oa, ob = po.Oracle(), po.Oracle()
    din, dout, dmax_in, dmax_out, same = [], [], 0, 0, []
    try:
        for t in range(48):
            j = pil_encode(clip.frame(t), 90, subsampling=2)
            hd, coef = mo.decode_coefficients(j)
            a, b = mo.reconstruct(hd, coef), mo.reconstruct(hd, coef, fancy=True)
            ra, pa = oa.process(a, P)
            rb, pb = ob.process(b, P)
            assert pa == pb
            d_in = np.abs(a.astype(int) - b.astype(int))
            dmax_in = max(dmax_in, int(d_in.max())); din.append(float(np.mean(d_in.astype(float) ** 2)))
            if pa:
                d = np.abs(ra.astype(int) - rb.astype(int))
                dmax_out = max(dmax_out, int(d.max())); dout.append(float(np.mean(d.astype(float) ** 2))); same.append(float((d == 0).mean()))
    finally:
        oa.close(); ob.close()
    psnr = lambda mse: 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))      # noqa: E731
    p_in, p_out = psnr(np.mean(din)), psnr(np.mean(dout))
    print("cfg%d decoded inputs (replicated vs triangle chroma): max %d levels, PSNR %.1f dB | magnified outputs: max %d levels, PSNR %.1f dB, "
          "%.1f %% identical bytes" % (cfg, dmax_in, p_in, dmax_out, p_out, 100 * np.mean(same)))
    assert dmax_in <= 12 and p_in >= 44.0
    if cfg == 0:
        # Laplace (alpha 20 on band-passed Lab pyramids): the decode difference passes through essentially unchanged (measured 55.1 -> 54.9 dB, max 6 -> 7 levels)
        assert p_out >= p_in - 1.5 and dmax_out <= 2 * dmax_in + 2, (p_in, p_out, dmax_in, dmax_out)
    else:
        # Riesz (alpha 50 on the PHASE of the luminance bands): the same 55 dB input difference comes out at ~40 dB, single pixels up to ~90 levels apart
        # (the ill-conditioned phase step of RieszPyramid.cpp:93-97 again): a file -> file export whose frames are decoded on the GPU matches the
        # reference's frame only as far as the two decoders match -- stated in DESIGN.md section 8.4; bounded here so that a regression shows
        assert p_out >= 36.0 and dmax_out <= 160, (p_in, p_out, dmax_out)
