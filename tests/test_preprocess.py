"""The two uint8 stages in front of the magnifier: PreprocessProcessor
(ROI crop + INTER_AREA decimation, processing/PreprocessProcessor.cpp:10-51) and GrayscaleProcessor
(processing/GrayscaleProcessor.cpp:7-16).

CPU part: the oracle's restatement against hand-computed known answers (the reference holds no fixture
for these stages: parity unpinned at the OpenCV boundary, see oracle/lvm_oracle.c), and the device
kernels' logic bit-exact against the oracle through the HIP emulation build.  GPU part: the same
comparison through the gfx950 library."""
import numpy as np
import pytest

from helpers import c_params

CASES = [  # (w, h, ch, downscale, roi (x, y, w, h) or None, grayscale)
    (64, 48, 3, 2, None, False),                       # 2x2 blocks: (sum + 2) >> 2
    (64, 48, 3, 4, None, True),                        # 4x4 blocks: round-half-even of sum / 16, then gray
    (64, 48, 1, 8, None, False),                       # gray input: the gray stage is a passthrough
    (67, 45, 3, 2, None, False),                       # 67 / 33: fractional cells (general INTER_AREA tables)
    (67, 45, 3, 4, (0.1, 0.2, 0.7, 0.55), True),       # ROI + fractional cells + gray
    (90, 60, 3, 1, (0.25, 0.25, 0.5, 0.5), False),     # crop only
    (90, 60, 3, 1, None, True),                        # gray only
    (90, 60, 3, 8, (0.0, 0.0, 0.05, 0.05), False),     # ROI smaller than the divisor: 1 x 1 output
    (90, 60, 3, 3, (0.9, 0.9, 0.5, 0.5), False),       # ROI clipped at the frame edge, divisor 3
    (50, 40, 3, 1, None, False),                       # identity: neither stage changes the frame
]


def _frame(w, h, ch, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w) if ch == 1 else (h, w, ch), dtype=np.uint8)


def _params(lvm, po, ds, roi, gray):
    kw = dict(downscale=ds, roiEnabled=roi is not None)
    if roi is not None:
        kw.update(roiX=roi[0], roiY=roi[1], roiW=roi[2], roiH=roi[3])
    pre = lvm.PreprocessParams(**kw)
    return lvm.to_c_preprocess(pre, gray), po.make_pre_params(grayscale=gray, **kw)


# ---- oracle known answers -------------------------------------------------------------------------
def test_oracle_geometry_rounding_and_clamps(po):
    pp = po.make_pre_params(downscale=2, roiEnabled=True, roiX=0.1, roiY=0.2, roiW=0.7, roiH=0.55)
    # lround(0.1 * 67) = 7, lround(0.2 * 45) = 9, lround(0.7 * 67) = 47, lround(0.55 * 45) = 25 (24.75)
    assert po.preprocess_geometry(pp, 67, 45, 3) == (7, 9, 47, 25, 23, 12, 3)
    pp = po.make_pre_params(downscale=8, roiEnabled=True, roiX=0.95, roiY=0.0, roiW=0.5, roiH=0.01)
    # x = lround(95) = 95, w clamped to 100 - 95 = 5 -> 5 / 8 = 0 -> max(1, .) = 1; h = lround(0.5) = 1 (half away from zero)
    assert po.preprocess_geometry(pp, 100, 50, 3) == (95, 0, 5, 1, 1, 1, 3)
    pp = po.make_pre_params(downscale=99, grayscale=True)                    # divisor clamped to 8
    assert po.preprocess_geometry(pp, 64, 48, 3) == (0, 0, 64, 48, 8, 6, 1)
    assert po.preprocess_geometry(pp, 64, 48, 1)[6] == 1


def test_oracle_area_fast_path_roundings(po):
    a = np.array([[1, 2], [3, 4]], np.uint8)                                  # sum 10 -> (10 + 2) >> 2 = 3 (2.5 rounds up)
    assert po.resize_area_u8(a, (1, 1))[0, 0] == 3
    a = np.array([[0, 1], [0, 1]], np.uint8)                                  # 0.5 -> 1 here, but ...
    assert po.resize_area_u8(a, (1, 1))[0, 0] == 1
    b = np.zeros((4, 4), np.uint8); b.flat[:8] = 1                            # ... 8 / 16 = 0.5 -> 0 on the 4x4 path (half-even)
    assert po.resize_area_u8(b, (1, 1))[0, 0] == 0
    b.flat[:] = 0; b.flat[:] = 1; b.flat[:8] = 2                              # 24 / 16 = 1.5 -> 2
    assert po.resize_area_u8(b, (1, 1))[0, 0] == 2
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(16, 24, 3), dtype=np.uint8)
    want = (img.reshape(4, 4, 6, 4, 3).astype(np.float32).sum(axis=(1, 3)) * np.float32(1.0 / 16)).round().astype(np.uint8)
    assert np.array_equal(po.resize_area_u8(img, (6, 4)), want)


def test_oracle_area_general_path_is_a_partition_of_unity(po):
    for (w, h, dw, dh) in [(67, 45, 33, 22), (10, 7, 3, 3), (101, 3, 50, 1)]:
        for v in (0, 1, 77, 255):
            out = po.resize_area_u8(np.full((h, w, 3), v, np.uint8), (dw, dh))
            assert out.shape == (dh, dw, 3) and (out == v).all()
    # 3 -> 2 columns: cells [0, 1.5) and [1.5, 3): weights (1, .5) / 1.5 and (.5, 1) / 1.5
    out = po.resize_area_u8(np.array([[30, 60, 90]], np.uint8), (2, 1))
    assert out.tolist() == [[40, 80]]


def test_oracle_gray_known_answers(po):
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [10, 200, 30]]], np.uint8)
    # (b * 3735 + g * 19235 + r * 9798 + 16384) >> 15
    assert po.bgr2gray_u8(px).tolist() == [[29, 150, 76, 255, 0, (10 * 3735 + 200 * 19235 + 30 * 9798 + 16384) >> 15]]


# ---- device kernels vs oracle ---------------------------------------------------------------------
def _check_stage(lvm, po, lib, torch_dev=None):
    ctx = lvm.Context(0, 1, lib)
    none = lvm.LvmParams(3, 1, 0, 0, 0, 0, 0, 30.0, 0)                         # mode None: the chain returns the preprocessed frame
    try:
        for i, (w, h, ch, ds, roi, gray) in enumerate(CASES):
            f = _frame(w, h, ch, 100 + i)
            cpre, opre = _params(lvm, po, ds, roi, gray)
            ref = po.preprocess(f, opre)
            assert ctx.preprocess_geometry(cpre, w, h, ch) == po.preprocess_geometry(opre, w, h, ch)
            out, produced = ctx.chain_process(f, cpre, none)
            assert not produced
            assert out.shape == ref.shape and np.array_equal(out, ref), "case %d" % i
    finally:
        ctx.close()


def test_preprocess_emu_bit_exact(lvm, po, emu):
    _check_stage(lvm, po, emu)


def _check_chain(lvm, po, lib, exact):
    """Preprocess -> Grayscale -> Laplace through lvm_chain_process against oracle(preprocess) -> oracle(magnify),
    with an ROI move at equal size in the middle (the magnifier must drop its state: MagnifyCore.hpp:55-56)."""
    ck, pk = lvm.synth.config(0, (96, 64, 3))
    clip = lvm.synth.Clip(**ck)
    ctx = lvm.Context(0, 1, lib)
    ctx.exact_lab(exact)
    orc = po.Oracle()
    P = po.make_params(**pk)
    try:
        for t in range(10):
            roi = (0.1, 0.1, 0.8, 0.75) if t < 6 else (0.15, 0.2, 0.8, 0.75)
            cpre, opre = _params(lvm, po, 2, roi, False)
            f = clip.frame(t)
            small = po.preprocess(f, opre)
            if t == 6:
                orc.reset()                                    # what the reference's tracker does on a moved ROI
            ref, pr = orc.process(small, P)
            out, pg = ctx.chain_process(f, cpre, c_params(lvm, pk))
            assert pr == pg
            if exact:
                assert np.array_equal(out, ref), "frame %d" % t
            else:
                d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
                assert d.max() <= 1 and (d == 0).mean() >= 0.999
    finally:
        ctx.close(); orc.close()


def test_chain_emu_preprocess_then_laplace(lvm, po, emu):
    _check_chain(lvm, po, emu, True)


def _check_chain_batch(lvm, po, lib, exact):
    """lvm_chain_process_batch: three independent streams (different clips), one launch per stage for all of them."""
    ck, pk = lvm.synth.config(0, (96, 64, 3))
    clips = [lvm.synth.Clip(seed=77 + s, **ck) for s in range(3)]
    ctx = lvm.Context(0, 3, lib)
    ctx.exact_lab(exact)
    orcs = [po.Oracle() for _ in range(3)]
    P = po.make_params(**pk)
    cpre, opre = _params(lvm, po, 2, (0.05, 0.1, 0.9, 0.8), True)
    try:
        for t in range(6):
            frames = [c.frame(t) for c in clips]
            outs, pg = ctx.chain_process_batch(frames, cpre, c_params(lvm, pk))
            for s in range(3):
                ref, pr = orcs[s].process(po.preprocess(frames[s], opre), P)
                assert pr == pg
                if exact:
                    assert np.array_equal(outs[s], ref), "frame %d stream %d" % (t, s)
                else:
                    d = np.abs(outs[s].astype(np.int32) - ref.astype(np.int32))
                    assert d.max() <= 1 and (d == 0).mean() >= 0.999
    finally:
        ctx.close()
        for o in orcs:
            o.close()


def test_chain_batch_emu_three_streams(lvm, po, emu):
    _check_chain_batch(lvm, po, emu, True)


def _check_original_tap(lvm, po, lib):
    """runChainOnce taps `original` after chain[0] = PreprocessProcessor, BEFORE GrayscaleProcessor (ChainBuilder.cpp:25): with
    grayscale on a BGR source the tap is the cropped / decimated COLOUR frame while the magnifier sees -- and returns -- gray."""
    cpre, opre = _params(lvm, po, 2, (0.1, 0.1, 0.8, 0.8), True)
    _, opre_tap = _params(lvm, po, 2, (0.1, 0.1, 0.8, 0.8), False)
    ck, pk = lvm.synth.config(0, (96, 64, 2))
    ctx = lvm.Context(0, 2, lib)
    try:
        for t in range(3):
            frames = [_frame(96, 64, 3, 10 * s + t) for s in range(2)]
            outs, taps, produced = ctx.chain_process_batch_ex(frames, cpre, c_params(lvm, pk))
            assert produced
            for s in range(2):
                want = po.preprocess(frames[s], opre_tap)
                assert taps[s].shape == want.shape == (outs[s].shape[0], outs[s].shape[1], 3)
                assert np.array_equal(taps[s], want), (t, s)
                assert outs[s].ndim == 2 and outs[s].shape == po.preprocess(frames[s], opre).shape
    finally:
        ctx.close()


def test_chain_batch_ex_original_tap_is_the_colour_frame_emu(lvm, po, emu):
    _check_original_tap(lvm, po, emu)


@pytest.mark.gpu
def test_chain_batch_ex_original_tap_is_the_colour_frame_gpu(lvm, po, hip):
    _check_original_tap(lvm, po, hip)


def _check_pinned_equals_pageable(lvm, lib, sizes):
    """lvm_process on page-locked frames (no staging copy: kernels read / write the caller's buffers) against the same call on
    pageable frames (staged copies): identical bytes, identical `produced`, nothing written outside the output rows (padded
    strides).  Laplace / Riesz on BGR take both aliases, the colour mode and gray frames the output alias only."""
    for cfg, size, gray, pad in sizes:
        ck, pk = lvm.synth.config(cfg, size)
        clip = lvm.synth.Clip(**ck)
        a, b = lvm.Context(0, 1, lib), lvm.Context(0, 1, lib)
        try:
            for t in range(6):
                f = clip.frame(t)
                if gray:
                    f = np.ascontiguousarray(f[:, :, 1])
                o1, p1 = a.process(f, c_params(lvm, pk))
                o2, p2 = b.process_pinned(f, c_params(lvm, pk), pad=pad)
                assert p1 == p2, (cfg, t)
                assert np.array_equal(o1, o2), (cfg, t, int(np.abs(o1.astype(int) - o2.astype(int)).max()))
        finally:
            a.close(); b.close()


def test_process_pinned_frames_zero_copy_emu(lvm, emu):
    _check_pinned_equals_pageable(lvm, emu, [(0, (64, 48, 3), False, 0), (2, (64, 48, 3), False, 20), (3, (64, 48, 2), False, 4), (0, (64, 48, 3), True, 8)])


@pytest.mark.gpu
def test_process_pinned_frames_zero_copy_gpu(lvm, hip):
    _check_pinned_equals_pageable(lvm, hip, [(0, (640, 360, 4), False, 0), (2, (320, 180, 4), False, 20), (3, (320, 180, 3), False, 4),
                                             (0, (320, 180, 3), True, 8), (1, (1920, 1080, 6), False, 0)])


def test_chain_rejects_bad_arguments(lvm, po, emu):
    ctx = lvm.Context(0, 2, emu)
    try:
        with pytest.raises(lvm.LvmError):
            ctx.chain_process(_frame(32, 32, 3, 1), lvm.to_c_preprocess(lvm.PreprocessParams()), lvm.LvmParams(3, 1, 0, 0, 0, 0, 0, 30.0, 0))
    finally:
        ctx.close()


@pytest.mark.gpu
def test_preprocess_gpu_bit_exact(lvm, po, hip):
    _check_stage(lvm, po, hip)


@pytest.mark.gpu
def test_chain_gpu_preprocess_then_laplace(lvm, po, hip):
    _check_chain(lvm, po, hip, False)
    _check_chain(lvm, po, hip, True)


@pytest.mark.gpu
def test_chain_batch_gpu_three_streams(lvm, po, hip):
    _check_chain_batch(lvm, po, hip, False)
    _check_chain_batch(lvm, po, hip, True)


@pytest.mark.gpu
def test_preprocess_device_multi_stream_1080p(lvm, po, hip):
    """lvm_preprocess_device on device-resident frames of two streams at the full size of BASELINE configs[1]."""
    import ctypes
    import torch
    w, h = 1920, 1080
    frames = np.stack([_frame(w, h, 3, 7), _frame(w, h, 3, 8)])
    cpre, opre = _params(lvm, po, 4, (0.1, 0.1, 0.8, 0.8), True)
    ctx = lvm.Context(0, 2, hip)
    try:
        _, _, _, _, ow, oh, och = ctx.preprocess_geometry(cpre, w, h, 3)
        d_in = torch.from_numpy(frames).cuda()
        d_out = torch.zeros((2, oh, ow), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.preprocess_device(cpre, ctypes.c_void_p(d_in.data_ptr()), w, h, 3, w * 3, w * h * 3, ctypes.c_void_p(d_out.data_ptr()),
                              ow * och, ow * oh * och)
        ctx.synchronize()
        got = d_out.cpu().numpy()
        for s in range(2):
            assert np.array_equal(got[s], po.preprocess(frames[s], opre))
    finally:
        ctx.close()
