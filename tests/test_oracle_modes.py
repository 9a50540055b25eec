"""Behavioural tests of the oracle at the MagnificationProcessor::process level
(reference: processing/MagnificationProcessor.cpp:17-67, MagnifyCore.hpp)."""
import numpy as np
import pytest


def mk(lvm, po, idx, small, **over):
    ck, pk = lvm.synth.config(idx, small)
    pk.update(over)
    return lvm.synth.Clip(**ck), pk, po.make_params(**pk)


def test_laplace_first_frame_is_lab_round_trip_and_produced(lvm, po):
    clip, pk, P = mk(lvm, po, 0, (96, 64, 3))
    o = po.Oracle()
    f = clip.frame(0)
    out, produced = o.process(f, P)
    assert produced                                   # MagnifyCore.hpp:100-103,159
    assert np.abs(out.astype(int) - f.astype(int)).max() <= 1
    assert (out == f).mean() > 0.99


def test_laplace_alpha_zero_and_static_clip(lvm, po):
    # alpha = 0 => gains min(0, currAlpha) <= 0 only where currAlpha < 0; use a huge wavelength so that
    # currAlpha > 0 on all levels => gain 0 => output == Lab round trip of the input
    clip, pk, P = mk(lvm, po, 0, (96, 64, 3), amplification=0.0, coWavelength=1.0)
    o = po.Oracle()
    for t in range(5):
        f = clip.frame(t)
        out, _ = o.process(f, P)
        assert np.abs(out.astype(int) - f.astype(int)).max() <= 1
    # static clip: band-pass of a constant is 0 after seeding => output == round trip
    clip2, pk2, P2 = mk(lvm, po, 0, (96, 64, 3))
    clip2.amp_px = 0.0
    o2 = po.Oracle()
    for t in range(4):
        f = clip2.frame(t)
        out, _ = o2.process(f, P2)
        assert np.abs(out.astype(int) - f.astype(int)).max() <= 1


def test_laplace_reset_and_structural_change(lvm, po):
    clip, pk, P = mk(lvm, po, 0, (96, 64, 3))
    o = po.Oracle()
    outs = [o.process(clip.frame(t), P)[0].copy() for t in range(4)]
    o.reset()                                         # MagnificationProcessor.cpp:10-15
    again = [o.process(clip.frame(t), P)[0].copy() for t in range(4)]
    for a, b in zip(outs, again):
        assert np.array_equal(a, b)
    # levels change => state dropped => next frame behaves as a first frame (round trip)
    pk2 = dict(pk); pk2["levels"] = 2
    out, _ = o.process(clip.frame(4), po.make_params(**pk2))
    assert np.abs(out.astype(int) - clip.frame(4).astype(int)).max() <= 1
    # preprocess key change at equal size => reset too (MagnifyCore.hpp:55-56)
    o.process(clip.frame(5), po.make_params(**pk2))
    pk3 = dict(pk2); pk3["preprocess_key"] = 77
    out, _ = o.process(clip.frame(6), po.make_params(**pk3))
    assert np.abs(out.astype(int) - clip.frame(6).astype(int)).max() <= 1


def test_mode_none_too_small_and_level_clamp(lvm, po):
    o = po.Oracle()
    f = np.full((40, 40, 3), 90, np.uint8)
    out, produced = o.process(f, po.make_params(mode=3, levels=4))
    assert not produced and out is f or np.array_equal(out, f)
    tiny = np.full((5, 40, 3), 90, np.uint8)            # maxLevels == 0 => identity (:32-33)
    _, produced = o.process(tiny, po.make_params(mode=0, levels=4, amplification=10, coWavelength=100, coLow=0.1, coHigh=0.4))
    assert not produced
    # levels far above maxLevels are clamped, not rejected (:34)
    clip, pk, _ = mk(lvm, po, 0, (96, 64, 3), levels=99)
    _, produced = o.process(clip.frame(0), po.make_params(**pk))
    assert produced


def test_riesz_passthrough_rules(lvm, po):
    clip, pk, P = mk(lvm, po, 2, (96, 64, 3))
    o = po.Oracle()
    _, p0 = o.process(clip.frame(0), P)
    _, p1 = o.process(clip.frame(1), P)
    assert (p0, p1) == (False, True)                  # MagnifyCore.hpp:226-240
    gray = lvm.synth.Clip(96, 64, channels=1)
    o2 = po.Oracle()
    for t in range(3):
        assert o2.process(gray.frame(t), P)[1] is False   # :212
    # degenerate Butterworth (fps == 0 -> all-zero coefficients, not NaN): still produces
    o3 = po.Oracle()
    pk0 = dict(pk); pk0["framerate"] = 0.0
    flags = [o3.process(clip.frame(t), po.make_params(**pk0))[1] for t in range(3)]
    assert flags == [False, True, True]


def test_riesz_static_scene_transient_decays(lvm, po):
    """RieszPyramid::init zeroes the prior pyramid's Riesz pair (RieszPyramid.cpp:192-213), so the
    first processed frame sees a spurious phase step even on a static scene; from the second
    processed frame on q1 = q2 = 0 exactly (0/0 -> NaN -> patched to 0) and the band-passed
    transient decays."""
    clip, pk, P = mk(lvm, po, 2, (96, 64, 3))
    clip.amp_px = 0.0
    o = po.Oracle()
    outs = []
    for t in range(90):
        out, produced = o.process(clip.frame(t), P)
        assert produced == (t > 0)
        outs.append(out.astype(np.int32))
        if produced:
            assert np.isfinite(o.last_float()).all()
    early = np.abs(outs[2] - outs[1]).max()
    late = np.abs(outs[89] - outs[88]).max()
    assert late <= 2 and late < early


def test_color_warmup_window_and_column_one(lvm, po):
    clip, pk, P = mk(lvm, po, 3, (96, 64, 3))
    o = po.Oracle()
    flags = [o.process(clip.frame(t), P)[1] for t in range(4)]
    assert flags == [False, True, True, True]         # MagnifyCore.hpp:180
    mn, mx = o.last_minmax()
    assert mx > mn
    fl = o.last_float()
    assert abs(float(fl.min()) - mn) < 1e-6 and abs(float(fl.max()) - mx) < 1e-6


def test_color_window_cap(lvm, po):
    # fps 7 -> 2*7 = 14 < 16 -> window caps at 16 columns
    clip, pk, P = mk(lvm, po, 3, (48, 32, 2), framerate=7.0, coLow=0.5, coHigh=2.0)
    o = po.Oracle()
    for t in range(24):
        o.process(clip.frame(t), P)
    assert po.lib().lvmo_optimal_buffer_size(7) == 16


def test_opencv_lut_forward_lab_gap_is_quantified(lvm, po):
    """OpenCV 4's default float BGR2Lab is a trilinear-interpolated 33^3 int16 LUT (RGB2Labfloat::useInterpolation): the
    oracle's and the library's default since round 3 (rounds 1-2 implemented the analytic path).  This test runs the oracle
    with both flavours on the reference's own configs (reduced size) and records how far the magnified frames move -- why an
    analytic implementation cannot meet the 1e-4 bar against a real OpenCV 4 build.  Bounds are loose
    on purpose: the gap is far above 1e-4 (Laplace ~5e-3; Riesz, whose phase step is ill-conditioned, ~0.3)."""
    res = {}
    for name, idx, nfr in (("laplace", 0, 12), ("riesz", 2, 8)):
        ck, pk = lvm.synth.config(idx, (320, 180, 4))
        clip = lvm.synth.Clip(**ck)
        P = po.make_params(**pk)
        outs = {}
        for lut in (0, 1):
            po.lib().lvmo_set_lab_lut(lut)
            try:
                o = po.Oracle()
                fl, u8 = [], []
                for t in range(nfr):
                    out, pr = o.process(clip.frame(t), P)
                    if pr:
                        fl.append(o.last_float().copy()); u8.append(out.copy())
                o.close()
            finally:
                po.lib().lvmo_set_lab_lut(1)
            outs[lut] = (np.stack(fl), np.stack(u8))
        rel = float(np.abs(outs[0][0] - outs[1][0]).max() / np.abs(outs[0][0]).max())
        du = np.abs(outs[0][1].astype(int) - outs[1][1].astype(int))
        res[name] = (rel, int(du.max()), float((du == 0).mean()))
        print("%s: analytic vs LUT forward Lab -> float frame max rel diff %.3e, u8 max diff %d, identical %.4f" % ((name,) + res[name]))
        if name == "laplace":
            assert 1e-4 < rel < 3e-2 and du.max() <= 2        # measured: 5.1e-3, u8 within 1 LSB, 80 % of the bytes identical
        else:
            # the Riesz path amplifies the LUT's 14-bit quantisation of L (phase differences x alpha = 50): measured 0.34 of
            # the frame's range, u8 differences up to 72 -- a real OpenCV 4 build and ANY analytic implementation differ visibly
            assert 1e-3 < rel < 1.0
