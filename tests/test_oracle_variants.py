"""The unpinned OpenCV build choices, BOUNDED instead of listed ("Next round" item 2).

The oracle restates OpenCV primitives from their published source; a few details depend on the OpenCV BUILD (CPU dispatch
of its SIMD loops, scalar type of `Mat * double`, the last bit of softfloat's pow).  Each is a switch of the oracle
(lvm_oracle.h LVMO_VAR_*).  These tests run the BASELINE configurations (reduced size, >= 64 frames so that the temporal
state drifts) under every switch and record how far the frames move from the default restatement: the ENVELOPE.  What they
assert is the claim DESIGN.md section 5 makes from it:
  * Laplace and Color stay inside the 1e-4 / 1-LSB parity bar under EVERY variant -- whichever build a maintainer links,
    the parity claim holds for those two modes;
  * Riesz (the ill-conditioned acos(q0 / |q|) step) stays inside it for the LUT / gamma variants and is reported, not
    asserted at 1e-4, for the tap-association variants (those move it by ~1e-4: the claim is stated for the AVX2-dispatch
    association only).
`-s` prints the table that DESIGN.md quotes; tests/test_gpu_parity.py::test_variant_envelope_gpu adds the HIP column.
"""
import importlib

import numpy as np
import pytest

lvm = importlib.import_module("live-video-magnification_amd")
from oracle import pyoracle as po   # noqa: E402

NFRAMES = 64
NFRAMES_CFG = {0: 64, 2: 64, 3: 150}     # colour: past the 128-frame window, so that the power-of-two transform length is reached too
SIZE = {0: (320, 180, 4), 2: (320, 180, 5), 3: (320, 180, 4)}        # cfg -> (w, h, levels)
# variants that can touch a mode at all (the others are skipped: identical code path)
APPLIES = {
    0: ["pyr_simd", "addw_fused", "gamma_f32", "lut_nudge_up", "lut_nudge_down", "spline_cv3"],
    2: ["filter_unfused", "filter_dft", "mul_f32", "gamma_f32", "lut_nudge_up", "lut_nudge_down", "spline_cv3"],
    3: ["pyr_simd", "dft_f32"],
}


def run_clip(cfg, mask, nframes=None):
    nframes = nframes or NFRAMES_CFG[cfg]
    ck, pk = lvm.synth.config(cfg, SIZE[cfg])
    clip = lvm.synth.Clip(**ck)
    P = po.make_params(**pk)
    po.set_variant(mask)
    po.lib().lvmo_set_lab_lut(1)
    orc = po.Oracle()
    fl, u8 = [], []
    try:
        for t in range(nframes):
            out, pr = orc.process(clip.frame(t), P)
            if pr:
                fl.append(orc.last_float()); u8.append(out.copy())
            else:
                fl.append(None); u8.append(None)
    finally:
        orc.close()
        po.set_variant(0)
    return fl, u8


def distance(a, b):
    """worst float metric (max|d| / max|ref|), worst u8 difference, smallest identical fraction over the produced frames"""
    rel, du, same = 0.0, 0, 1.0
    for (fa, ua), (fb, ub) in zip(zip(*a), zip(*b)):
        assert (fa is None) == (fb is None)
        if fa is None:
            continue
        rel = max(rel, float(np.abs(fa - fb).max() / np.abs(fa).max()))
        d = np.abs(ua.astype(np.int32) - ub.astype(np.int32))
        du = max(du, int(d.max())); same = min(same, float((d == 0).mean()))
    return rel, du, same


@pytest.fixture(scope="module")
def base():
    return {cfg: run_clip(cfg, 0) for cfg in SIZE}


BUILD_CHOICES = ("pyr_simd", "filter_unfused", "filter_dft", "addw_fused", "mul_f32", "dft_f32")      # depend on the OpenCV build a maintainer links
TABLE_RESIDUALS = ("gamma_f32", "lut_nudge_up", "lut_nudge_down")            # forward-table restatement: removed by lvm_set_lab_lut
RESTATEMENT_RESIDUALS = ("spline_cv3",)    # round 5: splineBuild in its OpenCV 3.x form (what rounds 1-4 restated) against OpenCV 4's (the default now)


@pytest.mark.parametrize("cfg", [0, 3])
def test_laplace_color_inside_the_bar_under_every_build_choice(cfg, base):
    rows = []
    mask = 0
    for name in APPLIES[cfg]:
        r = distance(base[cfg], run_clip(cfg, po.VARIANTS[name]))
        rows.append((name,) + r)
        if name in BUILD_CHOICES:
            mask |= po.VARIANTS[name]
            assert r[0] <= 1e-4 and r[1] <= 1 and r[2] >= 0.999, (cfg, name, r)
        else:
            # a table entry that differs by one unit is 6e-5 of L's range BEFORE the x20 amplification: single pixels leave
            # the float bar (measured 1.7e-4), the u8 frame stays within 1 LSB.  Hence the table must match entry by entry
            # (lvm_get_lab_lut / lvm_set_lab_lut, ref_recover_lab_lut); the test only bounds the damage of a mismatch.
            assert r[0] <= 1e-3 and r[1] <= 1 and r[2] >= 0.999, (cfg, name, r)
    # all build choices of the mode at once (a build differs in several places together)
    r = distance(base[cfg], run_clip(cfg, mask))
    rows.append(("all build choices",) + r)
    assert r[0] <= 1e-4 and r[1] <= 1 and r[2] >= 0.999, (cfg, "all", r)
    for r in rows:
        print("cfg%d %-18s float %.2e  u8 max %d  identical %.5f" % ((cfg,) + r))


def test_riesz_envelope(base):
    cfg = 2
    rows = {}
    for name in APPLIES[cfg]:
        rows[name] = distance(base[cfg], run_clip(cfg, po.VARIANTS[name]))
        print("cfg2 %-18s float %.2e  u8 max %d  identical %.5f" % ((name,) + rows[name]))
    # Reported, with a sanity bound only (a variant is a rounding-level change, never a different image): the
    # ill-conditioned acos(q0 / |q|) step turns last-bit differences of its inputs into ~1e-4 of the frame, which is why
    # DESIGN.md states the 1e-4 Riesz claim for ONE association (fma taps = the AVX2 / NEON dispatch, float64 scalar product)
    for name in APPLIES[cfg]:
        rel, du, same = rows[name]
        assert rel <= 5e-3 and du <= 3 and same >= 0.99, (name, rows[name])


def test_variant_switches_change_the_primitives():
    """every switch reaches the code it names (a dead switch would make the envelope vacuous)"""
    rng = np.random.default_rng(5)
    a = rng.uniform(0, 100, (37, 53)).astype(np.float32)
    k = np.zeros((9, 9), np.float32); k[:] = rng.uniform(-0.1, 0.1, (9, 9))
    d0, u0, f0 = po.pyr_down(a), po.pyr_up(a), po.filter2d(a, k)
    t0 = po.lab_lut_table()
    try:
        po.set_variant(po.VARIANTS["pyr_simd"])
        assert not np.array_equal(po.pyr_down(a), d0) and not np.array_equal(po.pyr_up(a), u0)
        assert np.abs(po.pyr_down(a) - d0).max() <= 2e-5 and np.abs(po.pyr_up(a) - u0).max() <= 2e-5
        po.set_variant(po.VARIANTS["filter_unfused"])
        f1 = po.filter2d(a, k)
        assert not np.array_equal(f1, f0) and np.abs(f1 - f0).max() <= 1e-4
        for name in ("gamma_f32", "lut_nudge_up", "lut_nudge_down"):
            po.set_variant(po.VARIANTS[name])
            t1 = po.lab_lut_table()
            nd = int((t1 != t0).sum())
            assert 0 < nd < 20000 and np.abs(t1.astype(int) - t0.astype(int)).max() == 1, (name, nd)
            print("table under %-15s: %d of %d entries differ, all by one unit" % (name, nd, t0.size))
    finally:
        po.set_variant(0)
    assert np.array_equal(po.lab_lut_table(), t0)
    # round 5: the spline form reaches the inverse conversion (every Lab frame), the binary32 transforms the colour band-pass
    ck, pk = lvm.synth.config(0, (64, 48, 2))
    f = lvm.synth.Clip(**ck).frame(0)
    P = po.make_params(**pk)

    def first_float(mask, cfg_pk=P, frame=f, n=1):
        po.set_variant(mask)
        o = po.Oracle()
        try:
            for _ in range(n):
                o.process(frame, cfg_pk)
            return o.last_float().copy()
        finally:
            o.close(); po.set_variant(0)
    # the two forms of splineBuild differ in the last bit of ~350 of the 4096 coefficients -- almost all of them the cubic terms
    # (|d| ~ 1e-5: invisible in a binary32 result) and the top three knots (x -> 1): a frame moves only where a channel saturates
    g0 = po.gamma_tab(True)
    po.set_variant(po.VARIANTS["spline_cv3"])
    try:
        g1 = po.gamma_tab(True)
    finally:
        po.set_variant(0)
    nd = int((g0 != g1).sum())
    assert 100 < nd < 1000 and np.abs(g0 - g1).max() <= 1e-6, (nd, float(np.abs(g0 - g1).max()))
    print("inverse-gamma spline under spline_cv3: %d of 4096 coefficients differ, max %.1e" % (nd, float(np.abs(g0 - g1).max())))
    a0, a1 = first_float(0), first_float(po.VARIANTS["spline_cv3"])
    assert np.abs(a0 - a1).max() <= 1e-5, float(np.abs(a0 - a1).max())
    rows = rng.uniform(0, 255, (5, 128)).astype(np.float32).reshape(5, 128, 1)
    y0 = po.ideal_filter(rows, 0.83, 1.0, 60.0)
    po.set_variant(po.VARIANTS["dft_f32"])
    try:
        y1 = po.ideal_filter(rows, 0.83, 1.0, 60.0)
    finally:
        po.set_variant(0)
    assert not np.array_equal(y0, y1) and np.abs(y0 - y1).max() <= 1e-4, float(np.abs(y0 - y1).max())


def test_filter_dft_variant_is_what_crosscorr_computes():
    """LVMO_VAR_FILTER_DFT restates filter2D's DFT path (templmatch.cpp crossCorr; every 9 x 9 kernel on builds without SSE3) as the float64
    sum of the 81 products rounded once.  Checked against the path itself: a float64 FFT correlation over blocks of crossCorr's sizes
    (blockScale 4.5, minBlockSize 256, getOptimalDFTSize), REFLECT_101 padding, one rounding to binary32 at the end."""
    rng = np.random.default_rng(11)
    w, h = 331, 207
    a = (rng.uniform(0, 100, (h, w)) + 20 * np.sin(np.arange(w) / 9.0)[None, :]).astype(np.float32)
    k = rng.uniform(-0.2, 0.2, (9, 9)).astype(np.float32)
    po.set_variant(po.VARIANTS["filter_dft"])
    try:
        got = po.filter2d(a, k)
        small = po.filter2d(a, k[4:5, 2:7].copy())          # 1 x 5: below dft_filter_size, the direct path even on such builds
    finally:
        po.set_variant(0)
    assert np.array_equal(small, po.filter2d(a, k[4:5, 2:7].copy()))
    # crossCorr's block decomposition
    def optimal_dft(n):
        while True:
            m = n
            for p in (2, 3, 5):
                while m % p == 0:
                    m //= p
            if m == 1:
                return n
            n += 1
    bw = min(max(int(round(9 * 4.5)), 256 - 9 + 1), w); bh = min(max(int(round(9 * 4.5)), 256 - 9 + 1), h)
    dw, dh = max(optimal_dft(bw + 8), 2), optimal_dft(bh + 8)
    bw, bh = min(dw - 8, w), min(dh - 8, h)
    pad = np.pad(a.astype(np.float64), 4, mode="reflect")               # numpy "reflect" = BORDER_REFLECT_101
    K = np.fft.rfft2(k.astype(np.float64), s=(dh, dw))
    want = np.empty((h, w), np.float32)
    for y0 in range(0, h, bh):
        for x0 in range(0, w, bw):
            y1, x1 = min(y0 + bh, h), min(x0 + bw, w)
            blk = pad[y0:y1 + 8, x0:x1 + 8]
            c = np.fft.irfft2(np.fft.rfft2(blk, s=(dh, dw)) * np.conj(K), s=(dh, dw))      # correlation: conj(kernel spectrum)
            want[y0:y1, x0:x1] = c[:y1 - y0, :x1 - x0].astype(np.float32)
    same = float((want == got).mean())
    print("float64 sum rounded once vs float64 FFT correlation in crossCorr's blocks: %.6f of the pixels identical, max |d| %.2e" % (same, float(np.abs(want - got).max())))
    assert same >= 0.9999 and np.abs(want - got).max() <= 1.6e-5          # (one binary32 step at magnitude <= 128 where the two disagree at all)
    d = po.filter2d(a, k)
    assert not np.array_equal(d, got) and np.abs(d - got).max() <= 1e-4


def test_table_entries_near_a_rounding_boundary():
    """How sensitive is the 33^3 table to the last bit of applyGamma?  Count the entries that flip when every interior
    gamma node moves one binary32 step: these are the only entries a correctly-rounded-to-1-ulp pow could change.  With
    the softdouble form (binary64 pow rounded once) a node is off only if the binary64 value lies within ~1e-15 of a
    rounding boundary, so the expected number of wrong ENTRIES is (entries that flip) x ~1e-8."""
    t0 = po.lab_lut_table()
    flips = []
    try:
        for name in ("lut_nudge_up", "lut_nudge_down"):
            po.set_variant(po.VARIANTS[name])
            flips.append(int((po.lab_lut_table() != t0).sum()))
    finally:
        po.set_variant(0)
    print("entries within one gamma ulp of a rounding boundary: up %d, down %d of %d" % (flips[0], flips[1], t0.size))
    assert max(flips) < t0.size // 10
