"""CPU tests of the oracle's primitives against independent numpy/scipy formulations, the
reference's own answers (tests/golden/ref_slices.json, produced by the reference code) and
the known-answer vectors of SURVEY.md 8c."""
import json
import math
import os

import numpy as np
import pytest
from scipy import ndimage

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_slices.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def rng_img(h, w, c=None, seed=0, lo=0.0, hi=1.0):
    r = np.random.default_rng(seed)
    return r.uniform(lo, hi, (h, w) if c is None else (h, w, c)).astype(np.float32)


# ---- reference-pinned scalars -------------------------------------------------------------------
def test_butterworth_matches_reference_code(po, gold, lvm):
    for e in gold["butterworth2"]:
        a, b = po.butterworth2(e["Wn"])
        ra = np.array([float(x) for x in e["a"]])
        rb = np.array([float(x) for x in e["b"]])
        np.testing.assert_allclose(a, ra, rtol=1e-13, atol=1e-300, equal_nan=True)
        np.testing.assert_allclose(b, rb, rtol=1e-13, atol=1e-300, equal_nan=True)


def test_butterworth_survey_known_answers(po):
    a, b = po.butterworth2(0.5 / 15)
    assert abs(a[1] - -1.8521464853959357) < 1e-14 and abs(a[2] - 0.862348626030081) < 1e-14
    assert abs(b[0] - 0.0025505351585362926) < 1e-16 and abs(b[1] - 0.0051010703170725853) < 1e-16
    a, b = po.butterworth2(10 / 15)
    assert abs(a[1] - 0.62020410288672845) < 1e-14 and abs(b[0] - 0.4651530771650465) < 1e-14
    from scipy.signal import butter
    for wn in (0.03, 0.2, 0.5, 0.9):
        sb, sa = butter(2, wn)
        a, b = po.butterworth2(wn)
        np.testing.assert_allclose(a, sa, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(b, sb, rtol=1e-12, atol=1e-15)


def test_optimal_buffer_size_and_max_levels(po, gold):
    L = po.lib()
    for fps, v in gold["optimal_buffer_size"].items():
        assert L.lvmo_optimal_buffer_size(int(fps)) == v
    for fps, v in [(15, 32), (24, 64), (30, 64), (33, 128), (60, 128), (64, 128), (65, 256), (120, 256)]:
        assert L.lvmo_optimal_buffer_size(fps) == v
    assert L.lvmo_max_levels(640, 360) == 7
    assert L.lvmo_max_levels(1920, 1080) == 8
    assert L.lvmo_max_levels(3840, 2160) == 9
    assert L.lvmo_max_levels(5, 100) == 0 and L.lvmo_max_levels(6, 6) == 1


def test_motion_hz_to_blend_matches_reference(lvm, gold):
    for e in gold["motion_hz_to_blend"]:
        assert lvm.synth.motion_hz_to_blend(e["hz"], e["fps"]) == pytest.approx(float(e["blend"]), rel=1e-15, abs=0)
    assert lvm.synth.motion_hz_to_blend(0.4, 30) == pytest.approx(0.08036258807421603, rel=1e-14)
    assert lvm.synth.motion_hz_to_blend(3, 30) == pytest.approx(0.4665119089088967, rel=1e-14)


def test_laplace_gains_table(po):
    # 1080p cfg-2 gains l5..l1 = 20, 13.4203, 5.7102, 1.8551, -0.0725; l0, l6 = 0
    g = po.laplace_gains(1920, 1080, 6, 20.0, 500.0)
    np.testing.assert_allclose(g[5:0:-1], [20, 13.4203, 5.7102, 1.8551, -0.0725], atol=6e-5)
    assert g[0] == 0 and g[6] == 0


# ---- pyramids -----------------------------------------------------------------------------------
K5 = np.array([1, 4, 6, 4, 1], np.float64) / 16.0


def np_pyr_down(a):
    a = a.astype(np.float64)
    p = np.pad(a, 2, mode="reflect")      # numpy 'reflect' == BORDER_REFLECT_101
    t = sum(K5[k] * p[:, k:k + a.shape[1]] for k in range(5))
    t = sum(K5[k] * t[k:k + a.shape[0], :] for k in range(5))
    return t[::2, ::2]


@pytest.mark.parametrize("h,w", [(16, 16), (17, 23), (9, 135), (6, 7), (68, 120)])
def test_pyr_down_vs_numpy(po, h, w):
    a = rng_img(h, w, seed=h * 100 + w)
    d = po.pyr_down(a)
    assert d.shape == ((h + 1) // 2, (w + 1) // 2)
    np.testing.assert_allclose(d, np_pyr_down(a), rtol=0, atol=2e-6)
    a3 = rng_img(h, w, 3, seed=7)
    d3 = po.pyr_down(a3)
    for c in range(3):
        np.testing.assert_array_equal(d3[:, :, c], po.pyr_down(np.ascontiguousarray(a3[:, :, c])))


def np_pyr_up_1d(s, dn):
    """OpenCV pyrUp_ border rules in float64 (x4 kernel, /8 per axis)."""
    n = len(s)
    out = np.zeros(2 * n)
    for i in range(n):
        if i == 0:
            e = 6 * s[0] + 2 * s[1]
        elif i == n - 1:
            e = s[n - 2] + 7 * s[n - 1]
        else:
            e = s[i - 1] + 6 * s[i] + s[i + 1]
        o = 8 * s[n - 1] if i == n - 1 else 4 * (s[i] + s[i + 1])
        out[2 * i], out[2 * i + 1] = e, o
    return out[:dn] / 8.0


@pytest.mark.parametrize("h,w,dh,dw", [(8, 8, 16, 16), (9, 12, 17, 23), (5, 68, 9, 135), (3, 3, 6, 6), (34, 60, 68, 120)])
def test_pyr_up_vs_numpy(po, h, w, dh, dw):
    a = rng_img(h, w, seed=h + w)
    u = po.pyr_up(a, (dw, dh))
    t = np.stack([np_pyr_up_1d(r.astype(np.float64), dw) for r in a])
    # vertical: row -1 -> 1 (same as the left rule), row h -> h-1 (same as the right rule)
    ref = np.stack([np_pyr_up_1d(t[:, x], dh) for x in range(dw)], axis=1)
    np.testing.assert_allclose(u, ref, rtol=0, atol=2e-6)


def test_pyr_up_constant_and_default_size(po):
    a = np.full((7, 9), 3.25, np.float32)
    u = po.pyr_up(a)
    assert u.shape == (14, 18)
    np.testing.assert_allclose(u, 3.25, atol=1e-6)


# ---- colour ---------------------------------------------------------------------------------------
def srgb_to_lab_f64(bgr):
    bgr = bgr.astype(np.float64)
    lin = np.where(bgr <= 0.04045, bgr / 12.92, ((bgr + 0.055) / 1.055) ** 2.4)
    B, G, R = lin[..., 0], lin[..., 1], lin[..., 2]
    X = (0.412453 * R + 0.357580 * G + 0.180423 * B) / 0.950456
    Y = 0.212671 * R + 0.715160 * G + 0.072169 * B
    Z = (0.019334 * R + 0.119193 * G + 0.950227 * B) / 1.088754
    f = lambda t: np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0)  # noqa: E731
    L = np.where(Y > 0.008856, 116 * f(Y) - 16, 903.3 * Y)
    return np.stack([L, 500 * (f(X) - f(Y)), 200 * (f(Y) - f(Z))], -1)


@pytest.fixture
def analytic_lab(po):
    """The oracle's analytic forward Lab (OpenCV with its interpolation switched off) for one test."""
    po.lib().lvmo_set_lab_lut(0)
    yield
    po.lib().lvmo_set_lab_lut(1)


def test_bgr2lab_matches_analytic_definition(po, analytic_lab):
    a = rng_img(40, 50, 3, seed=3)
    lab = po.bgr2lab(a)
    np.testing.assert_allclose(lab, srgb_to_lab_f64(a), rtol=0, atol=2e-4)
    white = po.bgr2lab(np.ones((1, 1, 3), np.float32))
    assert abs(white[0, 0, 0] - 100.0) < 1e-3 and abs(white[0, 0, 1]) < 2e-3 and abs(white[0, 0, 2]) < 2e-3
    black = po.bgr2lab(np.zeros((1, 1, 3), np.float32))
    np.testing.assert_allclose(black, 0, atol=1e-6)


def test_bgr2lab_lut_interpolates_the_definition(po):
    """OpenCV 4's default (the oracle's default): the 33^3 table, trilinear 4-bit weights.  Within the table's
    interpolation error of the CIE definition (L 0.3 of 100, a / b 0.7), exact at white and black."""
    a = rng_img(40, 50, 3, seed=3)
    d = np.abs(po.bgr2lab(a) - srgb_to_lab_f64(a)).reshape(-1, 3).max(axis=0)
    assert d[0] < 0.3 and d[1] < 0.7 and d[2] < 0.7, d
    white = po.bgr2lab(np.ones((1, 1, 3), np.float32))
    assert abs(white[0, 0, 0] - 100.0) < 0.01 and abs(white[0, 0, 1]) < 0.02 and abs(white[0, 0, 2]) < 0.02
    np.testing.assert_allclose(po.bgr2lab(np.zeros((1, 1, 3), np.float32)), 0, atol=1e-6)


def test_lab_round_trip(po, analytic_lab):
    u8 = np.random.default_rng(5).integers(0, 256, (64, 64, 3)).astype(np.float32) * np.float32(1.0 / 255.0)
    back = po.lab2bgr(po.bgr2lab(u8))
    np.testing.assert_allclose(back, u8, rtol=0, atol=2e-4)  # << 1/255: the 8-bit round trip is exact


def test_lab_round_trip_through_the_lut(po):
    """forward LUT + analytic inverse (what a first frame of the reference is): within the table's interpolation error,
    i.e. the 8-bit round trip is exact for most but not all colours."""
    u = np.random.default_rng(5).integers(0, 256, (64, 64, 3))
    back = po.lab2bgr(po.bgr2lab(u.astype(np.float32) * np.float32(1.0 / 255.0)))
    q = np.rint(back * 255.0).astype(np.int64)
    assert np.abs(q - u).max() <= 2 and (q == u).mean() > 0.8


def test_cube_root(po):
    x = np.random.default_rng(1).uniform(1e-4, 1.3, 5000).astype(np.float32)
    r = np.array([po.lib().lvmo_cube_root(float(v)) for v in x], np.float32)
    np.testing.assert_allclose(r, np.cbrt(x.astype(np.float64)), rtol=2e-7)
    assert po.lib().lvmo_cube_root(0.0) == 0.0


# ---- filters ------------------------------------------------------------------------------------
def test_filter2d_vs_scipy(po):
    lp, hp = po.riesz_kernels()
    assert abs(float(lp.sum()) - 0.9994) < 2e-4 and abs(float(hp.sum()) - 0.0057) < 2e-4 and hp[4, 4] == np.float32(-0.9455)
    np.testing.assert_array_equal(lp, lp.T)
    np.testing.assert_array_equal(hp, hp[::-1, ::-1])
    for (h, w) in [(20, 31), (6, 6), (9, 40)]:
        a = rng_img(h, w, seed=h, lo=-5, hi=100)
        for k in (lp, hp, np.array([[-0.2, -0.48, 0, 0.48, 0.2]], np.float32),
                  np.array([[-0.2, -0.48, 0, 0.48, 0.2]], np.float32).T):
            ref = ndimage.correlate(a.astype(np.float64), k.astype(np.float64), mode="mirror")
            np.testing.assert_allclose(po.filter2d(a, k), ref, rtol=0, atol=2e-4)


def test_gauss_kernel_and_sep_filter(po):
    k = po.gauss_kernel(13, 3.0)
    np.testing.assert_allclose(k[:7], [0.018544, 0.034167, 0.056332, 0.083109, 0.109719, 0.129618, 0.137023], atol=1e-6)
    assert abs(float(k.sum()) - 1.0) < 1e-6
    for (h, w) in [(30, 41), (6, 9), (14, 6)]:
        a = rng_img(h, w, seed=w, lo=0, hi=50)
        ref = ndimage.correlate1d(ndimage.correlate1d(a.astype(np.float64), k.astype(np.float64), axis=1, mode="mirror"),
                                  k.astype(np.float64), axis=0, mode="mirror")
        np.testing.assert_allclose(po.sep_filter(a, k), ref, rtol=0, atol=5e-5)


def np_resize_linear(a, dw, dh):
    h, w = a.shape[:2]
    def tab(d, s):
        scale = 1.0 / (d / s)
        ofs, al = [], []
        for i in range(d):
            f = np.float32((i + 0.5) * scale - 0.5)
            si = math.floor(f)
            f = np.float32(f - si)
            if si < 0:
                si, f = 0, np.float32(0)
            if si >= s - 1:
                si, f = s - 1, np.float32(0)
            ofs.append(si); al.append(f)
        return np.array(ofs), np.array(al, np.float64)
    xo, xa = tab(dw, w)
    yo, ya = tab(dh, h)
    a = a.astype(np.float64)
    x1 = np.minimum(xo + 1, w - 1); y1 = np.minimum(yo + 1, h - 1)
    hz = a[:, xo] * (1 - xa)[None, :, None] + a[:, x1] * xa[None, :, None]
    return hz[yo] * (1 - ya)[:, None, None] + hz[y1] * ya[:, None, None]


@pytest.mark.parametrize("h,w,dh,dw", [(1088, 64, 1080, 64), (368, 40, 360, 40), (32, 48, 27, 45), (16, 16, 16, 16)])
def test_resize_linear(po, h, w, dh, dw):
    a = rng_img(h, w, 3, seed=h, lo=0, hi=255)
    r = po.resize_linear(a, (dw, dh))
    np.testing.assert_allclose(r, np_resize_linear(a, dw, dh), rtol=0, atol=1e-4)
    if (h, w) == (dh, dw):
        np.testing.assert_array_equal(r, a)


# ---- DFT ----------------------------------------------------------------------------------------
def pack_ccs(F, n):
    p = np.zeros((F.shape[0], n))
    p[:, 0] = F[:, 0].real
    for k in range(1, (n - 1) // 2 + 1):
        p[:, 2 * k - 1] = F[:, k].real
        p[:, 2 * k] = F[:, k].imag
    if n % 2 == 0:
        p[:, n - 1] = F[:, n // 2].real
    return p


@pytest.mark.parametrize("n", [2, 3, 4, 5, 8, 13, 64, 128])
def test_dft_rows_ccs(po, n):
    x = rng_img(6, n, seed=n, lo=0, hi=255)
    X = po.dft_rows(x)
    np.testing.assert_allclose(X, pack_ccs(np.fft.rfft(x.astype(np.float64), axis=1) / n, n), rtol=0, atol=3e-5)
    np.testing.assert_allclose(po.idft_rows(X) * n, x, rtol=0, atol=1e-3)


def test_mul_spectrums_packed_complex(po):
    n = 8
    a = rng_img(3, n, seed=1); b = rng_img(3, n, seed=2)
    d = po.mul_spectrums_rows(a, b)
    np.testing.assert_allclose(d[:, 0], a[:, 0] * b[:, 0]); np.testing.assert_allclose(d[:, 7], a[:, 7] * b[:, 7])
    for j in (1, 3, 5):
        za = a[:, j] + 1j * a[:, j + 1]; zb = b[:, j] + 1j * b[:, j + 1]
        np.testing.assert_allclose(d[:, j] + 1j * d[:, j + 1], za * zb, rtol=1e-6)
    n = 7
    a = rng_img(2, n, seed=3); b = rng_img(2, n, seed=4)
    d = po.mul_spectrums_rows(a, b)
    za = a[:, 5] + 1j * a[:, 6]; zb = b[:, 5] + 1j * b[:, 6]
    np.testing.assert_allclose(d[:, 5] + 1j * d[:, 6], za * zb, rtol=1e-6)


@pytest.mark.parametrize("T,fps,lo,hi", [(2, 60, 0.83, 1.0), (3, 30, 0.0, 20.0), (17, 30, 0.5, 4.0), (64, 30, 0.84, 1.43), (128, 60, 0.83, 1.0)])
def test_ideal_filter_fast_equals_full_and_definition(po, T, fps, lo, hi):
    rows, cn = 37, 3
    win = rng_img(rows, T, cn, seed=T, lo=0, hi=255)
    fast = po.ideal_filter(win, lo, hi, fps, full=False)
    full = po.ideal_filter(win, lo, hi, fps, full=True)
    np.testing.assert_array_equal(fast, full)
    # independent restatement: dft -> packed 0/1 mask treated as a packed complex spectrum -> idft -> min-max
    lo2 = lo if lo != 0 else 0.01
    fl, fh = 2 * lo2 * np.float32(T) / fps, 2 * hi * np.float32(T) / fps
    mask = np.array([1.0 if (x >= fl and x <= fh) else 0.0 for x in range(T)], np.float32)
    filt = np.empty_like(win)
    for c in range(cn):
        X = po.dft_rows(np.ascontiguousarray(win[:, :, c]))
        Y = po.mul_spectrums_rows(X, np.tile(mask, (rows, 1)))
        filt[:, :, c] = po.idft_rows(Y)
    mn, mx = float(filt.min()), float(filt.max())
    sc = 1.0 / (mx - mn) if mx - mn > np.finfo(float).eps else 0.0
    ref = filt * np.float32(sc) + np.float32(-mn * sc)
    np.testing.assert_array_equal(fast, ref.astype(np.float32))
