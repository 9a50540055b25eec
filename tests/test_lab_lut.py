"""OpenCV 4's interpolated forward Lab table (what cv::cvtColor(COLOR_BGR2Lab) on CV_32F runs, MagnifyCore.hpp:90,219):
the library's table against the oracle's independent restatement, the closed-form cell index, the table's structure, and
the distance between the LUT and the analytic conversion."""
import numpy as np
import pytest


def _oracle_table(po):
    t = np.empty(33 * 33 * 33 * 3, np.int16)
    po.lib().lvmo_lab_lut_table(t.ctypes.data)
    return t


def test_fine_index_closed_form():
    """lab_lut.h: cvRound(float(u) * float(1/255) * 16384) >> 5 == (514 u + 4) >> 8 for every u8 value."""
    a255 = np.float32(1.0 / np.float32(255.0))
    u = np.arange(256, dtype=np.float32)
    c = np.rint((u * a255).astype(np.float32) * np.float32(16384)).astype(np.int64)
    assert np.array_equal(c >> 5, (np.arange(256) * 514 + 4) >> 8)
    assert c[0] == 0 and c[255] == 16384


def test_library_table_equals_oracle_table(lvm, po, emu):
    ctx = lvm.Context(0, 1, emu)
    try:
        lib_t = ctx.lab_lut()
    finally:
        ctx.close()
    assert np.array_equal(lib_t, _oracle_table(po))


def test_table_structure(po):
    t = _oracle_table(po).reshape(33, 33, 33, 3).astype(np.int64)      # [r][q][p][ch]
    assert t[0, 0, 0, 0] == 0 and abs(t[32, 32, 32, 0] - 16384) <= 1      # black: L = 0, white: L = 100
    assert abs(t[32, 32, 32, 1] - 8192) <= 2 and abs(t[32, 32, 32, 2] - 8192) <= 2   # white: a = b = 0
    for k in range(33):                                                   # the grey axis is neutral
        assert abs(t[k, k, k, 1] - 8192) <= 2 and abs(t[k, k, k, 2] - 8192) <= 2
    assert (np.diff(t[..., 0], axis=0) >= 0).all() and (np.diff(t[..., 0], axis=1) >= 0).all() and (np.diff(t[..., 0], axis=2) >= 0).all()
    assert t.min() >= 0 and t.max() <= 16384


def test_node_colours_return_the_table(po):
    """At a grid node every weight of the upper neighbours is 0: the conversion returns the table entry itself -- the
    property oracle/ref_driver.cpp uses to recover the table of a real OpenCV build."""
    t = _oracle_table(po).reshape(33, 33, 33, 3)
    g = np.arange(33, dtype=np.float32) / np.float32(32)
    bgr = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 1, 3)     # [r][q][p] -> (B, G, R)
    lab = po.bgr2lab(bgr).reshape(33, 33, 33, 3)
    iL = np.rint(lab[..., 0] * (16384.0 / 100.0)).astype(np.int64)
    ia = np.rint((lab[..., 1] + 128.0) * 64.0).astype(np.int64)
    ib = np.rint((lab[..., 2] + 128.0) * 64.0).astype(np.int64)
    assert np.array_equal(iL, t[..., 0]) and np.array_equal(ia, t[..., 1]) and np.array_equal(ib, t[..., 2])


def test_lut_against_analytic(po):
    """The table interpolates the analytic conversion: L within 0.3 (of 100), a / b within 0.7 on random u8 colours."""
    rng = np.random.default_rng(7)
    u = rng.integers(0, 256, size=(4096, 1, 3)).astype(np.float32) * np.float32(1.0 / np.float32(255.0))
    lut = po.bgr2lab(u)
    try:
        po.lib().lvmo_set_lab_lut(0)
        ana = po.bgr2lab(u)
    finally:
        po.lib().lvmo_set_lab_lut(1)
    d = np.abs(lut - ana).reshape(-1, 3).max(axis=0)
    assert d[0] <= 0.3 and d[1] <= 0.7 and d[2] <= 0.7, d
    assert d.max() > 0.01                                   # and they are NOT the same function


def test_set_lab_lut_round_trip(lvm, po, emu):
    """lvm_set_lab_lut / lvmo_lab_lut_override: a perturbed table in both gives bit-identical frames again."""
    from helpers import run_pair
    t = _oracle_table(po).copy()
    rng = np.random.default_rng(3)
    t2 = np.clip(t.astype(np.int32) + rng.integers(-1, 2, size=t.shape), 0, 16384).astype(np.int16)
    ck, pk = lvm.synth.config(0, (96, 64, 3))
    try:
        po.lib().lvmo_lab_lut_override(t2.ctypes.data)
        run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 3, 0.0, exact=True, lab_lut=t2)
    finally:
        po.lib().lvmo_lab_lut_override(None)


def _numpy_table():
    """initLabTabs' interpolation table restated a THIRD time, in numpy binary32 arithmetic (independent of the two C / C++
    builders apart from cv::cubeRoot, taken from the oracle)."""
    f32 = np.float32
    M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float64)
    D65 = np.array([0.950456, 1.0, 1.088754], np.float64)
    C = (M * np.array([1.0 / D65[0], 1.0, 1.0 / D65[2]])[:, None]).astype(f32)
    g = np.arange(33, dtype=f32) / f32(32)
    thr, low, shift, power = f32(809) / f32(20000), f32(323) / f32(25), f32(11) / f32(200), f32(12) / f32(5)
    # applyGamma: argument and binary32 constants promoted to binary64, pow in binary64, ONE rounding (color_lab.cpp's softdouble form)
    f64 = np.float64
    gd = g.astype(f64)
    hi = np.power((gd + f64(shift)) / (f64(1) + f64(shift)), f64(power))
    gam = np.where(gd <= f64(thr), gd / f64(low), hi).astype(f32)
    R, G, B = gam[None, None, :], gam[None, :, None], gam[:, None, None]          # [r][q][p]: p = R fastest
    def lin(c):
        return ((R * c[0]).astype(f32) + (G * c[1]).astype(f32)).astype(f32) + (B * c[2]).astype(f32)
    X, Y, Z = (lin(C[i]).astype(f32) for i in range(3))
    return X, Y, Z


def test_table_against_a_numpy_restatement(po):
    """Same formulas, third implementation (numpy float32, vectorised): every entry equal to the oracle's table."""
    f32 = np.float32
    X, Y, Z = _numpy_table()
    lth, lsc, lb = f32(216) / f32(24389), f32(841) / f32(108), f32(16) / f32(116)
    cbrt = np.vectorize(lambda v: po.lib().lvmo_cube_root(float(v)), otypes=[np.float32])
    def f(t):
        lin = (t.astype(np.float64) * np.float64(lsc) + np.float64(lb)).astype(f32)     # fma: one rounding (exact product in float64)
        return np.where(t > lth, cbrt(t), lin).astype(f32)
    FX, FY, FZ = f(X), f(Y), f(Z)
    L = np.where(Y > lth, (f32(116) * FY).astype(f32) - f32(16), (Y * (f32(24389) / f32(27))).astype(f32)).astype(f32)
    a = (f32(500) * (FX - FY).astype(f32)).astype(f32)
    b = (f32(200) * (FY - FZ).astype(f32)).astype(f32)
    tL = np.rint(((f32(16384) * L).astype(f32) / f32(100)).astype(f32)).astype(np.int64)
    ta = np.rint(((f32(16384) * (a + f32(128)).astype(f32)).astype(f32) / f32(256)).astype(f32)).astype(np.int64)
    tb = np.rint(((f32(16384) * (b + f32(128)).astype(f32)).astype(f32) / f32(256)).astype(f32)).astype(np.int64)
    t = _oracle_table(po).reshape(33, 33, 33, 3).astype(np.int64)
    assert np.array_equal(tL, t[..., 0]) and np.array_equal(ta, t[..., 1]) and np.array_equal(tb, t[..., 2])


def test_trilinear_interpolation_against_a_numpy_restatement(po):
    """trilinearInterpolate + the float scaling restated in vectorised numpy integer arithmetic on the oracle's table:
    bit-equal to lvmo_bgr2lab on 20 000 random u8 colours plus the corners of the cube."""
    rng = np.random.default_rng(11)
    u = rng.integers(0, 256, size=(20000, 3))
    u = np.concatenate([u, np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [254, 255, 1]])])
    s = (u.astype(np.float32) * np.float32(1.0 / np.float32(255.0))).astype(np.float32)       # B, G, R in [0, 1]
    c = np.rint(s * np.float32(16384)).astype(np.int64)[:, ::-1]                               # -> (R, G, B) at 1/16384
    t = _oracle_table(po).reshape(33, 33, 33, 3).astype(np.int64)                              # [r][q][p][ch]
    cell, w1 = c >> 9, (c >> 5) & 15
    acc = np.zeros((len(u), 3), np.int64)
    for dp in (0, 1):
        for dq in (0, 1):
            for dr in (0, 1):
                p = np.minimum(cell[:, 0] + dp, 32); q = np.minimum(cell[:, 1] + dq, 32); r = np.minimum(cell[:, 2] + dr, 32)
                wgt = (w1[:, 0] if dp else 16 - w1[:, 0]) * (w1[:, 1] if dq else 16 - w1[:, 1]) * (w1[:, 2] if dr else 16 - w1[:, 2])
                acc += t[r, q, p] * wgt[:, None]
    acc = (acc + 2048) >> 12
    want = np.stack([acc[:, 0].astype(np.float32) * np.float32(1.0 / 16384) * np.float32(100),
                     acc[:, 1].astype(np.float32) * np.float32(1.0 / 16384) * np.float32(256) - np.float32(128),
                     acc[:, 2].astype(np.float32) * np.float32(1.0 / 16384) * np.float32(256) - np.float32(128)], -1).astype(np.float32)
    got = po.bgr2lab(s.reshape(-1, 1, 3)).reshape(-1, 3)
    assert np.array_equal(got, want)
