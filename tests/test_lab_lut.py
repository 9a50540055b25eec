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
