"""The output quantiser of the Lab modes as a step table (round 6; lab_tables.cpp build_u8_steps, lvm_internal.h u8_step).

u8 = saturate_cast<uchar>(cvRound(255 * invGamma(clip01(c)) + 1/255)) per channel is what Lab2RGBfloat's splineInterpolate and
convertTo(CV_8U, 255, 1.0/255) do to a linear value c (MagnifyCore.hpp:152-153, :275-276).  The default flavour of the output
kernels replaces spline evaluation + scale + round + clamp by one table hit + one compare, which is exact iff that composite is
a monotone step function of c -- over EVERY float, rounding noise of the cubic included.  Checked here three ways:
  * CPU, exhaustively over the 1 065 353 217 floats of [0, 1] with the oracle's own code: no descent, exactly 255 steps of one level;
  * emulation build: the table against the EXACT flavour's code around every threshold, on a strided sample of all bit patterns and
    on the special values;
  * GPU: all 2^32 bit patterns (lvm_debug_sweep_u8_steps).
"""
import ctypes as C
import struct

import numpy as np
import pytest

ONE_BITS = 0x3F800000


def _bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def _sweep_oracle(po, first, count):
    L = po.lib()
    L.lvmo_u8_of_linear_sweep.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
    L.lvmo_u8_of_linear_sweep.restype = None
    d, s, j = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    thr = np.zeros(256, np.uint32)
    L.lvmo_u8_of_linear_sweep(first, count, C.byref(d), C.byref(s), C.byref(j), thr.ctypes.data)
    return int(d.value), int(s.value), int(j.value), thr


@pytest.fixture(scope="module")
def oracle_sweep(po):
    po.lib().lvmo_set_threads(8)
    return _sweep_oracle(po, 0, ONE_BITS + 1)


def test_quantiser_is_a_monotone_step_function_over_every_float_in_0_1(oracle_sweep):
    descents, steps, jumps, thr = oracle_sweep
    assert descents == 0, "the byte drops somewhere as c grows: a step table cannot be exact"
    assert steps == 255 and jumps == 0
    assert thr[0] == 0 and (np.diff(thr.astype(np.int64)) > 0).all()
    # at most one threshold per slice of 1/4096 (what one table entry can hold), with room to spare
    t = thr[1:].view(np.float32).astype(np.float64) * 4096.0
    assert len(set(np.floor(t).astype(int))) == 255
    assert np.diff(t).min() > 1.2


def test_quantiser_outside_0_1_and_specials(po):
    L = po.lib()
    L.lvmo_u8_of_linear.argtypes = [C.c_float]
    L.lvmo_u8_of_linear.restype = C.c_uint8
    for v, want in [(-0.0, 0), (-1e-30, 0), (-5.0, 0), (float("-inf"), 0), (1.0, 255), (1.0000001, 255), (7.0, 255), (float("inf"), 255), (float("nan"), 0)]:
        assert L.lvmo_u8_of_linear(v) == want, v


def _ranges_around_thresholds(thr, halo=1500):
    for k in range(1, 256):
        b = int(thr[k])
        yield max(0, b - halo), 2 * halo


def test_step_table_equals_spline_scale_round_emu(lvm, emu, oracle_sweep):
    """the emulation build's table + u8_step against its EXACT code (the operations the oracle runs): around every threshold, at the
    slice boundaries, on a strided sample of all patterns, and on negative / large / infinite / NaN patterns"""
    _, _, _, thr = oracle_sweep
    ctx = lvm.Context(0, 1, emu)
    try:
        for first, count in _ranges_around_thresholds(thr):
            assert ctx.sweep_u8_steps(first, count) == (0, 0), "threshold near pattern %08x" % first
        for i in list(range(1, 40)) + [511, 512, 1023, 2048, 4094, 4095]:          # both sides of slice boundaries i / 4096
            b = _bits(i / 4096.0)
            assert ctx.sweep_u8_steps(b - 64, 128) == (0, 0)
        rng = np.random.default_rng(6)
        for first in rng.integers(0, ONE_BITS, 60):
            assert ctx.sweep_u8_steps(int(first), 4096) == (0, 0)
        for first, count in [(0, 4096), (ONE_BITS - 4096, 8192), (0x7F7FFF00, 0x200),      # denormals, around 1.0, the largest floats and +inf, NaNs
                             (0x80000000, 4096), (0xBF800000 - 64, 128), (0xFF7FFF00, 0x200), (0xFFFFFF00, 0x100)]:
            assert ctx.sweep_u8_steps(first, count) == (0, 0), hex(first)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_step_table_equals_spline_scale_round_on_every_float_gpu(lvm, hip):
    """all 2^32 binary32 patterns on the device: the table path and OpenCV's operations one by one give the same byte"""
    ctx = lvm.Context(0, 1)
    try:
        bad, first_bad = ctx.sweep_u8_steps(0, 1 << 32)
        assert bad == 0, "first mismatch at pattern %08x" % first_bad
    finally:
        ctx.close()


@pytest.mark.gpu
def test_step_table_frames_equal_the_spline_frames_gpu(lvm, hip):
    """whole frames: the default flavour (step table) and the same flavour with the float frame kept (spline path: lvm_debug_keep_float
    switches the output kernels back to spline + scale + round) differ at most where fma(o, 255, 1/255) and o * 255 + 1/255 round apart"""
    ck, pk = lvm.synth.config(1, (640, 360, 4))
    clip = lvm.synth.Clip(**ck)
    cp = lvm.LvmParams(pk["mode"], pk["levels"], pk["amplification"], pk["coWavelength"], pk["coLow"], pk["coHigh"], pk["chromAttenuation"], pk["framerate"], 0)
    a, b = lvm.Context(0, 1), lvm.Context(0, 1)
    try:
        b.keep_float(True)
        worst = 1.0
        for t in range(6):
            f = clip.frame(t)
            oa, _ = a.process(f, cp)
            ob, _ = b.process(f, cp)
            d = np.abs(oa.astype(int) - ob.astype(int))
            assert d.max() <= 1
            worst = min(worst, float((d == 0).mean()))
        assert worst >= 0.9999
    finally:
        a.close(); b.close()
