"""ctypes binding of the CPU ORACLE (oracle/liblvm_oracle.so) and of the reference slices
(oracle/_ref/libref_slices.so).

TEST INFRASTRUCTURE ONLY: import from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODE_LAPLACE, MODE_PHASE, MODE_COLOR, MODE_NONE = 0, 1, 2, 3


class Params(C.Structure):
    """reference: src/processing/IProcessor.hpp:14-23 (+ preprocess key)."""
    _fields_ = [("mode", C.c_int32), ("levels", C.c_int32), ("amplification", C.c_double),
                ("coWavelength", C.c_double), ("coLow", C.c_double), ("coHigh", C.c_double),
                ("chromAttenuation", C.c_double), ("framerate", C.c_double),
                ("preprocess_key", C.c_uint64)]


def build(force=False):
    so = os.path.join(_HERE, "liblvm_oracle.so")
    src = os.path.join(_HERE, "lvm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)
    return so


_lib = None
_ref = None
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class PreParams(C.Structure):
    """lvmo_pre_params: PreprocessParams (IProcessor.hpp:26-35) + ProcessorConfig::grayscale."""
    _fields_ = [("downscale", C.c_int32), ("roi_enabled", C.c_int32), ("roiX", C.c_float), ("roiY", C.c_float),
                ("roiW", C.c_float), ("roiH", C.c_float), ("grayscale", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.lvmo_create.restype = C.c_void_p
        L.lvmo_destroy.argtypes = [C.c_void_p]
        L.lvmo_reset.argtypes = [C.c_void_p]
        L.lvmo_set_threads.argtypes = [C.c_int]
        L.lvmo_process.argtypes = [C.c_void_p, C.POINTER(Params), _u8p, C.c_int, C.c_int, C.c_int,
                                   C.c_ssize_t, _u8p, C.c_ssize_t, C.POINTER(C.c_int)]
        L.lvmo_process.restype = C.c_int
        L.lvmo_last_float.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.lvmo_last_float.restype = C.POINTER(C.c_float)
        L.lvmo_last_minmax.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.lvmo_max_levels.argtypes = [C.c_int, C.c_int]
        L.lvmo_optimal_buffer_size.argtypes = [C.c_int]
        L.lvmo_butterworth2.argtypes = [C.c_double, _f64p, _f64p]
        L.lvmo_laplace_gains.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _f32p]
        L.lvmo_pyr_down.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p]
        L.lvmo_pyr_up.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int]
        L.lvmo_bgr2lab.argtypes = [_f32p, C.c_int, _f32p]
        L.lvmo_lab2bgr.argtypes = [_f32p, C.c_int, _f32p]
        L.lvmo_filter2d.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p]
        L.lvmo_gauss_kernel.argtypes = [C.c_int, C.c_double, _f32p]
        L.lvmo_sep_filter.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, _f32p]
        L.lvmo_resize_linear.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int]
        L.lvmo_dft_rows.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
        L.lvmo_idft_rows.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
        L.lvmo_mul_spectrums_rows.argtypes = [_f32p, _f32p, C.c_int, C.c_int, _f32p]
        L.lvmo_ideal_filter.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                        C.c_double, _f32p, C.c_int]
        L.lvmo_riesz_kernels.argtypes = [_f32p, _f32p]
        L.lvmo_cube_root.argtypes = [C.c_float]
        L.lvmo_cube_root.restype = C.c_float
        L.lvmo_gamma_tab.argtypes = [C.c_int]
        L.lvmo_gamma_tab.restype = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        L.lvmo_preprocess_geometry.argtypes = [C.POINTER(PreParams), C.c_int, C.c_int, C.c_int, ip, ip, ip, ip, ip, ip, ip]
        L.lvmo_preprocess_geometry.restype = None
        L.lvmo_preprocess.argtypes = [C.POINTER(PreParams), _u8p, C.c_int, C.c_int, C.c_int, C.c_ssize_t, _u8p]
        L.lvmo_preprocess.restype = None
        L.lvmo_resize_area_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_ssize_t, _u8p, C.c_int, C.c_int]
        L.lvmo_resize_area_u8.restype = None
        L.lvmo_bgr2gray_u8.argtypes = [_u8p, C.c_int, _u8p]
        L.lvmo_bgr2gray_u8.restype = None
        L.lvmo_set_lab_lut.argtypes = [C.c_int]
        L.lvmo_set_lab_lut.restype = None
        L.lvmo_lab_lut_table.argtypes = [C.c_void_p]
        L.lvmo_lab_lut_table.restype = None
        L.lvmo_lab_lut_override.argtypes = [C.c_void_p]
        L.lvmo_lab_lut_override.restype = None
        L.lvmo_set_variant.argtypes = [C.c_uint]
        L.lvmo_set_variant.restype = None
        L.lvmo_get_variant.restype = C.c_uint
        L.lvmo_compose_geometry.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, ip]
        L.lvmo_compose.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_ssize_t, C.c_void_p, C.c_ssize_t]
        # a few threads only: on many-core hosts OpenMP fork/join over tiny pyramid levels dominates
        L.lvmo_set_threads(max(1, min(8, os.cpu_count() or 1)))
        _lib = L
    return _lib


# lvm_oracle.h LVMO_VAR_*: the unpinned OpenCV build choices as switches (0 = the restatement the parity tests use)
VARIANTS = {"pyr_simd": 1, "filter_unfused": 2, "addw_fused": 4, "mul_f32": 8, "gamma_f32": 16, "lut_nudge_up": 32,
            "lut_nudge_down": 64, "spline_cv3": 128, "dft_f32": 256, "filter_dft": 512}


def set_variant(mask):
    lib().lvmo_set_variant(int(mask))


def gamma_tab(inverse):
    """the 1024-knot spline (4 coefficients per knot) of the forward / inverse sRGB gamma (color_lab.cpp splineBuild)"""
    fn = lib().lvmo_gamma_tab
    fn.restype = C.POINTER(C.c_float)
    fn.argtypes = [C.c_int]
    return np.ctypeslib.as_array(fn(int(bool(inverse))), shape=(4096,)).copy()


def lab_lut_table():
    t = np.empty(33 * 33 * 33 * 3, np.int16)
    lib().lvmo_lab_lut_table(t.ctypes.data)
    return t


def ref_slices():
    """The OpenCV-free slices of the reference compiled in place (None when absent)."""
    global _ref
    if _ref is None:
        so = os.path.join(_HERE, "_ref", "libref_slices.so")
        if not os.path.exists(so):
            return None
        R = C.CDLL(so)
        R.ref_getOptimalBufferSize.argtypes = [C.c_int]
        R.ref_butterworth.argtypes = [C.c_uint, C.c_double, _f64p, _f64p]
        R.ref_motionHzToBlend.argtypes = [C.c_double, C.c_double]
        R.ref_motionHzToBlend.restype = C.c_double
        _ref = R
    return _ref


def make_params(mode, levels, amplification=0.0, coWavelength=0.0, coLow=0.0, coHigh=0.0,
                chromAttenuation=0.0, framerate=30.0, preprocess_key=0):
    return Params(mode, levels, amplification, coWavelength, coLow, coHigh, chromAttenuation,
                  framerate, preprocess_key)


class Oracle:
    """Mirror of MagnificationProcessor (reference: MagnificationProcessor.hpp:13-23)."""

    def __init__(self):
        self._l = lib()
        self._c = self._l.lvmo_create()

    def close(self):
        if self._c:
            self._l.lvmo_destroy(self._c)
            self._c = None

    __del__ = close

    def reset(self):
        self._l.lvmo_reset(self._c)

    def process(self, frame, params):
        """frame: HxWx3 or HxW uint8.  Returns (out_u8, produced)."""
        frame = np.ascontiguousarray(frame)
        h, w = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        out = np.empty_like(frame)
        produced = C.c_int(0)
        rc = self._l.lvmo_process(self._c, C.byref(params), frame.reshape(-1), w, h, ch, w * ch,
                                  out.reshape(-1), w * ch, C.byref(produced))
        if rc != 0:
            raise RuntimeError("oracle error %d" % rc)
        if not produced.value:
            return frame, False
        return out, True

    def last_float(self):
        w, h, c = C.c_int(), C.c_int(), C.c_int()
        p = self._l.lvmo_last_float(self._c, C.byref(w), C.byref(h), C.byref(c))
        n = w.value * h.value * c.value
        a = np.ctypeslib.as_array(p, shape=(n,)).copy()
        return a.reshape(h.value, w.value, c.value) if c.value > 1 else a.reshape(h.value, w.value)

    def last_minmax(self):
        a, b = C.c_double(), C.c_double()
        self._l.lvmo_last_minmax(self._c, C.byref(a), C.byref(b))
        return a.value, b.value


# ---- primitive wrappers (unit tests) ---------------------------------------------------------
def _img(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    h, w = a.shape[:2]
    cn = 1 if a.ndim == 2 else a.shape[2]
    return a, w, h, cn


def pyr_down(a):
    a, w, h, cn = _img(a)
    out = np.empty(((h + 1) // 2, (w + 1) // 2) + ((cn,) if a.ndim == 3 else ()), np.float32)
    lib().lvmo_pyr_down(a.reshape(-1), w, h, cn, out.reshape(-1))
    return out


def pyr_up(a, dsize=None):
    a, w, h, cn = _img(a)
    dw, dh = dsize if dsize else (2 * w, 2 * h)
    out = np.empty((dh, dw) + ((cn,) if a.ndim == 3 else ()), np.float32)
    lib().lvmo_pyr_up(a.reshape(-1), w, h, cn, out.reshape(-1), dw, dh)
    return out


def bgr2lab(a):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    lib().lvmo_bgr2lab(a.reshape(-1), a.size // 3, out.reshape(-1))
    return out


def lab2bgr(a):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    lib().lvmo_lab2bgr(a.reshape(-1), a.size // 3, out.reshape(-1))
    return out


def filter2d(a, k):
    a, w, h, _ = _img(a)
    k = np.ascontiguousarray(k, np.float32)
    out = np.empty_like(a)
    lib().lvmo_filter2d(a.reshape(-1), w, h, k.reshape(-1), k.shape[1], k.shape[0], out.reshape(-1))
    return out


def gauss_kernel(n, sigma):
    k = np.empty(n, np.float32)
    lib().lvmo_gauss_kernel(n, sigma, k)
    return k


def sep_filter(a, k):
    a, w, h, _ = _img(a)
    k = np.ascontiguousarray(k, np.float32)
    out = np.empty_like(a)
    lib().lvmo_sep_filter(a.reshape(-1), w, h, k, k.size, out.reshape(-1))
    return out


def resize_linear(a, dsize):
    a, w, h, cn = _img(a)
    dw, dh = dsize
    out = np.empty((dh, dw) + ((cn,) if a.ndim == 3 else ()), np.float32)
    lib().lvmo_resize_linear(a.reshape(-1), w, h, cn, out.reshape(-1), dw, dh)
    return out


def dft_rows(a):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    lib().lvmo_dft_rows(a.reshape(-1), a.shape[0], a.shape[1], out.reshape(-1))
    return out


def idft_rows(a):
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    lib().lvmo_idft_rows(a.reshape(-1), a.shape[0], a.shape[1], out.reshape(-1))
    return out


def mul_spectrums_rows(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty_like(a)
    lib().lvmo_mul_spectrums_rows(a.reshape(-1), b.reshape(-1), a.shape[0], a.shape[1], out.reshape(-1))
    return out


def ideal_filter(win, lo, hi, fps, full=False):
    """win: rows x cols x cn float32."""
    win = np.ascontiguousarray(win, np.float32)
    rows, cols, cn = win.shape
    out = np.empty_like(win)
    lib().lvmo_ideal_filter(win.reshape(-1), rows, cols, cn, lo, hi, fps, out.reshape(-1), int(full))
    return out


def butterworth2(Wn):
    a = np.zeros(3)
    b = np.zeros(3)
    lib().lvmo_butterworth2(Wn, a, b)
    return a, b


def laplace_gains(w, h, levels, amplification, coWavelength):
    g = np.zeros(levels + 1, np.float32)
    lib().lvmo_laplace_gains(w, h, levels, amplification, coWavelength, g)
    return g


def riesz_kernels():
    lp = np.zeros(81, np.float32)
    hp = np.zeros(81, np.float32)
    lib().lvmo_riesz_kernels(lp, hp)
    return lp.reshape(9, 9), hp.reshape(9, 9)


def gamma_tab(inverse=False):
    p = lib().lvmo_gamma_tab(int(inverse))
    return np.ctypeslib.as_array(p, shape=(4096,)).copy()


def make_pre_params(downscale=1, roiEnabled=False, roiX=0.0, roiY=0.0, roiW=1.0, roiH=1.0, grayscale=False):
    return PreParams(int(downscale), 1 if roiEnabled else 0, roiX, roiY, roiW, roiH, 1 if grayscale else 0)


def preprocess_geometry(pp, w, h, ch):
    v = [C.c_int() for _ in range(7)]
    lib().lvmo_preprocess_geometry(C.byref(pp), w, h, ch, *[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def preprocess(frame, pp):
    """PreprocessProcessor + GrayscaleProcessor of the reference on a uint8 frame."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w = frame.shape[:2]
    ch = 1 if frame.ndim == 2 else frame.shape[2]
    _, _, _, _, ow, oh, och = preprocess_geometry(pp, w, h, ch)
    out = np.empty((oh, ow) if och == 1 else (oh, ow, och), dtype=np.uint8)
    lib().lvmo_preprocess(C.byref(pp), frame, w, h, ch, w * ch, out)
    return out


def resize_area_u8(a, dsize):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    h, w = a.shape[:2]
    cn = 1 if a.ndim == 2 else a.shape[2]
    dw, dh = dsize
    out = np.empty((dh, dw) if cn == 1 else (dh, dw, cn), dtype=np.uint8)
    lib().lvmo_resize_area_u8(a, w, h, cn, w * cn, out, dw, dh)
    return out


def bgr2gray_u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    out = np.empty(a.shape[:2], dtype=np.uint8)
    lib().lvmo_bgr2gray_u8(a, a.shape[0] * a.shape[1], out)
    return out


def apply_overlay(canvas, labels):
    """What the tables of lvm_export_set_overlay say a label does to a canvas (the reference: drawLabel, export/Exporter.cpp:36-50, whose
    effect on a pixel is a function of that pixel's byte): labels = [(x, y, cls [h][w], fn [n_classes][256]), ...]; returns a new canvas."""
    out = np.array(canvas, dtype=np.uint8, copy=True)
    for x, y, cls, fn in labels:
        cls = np.asarray(cls); fn = np.asarray(fn, dtype=np.uint8)
        h, w = cls.shape
        region = out[y:y + h, x:x + w]
        region[...] = fn[cls[:, :, None].astype(np.int64), region.astype(np.int64)]
    return out


def standin_label_tables(text_w, text_h, pad, x, y, canvas_w, canvas_h, seed=0):
    """A STAND-IN for the reference-side renderer (INTEGRATION.md section 5 renders the real tables with cv::getTextSize / addWeighted /
    putText, none of which exists in this image): a rectangle of drawLabel's geometry (Exporter.cpp:41-44: text size + 2 pad, clipped to the
    canvas), darkened by the binary32 product v * 0.35f rounded half to even, with synthetic "strokes" of 255 coverage levels blended as
    d + ((255 - d) * a + 127 >> 8).  Only the PLUMBING is under test with it (tables -> device -> canvas == apply_overlay); whether the
    tables equal OpenCV's drawing is the renderer's business, and it reads them off OpenCV itself."""
    w = min(text_w + 2 * pad, canvas_w - x); h = min(text_h + 2 * pad, canvas_h - y)
    assert w > 0 and h > 0
    rng = np.random.default_rng(seed)
    v = np.arange(256, dtype=np.float32)
    dark = np.clip(np.rint(v * np.float32(0.35)), 0, 255).astype(np.int64)               # np.rint: half to even, like cvRound
    cov = np.zeros((h, w), np.int64)
    for k in range(6):                                                                   # a few anti-aliased "strokes"
        cx = rng.integers(pad, max(pad + 1, w - pad)); yy = np.arange(h)[:, None]; xx = np.arange(w)[None, :]
        d = np.abs(xx - cx - 0.37 * (yy - h / 2))
        cov = np.maximum(cov, np.clip(np.rint(255 * (1.6 - d)), 0, 255).astype(np.int64) * ((yy >= pad) & (yy < h - pad)))
    levels = np.unique(cov)
    cls = np.searchsorted(levels, cov).astype(np.uint16)
    fn = np.stack([np.clip(dark + (((255 - dark) * int(a) + 127) >> 8), 0, 255) for a in levels]).astype(np.uint8)
    return (x, y, cls, fn)


def compose(split, orig, proc):
    """Exporter::compose (export/Exporter.cpp:53-88, no text overlay): returns the BGR canvas or None (empty Mat)."""
    proc = np.ascontiguousarray(proc, dtype=np.uint8)
    ph, pw = proc.shape[:2]
    pch = 1 if proc.ndim == 2 else proc.shape[2]
    if orig is not None:
        orig = np.ascontiguousarray(orig, dtype=np.uint8)
        oh, ow = orig.shape[:2]
        och = 1 if orig.ndim == 2 else orig.shape[2]
    else:
        oh, ow, och = ph, pw, pch
    cw, ch = C.c_int(), C.c_int()
    if not lib().lvmo_compose_geometry(split, ow, oh, pw, ph, C.byref(cw), C.byref(ch)):
        return None
    canvas = np.zeros((ch.value, cw.value, 3), np.uint8)
    lib().lvmo_compose(split, orig.ctypes.data if orig is not None else None, ow, oh, och, ow * och, proc.ctypes.data, pw, ph, pch, pw * pch,
                       canvas.ctypes.data, cw.value * 3)
    return canvas


class RefOracle:
    """The REAL reference stage (oracle/_ref/libref_magnify.so: the reference's own sources + oracle/ref_driver.cpp),
    same interface as Oracle.  Only exists where OpenCV 4 was found at build time; available() says so."""

    _lib = None

    @classmethod
    def available(cls):
        if cls._lib is None:
            so = os.path.join(_HERE, "_ref", "libref_magnify.so")
            if not os.path.exists(so):
                return False
            try:
                R = C.CDLL(so)
            except OSError:
                return False
            R.ref_create.restype = C.c_void_p
            R.ref_destroy.argtypes = [C.c_void_p]
            R.ref_reset.argtypes = [C.c_void_p]
            R.ref_process.argtypes = [C.c_void_p, C.POINTER(Params), _u8p, C.c_int, C.c_int, C.c_int, C.c_ssize_t, _u8p, C.c_ssize_t,
                                      C.POINTER(C.c_int)]
            cls._lib = R
        return True

    @classmethod
    def recover_lab_lut(cls):
        """The forward Lab table of the OpenCV build behind the real reference (None: not built / not interpolating)."""
        if not cls.available() or not hasattr(cls._lib, "ref_recover_lab_lut"):
            return None
        t = np.empty(33 * 33 * 33 * 3, np.int16)
        return t if cls._lib.ref_recover_lab_lut(t.ctypes.data_as(C.c_void_p)) == 0 else None

    def __init__(self):
        if not self.available():
            raise RuntimeError("oracle/_ref/libref_magnify.so is not built (no OpenCV 4 on the build host)")
        self._c = self._lib.ref_create()

    def close(self):
        if getattr(self, "_c", None):
            self._lib.ref_destroy(self._c)
            self._c = None

    __del__ = close

    def reset(self):
        self._lib.ref_reset(self._c)

    def process(self, frame, params):
        frame = np.ascontiguousarray(frame)
        h, w = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        out = np.empty_like(frame)
        produced = C.c_int(0)
        if self._lib.ref_process(self._c, C.byref(params), frame.reshape(-1), w, h, ch, w * ch, out.reshape(-1), w * ch, C.byref(produced)) != 0:
            raise RuntimeError("the reference stage threw")
        return (out, True) if produced.value else (frame, False)
