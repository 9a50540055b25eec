"""ctypes binding of the C ABI (include/lvm_hip.h) and a Python mirror of the reference's
operator surface for this path:

    reference                                              here
    -----------------------------------------------------  ---------------------------------
    MagnificationMode   (processing/IProcessor.hpp:10)      MagnificationMode
    MagnificationParams (processing/IProcessor.hpp:14-23)   MagnificationParams
    PreprocessParams    (processing/IProcessor.hpp:26-41)   PreprocessParams
    ProcessorConfig     (processing/IProcessor.hpp:44-48)   ProcessorConfig
    MagnificationProcessor::process / reset                 MagnificationProcessor.process / reset
      (processing/MagnificationProcessor.cpp:10-67)

The library is the hipcc/gfx950 build of csrc/ (liblvm_hip.so next to this file).  There is no
CPU fallback: if the library or a GPU is missing, loading / creating a processor raises.
"""
import ctypes as C
import dataclasses
import enum
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LVM_HIP_LIB") or os.path.join(_HERE, "liblvm_hip.so")   # (override: A/B builds of the same sources)


class MagnificationMode(enum.IntEnum):
    Laplace = 0
    Phase = 1
    Color = 2
    None_ = 3


@dataclasses.dataclass
class MagnificationParams:
    mode: int = MagnificationMode.Laplace
    amplification: float = 0.0
    coWavelength: float = 0.0
    coLow: float = 0.0
    coHigh: float = 0.0
    chromAttenuation: float = 0.0
    levels: int = 4
    framerate: float = 30.0


@dataclasses.dataclass
class PreprocessParams:
    downscale: int = 1
    roiEnabled: bool = False
    roiX: float = 0.0
    roiY: float = 0.0
    roiW: float = 1.0
    roiH: float = 1.0

    def key(self):
        """64-bit value that changes iff the struct changes (operator==, IProcessor.hpp:36-39)."""
        raw = struct.pack("<i?ffff", self.downscale, self.roiEnabled, self.roiX, self.roiY, self.roiW, self.roiH)
        h = 0xcbf29ce484222325
        for b in raw:
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        return 0 if self == PreprocessParams() else (h or 1)


@dataclasses.dataclass
class ProcessorConfig:
    grayscale: bool = False
    preprocess: PreprocessParams = dataclasses.field(default_factory=PreprocessParams)
    magnification: MagnificationParams = dataclasses.field(default_factory=MagnificationParams)


LAB_LUT_ENTRIES = 33 * 33 * 33 * 3


class LvmParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("levels", C.c_int32), ("amplification", C.c_double),
                ("coWavelength", C.c_double), ("coLow", C.c_double), ("coHigh", C.c_double),
                ("chromAttenuation", C.c_double), ("framerate", C.c_double),
                ("preprocess_key", C.c_uint64)]


class LvmOverlayLabel(C.Structure):
    """lvm_overlay_label (include/lvm_hip.h): one caption of the export's text overlay as per-pixel tables"""
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("n_classes", C.c_int32),
                ("cls", C.c_void_p), ("fn", C.c_void_p)]


class LvmPreprocessParams(C.Structure):
    """lvm_preprocess_params (include/lvm_hip.h): PreprocessParams + ProcessorConfig::grayscale."""
    _fields_ = [("downscale", C.c_int32), ("roi_enabled", C.c_int32), ("roiX", C.c_float), ("roiY", C.c_float),
                ("roiW", C.c_float), ("roiH", C.c_float), ("grayscale", C.c_int32)]


def to_c_preprocess(pre, grayscale=False):
    return LvmPreprocessParams(int(pre.downscale), 1 if pre.roiEnabled else 0, pre.roiX, pre.roiY, pre.roiW, pre.roiH,
                               1 if grayscale else 0)


class LvmError(RuntimeError):
    pass


SYMBOLS = ["lvm_create", "lvm_destroy", "lvm_reset", "lvm_process", "lvm_process_device", "lvm_process_device_frames", "lvm_set_pipeline", "lvm_flush", "lvm_synchronize",
           "lvm_last_error", "lvm_max_levels", "lvm_optimal_buffer_size", "lvm_butterworth2",
           "lvm_debug_keep_float", "lvm_debug_read_float", "lvm_debug_exact_lab", "lvm_debug_sweep_u8_steps", "lvm_debug_clock_probe_start", "lvm_debug_clock_probe_stop", "lvm_debug_lab_analytic", "lvm_get_lab_lut", "lvm_set_lab_lut", "lvm_profile_enable", "lvm_profile_collect", "lvm_profile_only",
           "lvm_profile_entry", "lvm_algorithmic_bytes", "lvm_export_geometry", "lvm_export_frames", "lvm_export_set_overlay", "lvm_overlay_device", "lvm_tile_riesz_stage1", "lvm_tile_riesz_planes", "lvm_tile_riesz_stage2",
           "lvm_preprocess_geometry", "lvm_preprocess_device", "lvm_chain_process", "lvm_chain_process_batch",
           "lvm_set_max_frames", "lvm_host_alloc", "lvm_host_free", "lvm_compose_geometry", "lvm_compose_device", "lvm_chain_process_batch_ex",
           "lvm_chain_present", "lvm_mjpeg_bound", "lvm_mjpeg_encode_device", "lvm_export_frames_mjpeg", "lvm_mjpeg_decode_device", "lvm_export_mjpeg_frames", "lvm_mjpeg_set_restart_interval"]


def bind(lib):
    """Attach argtypes/restypes of include/lvm_hip.h to a loaded CDLL."""
    vp = C.c_void_p
    lib.lvm_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    lib.lvm_destroy.argtypes = [vp]
    lib.lvm_destroy.restype = None
    lib.lvm_reset.argtypes = [vp]
    lib.lvm_process.argtypes = [vp, C.POINTER(LvmParams), vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t,
                                vp, C.c_ssize_t, C.POINTER(C.c_int)]
    lib.lvm_process_device.argtypes = [vp, C.POINTER(LvmParams), vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t,
                                       C.c_ssize_t, vp, C.c_ssize_t, C.c_ssize_t, C.POINTER(C.c_int), vp]
    lib.lvm_synchronize.argtypes = [vp]
    lib.lvm_process_device_frames.argtypes = [vp, C.POINTER(LvmParams), C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t,
                                              C.c_ssize_t, C.c_ssize_t, vp, C.c_ssize_t, C.c_ssize_t, C.c_ssize_t,
                                              C.POINTER(C.c_int), vp]
    lib.lvm_set_pipeline.argtypes = [vp, C.c_int]
    lib.lvm_flush.argtypes = [vp, vp]
    lib.lvm_last_error.argtypes = [vp]
    lib.lvm_last_error.restype = C.c_char_p
    lib.lvm_max_levels.argtypes = [C.c_int, C.c_int]
    lib.lvm_optimal_buffer_size.argtypes = [C.c_int]
    lib.lvm_butterworth2.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.lvm_butterworth2.restype = None
    lib.lvm_debug_keep_float.argtypes = [vp, C.c_int]
    lib.lvm_debug_read_float.argtypes = [vp, vp, C.c_size_t]
    lib.lvm_debug_exact_lab.argtypes = [vp, C.c_int]
    lib.lvm_debug_lab_analytic.argtypes = [vp, C.c_int]
    lib.lvm_debug_clock_probe_start.argtypes = [vp, C.c_double]
    lib.lvm_debug_clock_probe_stop.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.lvm_debug_sweep_u8_steps.argtypes = [vp, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    lib.lvm_get_lab_lut.argtypes = [vp, vp]
    lib.lvm_set_lab_lut.argtypes = [vp, vp]
    lib.lvm_profile_enable.argtypes = [vp, C.c_int]
    lib.lvm_profile_collect.argtypes = [vp]
    lib.lvm_profile_only.argtypes = [vp, C.c_char_p]
    lib.lvm_profile_entry.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double),
                                      C.POINTER(C.c_longlong)]
    lib.lvm_algorithmic_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
    lib.lvm_algorithmic_bytes.restype = C.c_double
    ip = C.POINTER(C.c_int)
    lib.lvm_preprocess_geometry.argtypes = [C.POINTER(LvmPreprocessParams), C.c_int, C.c_int, C.c_int, ip, ip, ip, ip, ip, ip, ip]
    lib.lvm_preprocess_device.argtypes = [vp, C.POINTER(LvmPreprocessParams), vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t,
                                          C.c_ssize_t, vp, C.c_ssize_t, C.c_ssize_t, vp]
    lib.lvm_chain_process.argtypes = [vp, C.POINTER(LvmPreprocessParams), C.POINTER(LvmParams), vp, C.c_int, C.c_int, C.c_int,
                                      C.c_ssize_t, vp, C.c_ssize_t, ip]
    lib.lvm_chain_process_batch.argtypes = [vp, C.POINTER(LvmPreprocessParams), C.POINTER(LvmParams), C.POINTER(vp), C.c_int, C.c_int,
                                            C.c_int, C.c_ssize_t, C.POINTER(vp), C.c_ssize_t, ip]
    lib.lvm_chain_process_batch_ex.argtypes = [vp, C.POINTER(LvmPreprocessParams), C.POINTER(LvmParams), C.POINTER(vp), C.c_int, C.c_int,
                                               C.c_int, C.c_ssize_t, C.POINTER(vp), C.c_ssize_t, C.POINTER(vp), C.c_ssize_t, ip]
    lib.lvm_chain_present.argtypes = [vp, C.POINTER(LvmPreprocessParams), C.POINTER(LvmParams), vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t,
                                      vp, C.c_ssize_t, vp, C.c_ssize_t, ip]
    lib.lvm_export_geometry.argtypes = [C.POINTER(LvmPreprocessParams), C.c_int, C.c_int, C.c_int, C.c_int, ip, ip]
    lib.lvm_export_frames.argtypes = [vp, C.POINTER(LvmPreprocessParams), C.POINTER(LvmParams), C.c_int, C.c_int, C.POINTER(vp), C.c_int, C.c_int,
                                      C.c_int, C.c_ssize_t, C.POINTER(vp), C.c_ssize_t, ip]
    lib.lvm_mjpeg_bound.argtypes = [C.c_int, C.c_int]
    lib.lvm_mjpeg_bound.restype = C.c_size_t
    lib.lvm_mjpeg_encode_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.lvm_mjpeg_set_restart_interval.argtypes = [vp, C.c_int]
    lib.lvm_mjpeg_decode_device.argtypes = [vp, vp, C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, vp, C.c_ssize_t, C.c_ssize_t]
    lib.lvm_export_mjpeg_frames.argtypes = [vp, C.POINTER(LvmPreprocessParams), C.POINTER(LvmParams), C.c_int, C.c_int, vp, C.POINTER(C.c_size_t), C.c_int, C.c_int,
                                            C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t), ip]
    lib.lvm_export_frames_mjpeg.argtypes = [vp, C.POINTER(LvmPreprocessParams), C.POINTER(LvmParams), C.c_int, C.c_int, C.POINTER(vp), C.c_int, C.c_int,
                                            C.c_int, C.c_ssize_t, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t), ip]
    lib.lvm_compose_geometry.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, ip, ip, ip]
    lib.lvm_compose_device.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, vp, C.c_int, C.c_int, C.c_int,
                                       C.c_ssize_t, C.c_ssize_t, vp, C.c_ssize_t, C.c_ssize_t, vp]
    lib.lvm_set_max_frames.argtypes = [vp, C.c_int]
    lib.lvm_tile_riesz_stage1.argtypes = [vp, C.POINTER(LvmParams), vp, C.c_int, C.c_int, C.c_ssize_t, ip, vp, ip, ip, vp]
    lib.lvm_tile_riesz_planes.argtypes = [vp, C.POINTER(LvmParams), vp, C.c_int, C.c_int, vp, ip, vp]
    lib.lvm_tile_riesz_stage2.argtypes = [vp, C.POINTER(LvmParams), vp, C.c_int, C.c_int, C.c_ssize_t, vp, vp, C.c_ssize_t, vp]
    lib.lvm_export_set_overlay.argtypes = [vp, C.c_int, C.POINTER(LvmOverlayLabel)]
    lib.lvm_overlay_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int, vp]
    lib.lvm_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.lvm_host_free.argtypes = [vp]
    lib.lvm_host_free.restype = None
    return lib


_lib = None


def load():
    """Load liblvm_hip.so (the gfx950 build).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LvmError("liblvm_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        try:
            import torch  # noqa: F401  -- share torch's HIP runtime instance when torch is in the process
        except Exception:
            pass
        _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def to_c_params(p, preprocess_key=0):
    return LvmParams(int(p.mode), int(p.levels), float(p.amplification), float(p.coWavelength), float(p.coLow),
                     float(p.coHigh), float(p.chromAttenuation), float(p.framerate), int(preprocess_key))


class Context:
    """Thin RAII wrapper over lvm_ctx."""

    def __init__(self, device=0, n_streams=1, lib=None):
        self.lib = lib if lib is not None else load()
        h = C.c_void_p()
        rc = self.lib.lvm_create(device, n_streams, C.byref(h))
        if rc != 0:
            raise LvmError("lvm_create failed (%d): no usable HIP device / out of memory; no CPU fallback" % rc)
        self.h = h
        self.n_streams = n_streams

    def close(self):
        if getattr(self, "h", None):
            self.lib.lvm_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise LvmError("lvm error %d: %s" % (rc, self.lib.lvm_last_error(self.h).decode()))

    def reset(self):
        self._check(self.lib.lvm_reset(self.h))

    def preprocess_geometry(self, cpre, w, h, ch):
        """(roi_x, roi_y, roi_w, roi_h, out_w, out_h, out_channels) of the two stages in front of the magnifier."""
        v = [C.c_int() for _ in range(7)]
        rc = self.lib.lvm_preprocess_geometry(C.byref(cpre), w, h, ch, *[C.byref(x) for x in v])
        if rc != 0:
            raise LvmError("lvm_preprocess_geometry: invalid arguments")
        return tuple(x.value for x in v)

    def preprocess_device(self, cpre, d_in, w, h, ch, in_stride, in_sstride, d_out, out_stride, out_sstride, stream=None):
        self._check(self.lib.lvm_preprocess_device(self.h, C.byref(cpre), d_in, w, h, ch, in_stride, in_sstride, d_out, out_stride,
                                                   out_sstride, stream))

    def chain_process_batch(self, frames, cpre, cparams):
        """lvm_chain_process_batch: one host frame per stream of this context (same geometry); returns (outs, produced)."""
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        if len(frames) != self.n_streams:
            raise LvmError("chain_process_batch needs one frame per stream")
        h, w = frames[0].shape[:2]
        ch = 1 if frames[0].ndim == 2 else frames[0].shape[2]
        _, _, _, _, ow, oh, och = self.preprocess_geometry(cpre, w, h, ch)
        outs = [np.empty((oh, ow) if och == 1 else (oh, ow, och), dtype=np.uint8) for _ in frames]
        pin = (C.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        pout = (C.c_void_p * len(frames))(*[o.ctypes.data for o in outs])
        produced = C.c_int(0)
        self._check(self.lib.lvm_chain_process_batch(self.h, C.byref(cpre), C.byref(cparams), pin, w, h, ch, w * ch, pout, ow * och,
                                                     C.byref(produced)))
        return outs, bool(produced.value)

    def chain_process_batch_ex(self, frames, cpre, cparams):
        """lvm_chain_process_batch_ex: the batch call plus runChainOnce's `original` tap (PreprocessProcessor's output, before
        GrayscaleProcessor: the source's channel count).  Returns (outs, originals, produced)."""
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        if len(frames) != self.n_streams:
            raise LvmError("chain_process_batch_ex needs one frame per stream")
        h, w = frames[0].shape[:2]
        ch = 1 if frames[0].ndim == 2 else frames[0].shape[2]
        _, _, _, _, ow, oh, och = self.preprocess_geometry(cpre, w, h, ch)
        outs = [np.empty((oh, ow) if och == 1 else (oh, ow, och), dtype=np.uint8) for _ in frames]
        taps = [np.empty((oh, ow) if ch == 1 else (oh, ow, ch), dtype=np.uint8) for _ in frames]
        pin = (C.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        pout = (C.c_void_p * len(frames))(*[o.ctypes.data for o in outs])
        ptap = (C.c_void_p * len(frames))(*[o.ctypes.data for o in taps])
        produced = C.c_int(0)
        self._check(self.lib.lvm_chain_process_batch_ex(self.h, C.byref(cpre), C.byref(cparams), pin, w, h, ch, w * ch, pout, ow * och,
                                                        ptap, ow * ch, C.byref(produced)))
        return outs, taps, bool(produced.value)

    def chain_present(self, frame, cpre, cparams, d_proc, proc_stride, d_orig, orig_stride):
        """lvm_chain_present: host frame in, processed frame and `original` tap left in DEVICE buffers (addresses).  Returns produced."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        produced = C.c_int(0)
        self._check(self.lib.lvm_chain_present(self.h, C.byref(cpre), C.byref(cparams), frame.ctypes.data, w, h, ch, w * ch,
                                               d_proc, proc_stride, d_orig, orig_stride, C.byref(produced)))
        return bool(produced.value)

    def process_pinned(self, frame, cparams, pad=0):
        """lvm_process on PAGE-LOCKED frames (lvm_host_alloc): the zero-copy surface.  `pad` extra bytes per row on both sides."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        stride = w * ch + pad
        pin, pout = C.c_void_p(), C.c_void_p()
        self._check(self.lib.lvm_host_alloc(stride * h, C.byref(pin)))
        self._check(self.lib.lvm_host_alloc(stride * h, C.byref(pout)))
        try:
            src = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(h, stride))
            dst = np.ctypeslib.as_array(C.cast(pout, C.POINTER(C.c_uint8)), shape=(h, stride))
            src[:, :w * ch] = frame.reshape(h, w * ch)
            dst[:] = 0xA5
            produced = C.c_int(0)
            self._check(self.lib.lvm_process(self.h, C.byref(cparams), pin.value, w, h, ch, stride, pout.value, stride, C.byref(produced)))
            out = dst[:, :w * ch].reshape(frame.shape).copy()
            untouched = bool((dst[:, w * ch:] == 0xA5).all())
            if not untouched:
                raise LvmError("lvm_process wrote outside the output rows")
        finally:
            self.lib.lvm_host_free(pin); self.lib.lvm_host_free(pout)
        return (out, True) if produced.value else (frame, False)

    def export_frames(self, frames, cpre, cparams, split):
        """lvm_export_frames: Exporter::run's loop body (runChainOnce + Exporter::compose) for consecutive host frames of a 1-stream
        context.  Returns (canvases, produced flags)."""
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        h, w = frames[0].shape[:2]
        ch = 1 if frames[0].ndim == 2 else frames[0].shape[2]
        cw, chh = C.c_int(0), C.c_int(0)
        self._check(self.lib.lvm_export_geometry(C.byref(cpre), int(split), w, h, ch, C.byref(cw), C.byref(chh)))
        if cw.value <= 0 or chh.value <= 0:
            raise LvmError("export_frames: empty canvas for this geometry (Exporter::compose returns an empty Mat)")
        canvases = [np.empty((chh.value, cw.value, 3), dtype=np.uint8) for _ in frames]
        pin = (C.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        pout = (C.c_void_p * len(frames))(*[o.ctypes.data for o in canvases])
        produced = (C.c_int * len(frames))()
        self._check(self.lib.lvm_export_frames(self.h, C.byref(cpre), C.byref(cparams), int(split), len(frames), pin, w, h, ch, w * ch,
                                               pout, cw.value * 3, produced))
        return canvases, [bool(x) for x in produced]

    def export_set_overlay(self, labels):
        """lvm_export_set_overlay: labels = [(x, y, cls uint16 [h][w], fn uint8 [n_classes][256]), ...] (at most 4; [] switches the overlay
        off).  The tables are copied into device memory; every later export call of this context applies them to its canvases."""
        arr = (LvmOverlayLabel * max(len(labels), 1))()
        keep = []
        for i, (x, y, cls, fn) in enumerate(labels):
            cls = np.ascontiguousarray(cls, dtype=np.uint16)
            fn = np.ascontiguousarray(fn, dtype=np.uint8)
            assert cls.ndim == 2 and fn.ndim == 2 and fn.shape[1] == 256
            keep += [cls, fn]
            arr[i] = LvmOverlayLabel(int(x), int(y), cls.shape[1], cls.shape[0], fn.shape[0], cls.ctypes.data, fn.ctypes.data)
        self._check(self.lib.lvm_export_set_overlay(self.h, len(labels), arr))

    def overlay_device(self, d_ptr, cw, chh, n_frames=1, stride=None, frame_stride=None, stream=0):
        """lvm_overlay_device: the labels onto device-resident canvases (address)."""
        stride = cw * 3 if stride is None else stride
        frame_stride = stride * chh if frame_stride is None else frame_stride
        self._check(self.lib.lvm_overlay_device(self.h, d_ptr, cw, chh, stride, frame_stride, n_frames, stream))

    def mjpeg_encode_device(self, d_ptr, w, h, n_frames, quality=75, stride=None, frame_stride=None, capacity=None):
        """lvm_mjpeg_encode_device: device-resident BGR frames (address) -> list of JPEG frames (bytes)."""
        stride = w * 3 if stride is None else stride
        frame_stride = stride * h if frame_stride is None else frame_stride
        cap = int(self.lib.lvm_mjpeg_bound(w, h)) * n_frames if capacity is None else int(capacity)
        out = np.empty(cap, dtype=np.uint8)
        offs = (C.c_size_t * (n_frames + 1))()
        self._check(self.lib.lvm_mjpeg_encode_device(self.h, d_ptr, w, h, stride, frame_stride, n_frames, int(quality), out.ctypes.data, cap, offs))
        return [out[offs[i]:offs[i + 1]].tobytes() for i in range(n_frames)]

    def mjpeg_set_restart_interval(self, mcus):
        self._check(self.lib.lvm_mjpeg_set_restart_interval(self.h, int(mcus)))

    def mjpeg_decode_device(self, jpegs, w, h, d_ptr, stride=None, frame_stride=None):
        """lvm_mjpeg_decode_device: a list of JPEG frames (bytes) -> BGR frames in device memory at d_ptr."""
        stride = w * 3 if stride is None else stride
        frame_stride = stride * h if frame_stride is None else frame_stride
        blob = np.frombuffer(b"".join(jpegs), dtype=np.uint8)
        offs = (C.c_size_t * (len(jpegs) + 1))(*np.concatenate([[0], np.cumsum([len(j) for j in jpegs])]).tolist())
        self._check(self.lib.lvm_mjpeg_decode_device(self.h, blob.ctypes.data, offs, len(jpegs), w, h, d_ptr, stride, frame_stride))

    def export_mjpeg_frames(self, jpegs, w, h, cpre, cparams, split, quality=75, capacity=None):
        """lvm_export_mjpeg_frames: JPEG frames in, JPEG frames (of the composed canvases) out.  Returns (JPEG frames, produced flags)."""
        cw, chh = C.c_int(0), C.c_int(0)
        self._check(self.lib.lvm_export_geometry(C.byref(cpre), int(split), w, h, 3, C.byref(cw), C.byref(chh)))
        if cw.value <= 0 or chh.value <= 0:
            raise LvmError("export_mjpeg_frames: empty canvas for this geometry")
        n = len(jpegs)
        blob = np.frombuffer(b"".join(jpegs), dtype=np.uint8)
        ioffs = (C.c_size_t * (n + 1))(*np.concatenate([[0], np.cumsum([len(j) for j in jpegs])]).tolist())
        cap = int(self.lib.lvm_mjpeg_bound(cw.value, chh.value)) * n if capacity is None else int(capacity)
        out = np.empty(cap, dtype=np.uint8)
        offs = (C.c_size_t * (n + 1))()
        produced = (C.c_int * n)()
        self._check(self.lib.lvm_export_mjpeg_frames(self.h, C.byref(cpre), C.byref(cparams), int(split), n, blob.ctypes.data, ioffs, w, h, int(quality),
                                                     out.ctypes.data, cap, offs, produced))
        return [out[offs[i]:offs[i + 1]].tobytes() for i in range(n)], [bool(x) for x in produced]

    def export_frames_mjpeg(self, frames, cpre, cparams, split, quality=75, capacity=None):
        """lvm_export_frames_mjpeg: Exporter::run's loop body with the canvases encoded on the device.  Returns (JPEG frames, produced flags)."""
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        h, w = frames[0].shape[:2]
        ch = 1 if frames[0].ndim == 2 else frames[0].shape[2]
        cw, chh = C.c_int(0), C.c_int(0)
        self._check(self.lib.lvm_export_geometry(C.byref(cpre), int(split), w, h, ch, C.byref(cw), C.byref(chh)))
        if cw.value <= 0 or chh.value <= 0:
            raise LvmError("export_frames_mjpeg: empty canvas for this geometry (Exporter::compose returns an empty Mat)")
        n = len(frames)
        cap = int(self.lib.lvm_mjpeg_bound(cw.value, chh.value)) * n if capacity is None else int(capacity)
        out = np.empty(cap, dtype=np.uint8)
        offs = (C.c_size_t * (n + 1))()
        pin = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        produced = (C.c_int * n)()
        self._check(self.lib.lvm_export_frames_mjpeg(self.h, C.byref(cpre), C.byref(cparams), int(split), n, pin, w, h, ch, w * ch, int(quality),
                                                     out.ctypes.data, cap, offs, produced))
        return [out[offs[i]:offs[i + 1]].tobytes() for i in range(n)], [bool(x) for x in produced]

    def chain_process(self, frame, cpre, cparams):
        """Preprocess -> Grayscale -> Magnification on a host frame (lvm_chain_process).  Returns (out, produced);
        `out` is the magnified frame, or the preprocessed frame on passthrough."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        _, _, _, _, ow, oh, och = self.preprocess_geometry(cpre, w, h, ch)
        out = np.empty((oh, ow) if och == 1 else (oh, ow, och), dtype=np.uint8)
        produced = C.c_int(0)
        self._check(self.lib.lvm_chain_process(self.h, C.byref(cpre), C.byref(cparams), frame.ctypes.data, w, h, ch, w * ch,
                                               out.ctypes.data, ow * och, C.byref(produced)))
        return out, bool(produced.value)

    def process(self, frame, cparams):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w = frame.shape[:2]
        ch = 1 if frame.ndim == 2 else frame.shape[2]
        out = np.empty_like(frame)
        produced = C.c_int(0)
        self._check(self.lib.lvm_process(self.h, C.byref(cparams), frame.ctypes.data, w, h, ch, w * ch,
                                         out.ctypes.data, w * ch, C.byref(produced)))
        return (out, True) if produced.value else (frame, False)

    def process_device(self, cparams, d_in, w, h, ch, in_stride, in_sstride, d_out, out_stride, out_sstride,
                       stream=0):
        produced = C.c_int(0)
        self._check(self.lib.lvm_process_device(self.h, C.byref(cparams), d_in, w, h, ch, in_stride, in_sstride,
                                                d_out, out_stride, out_sstride, C.byref(produced), stream))
        return bool(produced.value)

    def process_device_frames(self, cparams, n_frames, d_in, w, h, ch, in_stride, in_sstride, in_fstride, d_out,
                              out_stride, out_sstride, out_fstride, stream=0):
        produced = (C.c_int * n_frames)()
        self._check(self.lib.lvm_process_device_frames(self.h, C.byref(cparams), n_frames, d_in, w, h, ch, in_stride, in_sstride,
                                                       in_fstride, d_out, out_stride, out_sstride, out_fstride, produced, stream))
        return [bool(x) for x in produced]

    def compose_geometry(self, split, ow, oh, pw, ph):
        """(pane_w, pane_h, canvas_w, canvas_h) of Exporter::compose; zeros = the reference's empty Mat."""
        v = [C.c_int() for _ in range(4)]
        if self.lib.lvm_compose_geometry(int(split), ow, oh, pw, ph, *[C.byref(x) for x in v]) != 0:
            raise LvmError("lvm_compose_geometry: invalid split mode")
        return tuple(x.value for x in v)

    def compose_device(self, split, d_orig, ow, oh, och, ostride, osstride, d_proc, pw, ph, pch, pstride, psstride, d_canvas, cstride,
                       csstride, stream=None):
        self._check(self.lib.lvm_compose_device(self.h, int(split), d_orig, ow, oh, och, ostride, osstride, d_proc, pw, ph, pch, pstride,
                                                psstride, d_canvas, cstride, csstride, stream))

    def set_max_frames(self, n):
        self._check(self.lib.lvm_set_max_frames(self.h, int(n)))

    def set_pipeline(self, depth):
        self._check(self.lib.lvm_set_pipeline(self.h, int(depth)))

    def flush(self, stream=0):
        self._check(self.lib.lvm_flush(self.h, stream))

    def make_stepper(self, cparams, w, h, ch, in_stride, in_sstride, out_stride, out_sstride, stream=0):
        """Returns f(d_in_ptr, d_out_ptr) -> rc with every constant argument pre-converted (the per-call
        Python overhead of the generic wrapper is comparable to a whole frame's GPU time)."""
        fn = self.lib.lvm_process_device
        h_ctx, p_ref, produced = self.h, C.byref(cparams), C.c_int(0)
        p_prod = C.byref(produced)
        cw, chh, cch = C.c_int(w), C.c_int(h), C.c_int(ch)
        a, b2, c2, d = C.c_ssize_t(in_stride), C.c_ssize_t(in_sstride), C.c_ssize_t(out_stride), C.c_ssize_t(out_sstride)
        st = C.c_void_p(stream)
        self._keep = (cparams, produced)

        def step(d_in, d_out):
            return fn(h_ctx, p_ref, d_in, cw, chh, cch, a, b2, d_out, c2, d, p_prod, st)
        return step

    def synchronize(self):
        self._check(self.lib.lvm_synchronize(self.h))

    def keep_float(self, on=True):
        self._check(self.lib.lvm_debug_keep_float(self.h, int(on)))

    def read_float(self, shape):
        a = np.empty(shape, np.float32)
        self._check(self.lib.lvm_debug_read_float(self.h, a.ctypes.data, a.size))
        return a

    def exact_lab(self, on=True):
        self._check(self.lib.lvm_debug_exact_lab(self.h, int(on)))

    def clock_probe_start(self, max_seconds=2.0):
        """starts the shader-clock probe beside the work on the other streams (lvm_debug_clock_probe_start)"""
        self._check(self.lib.lvm_debug_clock_probe_start(self.h, float(max_seconds)))

    def clock_probe_stop(self):
        """-> (average shader clock in MHz, seconds covered)"""
        mhz, sec = C.c_double(0), C.c_double(0)
        self._check(self.lib.lvm_debug_clock_probe_stop(self.h, C.byref(mhz), C.byref(sec)))
        return float(mhz.value), float(sec.value)

    def sweep_u8_steps(self, first_bits=0, count=1 << 32):
        """(mismatches, first bad bit pattern) of the u8 step table against spline + scale + round over the floats with bit patterns
        first_bits .. first_bits + count - 1 (lvm_debug_sweep_u8_steps)."""
        bad, fb = C.c_uint64(0), C.c_uint32(0)
        self._check(self.lib.lvm_debug_sweep_u8_steps(self.h, first_bits, count, C.byref(bad), C.byref(fb)))
        return int(bad.value), int(fb.value)

    def lab_analytic(self, on=True):
        """Analytic forward Lab (OpenCV with its interpolation switched off) instead of the 33^3 table."""
        self._check(self.lib.lvm_debug_lab_analytic(self.h, int(on)))

    def lab_lut(self):
        """The forward table in use: int16 [33*33*33*3], index 3 (p + 33 q + 1089 r) + channel."""
        a = np.empty(LAB_LUT_ENTRIES, np.int16)
        self._check(self.lib.lvm_get_lab_lut(self.h, a.ctypes.data))
        return a

    def set_lab_lut(self, table):
        a = np.ascontiguousarray(table, np.int16).reshape(-1)
        assert a.size == LAB_LUT_ENTRIES
        self._check(self.lib.lvm_set_lab_lut(self.h, a.ctypes.data))

    def profile(self, on=True):
        self._check(self.lib.lvm_profile_enable(self.h, int(on)))

    def profile_only(self, name=None):
        """Clear the totals and bracket only launches with this report name (None: all)."""
        self._check(self.lib.lvm_profile_only(self.h, name.encode() if name else None))

    def profile_collect(self):
        n = self.lib.lvm_profile_collect(self.h)
        out = {}
        for i in range(max(n, 0)):
            name = C.create_string_buffer(64)
            ms = C.c_double()
            cnt = C.c_longlong()
            self.lib.lvm_profile_entry(self.h, i, name, 64, C.byref(ms), C.byref(cnt))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out


class MagnificationProcessor:
    """Mirror of the reference stage (MagnificationProcessor.hpp:13-23): process(frame, cfg)
    returns the magnified frame, or the input frame itself on passthrough; reset() drops all
    temporal state.  `frame` is an HxWx3 (BGR) or HxW (gray) uint8 array."""

    def __init__(self, device=0, lib=None):
        self.ctx = Context(device, 1, lib)

    def process(self, frame, cfg):
        cp = to_c_params(cfg.magnification, cfg.preprocess.key())
        out, _ = self.ctx.process(frame, cp)
        return out

    def process_ex(self, frame, cfg):
        cp = to_c_params(cfg.magnification, cfg.preprocess.key())
        return self.ctx.process(frame, cp)

    def reset(self):
        self.ctx.reset()


class ProcessingChain:
    """Mirror of runChainOnce (processing/ChainBuilder.cpp:19-29) for the three per-frame stages
    PreprocessProcessor -> GrayscaleProcessor -> MagnificationProcessor, all on the device: process(frame, cfg)
    returns what the reference chain hands to the display (the magnified frame, or the preprocessed frame
    when the magnifier passes its input through)."""

    def __init__(self, device=0, lib=None):
        self.ctx = Context(device, 1, lib)

    def process(self, frame, cfg):
        out, _ = self.process_ex(frame, cfg)
        return out

    def process_ex(self, frame, cfg):
        return self.ctx.chain_process(frame, to_c_preprocess(cfg.preprocess, cfg.grayscale), to_c_params(cfg.magnification, 0))

    def reset(self):
        self.ctx.reset()
