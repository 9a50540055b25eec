"""Deterministic synthetic clips and per-config parameters.

The reference ships no sample clips; benchmarks and parity tests use these generators:
a static texture (never flat/black, so the Riesz quotients stay away from 0/0), sampled
with a sub-pixel horizontal sine motion, plus an optional global colour pulse.
"""
import math

import numpy as np

MODE_LAPLACE, MODE_PHASE, MODE_COLOR, MODE_NONE = 0, 1, 2, 3


def motion_hz_to_blend(hz, fps):
    """reference: src/processing/MagnificationParamsUi.hpp:29-34."""
    if fps <= 0.0:
        fps = 30.0
    if hz <= 0.0:
        return 0.0
    a = 1.0 - math.exp(-6.283185307179586 * hz / fps)
    return min(max(a, 0.0), 0.999999)


def texture(w, h, seed=1234, pad=2, noise=12.0):
    """float64 texture (h, w+2*pad, 3) in [4, 251]: two sine gratings + uniform per-channel noise of +- `noise` levels (12: far
    above a camera's -- neighbouring pixels land in different cells of the Lab table, the worst case for its gathers)."""
    rng = np.random.default_rng(seed)
    x = np.arange(-pad, w + pad, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    base = 96.0 + 48.0 * np.sin(2 * np.pi * (x / 37.0 + y / 53.0)) + 32.0 * np.sin(2 * np.pi * (x / 11.0 - y / 7.0))
    noise = rng.uniform(-noise, noise, size=(h, w + 2 * pad, 3))
    return np.clip(base[:, :, None] + noise, 4.0, 251.0)


class Clip:
    """Frame t = texture sampled at (x + A sin(2 pi f_m t / fps), y) (bilinear) + colour pulse."""

    def __init__(self, w, h, fps=30.0, f_motion=1.5, amp_px=0.5, f_color=0.0, amp_color=0.0,
                 seed=1234, channels=3, noise=12.0):
        self.w, self.h, self.fps = w, h, fps
        self.f_motion, self.amp_px = f_motion, amp_px
        self.f_color, self.amp_color = f_color, amp_color
        self.channels = channels
        self.pad = 2
        self.tex = texture(w, h, seed, self.pad, noise)

    def frame(self, t):
        d = self.amp_px * math.sin(2 * math.pi * self.f_motion * t / self.fps)
        i0 = math.floor(d)
        fr = d - i0
        a = self.tex[:, self.pad + i0: self.pad + i0 + self.w]
        b = self.tex[:, self.pad + i0 + 1: self.pad + i0 + 1 + self.w]
        img = (1.0 - fr) * a + fr * b
        if self.amp_color:
            img = img + self.amp_color * math.sin(2 * math.pi * self.f_color * t / self.fps)
        img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        if self.channels == 1:
            return np.ascontiguousarray(img[:, :, 1])
        return np.ascontiguousarray(img)

    def frame_torch(self, t, device):
        """frame(t) computed on `device` with the same float64 operations in the same order (IEEE: bit-identical to
        frame(t)); staging a 4K clip through numpy costs seconds per frame on the host."""
        import torch
        if getattr(self, "_tex_dev", None) is None or self._tex_dev.device != torch.device(device):
            self._tex_dev = torch.from_numpy(self.tex).to(device)
        d = self.amp_px * math.sin(2 * math.pi * self.f_motion * t / self.fps)
        i0 = math.floor(d)
        fr = d - i0
        a = self._tex_dev[:, self.pad + i0: self.pad + i0 + self.w]
        b = self._tex_dev[:, self.pad + i0 + 1: self.pad + i0 + 1 + self.w]
        img = (1.0 - fr) * a + fr * b
        if self.amp_color:
            img = img + self.amp_color * math.sin(2 * math.pi * self.f_color * t / self.fps)
        img = torch.clip(torch.round(img), 0, 255).to(torch.uint8)
        if self.channels == 1:
            return img[:, :, 1].contiguous()
        return img.contiguous()

    def frames(self, n, start=0):
        return np.stack([self.frame(t) for t in range(start, start + n)])


# BASELINE.json configs -> (clip kwargs, params kwargs).  cfg ids follow BASELINE.json order.
def config(idx, small=None):
    """small=(w,h,levels) overrides the geometry (parity tests run reduced sizes)."""
    if idx in (0, 1):
        w, h, lv = (640, 360, 4) if idx == 0 else (1920, 1080, 6)
        if small:
            w, h, lv = small
        clip = dict(w=w, h=h, fps=30.0, f_motion=1.5, amp_px=0.5)
        par = dict(mode=MODE_LAPLACE, levels=lv, amplification=20.0, coWavelength=500.0,
                   coLow=motion_hz_to_blend(0.4, 30.0), coHigh=motion_hz_to_blend(3.0, 30.0),
                   chromAttenuation=0.1, framerate=30.0)
    elif idx in (2, 4):
        w, h, lv = (1920, 1080, 6) if idx == 2 else (3840, 2160, 8)
        if small:
            w, h, lv = small
        clip = dict(w=w, h=h, fps=30.0, f_motion=2.0, amp_px=0.5)
        par = dict(mode=MODE_PHASE, levels=lv, amplification=50.0, coWavelength=50.0,
                   coLow=0.5, coHigh=10.0, chromAttenuation=0.0, framerate=30.0)
    elif idx == 3:
        w, h, lv = 1920, 1080, 6
        if small:
            w, h, lv = small
        clip = dict(w=w, h=h, fps=60.0, f_motion=0.0, amp_px=0.0, f_color=0.9, amp_color=2.0)
        par = dict(mode=MODE_COLOR, levels=lv, amplification=100.0, coWavelength=0.0,
                   coLow=0.83, coHigh=1.0, chromAttenuation=0.0, framerate=60.0)
    else:
        raise ValueError(idx)
    return clip, par
