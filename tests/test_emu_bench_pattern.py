"""The call pattern of `bench.py --width 320 --height 180 --levels 4 --frames-per-call 4 --ring 8` on the emulation build, whose
allocations are filled with 0xFF (NaN as float): a context with lvm_set_max_frames(T), calls of T consecutive frames from the very
first frame on, the ring wrapping, and one call with the float frame kept at the end.  On the GPU box one in four full-suite runs had a
rank of this configuration turn entirely NaN (recycled device memory is not zero; a fresh box's is): every frame must be finite and
match the oracle here."""
import ctypes as C

import numpy as np
import pytest

from helpers import c_params


@pytest.mark.parametrize("size,T,ring,nframes", [((320, 180, 4), 4, 8, 24), ((320, 180, 4), 3, 6, 18), ((162, 90, 4), 4, 8, 16)])
def test_emu_laplace_batches_from_the_first_frame_on_poisoned_memory(lvm, po, emu, size, T, ring, nframes):
    w, h, levels = size
    ck, pk = lvm.synth.config(1, size)
    clip = lvm.synth.Clip(**ck)
    frames = np.stack([clip.frame(t) for t in range(ring)])
    fb = w * h * 3
    out = np.full((nframes, h, w, 3), 0xAB, np.uint8)
    ctx = lvm.Context(0, 1, emu)
    ctx.set_max_frames(T)
    orc = po.Oracle()
    P = po.make_params(**pk)
    cp = c_params(lvm, pk)
    try:
        i = 0
        while i < nframes:
            t = i % ring
            nf = min(T, nframes - i, ring - t)
            if i + nf >= nframes:
                ctx.keep_float(True)
            prod = ctx.process_device_frames(cp, nf, frames[t:].ctypes.data, w, h, 3, w * 3, fb, fb, out[i:].ctypes.data, w * 3, fb, fb)
            ctx.synchronize()
            first_of_last_call = i
            i += nf
        fl = ctx.read_float((h, w, 3))
        assert np.isfinite(fl).all()
        for k in range(nframes):
            ref, pr = orc.process(frames[k % ring], P)
            if k == first_of_last_call:
                fr = orc.last_float()
                assert float(np.abs(fr - fl).max() / max(float(np.abs(fr).max()), 1e-30)) <= 1e-4
            if pr:
                du = np.abs(ref.astype(int) - out[k].astype(int))
                assert du.max() <= 1 and (du == 0).mean() >= 0.999, (k, int(du.max()), float((du == 0).mean()))
    finally:
        ctx.close(); orc.close()


def test_emu_clock_probe_plumbing(lvm, emu):
    """lvm_debug_clock_probe_start / _stop (bench.py's `clock_mhz`): start, stop, the ratio of the two counters x 100 MHz; one probe at a
    time; a context destroyed with a probe running.  (The emulation's counters tick at a fixed 20 : 1, i.e. "2000 MHz".)"""
    ctx = lvm.Context(0, 1, emu)
    ctx.clock_probe_start(0.0005)
    with pytest.raises(lvm.LvmError):
        ctx.clock_probe_start(0.0005)
    mhz, sec = ctx.clock_probe_stop()
    assert abs(mhz - 2000.0) < 200.0 and sec > 0.0        # (the two reads at each end are not simultaneous)
    with pytest.raises(lvm.LvmError):
        ctx.clock_probe_stop()
    ctx.clock_probe_start(0.0005)
    ctx.close()
