"""Shared helpers of the parity tests (test infrastructure)."""
import numpy as np


def c_params(lvm, pk, key=0):
    return lvm.LvmParams(pk["mode"], pk["levels"], pk["amplification"], pk["coWavelength"], pk["coLow"],
                         pk["coHigh"], pk["chromAttenuation"], pk["framerate"], key)


def run_pair(lvm, po, lib, clip, pk, nframes, float_tol, n_streams=1, u8_max=1, u8_frac=0.999, exact=False,
             param_fn=None, exact_lab=None, analytic=False, lab_lut=None):
    """Feeds the same frames to the CPU oracle and to the library behind the C ABI `lib`
    (gfx950 build or the CPU emulation build) and checks, frame by frame:
      (i)   produced / passthrough flags identical,
      (ii)  pre-quantisation float frame: max|d| / max|ref| <= float_tol  (exact => bit-equal),
      (iii) u8 frame: max abs diff <= u8_max LSB and >= u8_frac identical pixels.
    Both sides run OpenCV 4's default forward Lab (the interpolated 33^3 table) unless analytic=True (the cube-root form
    OpenCV computes with its interpolation switched off: oracle lvmo_set_lab_lut(0), library lvm_debug_lab_analytic).
    Returns the worst observed (rel, u8 diff, identical fraction)."""
    P = po.make_params(**pk)
    ctx = lvm.Context(0, n_streams, lib)
    ctx.keep_float(True)
    ctx.exact_lab(exact if exact_lab is None else exact_lab)   # bit-exact checks need OpenCV-order Lab math
    ctx.lab_analytic(analytic)
    if lab_lut is not None:
        ctx.set_lab_lut(lab_lut)
    po.lib().lvmo_set_lab_lut(0 if analytic else 1)
    orc = po.Oracle()
    worst = [0.0, 0, 1.0]
    try:
        for t in range(nframes):
            if param_fn:
                pk2 = param_fn(t, dict(pk))
                P = po.make_params(**pk2)
                cp = c_params(lvm, pk2)
            else:
                cp = c_params(lvm, pk)
            f = clip.frame(t)
            ref, pr = orc.process(f, P)
            out, pg = ctx.process(f, cp)
            assert pr == pg, "produced flag differs at frame %d: oracle %s, lib %s" % (t, pr, pg)
            if not pr:
                assert out is f or np.array_equal(out, f)
                continue
            fr = orc.last_float()
            fg = ctx.read_float(fr.shape)
            assert np.isfinite(fg).all(), "non-finite values in frame %d" % t
            rel = float(np.abs(fr - fg).max() / max(float(np.abs(fr).max()), 1e-30))
            du = np.abs(ref.astype(np.int32) - out.astype(np.int32))
            worst = [max(worst[0], rel), max(worst[1], int(du.max())), min(worst[2], float((du == 0).mean()))]
            if exact:
                assert np.array_equal(fr, fg), "frame %d: float frames differ (max rel %.3e)" % (t, rel)
                assert np.array_equal(ref, out)
            else:
                assert rel <= float_tol, "frame %d: float rel err %.3e > %.1e" % (t, rel, float_tol)
                assert du.max() <= u8_max, "frame %d: u8 diff %d" % (t, du.max())
                assert (du == 0).mean() >= u8_frac, "frame %d: identical fraction %.5f" % (t, (du == 0).mean())
    finally:
        po.lib().lvmo_set_lab_lut(1)
        ctx.close()
        orc.close()
    return worst
