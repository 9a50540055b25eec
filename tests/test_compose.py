"""Export pane composition on the device: lvm_compose_device against the oracle's restatement
of Exporter::compose (export/Exporter.cpp:53-88: toBgr, crop to the common even size, LeftRight / TopBottom / None).
Byte work: bit-exact.  CPU part: oracle known answers + the kernel's logic through the HIP emulation build; GPU part:
the gfx950 library on device memory, incl. two 1080p streams fed by the magnifier's own device output."""
import ctypes

import numpy as np
import pytest

from helpers import c_params

NONE, LR, TB = 0, 1, 2
CASES = [  # (split, (ow, oh, och), (pw, ph, pch))
    (LR, (64, 48, 3), (64, 48, 3)),
    (TB, (64, 48, 3), (64, 48, 3)),
    (NONE, (64, 48, 3), (65, 49, 3)),        # odd size: cropped to 64 x 48
    (LR, (67, 45, 3), (64, 48, 1)),          # different sizes and a gray processed pane: common even size 64 x 44
    (TB, (33, 21, 1), (40, 30, 3)),          # gray original
    (LR, (30, 20, 3), (30, 20, 3)),          # 3 * w not a multiple of 4: the byte path for the right pane
    (LR, None, (50, 40, 3)),                 # original tap missing: both panes show the processed frame (:62)
    (NONE, None, (1, 37, 3)),                # even width 0: empty Mat
    (TB, (520, 31, 3), (520, 31, 3)),        # several 256-pixel groups per row, odd height
]


def _img(w, h, ch, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w) if ch == 1 else (h, w, ch), dtype=np.uint8)


def test_oracle_compose_known_answers(po):
    o = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)            # 3 x 2 BGR
    p = (100 + np.arange(2 * 3, dtype=np.uint8)).reshape(2, 3)           # 3 x 2 gray
    c = po.compose(LR, o, p)                                             # common even size 2 x 2
    assert c.shape == (2, 4, 3)
    assert c[0, 0].tolist() == [0, 1, 2] and c[1, 1].tolist() == [12, 13, 14]
    assert c[0, 2].tolist() == [100, 100, 100] and c[1, 3].tolist() == [104, 104, 104]
    c = po.compose(TB, o, p)
    assert c.shape == (4, 2, 3) and c[2, 1].tolist() == [101, 101, 101] and c[1, 0].tolist() == [9, 10, 11]
    assert po.compose(NONE, o, p).shape == (2, 2, 3)
    assert po.compose(NONE, None, np.zeros((5, 1, 3), np.uint8)) is None


def _check(lvm, po, lib, to_dev, from_dev):
    ctx = lvm.Context(0, 1, lib)
    try:
        for i, (split, og, pg) in enumerate(CASES):
            proc = _img(*pg, 200 + i)
            orig = _img(*og, 300 + i) if og else None
            ref = po.compose(split, orig, proc)
            ow, oh, och = og if og else (0, 0, 0)
            pw, ph, pch = pg
            geo = ctx.compose_geometry(split, ow if og else pw, oh if og else ph, pw, ph)
            if ref is None:
                assert geo == (0, 0, 0, 0)
                continue
            assert (geo[3], geo[2]) == ref.shape[:2]
            pad = 5 if i % 2 else 0                                           # ragged canvas rows on every other case
            canvas = np.full((geo[3], geo[2] * 3 + pad), 0xEE, np.uint8)
            d_p, d_o, d_c = to_dev(proc), (to_dev(orig) if og else None), to_dev(canvas)
            ctx.compose_device(split, d_o[0] if og else None, ow, oh, och, ow * och, ow * oh * och, d_p[0], pw, ph, pch, pw * pch,
                               pw * ph * pch, d_c[0], geo[2] * 3 + pad, canvas.size)
            ctx.synchronize()
            got = from_dev(d_c)
            assert np.array_equal(got[:, :geo[2] * 3].reshape(ref.shape), ref), "case %d" % i
            assert (got[:, geo[2] * 3:] == 0xEE).all(), "padding written in case %d" % i
    finally:
        ctx.close()


def test_compose_emu_bit_exact(lvm, po, emu):
    keep = []

    def to_dev(a):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return (ctypes.c_void_p(a.ctypes.data), a)
    _check(lvm, po, emu, to_dev, lambda d: d[1])


@pytest.mark.gpu
def test_compose_gpu_bit_exact(lvm, po, hip):
    import torch

    def to_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return (ctypes.c_void_p(t.data_ptr()), t)
    _check(lvm, po, hip, to_dev, lambda d: d[1].cpu().numpy())


@pytest.mark.gpu
def test_compose_gpu_1080p_two_streams_after_the_magnifier(lvm, po, hip):
    """Device-resident export hand-off: two 1080p streams, Laplace output left in HBM, side-by-side canvas composed
    from the device input and the device output; only the canvas is downloaded."""
    import torch
    ck, pk = lvm.synth.config(1)
    w, h = ck["w"], ck["h"]
    clips = [lvm.synth.Clip(seed=1234 + s, **ck) for s in range(2)]
    ctx = lvm.Context(0, 2, hip)
    orcs = [po.Oracle() for _ in range(2)]
    P = po.make_params(**pk)
    cp = c_params(lvm, pk)
    st = torch.cuda.current_stream().cuda_stream
    fb = w * h * 3
    try:
        for t in range(3):
            fin = np.stack([c.frame(t) for c in clips])
            d_in = torch.from_numpy(fin).cuda()
            d_out = torch.zeros_like(d_in)
            d_canvas = torch.zeros((2, h, 2 * w, 3), dtype=torch.uint8, device="cuda")
            assert ctx.process_device(cp, d_in.data_ptr(), w, h, 3, w * 3, fb, d_out.data_ptr(), w * 3, fb, st)
            ctx.compose_device(LR, ctypes.c_void_p(d_in.data_ptr()), w, h, 3, w * 3, fb, ctypes.c_void_p(d_out.data_ptr()), w, h, 3, w * 3, fb,
                               ctypes.c_void_p(d_canvas.data_ptr()), 2 * w * 3, 2 * fb, ctypes.c_void_p(st))
            torch.cuda.synchronize()
            got = d_canvas.cpu().numpy()
            out = d_out.cpu().numpy()
            for s in range(2):
                ref, _ = orcs[s].process(fin[s], P)
                assert np.array_equal(got[s], po.compose(LR, fin[s], out[s]))          # the composition itself: bit-exact
                du = np.abs(ref.astype(int) - got[s][:, w:].astype(int))               # and the right pane is the magnified frame
                assert du.max() <= 1 and (du == 0).mean() >= 0.999
    finally:
        ctx.close()


def test_compose_rejects_bad_strides(lvm, emu):
    """The kernel indexes with the caller's strides: rows that cannot hold their pixels and overlapping streams are refused."""
    import ctypes
    w, h = 32, 16
    o, p = _img(w, h, 3, 1), _img(w, h, 3, 2)
    canvas = np.zeros((2, h, 2 * w, 3), np.uint8)
    oo, pp = np.stack([o, o]), np.stack([p, p])
    LR = 1
    for ns, kw in ((1, dict(pstride=w * 3 - 1)), (1, dict(ostride=w * 3 - 3)), (1, dict(ostride=-w * 3)), (2, dict(csstride=h * 2 * w * 3 - 4)),
                   (2, dict(psstride=w * 3))):
        ctx = lvm.Context(0, ns, emu)
        a = dict(ostride=w * 3, osstride=w * h * 3, pstride=w * 3, psstride=w * h * 3, cstride=2 * w * 3, csstride=h * 2 * w * 3)
        a.update(kw)
        with pytest.raises(lvm.LvmError):
            ctx.compose_device(LR, ctypes.c_void_p(oo.ctypes.data), w, h, 3, a["ostride"], a["osstride"], ctypes.c_void_p(pp.ctypes.data), w, h, 3,
                               a["pstride"], a["psstride"], ctypes.c_void_p(canvas.ctypes.data), a["cstride"], a["csstride"])
        ctx.close()
