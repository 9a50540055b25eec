"""The oracle's imaging primitives against PyTorch's OWN implementations of the same published operations (a third party's
code, not another restatement by this repository's author): bilinear resize with half-pixel centres (= cv::resize
INTER_LINEAR), reflect-padded separable convolutions (pyrDown's 5 x 5 Gaussian, GaussianBlur 13 x 13 sigma 3, filter2D with
BORDER_REFLECT_101 = torch 'reflect'), the real DFT, and -- in the interior, where OpenCV's border rules play no part --
pyrUp as a stride-2 transposed convolution.  OpenCV itself is not installable here; this pins everything about these
primitives that is not OpenCV-specific."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

K5 = torch.tensor([1, 4, 6, 4, 1], dtype=torch.float64) / 16.0


def img(h, w, c=None, seed=0, lo=-50.0, hi=200.0):
    r = np.random.default_rng(seed)
    shape = (h, w) if c is None else (h, w, c)
    return r.uniform(lo, hi, size=shape).astype(np.float32)


def sep_conv_reflect(a, k):
    """2-D separable correlation of an (h, w) float64 tensor with the 1-D kernel k, reflect (101) borders."""
    r = len(k) // 2
    t = F.pad(a[None, None], (r, r, r, r), mode="reflect")
    t = F.conv2d(t, k.view(1, 1, 1, -1))
    t = F.conv2d(t, k.view(1, 1, -1, 1))
    return t[0, 0]


@pytest.mark.parametrize("h,w", [(16, 16), (17, 23), (68, 120), (135, 240)])
def test_pyr_down_against_torch_conv(po, h, w):
    a = img(h, w, seed=h + w)
    ref = sep_conv_reflect(torch.from_numpy(a).double(), K5)[::2, ::2].numpy()
    np.testing.assert_allclose(po.pyr_down(a), ref, rtol=0, atol=3e-5)


@pytest.mark.parametrize("h,w", [(8, 8), (34, 60), (17, 30)])
def test_pyr_up_interior_against_torch_transposed_conv(po, h, w):
    a = img(h, w, seed=3 * h + w)
    k2 = torch.outer(K5, K5) * 4.0                       # zero-insertion x2 followed by the kernel x4
    up = F.conv_transpose2d(torch.from_numpy(a).double()[None, None], k2[None, None], stride=2, padding=2, output_padding=1)[0, 0].numpy()
    got = po.pyr_up(a, (2 * w, 2 * h))
    np.testing.assert_allclose(got[2:-3, 2:-3], up[2:-3, 2:-3], rtol=0, atol=3e-5)    # interior: no border rule involved


@pytest.mark.parametrize("h,w,dh,dw", [(1088, 64, 1080, 64), (368, 40, 360, 40), (32, 48, 27, 45), (20, 30, 40, 60)])
def test_resize_linear_against_torch_interpolate(po, h, w, dh, dw):
    a = img(h, w, 3, seed=h, lo=0, hi=255)
    t = torch.from_numpy(a).double().permute(2, 0, 1)[None]
    ref = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
    # cv::resize computes the interpolation weight in float32 -- fx = (float)((dx + 0.5) * scale - 0.5), one ulp of ~1000 is
    # 6e-5 -- which the oracle restates; torch does it in the tensor's float64 here.  6e-5 x a 255 step = 0.015: the
    # tolerance.  A wrong sampling convention (corner-aligned, no half-pixel shift) is off by tens of grey levels.
    np.testing.assert_allclose(po.resize_linear(a, (dw, dh)), ref, rtol=0, atol=2e-2)


def test_gaussian_blur_against_torch(po):
    a = img(40, 52, seed=11, lo=0, hi=4)
    x = torch.arange(13, dtype=torch.float64) - 6
    g = torch.exp(-x * x / 18.0)
    g = g / g.sum()                                       # getGaussianKernel(13, 3)
    np.testing.assert_allclose(po.gauss_kernel(13, 3.0), g.numpy(), rtol=0, atol=1e-7)
    np.testing.assert_allclose(po.sep_filter(a, po.gauss_kernel(13, 3.0)), sep_conv_reflect(torch.from_numpy(a).double(), g).numpy(), rtol=0, atol=3e-6)


def test_filter2d_against_torch(po):
    a = img(33, 47, seed=5, lo=-1, hi=1)
    k = img(9, 9, seed=6, lo=-0.2, hi=0.2)
    t = F.pad(torch.from_numpy(a).double()[None, None], (4, 4, 4, 4), mode="reflect")
    ref = F.conv2d(t, torch.from_numpy(k).double()[None, None])[0, 0].numpy()          # conv2d IS correlation, like filter2D
    np.testing.assert_allclose(po.filter2d(a, k), ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("n", [2, 5, 13, 64, 128])
def test_dft_rows_against_torch_rfft(po, n):
    x = img(6, n, seed=n, lo=0, hi=255)
    X = po.dft_rows(x)                                    # CCS-packed, scaled by 1 / n
    Fq = (torch.fft.rfft(torch.from_numpy(x).double(), dim=1) / n).numpy()
    assert np.allclose(X[:, 0], Fq[:, 0].real, atol=3e-5)
    for k in range(1, (n - 1) // 2 + 1):
        assert np.allclose(X[:, 2 * k - 1], Fq[:, k].real, atol=3e-5) and np.allclose(X[:, 2 * k], Fq[:, k].imag, atol=3e-5)
    if n % 2 == 0:
        assert np.allclose(X[:, n - 1], Fq[:, n // 2].real, atol=3e-5)
