"""The display half of SURVEY.md 8(f) rank 2: lvm_chain_present (host frame in, processed frame + `original` tap left in DEVICE buffers) and
host/HipDisplayPresenter.hpp, the reference-side presenter around it (pixel-unpack buffers mapped through HIP's GL interop; the core is a
template over the GL calls).  CPU: the entry point through the emulation build (device memory = host memory there), both panes against
the CPU oracle (po.preprocess + Oracle.process); the presenter compiled and linked against a mock traits type; its GL binding syntax-checked where <GL/gl.h>
exists.  GPU: the same entry point with real device buffers, and the presenter run with a mock whose pixel buffer is a hipMalloc'd buffer."""
import os
import subprocess

import numpy as np
import pytest

from helpers import c_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "live-video-magnification_amd")


def _pre(lvm, ds, roi, gray):
    return lvm.LvmPreprocessParams(ds, 1 if roi else 0, *(roi or (0.0, 0.0, 1.0, 1.0)), 1 if gray else 0)


CASES = [(0, (128, 96, 3), (2, (0.1, 0.1, 0.8, 0.8), True)),      # ROI + INTER_AREA + gray: colour tap next to a gray result
         (0, (96, 64, 2), (1, None, False)),                      # identity stages: the tap is the frame itself
         (2, (96, 64, 3), (2, None, False)),                      # Riesz: first frame passes through (d_proc = what the magnifier saw)
         (3, (96, 64, 2), (1, None, True))]                       # Color on a grayed frame


def _check_present(lvm, po, lib, alloc, exact):
    """Both panes of lvm_chain_present against the ORACLE (round 6; until then against the sibling entry point): the `original` pane is
    PreprocessProcessor's output -- what runChainOnce taps after chain[0], before GrayscaleProcessor (ChainBuilder.cpp:25) -- and must be
    byte-equal (integer work); the processed pane is Oracle.process of the fully preprocessed frame, or that frame itself when the magnifier
    passes through (MagnificationProcessor.cpp:61): bit-exact in the exact flavour (emulation build), within the u8 bars otherwise."""
    for cfg, size, (ds, roi, gray) in CASES:
        ck, pk = lvm.synth.config(cfg, size)
        clip = lvm.synth.Clip(**ck)
        cpre = _pre(lvm, ds, roi, gray)
        opre = po.PreParams(ds, 1 if roi else 0, *(roi or (0.0, 0.0, 1.0, 1.0)), 1 if gray else 0)
        opre_tap = po.PreParams(ds, 1 if roi else 0, *(roi or (0.0, 0.0, 1.0, 1.0)), 0)
        P = po.make_params(**pk)
        b = lvm.Context(0, 1, lib)
        b.exact_lab(exact)
        orc = po.Oracle()
        try:
            for t in range(6):
                f = clip.frame(t)
                small = po.preprocess(f, opre)                 # what the magnifier sees
                tap = po.preprocess(f, opre_tap)               # the `original` pane
                ref, pr_ref = orc.process(small, P)
                want = ref if pr_ref else small
                oh, ow = small.shape[:2]
                och = 1 if small.ndim == 2 else small.shape[2]
                dp, read_p = alloc(oh * ow * och)
                do, read_o = alloc(oh * ow * 3)
                pr = b.chain_present(f, cpre, c_params(lvm, pk), dp, ow * och, do, ow * 3)
                assert bool(pr) == bool(pr_ref), (cfg, t)
                got = read_p().reshape(want.shape)
                if exact or not pr_ref:
                    assert np.array_equal(got, want), (cfg, t, "processed pane")
                else:
                    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
                    assert d.max() <= 1 and (d == 0).mean() >= 0.999, (cfg, t, int(d.max()), float((d == 0).mean()))
                assert np.array_equal(read_o().reshape(tap.shape), tap), (cfg, t, "original pane")
        finally:
            b.close(); orc.close()


def test_chain_present_emu(lvm, po, emu):
    keep = []

    def alloc(n):
        buf = np.full(n, 0x5A, np.uint8)
        keep.append(buf)
        return buf.ctypes.data, (lambda: buf.copy())
    _check_present(lvm, po, emu, alloc, True)


@pytest.mark.gpu
def test_chain_present_gpu(lvm, po, hip):
    import torch

    def alloc(n):
        t = torch.full((n,), 0x5A, dtype=torch.uint8, device="cuda")
        return t.data_ptr(), (lambda: t.cpu().numpy())
    _check_present(lvm, po, hip, alloc, False)


PRESENTER_SRC = r'''
#include <cstdio>
#include <cstring>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "HipDisplayPresenter.hpp"
// mock traits: the "pixel-unpack buffer" is a device buffer, the "texture" a host vector filled by a device -> host copy
struct MockGl {
    struct Buffer { void* d = nullptr; std::size_t bytes = 0; int maps = 0; };
    struct Texture { std::vector<unsigned char> px; int w = 0, h = 0, channels = 0, uploads = 0; };
    static void create(Buffer& b, std::size_t n) { if (hipMalloc(&b.d, n) != hipSuccess) throw lvm::Error(LVM_ERR_OOM, "hipMalloc"); b.bytes = n; }
    static void destroy(Buffer& b) { if (b.d) (void)hipFree(b.d); b = Buffer{}; }
    static std::uint8_t* map(Buffer& b) { ++b.maps; return static_cast<std::uint8_t*>(b.d); }
    static void unmap(Buffer& b) { --b.maps; }
    static void upload(Buffer& b, Texture& t, int w, int h, int c) {
        if (b.maps != 0) throw lvm::Error(LVM_ERR_INVALID, "upload from a mapped buffer");
        t.px.resize((size_t)w * h * c); t.w = w; t.h = h; t.channels = c; ++t.uploads;
        if (hipMemcpy(t.px.data(), b.d, t.px.size(), hipMemcpyDeviceToHost) != hipSuccess) throw lvm::Error(LVM_ERR_HIP, "hipMemcpy");
    }
};
int main() {
    try {
        const int W = 96, H = 64;
        lvm::DisplayPresenter<MockGl> pres(0);
        lvm::Magnifier single(0, 1);
        lvm_preprocess_params pre{}; pre.downscale = 2; pre.roiW = pre.roiH = 1.f; pre.grayscale = 1;
        lvm::MagnificationParams mag; mag.mode = lvm::MagnificationMode::Laplace; mag.levels = 3; mag.amplification = 15; mag.coWavelength = 100;
        mag.coLow = 0.1; mag.coHigh = 0.4; mag.chromAttenuation = 0.2;
        MockGl::Texture tp, to;
        int bad = 0;
        for (int t = 0; t < 5; ++t) {
            std::vector<unsigned char> f((size_t)W * H * 3);
            for (size_t i = 0; i < f.size(); ++i) f[i] = (unsigned char)(40 + ((i * 7 + (size_t)t * 13 + (i / 97)) % 150));
            const auto s = pres.present(f.data(), W, H, 3, W * 3, pre, mag, tp, to);
            std::vector<unsigned char> ref((size_t)(W / 2) * (H / 2));
            const bool produced = single.chain_process(pre, mag, f.data(), W, H, 3, W * 3, ref.data(), W / 2);
            if (s.w != W / 2 || s.h != H / 2 || s.proc_channels != 1 || s.orig_channels != 3 || s.produced != produced || tp.px != ref) { std::printf("frame %d: processed pane differs\n", t); ++bad; }
            if (to.channels != 3 || (int)to.px.size() != (W / 2) * (H / 2) * 3) { std::printf("frame %d: bad original pane\n", t); ++bad; }
        }
        std::printf("uploads %d %d bad %d\n", tp.uploads, to.uploads, bad);
        return bad ? 4 : 0;
    } catch (const lvm::Error& e) { std::printf("lvm::Error %d: %s\n", e.status(), e.what()); return 3; }
}
'''


def _build_and_run(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(PRESENTER_SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__=1", str(src), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "host"),
                           "-I", "/opt/rocm/include", "-L", PKG, "-llvm_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    return subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)


def test_display_presenter_compiles_and_fails_loudly_without_a_device(tmp_path):
    r = _build_and_run(tmp_path)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0 and "uploads 5 5 bad 0" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 3 and "lvm::Error -3" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_display_presenter_runs_on_the_gpu(tmp_path):
    r = _build_and_run(tmp_path)
    assert r.returncode == 0 and "uploads 5 5 bad 0" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not (os.path.exists("/usr/include/GL/gl.h") and os.path.exists("/opt/rocm/include/hip/hip_gl_interop.h")), reason="no GL headers")
def test_gl_interop_binding_syntax_checks(tmp_path):
    """GlInteropTraits (OpenGL pixel-unpack buffers + hip_gl_interop.h) against the real headers: compiled, not run -- no GL context here"""
    src = tmp_path / "g.cpp"
    src.write_text('#define LVM_WITH_GL_INTEROP 1\n#include "HipDisplayPresenter.hpp"\nint use(lvm::GlDisplayPresenter& p) { p.reset(); return 0; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", str(src), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "host"), "-I", "/opt/rocm/include"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
