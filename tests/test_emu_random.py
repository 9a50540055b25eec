"""Seeded random sweep of the HIP kernels' logic on the CPU emulation build (tests/emu): geometry (odd and even sizes, sizes
around the tile / strip / vector-width boundaries), level count, channel count and every numeric parameter drawn at random,
a parameter change in mid-clip, and the frames compared with the CPU oracle BIT FOR BIT in the exact flavour -- the
fixed-size cases of test_emu_parity.py cover the variants by construction, this one covers what nobody thought of."""
import numpy as np
import pytest

from helpers import run_pair

MODES = (0, 2, 3)          # synth.config indices: Laplace, Riesz, Color


def draw(seed):
    r = np.random.default_rng(1000 + seed)
    idx = MODES[seed % 3]
    w = int(r.integers(20, 220))
    h = int(r.integers(20, 150))
    if r.random() < 0.4:
        w = int(r.choice([64, 128, 132, 192, 256, 260]))          # multiples of 4 / tile widths: the vector and strip kernels
    if r.random() < 0.3:
        h = int(r.choice([32, 48, 64, 96, 128]))
    levels = int(r.integers(1, 7))
    ch = 1 if (idx != 2 and r.random() < 0.2) else 3             # Riesz on one channel is a passthrough (covered elsewhere)
    fps = float(r.choice([15.0, 24.0, 30.0, 60.0]))
    return r, idx, w, h, levels, ch, fps


def configure(lvm, seed, scale=1):
    r, idx, w, h, levels, ch, fps = draw(seed)
    w, h = w * scale, h * scale
    ck, pk = lvm.synth.config(idx, (w, h, levels))
    ck["fps"] = fps
    ck["channels"] = ch
    pk["framerate"] = fps
    pk["amplification"] = float(r.uniform(1.0, 120.0))
    pk["chromAttenuation"] = float(r.uniform(0.0, 1.0))
    if idx == 0:
        pk["coWavelength"] = float(r.uniform(5.0, 900.0))
        lo = float(r.uniform(0.0, 0.5))                           # (0 exercises the lo == 0 -> 0.01 rule, TemporalFilter.cpp:12)
        pk["coLow"], pk["coHigh"] = (0.0 if r.random() < 0.15 else lo), float(r.uniform(lo + 0.05, 0.999))
    elif idx == 2:
        pk["coWavelength"] = float(r.uniform(1.0, 99.0))
        lo = float(r.uniform(0.1, 0.3 * fps))
        pk["coLow"], pk["coHigh"] = lo, float(r.uniform(lo + 0.1, 0.49 * fps))
    else:
        lo = float(r.uniform(0.1, 0.2 * fps))
        pk["coLow"], pk["coHigh"] = lo, float(r.uniform(lo + 0.05, 0.45 * fps))
    change_at = int(r.integers(3, 6))
    new_amp = float(r.uniform(1.0, 80.0))

    def vary(t, p):
        if t >= change_at:
            p["amplification"] = new_amp                          # non-structural change: state must survive it
        return p
    return ck, pk, vary


@pytest.mark.parametrize("seed", range(24))
def test_random_configurations_bit_exact(lvm, po, emu, seed):
    ck, pk, vary = configure(lvm, seed)
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 7, 0.0, exact=True, param_fn=vary)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_configurations_gpu(lvm, po, hip, seed):
    """The same draw at four times the size on the gfx950 build, default flavour, at the parity bars of SURVEY.md 8c."""
    ck, pk, vary = configure(lvm, seed, scale=4)
    worst = run_pair(lvm, po, hip, lvm.synth.Clip(**ck), pk, 7, 1e-4, param_fn=vary)
    print("random", seed, (ck["w"], ck["h"], pk["levels"], ck.get("channels", 3), pk["mode"]), "worst rel/u8/frac", worst)
