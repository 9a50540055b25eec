"""Kernel-logic parity on CPU: the product sources (csrc/*.hip) compiled against the HIP
emulation header (tests/emu) vs the CPU oracle, through the same C ABI.  Because the kernels
follow the oracle's operation order, the float frames must be BIT-IDENTICAL wherever no
transcendental is involved (the emulation build uses the host libm like the oracle does).
This validates tiling, border rules, odd sizes and state handling without a GPU; the real
gfx950 build is checked by tests/test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest

from helpers import c_params, run_pair


@pytest.mark.parametrize("w,h,levels,ch", [(160, 90, 3, 3), (135, 77, 4, 3), (100, 64, 2, 1), (64, 48, 1, 3),
                                            (67, 131, 3, 3), (40, 23, 2, 3), (320, 180, 4, 3), (404, 300, 5, 3),
                                            (330, 200, 6, 3)])
def test_laplace_emu_bit_exact(lvm, po, emu, w, h, levels, ch):
    ck, pk = lvm.synth.config(0, (w, h, levels))
    ck["channels"] = ch
    clip = lvm.synth.Clip(**ck)
    run_pair(lvm, po, emu, clip, pk, 6, 0.0, exact=True)


@pytest.mark.parametrize("idx,w,h,levels", [(0, 135, 77, 4), (0, 328, 109, 3), (2, 135, 77, 3), (2, 264, 150, 3)])
def test_analytic_flavour_emu_bit_exact(lvm, po, emu, idx, w, h, levels, monkeypatch):
    """lvm_debug_lab_analytic: the cube-root forward Lab (OpenCV with its interpolation switched off) in the kernels that
    convert from the u8 frame themselves, against the oracle with lvmo_set_lab_lut(0); scalar and 4-pixel variants, strip
    first kernel forced on."""
    monkeypatch.setenv("LVM_D0_MIN_TASKS", "0")
    ck, pk = lvm.synth.config(idx, (w, h, levels))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 5, 0.0, exact=True, analytic=True)


def test_laplace_emu_param_changes_and_reset(lvm, po, emu):
    ck, pk = lvm.synth.config(0, (96, 64, 3))
    clip = lvm.synth.Clip(**ck)

    def vary(t, p):
        if t >= 3:
            p["amplification"] = 35.0
            p["coLow"] = 0.0            # exercises the lo == 0 -> 0.01 rule (TemporalFilter.cpp:11-12)
        if t >= 5:
            p["levels"] = 2             # structural change -> state reset
        return p
    run_pair(lvm, po, emu, clip, pk, 8, 0.0, exact=True, param_fn=vary)


def test_laplace_emu_two_streams_are_independent(lvm, po, emu):
    import ctypes as C
    ck, pk = lvm.synth.config(0, (96, 64, 3))
    clips = [lvm.synth.Clip(seed=1234 + s, **ck) for s in range(2)]
    h, w = 64, 96
    ctx = lvm.Context(0, 2, emu)
    ctx.exact_lab(True)
    orcs = [po.Oracle(), po.Oracle()]
    P = po.make_params(**pk)
    cp = c_params(lvm, pk)
    for t in range(5):
        fin = np.stack([c.frame(t) for c in clips])
        fout = np.zeros_like(fin)
        produced = ctx.process_device(cp, fin.ctypes.data, w, h, 3, w * 3, w * h * 3, fout.ctypes.data, w * 3, w * h * 3)
        assert produced
        for s in range(2):
            ref, _ = orcs[s].process(fin[s], P)
            assert np.array_equal(ref, fout[s])
    ctx.close()


# ---- Riesz (phase) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,levels", [(96, 64, 3), (135, 77, 4), (64, 48, 1), (67, 131, 2), (160, 90, 5), (134, 78, 2)])
def test_riesz_emu_bit_exact(lvm, po, emu, w, h, levels):
    ck, pk = lvm.synth.config(2, (w, h, levels))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 6, 0.0, exact=True)


@pytest.mark.parametrize("w,h,levels,calls", [(264, 150, 3, (2, 1, 2)), (160, 90, 5, (2, 3)), (96, 64, 3, (2, 1, 1))])
@pytest.mark.parametrize("wide", ["1", "0"])
def test_riesz_emu_wide_and_narrow_tile_kernels(lvm, po, emu, w, h, levels, calls, wide, monkeypatch):
    """Round 5 picks the tile kernels by launch size: k_rz_split2 (4 x 2 outputs per thread) from 600 000 plane-pixels per launch, else
    k_rz_split (one output per thread); k_rz_phase4 (four pixels per thread) from two frames x streams per launch, else k_rz_phase.  Both
    choices forced onto small frames -- per-frame calls between temporal batches, so that the state written by one phase kernel is read
    by the other."""
    monkeypatch.setenv("LVM_RZ_SPLIT2_MIN", "0" if wide == "1" else "1000000000")
    monkeypatch.setenv("LVM_RZ_PHASE4_MIN_FRAMES", "1" if wide == "1" else "2")
    monkeypatch.setenv("LVM_RZ_SPLIT_ROWS", "0")
    _frames_clip(lvm, po, emu, 2, w, h, levels, 1, calls)


@pytest.mark.parametrize("blur4", ["1", "0"])
def test_riesz_emu_register_blocked_blur(lvm, po, emu, blur4, monkeypatch):
    """k_rz_blur_amp4 (64 x 32 tiles, vector staging of interior tiles, 4 x 2 outputs per thread) against the
    scalar kernel's arithmetic: a frame with interior, edge and partial tiles on two levels."""
    monkeypatch.setenv("LVM_RZ_BLUR4", blur4)
    monkeypatch.setenv("LVM_RZ_BLUR_STRIPS", "0")         # (the strip form is the default for every even-width level since round 5)
    ck, pk = lvm.synth.config(2, (264, 150, 3))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 4, 0.0, exact=True)


@pytest.mark.parametrize("w,h,levels,rows,exact", [(264, 150, 3, "16", True), (134, 78, 2, "64", True), (520, 70, 3, "32", True), (264, 150, 3, "16", False)])
def test_riesz_emu_strip_blur(lvm, po, emu, w, h, levels, rows, exact, monkeypatch):
    """k_rz_blur_strips (three 13-tap Gaussians + amplify as wave strips: DPP halo exchange over three lanes either side, a
    13-row register window per plane) forced onto small levels: several strips per row with mirrored edge columns, a last
    strip ending short of the wave, strips shorter than the 12 halo rows, odd heights, level widths 2 mod 4.
    exact=False: the default flavour (hardware sine / cosine) against the parity bars."""
    monkeypatch.setenv("LVM_RZ_BLUR_STRIPS_MIN", "0")
    monkeypatch.setenv("LVM_RZ_BLUR_STRIP_ROWS", rows)
    ck, pk = lvm.synth.config(2, (w, h, levels))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 4, 0.0 if exact else 1e-4, exact=exact)


def test_riesz_emu_strip_blur_in_temporal_batches(lvm, po, emu, monkeypatch):
    """Batched frames with the strip form forced onto every level: the phase kernel then stores no per-frame Riesz pair for
    those levels and the amplify stage recomputes it from the band (two streams, calls of several lengths)."""
    monkeypatch.setenv("LVM_RZ_BLUR_STRIPS_MIN", "0")
    monkeypatch.setenv("LVM_RZ_BLUR_STRIP_ROWS", "32")
    _frames_clip(lvm, po, emu, 2, 264, 150, 3, 2, (2, 4, 1))


def test_riesz_emu_tiled_blur_still_matches(lvm, po, emu, monkeypatch):
    monkeypatch.setenv("LVM_RZ_BLUR_STRIPS", "0")
    ck, pk = lvm.synth.config(2, (264, 150, 3))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 3, 0.0, exact=True)


@pytest.mark.parametrize("compact", ["1", "0"])
def test_riesz_emu_collapse_tile_variants(lvm, po, emu, compact, monkeypatch):
    """The collapse kernels' zero-injected tile, compact (even rows / columns only; planes with even width and height) and
    full: interior (vector-staged) and border tiles on level 0, a plane with an odd height on level 1."""
    monkeypatch.setenv("LVM_RZ_COMPACT", compact)
    ck, pk = lvm.synth.config(2, (264, 150, 3))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 3, 0.0, exact=True)


def test_riesz_emu_cutoff_change_gray_and_reset(lvm, po, emu):
    ck, pk = lvm.synth.config(2, (96, 64, 3))

    def vary(t, p):
        if t >= 3:
            p["coLow"] = 1.0            # MagnifyCore.hpp:243-248: new coefficients, filters cleared, prior rebuilt
        if t >= 5:
            p["coHigh"] = 5.0
        if t >= 7:
            p["levels"] = 2             # structural change: re-init => passthrough frame
        return p
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 10, 0.0, exact=True, param_fn=vary)
    gray = lvm.synth.Clip(96, 64, channels=1)
    run_pair(lvm, po, emu, gray, pk, 3, 0.0, exact=True)    # < 3 channels: always passthrough (:212)


# ---- Colour ----------------------------------------------------------------------------------------
# (at least 12 frames per clip: with a window of a few columns the ideal band-pass passes nothing, the magnified signal is zero
# and neither the pyramid nor the up-chain arithmetic would influence the output -- checked by mutating the kernels)
@pytest.mark.parametrize("w,h,levels,ch,fps", [(96, 64, 3, 3, 60.0), (135, 77, 4, 3, 30.0), (64, 48, 1, 3, 7.0),
                                                (67, 131, 2, 1, 15.0)])
def test_color_emu_bit_exact(lvm, po, emu, w, h, levels, ch, fps):
    ck, pk = lvm.synth.config(3, (w, h, levels))
    ck["channels"] = ch; ck["fps"] = fps; pk["framerate"] = fps
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 20, 0.0, exact=True)


@pytest.mark.parametrize("w,h,levels,rows", [(264, 90, 3, "0"), (264, 90, 3, "2"), (264, 90, 3, "8"), (264, 90, 3, "36"), (96, 77, 2, "16")])
def test_color_emu_output_kernel_variants(lvm, po, emu, w, h, levels, rows, monkeypatch):
    """The vectorised output kernels: tiled (rows = 0) and wave strips of 2 ... 36 rows (k_col_out_strips: both pyrUps inside, window
    positions of two row slots each, U2 window in an LDS ring, buffer loads / stores), on heights the up chain overshoots (the
    bilinear row map skips source rows) and widths with a partly filled wave."""
    monkeypatch.setenv("LVM_COL_OUT_ROWS", rows)
    monkeypatch.setenv("LVM_COL_OUT_MIN_TASKS", "0")
    ck, pk = lvm.synth.config(3, (w, h, levels))
    ck["fps"] = 15.0; pk["framerate"] = 15.0
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 12, 0.0, exact=True)


@pytest.mark.parametrize("w,h,levels", [(516, 40, 2), (772, 24, 2), (256, 64, 3)])
def test_color_emu_strip_kernel_border_lanes(lvm, po, emu, w, h, levels, monkeypatch):
    """k_col_out_strips: widths whose last strip holds one group (516: the U2 border column vw - 4 sits in the strip BEFORE the last
    one), interior strips (772), an exact multiple of the strip width."""
    monkeypatch.setenv("LVM_COL_OUT_MIN_TASKS", "0")
    ck, pk = lvm.synth.config(3, (w, h, levels))
    ck["fps"] = 15.0; pk["framerate"] = 15.0
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 12, 0.0, exact=True)


@pytest.mark.parametrize("w,h,levels", [(264, 90, 3), (512, 128, 4)])
def test_color_emu_previous_strip_kernels_still_match(lvm, po, emu, w, h, levels, monkeypatch):
    """LVM_COL_OUT_LEAN=0: k_col_out_rows (the fallback for row maps that are not strictly increasing / one-level pyramids)."""
    monkeypatch.setenv("LVM_COL_OUT_LEAN", "0")
    monkeypatch.setenv("LVM_COL_OUT_MIN_TASKS", "0")
    ck, pk = lvm.synth.config(3, (w, h, levels))
    ck["fps"] = 15.0; pk["framerate"] = 15.0
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 12, 0.0, exact=True)


def test_color_emu_one_level_uses_the_single_pyrup_kernels(lvm, po, emu, monkeypatch):
    """levels = 1: no level-2 image exists, the strip kernels with one pyrUp inside run."""
    monkeypatch.setenv("LVM_COL_OUT_MIN_TASKS", "0")
    ck, pk = lvm.synth.config(3, (128, 48, 1))
    ck["fps"] = 15.0; pk["framerate"] = 15.0
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 12, 0.0, exact=True)


@pytest.mark.parametrize("w,h,levels,rows", [(264, 90, 3, "7"), (264, 90, 3, "17"), (96, 77, 2, "1"), (96, 77, 2, "4"), (520, 52, 3, "7"),
                                               (128, 37, 3, "17"), (512, 128, 4, "4")])
def test_color_emu_first_two_levels_in_one_pass(lvm, po, emu, w, h, levels, rows, monkeypatch):
    """k_down01_rows (u8 -> level 2 without writing level 1; large launches only in production, forced here): strips of 1 ... 17
    level-2 rows -- top strip (mirrored level-1 rows -2, -1), interior strips, bottom rows with an even and an odd number of level-1
    rows (rows h1, h1 + 1 are window copies), one and several strips per row, the level-1 border columns."""
    monkeypatch.setenv("LVM_D0_MIN_TASKS", "0")
    monkeypatch.setenv("LVM_COL_DOWN01_ROWS", rows)
    monkeypatch.setenv("LVM_COL_OUT_MIN_TASKS", "0")
    ck, pk = lvm.synth.config(3, (w, h, levels))
    ck["fps"] = 15.0; pk["framerate"] = 15.0
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 12, 0.0, exact=True)    # (12 frames: the band-pass passes something)


def test_color_emu_two_level_pass_can_be_switched_off(lvm, po, emu, monkeypatch):
    monkeypatch.setenv("LVM_D0_MIN_TASKS", "0")
    monkeypatch.setenv("LVM_COL_DOWN01", "0")
    ck, pk = lvm.synth.config(3, (264, 90, 3))
    ck["fps"] = 15.0; pk["framerate"] = 15.0
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 12, 0.0, exact=True)


def test_color_emu_wide_band_and_fps_change(lvm, po, emu):
    ck, pk = lvm.synth.config(3, (64, 48, 2))
    pk["coLow"] = 0.0; pk["coHigh"] = 40.0                   # every packed element passes (lo == 0 -> 0.01)

    def vary(t, p):
        if t >= 12:
            p["framerate"] = 7.0                              # window cap shrinks 128 -> 16: one column dropped per frame
        return p
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 24, 0.0, exact=True, param_fn=vary)


def test_mode_switch_drops_state(lvm, po, emu):
    ck, pk0 = lvm.synth.config(0, (96, 64, 3))
    _, pk2 = lvm.synth.config(2, (96, 64, 3))
    _, pk3 = lvm.synth.config(3, (96, 64, 3))

    def vary(t, p):
        return dict([pk0, pk2, pk3, pk0][(t // 3) % 4])
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk0, 12, 0.0, exact=True, param_fn=vary)


def test_fast_lab_flavour_stays_within_tolerance(lvm, po, emu):
    """Default arithmetic (float32 cube root, reciprocal multiplies) vs the oracle: within the
    1e-4 / 1 LSB parity bar for every mode."""
    for idx in (0, 2, 3):
        ck, pk = lvm.synth.config(idx, (96, 64, 3))
        run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 8, 1e-4, exact=False, exact_lab=False)


def _pipelined_clip(lvm, po, lib, w, h, levels, nframes, ring=4):
    """lvm_process_device with pipeline depth 1 over a ring of in/out buffers + flush: every frame's
    output must equal the oracle's (and therefore the depth-0 schedule's) bit for bit."""
    ck, pk = lvm.synth.config(0, (w, h, levels))
    clip = lvm.synth.Clip(**ck)
    P = po.make_params(**pk)
    cp = c_params(lvm, pk)
    ctx = lvm.Context(0, 1, lib)
    ctx.exact_lab(True)
    ctx.set_pipeline(1)
    orc = po.Oracle()
    ins = [np.zeros((h, w, 3), np.uint8) for _ in range(ring)]
    outs = [np.zeros((h, w, 3), np.uint8) for _ in range(ring)]
    refs = {}
    for t in range(nframes):
        k = t % ring
        if t >= ring:        # slot k is about to be reused: frame t-ring must already be complete
            assert np.array_equal(outs[k], refs[t - ring]), "frame %d" % (t - ring)
        ins[k][...] = clip.frame(t)
        refs[t], _ = orc.process(ins[k].copy(), P)
        assert ctx.process_device(cp, ins[k].ctypes.data, w, h, 3, w * 3, w * h * 3, outs[k].ctypes.data, w * 3, w * h * 3)
    ctx.flush()
    ctx.synchronize()
    for t in range(max(0, nframes - ring), nframes):
        assert np.array_equal(outs[t % ring], refs[t]), "frame %d" % t
    ctx.close()


@pytest.mark.parametrize("w,h,levels", [(160, 90, 3), (320, 180, 4), (135, 77, 4), (64, 48, 1)])
def test_laplace_emu_pipelined_schedule(lvm, po, emu, w, h, levels):
    _pipelined_clip(lvm, po, emu, w, h, levels, 11)


@pytest.mark.parametrize("w,h,levels", [(1000, 760, 6), (800, 600, 4), (1001, 763, 5)])
def test_laplace_emu_fused_multi_level_pyrdown(lvm, po, emu, w, h, levels):
    """Sizes large enough that the tail starts at level 3-4, so G_1 -> G_2..G_4 goes through the
    fused k_pyr_down_multi<2|3> kernel (vector and generic first/last kernels)."""
    ck, pk = lvm.synth.config(0, (w, h, levels))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 3, 0.0, exact=True)


@pytest.mark.parametrize("rows", [4, 8, 16])
def test_laplace_emu_final_kernel_strip_heights(lvm, po, emu, rows, monkeypatch):
    """k_lap_final_v4 walks strips of `rows` output rows per wave (the launch code shortens them for small
    frames): force the long strips, on a height that leaves a partial last strip and a width with a
    partly filled last wave."""
    monkeypatch.setenv("LVM_FIN_ROWS", str(rows))
    monkeypatch.setenv("LVM_FIN_MIN_TASKS", "0")
    ck, pk = lvm.synth.config(0, (328, 90 + 2 * rows + 3, 3))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 4, 0.0, exact=True)


@pytest.mark.parametrize("w,h,levels", [(328, 109, 3), (1000, 760, 5), (520, 77, 4)])
def test_laplace_emu_wave_strip_pyrdown(lvm, po, emu, w, h, levels, monkeypatch):
    """k_pyr_down_rows (the pyrDown of large planes) forced onto every level whose width allows it:
    edge lanes, partial strips, odd heights."""
    monkeypatch.setenv("LVM_ROWS_MIN_ELEMS", "0")
    ck, pk = lvm.synth.config(0, (w, h, levels))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 3, 0.0, exact=True)


@pytest.mark.parametrize("idx,w,h,levels", [(0, 328, 109, 3), (0, 1000, 70, 4), (0, 124 * 2 * 2, 40, 2), (3, 264, 90, 3), (3, 96, 77, 2)])
def test_emu_wave_strip_first_kernel(lvm, po, emu, idx, w, h, levels, monkeypatch):
    """k_down0_rows (u8 -> Lab / float -> pyrDown with DPP halo exchange between lanes) forced onto small frames:
    several strips per row (mirrored left / right edge groups, a strip ending exactly at the image edge), partly
    filled last strips, odd heights; Laplace (Lab) and Color (unscaled planes)."""
    monkeypatch.setenv("LVM_D0_MIN_TASKS", "0")
    ck, pk = lvm.synth.config(idx, (w, h, levels))
    if idx == 3:
        ck["fps"] = 15.0; pk["framerate"] = 15.0
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 4, 0.0, exact=True)


@pytest.mark.parametrize("w,h,levels,exact", [(328, 109, 3, True), (1000, 70, 4, True), (124 * 2 * 2, 40, 2, True), (264, 90, 3, False)])
def test_emu_fused_table_conversion_and_first_kernel(lvm, po, emu, w, h, levels, exact, monkeypatch):
    """k_down0_lut_rows (OpenCV's forward Lab table + pyrDown + the integer planes of the owned pixels in one pass)
    forced onto small frames: mirrored edge groups, a strip ending at the image edge, partly filled last strips, odd
    heights (the last source row owned by the last strip); per-frame calls, so the output kernel of every frame reads the
    planes this kernel stored.  exact=False: the default flavour's fma tap sums against the 1e-4 bar."""
    monkeypatch.setenv("LVM_D0_FUSED_WAVES", "1")
    ck, pk = lvm.synth.config(0, (w, h, levels))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 4, 0.0 if exact else 1e-4, exact=exact)


def test_emu_unfused_conversion_in_batches(lvm, po, emu, monkeypatch):
    """LVM_D0_FUSED=0: labconv.hip's conversion kernel + the plane-reading first kernels in temporal batches."""
    monkeypatch.setenv("LVM_D0_FUSED", "0")
    _frames_clip(lvm, po, emu, 0, 320, 180, 4, 1, (1, 6, 5))


@pytest.mark.parametrize("w,h,levels,ns,calls", [
    (320, 180, 2, 1, (1, 4, 3)),        # two levels: level 1 is the top live level (no cur_2), too large for the tail kernel
    (264, 74, 3, 1, (1, 6, 1, 2)),      # partial tiles right and below, per-frame calls in between
    (132, 70, 3, 2, (1, 5, 3)),         # level 2 with an odd width: level chain for level 2; two streams
    (160, 91, 3, 1, (1, 4, 4)),         # odd frame height (pyrUp with dsize = 2 n - 1 on both steps)
    (520, 150, 4, 1, (1, 9)),           # five tiles across: interior tiles without any border lane
    (128, 16, 2, 1, (1, 3, 3)),         # exactly one tile
])
def test_laplace_emu_level1_step_and_last_kernel_geometries(lvm, po, emu, w, h, levels, ns, calls):
    """k_lap_up at level 1 + k_lap_final_v4 over the geometries that the (deleted, round 5) fused level-1 kernel was checked on:
    bit-identical to the oracle over several calls, per-frame calls between temporal batches included."""
    _frames_clip(lvm, po, emu, 0, w, h, levels, ns, calls)


def test_laplace_emu_parameter_change_between_calls(lvm, po, emu):
    """amplification / chromAttenuation change between calls: the level-1 states carry over, frames keep matching the oracle"""
    ck, pk = lvm.synth.config(0, (264, 74, 3))
    def vary(t, q):
        if t >= 5:
            q["amplification"] = 35.0; q["chromAttenuation"] = 0.4
        return q
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 9, 0.0, exact=True, param_fn=vary)


def test_emu_fused_conversion_in_batches_two_streams(lvm, po, emu, monkeypatch):
    monkeypatch.setenv("LVM_D0_FUSED_WAVES", "1")
    _frames_clip(lvm, po, emu, 0, 264, 90, 3, 2, (1, 5, 4))


def _frames_clip(lvm, po, lib, idx, w, h, levels, n_streams, calls, over=None, clip_over=None):
    """lvm_process_device_frames: batches of consecutive frames (sizes in `calls`) of n_streams streams
    must give exactly the frames the oracle produces one by one."""
    ck, pk = lvm.synth.config(idx, (w, h, levels))
    pk.update(over or {})
    ck.update(clip_over or {})
    clips = [lvm.synth.Clip(seed=1234 + s, **ck) for s in range(n_streams)]
    P = po.make_params(**pk)
    cp = c_params(lvm, pk)
    ctx = lvm.Context(0, n_streams, lib)
    ctx.exact_lab(True)
    orcs = [po.Oracle() for _ in range(n_streams)]
    t = 0
    fb = w * h * 3
    for nf in calls:
        fin = np.stack([np.stack([c.frame(t + f) for c in clips]) for f in range(nf)])      # [frame][stream][h][w][3]
        fout = np.zeros_like(fin)
        produced = ctx.process_device_frames(cp, nf, fin.ctypes.data, w, h, 3, w * 3, fb, fb * n_streams,
                                             fout.ctypes.data, w * 3, fb, fb * n_streams)
        ctx.synchronize()
        for f in range(nf):
            for s_ in range(n_streams):
                ref, pr = orcs[s_].process(fin[f, s_], P)
                assert produced[f] == pr, (t + f, produced[f], pr)
                if pr:
                    assert np.array_equal(ref, fout[f, s_]), "frame %d stream %d" % (t + f, s_)
        t += nf
    ctx.close()


@pytest.mark.parametrize("w,h,levels,ns,calls", [(160, 90, 3, 1, (1, 4, 3, 1, 5)), (320, 180, 4, 1, (5, 6)),
                                                  (135, 77, 4, 2, (3, 3, 2)), (404, 300, 5, 1, (2, 7)), (64, 48, 1, 1, (3, 3)),
                                                  (200, 120, 4, 1, (1, 11, 17, 9))])   # deeper than the prefetch ring of k_lap_up
def test_laplace_emu_temporal_batches(lvm, po, emu, w, h, levels, ns, calls):
    _frames_clip(lvm, po, emu, 0, w, h, levels, ns, calls)


@pytest.mark.parametrize("w,h,levels,ns,calls", [(640, 360, 5, 1, (1, 8, 4)), (256, 256, 6, 1, (1, 4, 6)), (320, 182, 5, 2, (1, 5, 16)),
                                                  (576, 72, 4, 1, (1, 4, 4)), (512, 384, 7, 1, (1, 4))])   # 7 levels: five decoupled levels, the top one 8 x 6
def test_laplace_emu_split_levels_iir_and_collapse(lvm, po, emu, w, h, levels, ns, calls):
    """Temporal batches of >= 4 frames: levels 2 .. L-1 as ONE k_lap_iir_levels launch + ONE k_lap_collapse launch
    (several 64 x 32 tiles of level 2, odd level heights, levels of a few pixels, two streams, every ring depth)."""
    _frames_clip(lvm, po, emu, 0, w, h, levels, ns, calls)


@pytest.mark.parametrize("w,h,levels,ns,calls", [(640, 360, 5, 1, (1, 8, 4)), (256, 256, 6, 1, (1, 4, 6)), (320, 182, 5, 2, (1, 5, 16))])
def test_laplace_emu_split_from_level_2_still_matches(lvm, po, emu, w, h, levels, ns, calls, monkeypatch):
    """LVM_LAP_SPLIT_FROM=2: levels 2 .. L-1 all in the IIR + collapse launches (the default until round 6; since then level 2 is a fused
    band / IIR / collapse step when the pyramid has >= 5 levels and the two launches start at level 3)."""
    monkeypatch.setenv("LVM_LAP_SPLIT_FROM", "2")
    _frames_clip(lvm, po, emu, 0, w, h, levels, ns, calls)


def test_laplace_emu_level_chain_still_matches(lvm, po, emu, monkeypatch):
    """LVM_LAP_SPLIT=0 keeps the level-by-level chain of fused launches in temporal batches."""
    monkeypatch.setenv("LVM_LAP_SPLIT", "0")
    _frames_clip(lvm, po, emu, 0, 320, 180, 4, 1, (1, 8, 4))


@pytest.mark.parametrize("w,h,levels,calls", [(328, 109, 3, (1, 4, 8, 2, 3)), (200, 120, 4, (1, 16, 6))])
def test_laplace_emu_block_up_kernel_variants(lvm, po, emu, w, h, levels, calls):
    """k_lap_up_rows over batch lengths that select every ring depth (4, 2, 1), on odd heights
    (half-filled last block row)."""
    _frames_clip(lvm, po, emu, 0, w, h, levels, 1, calls)


def test_laplace_emu_tiled_up_kernel_still_matches(lvm, po, emu, monkeypatch):
    """LVM_UP_ROWS=0 selects the LDS-tiled k_lap_up (the kernel odd-width levels always use)."""
    monkeypatch.setenv("LVM_UP_ROWS", "0")
    _frames_clip(lvm, po, emu, 0, 200, 120, 4, 1, (1, 8, 5))


def test_frames_api_other_modes_fall_back_frame_by_frame(lvm, po, emu):
    _frames_clip(lvm, po, emu, 2, 96, 64, 3, 1, (4, 3))
    _frames_clip(lvm, po, emu, 3, 96, 64, 3, 1, (4, 3))


@pytest.mark.parametrize("w,h,levels,ns,calls", [(96, 64, 3, 1, (2, 5, 3)), (135, 77, 4, 2, (3, 4)), (160, 90, 5, 1, (6, 2)), (64, 48, 1, 1, (4,))])
def test_riesz_emu_temporal_batches(lvm, po, emu, w, h, levels, ns, calls):
    _frames_clip(lvm, po, emu, 2, w, h, levels, ns, calls)


@pytest.mark.parametrize("w,h,levels,ns,calls", [(64, 48, 2, 1, (18, 5, 7, 3)), (40, 30, 1, 2, (20, 6)), (80, 52, 3, 1, (17, 16, 9)),
                                                  (48, 32, 2, 1, (18, 40, 35))])   # calls longer than one batch (32 frames) are cut
def test_color_emu_temporal_batches(lvm, po, emu, w, h, levels, ns, calls):
    """fps 7 -> the window caps at 16 columns: once it is full the remaining frames of a call share
    launches (every frame of a batch sees the ring shifted by one column)."""
    _frames_clip(lvm, po, emu, 3, w, h, levels, ns, calls, over={"framerate": 7.0, "coLow": 0.4, "coHigh": 2.0}, clip_over={"fps": 7.0})


@pytest.mark.parametrize("lo,hi", [(0.8, 1.6), (2.9, 3.5), (0.8, 0.95)])
@pytest.mark.parametrize("thin8", ["1", "0"])
def test_color_emu_narrow_band_dft_eight_lanes_per_row(lvm, po, emu, lo, hi, thin8, monkeypatch):
    """k_col_dft_thin8 (at most four spectrum entries: eight lanes per window row) on temporal batches: three complex bins (0.8-1.6 Hz
    at 7 fps, 16-frame window), one complex bin + the Nyquist element (2.9-3.5 Hz), a single bin (0.8-0.95 Hz); LVM_COL_THIN8_DFT=0
    runs the one-thread-per-row kernel on the same clips.  Bit-exact against the oracle either way."""
    monkeypatch.setenv("LVM_COL_THIN8_DFT", thin8)
    _frames_clip(lvm, po, emu, 3, 64, 48, 2, 1, (18, 5, 7, 9), over={"framerate": 7.0, "coLow": lo, "coHigh": hi}, clip_over={"fps": 7.0})


# ---- ragged rows: strides larger than the row ---------------------------------------------------------
@pytest.mark.parametrize("idx,pad_in,pad_out", [(0, 4, 8), (0, 1, 3), (2, 4, 4), (2, 7, 1), (3, 8, 4), (3, 5, 5)])
def test_emu_padded_row_strides(lvm, po, emu, idx, pad_in, pad_out):
    """lvm_process_device on frames whose rows are padded (a cv::Mat ROI view has step > cols * channels):
    dword-aligned paddings keep the vectorised kernels, odd ones select the generic byte kernels; the padding
    bytes of the output must stay untouched."""
    w, h, levels = 96, 64, 3
    ck, pk = lvm.synth.config(idx, (w, h, levels))
    clip = lvm.synth.Clip(**ck)
    P = po.make_params(**pk)
    cp = c_params(lvm, pk)
    ctx = lvm.Context(0, 1, emu)
    ctx.exact_lab(True)
    orc = po.Oracle()
    si, so = w * 3 + pad_in, w * 3 + pad_out
    try:
        for t in range(6):
            f = clip.frame(t)
            buf_in = np.full((h, si), 0xAB, np.uint8)
            buf_in[:, :w * 3] = f.reshape(h, w * 3)
            buf_out = np.full((h, so), 0xCD, np.uint8)
            ref, pr = orc.process(f, P)
            pg = ctx.process_device(cp, buf_in.ctypes.data, w, h, 3, si, si * h, buf_out.ctypes.data, so, so * h)
            ctx.synchronize()
            assert pr == pg
            assert (buf_out[:, w * 3:] == 0xCD).all(), "padding bytes of the output were written"
            if pr:
                assert np.array_equal(buf_out[:, :w * 3].reshape(h, w, 3), ref), "frame %d" % t
    finally:
        ctx.close(); orc.close()


# ---- degenerate content: flat black / white regions and constant frames -----------------------------------
class _PatchedClip:
    """The synthetic clip with a flat black block, a flat white block and (from frame `const_from`) a constant frame:
    0/0 in the Riesz phase and amplitude steps (NaN patches, RieszPyramid.cpp:105-106,141), max == min in the colour
    normalisations (TemporalFilter.cpp:55, MagnifyCore.hpp:200-203)."""

    def __init__(self, clip, const_from=None):
        self.clip, self.const_from = clip, const_from

    def frame(self, t):
        f = self.clip.frame(t).copy()
        h, w = f.shape[:2]
        f[h // 8:h // 2, w // 8:w // 3] = 0
        f[h // 2:h - h // 8, w // 2:w - w // 8] = 255
        if self.const_from is not None and t >= self.const_from:
            f[...] = 77
        return f


@pytest.mark.parametrize("idx,const_from,size", [(0, None, (96, 64, 3)), (2, None, (96, 64, 3)), (3, None, (96, 64, 3)), (0, 5, (96, 64, 3)),
                                                   (2, 5, (96, 64, 3)), (3, 5, (96, 64, 3)),
                                                   (2, None, (200, 120, 3))])   # black block wider than the 9x9 + 13x13 supports: exact 0/0
def test_emu_flat_regions_and_constant_frames(lvm, po, emu, idx, const_from, size):
    ck, pk = lvm.synth.config(idx, size)
    if idx == 3:
        ck["fps"] = 15.0; pk["framerate"] = 15.0
    run_pair(lvm, po, emu, _PatchedClip(lvm.synth.Clip(**ck), const_from), pk, 9, 0.0, exact=True)


class _ConstClip:
    def __init__(self, h, w, v):
        self.f = np.full((h, w, 3), v, np.uint8)

    def frame(self, t):
        return self.f


@pytest.mark.parametrize("idx", [0, 2, 3])
def test_emu_fully_constant_clip(lvm, po, emu, idx):
    """Every frame the same constant: Color's output range collapses (max == min, 255 / 0 in convertTo:
    MagnifyCore.hpp:200-203), Riesz sees 0/0 in every phase difference, Laplace must return the Lab round trip."""
    ck, pk = lvm.synth.config(idx, (96, 64, 3))
    if idx == 3:
        pk["framerate"] = 15.0
    run_pair(lvm, po, emu, _ConstClip(64, 96, 131), pk, 8, 0.0, exact=True)


@pytest.mark.parametrize("idx,over", [(3, dict(coLow=5.0, coHigh=1.0)),                # colour: empty pass band (mask all zero)
                                       (3, dict(coLow=0.0, coHigh=0.3)),                # colour: lo == 0 -> 0.01, DC excluded, first bins
                                       (2, dict(coLow=0.5, coHigh=20.0)),               # Riesz: cutoff above Nyquist (Wn > 1)
                                       (2, dict(coLow=0.5, coHigh=15.0)),               # Riesz: cutoff exactly at Nyquist (Wn == 1)
                                       (2, dict(coLow=3.0, coHigh=1.0)),                # Riesz: hi < lo
                                       (0, dict(amplification=1000.0, chromAttenuation=1.0)),   # Laplace: far out of gamut
                                       (0, dict(amplification=0.0)),
                                       (2, dict(amplification=0.0, coWavelength=0.0))])  # Riesz: zero gain / zero threshold
def test_emu_extreme_parameters(lvm, po, emu, idx, over):
    ck, pk = lvm.synth.config(idx, (96, 64, 3))
    if idx == 3:
        ck["fps"] = 15.0; pk["framerate"] = 15.0
    pk.update(over)
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 8, 0.0, exact=True)


class _ShapeShifter:
    """Frames whose size / channel count changes mid-stream (the structural tracker must drop all state:
    MagnifyCore.hpp:53-65) and changes back."""

    def __init__(self, lvm, ck):
        self.a = lvm.synth.Clip(**ck)
        k2 = dict(ck); k2["w"], k2["h"] = 80, 48
        self.b = lvm.synth.Clip(**k2)

    def frame(self, t):
        if t < 4:
            return self.a.frame(t)
        if t < 7:
            return self.b.frame(t)                       # smaller frame
        if t < 10:
            return np.ascontiguousarray(self.a.frame(t)[:, :, 1])   # gray frame of the first size
        return self.a.frame(t)


@pytest.mark.parametrize("idx", [0, 2, 3])
def test_emu_size_and_channel_changes(lvm, po, emu, idx):
    ck, pk = lvm.synth.config(idx, (96, 64, 2))
    if idx == 3:
        ck["fps"] = 15.0; pk["framerate"] = 15.0
    run_pair(lvm, po, emu, _ShapeShifter(lvm, ck), pk, 13, 0.0, exact=True)


@pytest.mark.parametrize("w,h,levels", [(520, 40, 3), (260, 36, 2), (256, 34, 2), (772, 22, 2)])
def test_fast_final_kernel_strips_and_shortcut_paths(lvm, po, emu, w, h, levels):
    """Default-flavour last Laplace kernel (k_lap_final_fast): several wave strips per row, a last strip with a
    single lane / a half wave, DPP halo exchange with the halo loads of the first and last lane; a clip with very
    dark and saturated patches so that the general (select / spline) paths and the wave-uniform shortcuts both run,
    strong amplification so that outputs clamp at 0 and 255."""
    ck, pk = lvm.synth.config(0, (w, h, levels))
    pk["amplification"] = 60.0
    base = lvm.synth.Clip(**ck)

    class Patched:
        def frame(self, t):
            f = base.frame(t).copy()
            f[2:h // 2, 8:w // 4] = (f[2:h // 2, 8:w // 4] // 12)           # values 0..20: below the CIE threshold
            f[h // 2:h - 1, w // 2:w - 5] = 255 - (255 - f[h // 2:h - 1, w // 2:w - 5]) // 16
            f[:, w - 4:] = f[:, w - 4:] // 3
            return f
    worst = run_pair(lvm, po, emu, Patched(), pk, 7, 1e-4, exact=False, exact_lab=False, u8_frac=0.998)
    print("fast final kernel", (w, h, levels), worst)


@pytest.mark.parametrize("strip", ["0", "10", "54"])
@pytest.mark.parametrize("w,h,levels", [(520, 70, 3), (256, 41, 2), (1000, 24, 2), (128, 5, 2)])
def test_riesz_emu_wave_strip_stencils(lvm, po, emu, w, h, levels, strip, monkeypatch):
    """The LDS-free 9x9 strip kernels (forced onto these small planes): several 248-column strips per row (a full one, a
    partial one, a strip whose last lane owns the image's last column group), strips cut by the image height, odd heights,
    the strip height chosen by the launch code (0) or forced (10 rows: several strips per column; 54: the 1080p choice), a
    plane of five rows (the row index of the loads bounces off both edges inside one group of steps), bit-exact against the oracle."""
    monkeypatch.setenv("LVM_RZ_SPLIT_ROWS_MIN", "1")
    monkeypatch.setenv("LVM_RZ_SPLIT_STRIP", strip)
    ck, pk = lvm.synth.config(2, (w, h, levels))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 4, 0.0, exact=True)


@pytest.mark.parametrize("strip", ["0", "10", "64"])
@pytest.mark.parametrize("w,h,levels", [(512, 64, 4), (520, 70, 2), (1000, 24, 2), (8, 4, 2)])
def test_riesz_emu_wave_strip_collapse_and_output(lvm, po, emu, w, h, levels, strip, monkeypatch):
    """k_rz_collapse_strips forced onto small planes with even sizes: the collapse of levels 1 and 2 and the output kernel of a
    512 x 64 frame, several 248-column strips per row, strips cut by the image height, the smallest plane the kernel accepts
    (two column groups, four rows: every row index is a reflection), strip height chosen by the launch code or forced;
    OpenCV-order Lab arithmetic: bit-exact against the oracle."""
    monkeypatch.setenv("LVM_RZ_COLLAPSE_STRIPS_MIN", "1")
    monkeypatch.setenv("LVM_RZ_COLLAPSE_STRIP", strip)
    ck, pk = lvm.synth.config(2, (w, h, levels))
    run_pair(lvm, po, emu, lvm.synth.Clip(**ck), pk, 4, 0.0, exact=True)


def test_riesz_emu_strip_output_fast_flavour_equals_the_tiled_kernel(lvm, emu, monkeypatch):
    """The default flavour (reciprocal multiplies, packed Lab2BGR) of the strip output kernel against the tiled k_rz_final on the
    same frames: identical bytes and identical float frames (both evaluate the same operations per pixel)."""
    ck, pk = lvm.synth.config(2, (256, 48, 3))
    clip = lvm.synth.Clip(**ck)
    from helpers import c_params
    outs = []
    for strips in ("1", "0"):
        monkeypatch.setenv("LVM_RZ_COLLAPSE_STRIPS_MIN", "1")
        monkeypatch.setenv("LVM_RZ_COLLAPSE_STRIPS", strips)
        ctx = lvm.Context(0, 1, emu)
        ctx.keep_float(True)
        try:
            got = []
            for t in range(5):
                out, produced = ctx.process(clip.frame(t), c_params(lvm, pk))
                got.append((np.array(out, copy=True), ctx.read_float((48, 256, 3)).copy() if produced else None))
        finally:
            ctx.close()
        outs.append(got)
    for (a, fa), (b, fb) in zip(*outs):
        assert np.array_equal(a, b)
        assert (fa is None and fb is None) or np.array_equal(fa, fb)
