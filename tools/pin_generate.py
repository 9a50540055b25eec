#!/usr/bin/env python3
"""Writes the golden vectors that PIN the oracle at the OpenCV boundary -- run on any box that has OpenCV 4 (see tools/pin_with_opencv.sh).

For BASELINE configs 0-3 (Laplace 640x360 L4 plumbing config, Laplace / Riesz / Color at their 1080p parameters) rendered at the sizes of
tests/test_refpin.py by the REAL reference stage (oracle/_ref/libref_magnify.so = the reference's own MagnificationProcessor.cpp +
magnification/*.cpp compiled where they lie and linked with this box's OpenCV), tests/golden/frames_cfg<k>.npz holds

    lut          the forward Lab table of THIS OpenCV build, recovered through cv::cvtColor (33^3 x 3 int16, RGB2Labprev order)
    sha256       one digest per frame of the u8 output (64 frames; "-" where the reference returned its input)
    produced     0 / 1 per frame
    idx, frames  eight output frames in full (indices spread over the clip)
    meta         size, levels, parameters, OpenCV version string, the generator's own sha256

The inputs are NOT stored: live-video-magnification_amd/synth.py regenerates them bit for bit (seed 1234).
tests/test_refpin.py (`-m refpin`) then checks BOTH the oracle and the library against these files -- loudly skipped while they do
not exist.  Nothing here runs in the product."""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

lvm = importlib.import_module("live-video-magnification_amd")

SIZES = {0: (640, 360, 4), 1: (480, 270, 5), 2: (480, 270, 5), 3: (480, 270, 4)}     # cfg -> (w, h, levels); must match tests/test_refpin.py
NFRAMES = {0: 64, 1: 64, 2: 64, 3: 160}                                                # colour: past its 128-frame window


def generate(out_dir, stand_in=False, cfgs=None, sizes=None, nframes=None):
    """stand_in=True renders with the RESTATEMENT instead of the real reference: only tests/test_refpin.py's plumbing test does that (into a
    temporary directory); the files say so in `meta` and the pin tests refuse them."""
    sizes = sizes or SIZES
    nframes = nframes or NFRAMES
    if not stand_in and not po.RefOracle.available():
        raise SystemExit("oracle/_ref/libref_magnify.so is not built: run `make -C oracle ref_full` on a box with OpenCV 4 first")
    lut = po.lab_lut_table() if stand_in else po.RefOracle.recover_lab_lut()
    ver = "unknown"
    try:
        import cv2
        ver = cv2.__version__
    except Exception:
        pass
    os.makedirs(out_dir, exist_ok=True)
    for cfg in (cfgs or sorted(sizes)):
        size = sizes[cfg]
        ck, pk = lvm.synth.config(cfg, size)
        clip = lvm.synth.Clip(**ck)
        P = po.make_params(**pk)
        ref = po.Oracle() if stand_in else po.RefOracle()
        n = nframes[cfg]
        keep_idx = sorted(set(int(round(x)) for x in np.linspace(n // 4, n - 1, 8)))
        digests, produced, kept = [], [], []
        for t in range(n):
            out, pr = ref.process(clip.frame(t), P)
            produced.append(int(pr))
            digests.append(hashlib.sha256(out.tobytes()).hexdigest() if pr else "-")
            if t in keep_idx:
                kept.append(out.copy())
        ref.close()
        meta = {"cfg": cfg, "size": list(size), "frames": n, "params": {k: (float(v) if not isinstance(v, int) else v) for k, v in pk.items()},
                "opencv": ver, "renderer": "oracle restatement (PLUMBING TEST ONLY -- pins nothing)" if stand_in else "reference + OpenCV",
                "generator_sha256": hashlib.sha256(open(__file__, "rb").read()).hexdigest()}
        np.savez_compressed(os.path.join(out_dir, "frames_cfg%d.npz" % cfg), lut=(lut if lut is not None else np.zeros(0, np.int16)),
                            sha256=np.array(digests), produced=np.array(produced, np.uint8), idx=np.array(keep_idx, np.int32),
                            frames=np.stack(kept), meta=np.array(json.dumps(meta)))
        print("cfg%d: %d frames, %d produced, LUT %s, %s" % (cfg, n, sum(produced), "recovered" if lut is not None else "not interpolating", meta["renderer"]))


def main():
    generate(os.path.join(ROOT, "tests", "golden"))


if __name__ == "__main__":
    main()
