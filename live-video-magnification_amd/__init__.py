"""live-video-magnification_amd: MI355X-native Eulerian video magnification core.

The product is liblvm_hip.so (csrc/, C ABI in include/lvm_hip.h); this package is the
Python-side mirror of the reference's operator interface plus bench/test input generators.
Import with importlib (the directory name contains a hyphen):

    lvm = importlib.import_module("live-video-magnification_amd")
"""
from .binding import (Context, LvmError, LvmOverlayLabel, LvmParams, LvmPreprocessParams, MagnificationMode, MagnificationParams,  # noqa: F401
                      MagnificationProcessor, PreprocessParams, ProcessingChain, ProcessorConfig, bind, load, to_c_params,
                      to_c_preprocess)
from . import sharding, synth, tiling  # noqa: F401
