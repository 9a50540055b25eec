"""The C-ABI library loads and exports every symbol include/lvm_hip.h declares; without a GPU
it refuses to create a context (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lvm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lvm_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface(lvm):
    syms = declared_symbols()
    for s in ("lvm_create", "lvm_destroy", "lvm_reset", "lvm_process", "lvm_process_device", "lvm_last_error"):
        assert s in syms
    assert sorted(lvm.binding.SYMBOLS) == syms


def test_library_exports_every_declared_symbol(lvm):
    path = lvm.binding.LIB_PATH
    assert os.path.exists(path), "liblvm_hip.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    for s in declared_symbols():
        assert hasattr(lib, s), "missing export: " + s


def test_params_struct_layout_matches_header(lvm):
    # int32 x2, double x6, uint64: 64 bytes, no padding surprises
    assert ctypes.sizeof(lvm.LvmParams) == 64
    assert lvm.LvmParams.amplification.offset == 8 and lvm.LvmParams.preprocess_key.offset == 56


def test_scalar_helpers_match_reference_answers(lvm):
    lib = lvm.load()
    assert lib.lvm_max_levels(1920, 1080) == 8 and lib.lvm_max_levels(640, 360) == 7 and lib.lvm_max_levels(3840, 2160) == 9
    for fps, v in [(15, 32), (24, 64), (30, 64), (33, 128), (60, 128), (65, 256)]:
        assert lib.lvm_optimal_buffer_size(fps) == v
    a = (ctypes.c_double * 3)(); b = (ctypes.c_double * 3)()
    lib.lvm_butterworth2(0.5 / 15, a, b)
    assert abs(a[1] - -1.8521464853959357) < 1e-14 and abs(b[0] - 0.0025505351585362926) < 1e-16
    # B_alg Laplace 1080p L6 = 45.59 MB, Riesz = 255.5 MB, Color T=128 = 13.23 MB
    assert abs(lib.lvm_algorithmic_bytes(0, 1920, 1080, 3, 6, 30.0) / 1e6 - 45.59) < 0.01
    assert abs(lib.lvm_algorithmic_bytes(1, 1920, 1080, 3, 6, 30.0) / 1e6 - 255.5) < 0.1
    assert abs(lib.lvm_algorithmic_bytes(2, 1920, 1080, 3, 6, 60.0) / 1e6 - 13.23) < 0.01


def test_no_gpu_means_loud_failure(lvm):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lvm.LvmError):
        lvm.Context(0, 1)
