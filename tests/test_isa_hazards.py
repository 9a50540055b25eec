"""Static check of the gfx950 ISA of every device translation unit for the one instruction form that bit this library on hardware:
a buffer store of more than 64 bits with a REGISTER in its soffset field.  The store reads its data registers some cycles after it
issues; the compiler pads a following write to those registers with wait states only when the soffset field holds no register
(GCNHazardRecognizer), and on gfx950 the hazard exists either way -- `buffer_store_dwordx3 v[4:6], v64, s[40:43], s14 offen` followed by
`v_mov_b64 v[6:7], ...` lost the third dword for the last four lanes of every 16 ("Riesz, round 4").  csrc/lvm_gfx950.h
keeps the whole offset in the vector register for such stores; this test keeps it that way for code that does not go through the helpers."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "live-video-magnification_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
UNITS = ["lvm_api.hip", "labconv.hip", "laplace.hip", "riesz.hip", "color.hip", "preprocess.hip", "compose.hip"]
NOSLP = {"laplace.hip", "riesz.hip"}                      # as in csrc/Makefile


def _isa(unit, out_dir):
    out = os.path.join(out_dir, unit + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-I" + os.path.join(ROOT, "include"), "-I" + SRC, "-S", "--cuda-device-only", "-o", out, os.path.join(SRC, unit)]
    if unit in NOSLP:
        cmd.insert(1, "-fno-slp-vectorize")
    subprocess.run(cmd, check=True, capture_output=True, timeout=1200)
    return open(out).read()


_CACHE = {}


def _all_isa(tmp_path):
    if "t" not in _CACHE:
        with ThreadPoolExecutor(max_workers=4) as ex:
            _CACHE["t"] = list(ex.map(lambda u: _isa(u, str(tmp_path)), UNITS))
    return _CACHE["t"]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_kernel_spills_to_scratch(tmp_path):
    """Every kernel of the library fits its registers: `.amdhsa_private_segment_fixed_size 0`.  A spill puts scratch loads and stores into
    the loops of the strip kernels (and their s_waitcnt vmcnt(0)); it has crept in twice unnoticed.  Allowed: the OpenCV-order debug
    flavour of the Riesz output strips (lvm_debug_exact_lab: IEEE divisions, ~100 bytes), which is not a production path."""
    allowed = re.compile(r"k_rz_collapse_stripsILb1ELi1E")          # <FINAL = true, FL = FL_LUT_EXACT, *>
    spills = []
    for unit, text in zip(UNITS, _all_isa(tmp_path)):
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
            size = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2)).group(1))
            if size and not allowed.search(m.group(1)):
                spills.append((unit, m.group(1), size))
    assert not spills, spills


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_wide_buffer_store_with_a_register_soffset(tmp_path):
    texts = _all_isa(tmp_path)
    wide = re.compile(r"^\s*buffer_store_(dwordx3|dwordx4)\s+(.*)$", re.M)
    seen, bad = 0, []
    for unit, text in zip(UNITS, texts):
        for m in wide.finditer(text):
            seen += 1
            ops = [o.strip() for o in m.group(2).split(",")]
            soffset = ops[3].split()[0]                   # vdata, vaddr, srsrc, soffset [modifiers]
            if soffset != "0":
                bad.append((unit, m.group(0).strip()))
    assert seen > 0, "the strip kernels' stores were expected in the ISA"
    assert not bad, bad[:5]
