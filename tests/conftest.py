import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "refpin: golden vectors rendered by the REAL reference + OpenCV 4 (tools/pin_with_opencv.sh); skipped, loudly, while they do not exist")


@pytest.fixture(scope="session")
def lvm():
    import importlib
    return importlib.import_module("live-video-magnification_amd")


@pytest.fixture(scope="session")
def po():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def emu(lvm):
    """The product sources compiled against the HIP emulation header (tests/emu): kernel-logic
    checks on CPU.  Test infrastructure only."""
    import ctypes
    import subprocess
    here = os.path.join(ROOT, "tests", "emu")
    alt = os.environ.get("LVM_EMU_LIB")          # e.g. the AddressSanitizer build of tools/emu_asan.sh
    if alt:
        return lvm.bind(ctypes.CDLL(alt))
    import fcntl
    os.makedirs(os.path.join(here, "_build"), exist_ok=True)
    with open(os.path.join(here, "_build", ".lock"), "w") as lk:       # pytest-xdist workers build one at a time
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.check_call([os.path.join(here, "build_emu.sh")])
    return lvm.bind(ctypes.CDLL(os.path.join(here, "_build", "liblvm_emu.so")))


@pytest.fixture(scope="session")
def hip(lvm):
    """The gfx950 library through the C ABI; fails loudly when missing."""
    return lvm.load()
