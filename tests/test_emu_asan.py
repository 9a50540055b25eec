"""One small clip of every mode through the AddressSanitizer build of the CPU emulation (tools/emu_asan.sh): each global
access of each kernel and the host code of the C ABI checked against the bounds of its heap block.  The full run of every
emulation-based test under ASan (147 tests, ~17 min; clean at the end of round 2) is `tools/emu_asan.sh` without arguments."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulation_build_is_clean_under_address_sanitizer(tmp_path):
    env = dict(os.environ, LVM_ASAN_DIR=str(tmp_path))
    r = subprocess.run([os.path.join(ROOT, "tools", "emu_asan.sh"), "tests/test_emu_parity.py", "-k",
                        "laplace_emu_bit_exact and 135 or riesz_emu_bit_exact and 135 or color_emu_bit_exact and 135 or temporal_batches and 64-48"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "passed" in r.stdout and "AddressSanitizer" not in r.stdout + r.stderr
