"""One small clip of every mode (and the bench's batch pattern) through the emulation build whose automatic variables are pre-filled
with a byte pattern (tools/emu_uninit.sh, clang -ftrivial-auto-var-init=pattern): a local that is read before it is written -- in host
code or in a kernel -- computes garbage there every time instead of once in a while.  The full run of every emulation-based test on
that build (226 tests, ~7 min; clean at the end of round 4) is `tools/emu_uninit.sh` without arguments."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = os.environ.get("LVM_CLANGXX", "/opt/rocm/lib/llvm/bin/clang++")


@pytest.mark.skipif(not os.path.exists(CLANG), reason="no clang++")
def test_emulation_build_with_pattern_filled_locals_matches_the_oracle(tmp_path):
    env = dict(os.environ, LVM_UNINIT_DIR=str(tmp_path))
    r = subprocess.run([os.path.join(ROOT, "tools", "emu_uninit.sh"), "tests/test_emu_parity.py", "tests/test_emu_bench_pattern.py", "-k",
                        "laplace_emu_bit_exact and 135 or riesz_emu_bit_exact and 135 or color_emu_bit_exact and 135 or temporal_batches and 64-48 "
                        "or poisoned_memory and size2"],
                       capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "passed" in r.stdout
