"""The emulation executes the workgroups of a launch one after the other and the work-items of a workgroup in ascending order between
barriers.  HIPEMU_ORDER=reverse turns both around: a kernel whose output depends on that order -- a missing barrier, a workgroup that
reads what another workgroup of the same launch writes (how the level-1 states of the fused last kernel raced in round 4) -- then
differs from the oracle.  One clip per mode, the temporal batches and the strip kernels here; the whole emulation suite (184 tests,
two minutes) ran green in reverse order at the end of round 5: `HIPEMU_ORDER=reverse python -m pytest tests/test_emu_parity.py
tests/test_emu_bench_pattern.py`."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulation_results_do_not_depend_on_the_execution_order():
    env = dict(os.environ, HIPEMU_ORDER="reverse")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "tests/test_emu_parity.py", "tests/test_emu_bench_pattern.py", "-k",
                        "laplace_emu_bit_exact and 135 or riesz_emu_bit_exact and 135 or color_emu_bit_exact and 135 or temporal_batches and 64-48 "
                        "or wave_strip_collapse and 512 or wave_strip_stencils or strip_blur_in_temporal_batches or wave_strip_pyrdown or eight_lanes or poisoned_memory and size2"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "passed" in r.stdout
