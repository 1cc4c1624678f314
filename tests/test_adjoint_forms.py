"""
r6: the staggered advection adjoints run ALL components per launch (pass A, B, C and the MacCormack correction's adjoint: three launches per call instead
of nine). The bodies are the per-component kernels': both launch forms must give the same BITS. The form is read from the environment once per process (PHIHIP_ADJOINT_ALL),
so each side runs in its own subprocess (tests/adjoint_forms_probe.py prints a SHA-256 per seeded case).
"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _hashes(where, all_components):
    env = dict(os.environ, PHIHIP_ADJOINT_ALL="1" if all_components else "0")
    out = subprocess.run([sys.executable, os.path.join(HERE, "adjoint_forms_probe.py"), where], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("(")]
    assert len(lines) == 3, out.stdout
    return lines


def test_adjoint_launch_forms_same_bits_emulation(emu_library):
    assert _hashes("emu", True) == _hashes("emu", False)


@pytest.mark.gpu
def test_adjoint_launch_forms_same_bits_gpu():
    assert _hashes("gpu", True) == _hashes("gpu", False)
