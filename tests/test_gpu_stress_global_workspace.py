"""Repeated-run stress of the generic kernel's global-workspace path (198 < N <= 512) on poisoned device memory
(FCCQP_DEBUG_POISON=1: fresh allocations are NaN-filled), in a child process because the poison switch is read once per
process: 40 fresh handles per shape, a cold and a warm solve each, every one within 1e-9 of the oracle and bit-identical
to the first.  Before the fence fix (agent_release / agent_acquire in fcc_qp_amd/csrc/fccqp_device.h) this failed on
EVERY repetition for N > 256.  Reference behaviour: FCCQP::Solve, /root/reference/src/fcc_qp.cpp:114-191."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forty_fresh_handles_per_shape_on_poisoned_memory():
    env = dict(os.environ, FCCQP_DEBUG_POISON="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "n512_probe.py"), "40"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("(")]
    assert r.returncode == 0 and len(lines) == 7, r.stdout[-3000:] + r.stderr[-2000:]
    assert all("wrong 0/40" in l and "bit-identical to the first 0/40" in l for l in lines), "\n".join(lines)
