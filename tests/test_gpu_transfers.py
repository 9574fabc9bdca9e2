"""The GPU never addresses pageable caller memory (fccqp_capi.cpp: copy_in / copy_out; include/fccqp.h at fccqp_solve).

For a copy of 1 MiB or more between pageable host memory and the device the HIP runtime page-locks the CALLER's pages on
the fly and lets a copy kernel read / write them in place, and it caches such locks by address.  The two process aborts
this library has seen (one GPU-suite run in eleven in round 4, one in six in round 5: "Memory access fault by GPU ... on
address 0x64bef7d16000", a malloc-heap address) were host-pointer solves on exactly that path.  The library now bounces
every pageable caller array through page-locked buffers of its own.  The runtime says in its log which path each of its
copies took; this test runs host-pointer solves, result fetches and state transplants with arrays of up to 190 MB in a
child process under AMD_LOG_LEVEL=4 and requires that NO copy used the caller-pinning path -- and, as a control that the
log is being read, that the small staging copies the probe does not make are absent too while the API trace is there.
The reference has no counterpart: its inputs never leave the host (src/fcc_qp.cpp:146-150)."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_transfer_lets_the_runtime_page_lock_caller_memory():
    with tempfile.TemporaryDirectory(prefix="fccqp_copy_probe_") as tmp:
        log = os.path.join(tmp, "hip.log")
        env = dict(os.environ, AMD_LOG_LEVEL="4", AMD_LOG_LEVEL_FILE=log)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pageable_copy_probe.py")], env=env, cwd=ROOT,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "MARK done" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
        text = ""
        for f in os.listdir(tmp):                       # (the runtime appends the pid to the file name)
            if f.startswith("hip.log"):
                with open(os.path.join(tmp, f), errors="replace") as fh:
                    text += fh.read()
    assert "hipMemcpyAsync" in text, "control: the runtime's log is being captured (API trace at level 4)"
    pinned = [l for l in text.splitlines() if "HSA Copy Using Pinned resource" in l]
    assert not pinned, f"the runtime page-locked caller memory for {len(pinned)} copies, e.g. {pinned[0][-80:]}"
    staged = [l for l in text.splitlines() if "HSA Copy Using Staging resource" in l]
    assert not staged, f"{len(staged)} copies went through the runtime's staging buffers: a caller array reached hipMemcpy"
