"""ISA-level invariants of the persistent conv kernel that the compiler does not guarantee (see tools/check_lds_hazard.py):
no instruction touches a register of a hand-issued LDS read that is still in flight, and every DMA-publishing barrier
has its `s_waitcnt vmcnt(0)`.  Needs hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_conv_kernel_isa_invariants():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_hazard.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 hazards" in r.stdout
