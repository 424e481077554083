"""CPU-side audit of the compiled K2 kernels (scripts/audit_k2_isa.py): no spills in the one-pass kernels, the hand-counted
asm Q loads untouched until their wait, no compiler-inserted full drain in front of the first score MFMA.  Cross-compiles
csrc/infonce_fused.hip for gfx950 (about 15 s); skipped when hipcc is absent."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_k2_isa_audit():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "audit_k2_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
