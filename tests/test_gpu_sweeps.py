"""Randomised parity sweeps on the GPU (scripts/sweep_k2.py, scripts/sweep_k1.py, scripts/sweep_k2_f32.py): the bf16-policy K2 call,
the K1 module, the exact-fp32 one-pass K2 and the multi-term entry against fp64 restatements over ragged shapes (K = 1 included),
large logits and every kernel width; K3 / K4 bit-exact against the numpy oracle (n > K, wraps, empty and block-edge tensors).  Each sweep runs in a child process (a few seconds)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,cases,seed", [("sweep_k2.py", 120, 11), ("sweep_k1.py", 80, 5), ("sweep_k2_f32.py", 150, 2024),
                                               ("sweep_k3_k4.py", 120, 1)])
def test_randomised_sweep(script, cases, seed):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(cases), str(seed)], capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    bad = [l for l in r.stdout.splitlines() if l.startswith("BAD")]
    assert r.returncode == 0 and not bad, "\n".join(bad[:10]) + r.stderr[-2000:]
