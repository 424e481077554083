"""Why do two `--reproducible` CLI runs differ?  The in-process loop at the CLI test's shapes, three runs per setting.
usage: python scripts/diag_cli_nondeterminism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from test_gpu_step_graph import _run

def report(label, **kw):
    runs = [_run(False, "resnet8x4", True, "bf16", "bf16", None, **kw) for _ in range(3)]
    first = next((i for i in range(len(runs[0]["loss"])) if len({r["loss"][i].tobytes() for r in runs}) > 1), None)
    d = max(float(np.abs(runs[0]["delta"] - r["delta"]).max()) for r in runs[1:])
    print(f"{label}: distinct {len({r['loss'].tobytes() for r in runs})}, first differing step {first}, max |delta diff| {d:.3e}", flush=True)

report("resnet8x4 B=8 K=256 d=64")
report("resnet8x4 B=32 K=1024 d=128", B=32, K=1024, d=128)
report("resnet8x4 B=32 K=1024 d=128 lr 0.05", B=32, K=1024, d=128, lr=0.05)
torch.use_deterministic_algorithms(True, warn_only=True)
report("same + torch.use_deterministic_algorithms", B=32, K=1024, d=128, lr=0.05)
