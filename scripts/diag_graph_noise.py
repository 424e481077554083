"""Run-to-run noise of the eager loop against itself, next to graph-served vs eager, for the loop variants of tests/test_gpu_step_graph.py
(MIOpen's weight-gradient kernels are not bitwise reproducible: how far do two EAGER runs part over 15 steps?).
usage: python scripts/diag_graph_noise.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_gpu_step_graph import _run_impl as _run       # (the searched, non-deterministic MIOpen algorithms: the noise this script measures)

for attn, mem, prec, qd in (("self_nomix", "MoCo", "bf16", "bf16"), ("self_mix", "MoCo", "bf16", "bf16"), ("all", "MoCoAtt", "bf16", "fp32"),
                            ("dual", "MoCoAtt", "fp32", "fp32"), ("self", "MoCo", "bf16", "bf16")):
    for lr in (0.02, 2e-3):
        e1 = _run(False, "resnet8", True, prec, qd, None, attn=attn, mem=mem, lr=lr)
        e2 = _run(False, "resnet8", True, prec, qd, None, attn=attn, mem=mem, lr=lr)
        g = _run(True, "resnet8", True, prec, qd, None, attn=attn, mem=mem, lr=lr)
        rel = lambda a, b: float(np.abs(a["loss"] - b["loss"]).max() / np.abs(b["loss"]).max())
        dl = lambda a, b: float(np.linalg.norm(a["delta"] - b["delta"]) / np.linalg.norm(b["delta"]))
        print(f"{attn:10s} {mem:8s} {prec} lr {lr:g}: loss eager/eager {rel(e1, e2):.2e}  graph/eager {rel(g, e1):.2e} | delta eager/eager {dl(e1, e2):.3f} "
              f"graph/eager {dl(g, e1):.3f} | replays {g['replays']}", flush=True)
