"""GPU-box diagnostic: where do a graph-served run and an eager run of the same seeded loop part?  Prints per-step |loss| differences
eager-vs-eager (run-to-run noise) and graph-vs-eager for the cases of tests/test_gpu_step_graph.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_step_graph import _run

cases = [("resnet8", True, "fp32", "fp32", None), ("resnet8", False, "bf16", "bf16", None), ("resnet8", True, "bf16", "fp32", None)]
for c in cases:
    e1, e2, g1 = _run(False, *c), _run(False, *c), _run(True, *c)
    rel = lambda a, b: np.linalg.norm(a["delta"] - b["delta"]) / np.linalg.norm(b["delta"])
    print(c, "update rel: eager/eager", rel(e1, e2), "graph/eager", rel(g1, e1))
    print("  eager/eager |dloss|:", np.array2string(np.abs(e1["loss"] - e2["loss"]), precision=2))
    print("  graph/eager |dloss|:", np.array2string(np.abs(g1["loss"] - e1["loss"]), precision=2))
    print("  graph/eager |dloss_kd|:", np.array2string(np.abs(g1["loss_kd"] - e1["loss_kd"]), precision=2))
