"""Round-4 diagnostic: loss trajectories of the 32-sample / 64-pixel miniature (the rehearsal tests' configuration) at lr 0.05,
step graphs on and off, a few repetitions each -- is an occasional NaN there a property of the configuration or of the graph path?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_step_graph import _run
for rep in range(3):
    for g in (True, False):
        r = _run(g, "effiB0", True, "bf16", "bf16", "bf16", epochs=2, steps=15, B=32, K=4096, d=512, size=64, lr=0.05)
        l = r["loss"]
        print("graphs" if g else "eager ", "rep", rep, "finite" if np.all(np.isfinite(l)) else "NON-FINITE", "max %.2f" % np.nanmax(l),
              "last5", np.round(l[-5:], 2), "kd last", np.round(r["loss_kd"][-3:], 2), flush=True)
