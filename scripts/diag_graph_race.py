"""Are graph-served runs repeatable when the host can run ahead (batches resident on the device, as with the CLI's synthetic loader)?
usage: python scripts/diag_graph_race.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_gpu_step_graph import _run
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
KW = dict(B=32, K=1024, d=128, lr=0.05)
for label, graph, overlap, dev in (("graphs, device data, two streams", True, True, True), ("graphs, device data, one stream", True, False, True),
                                   ("graphs, host data, two streams", True, True, False), ("eager, device data, two streams", False, True, True)):
    runs = [_run(graph, "resnet8x4", overlap, "bf16", "bf16", None, data_on_device=dev, **KW) for _ in range(N)]
    keys = [r["loss"].tobytes() + r["delta"].tobytes() for r in runs]
    first = next((i for i in range(len(runs[0]["loss"])) if len({r["loss"][i].tobytes() for r in runs}) > 1), None)
    print(f"{label}: {N} runs, {len(set(keys))} distinct outcomes, first step whose loss differs {first}, replays {runs[0]['replays']}", flush=True)
