"""--attn self_mix under HIP graphs went NaN in a long CLI run where the eager loop did not: graph-served vs eager, in one process, bit for bit,
over more steps than the tests take.   usage: python scripts/diag_selfmix_graph.py [attn] [epochs] [steps] [mem]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_gpu_step_graph import _run
attn = sys.argv[1] if len(sys.argv) > 1 else "self_mix"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
mem = sys.argv[4] if len(sys.argv) > 4 else "MoCo"
qd = "fp32" if mem == "MoCoAtt" else "bf16"
for model, amp, kw in (("resnet8", None, dict(B=8, K=256, d=64, size=32, lr=0.05)), ("effiB0", "bf16", dict(B=16, K=1024, d=128, size=64, lr=0.05)),
                       ("effiB0", "bf16", dict(B=64, K=16384, d=512, size=128, lr=0.05))):
    VAL = os.environ.get("VALIDATE", "0") == "1"
    g = _run(True, model, True, "bf16", qd, amp, attn=attn, mem=mem, epochs=epochs, steps=steps, validate=VAL, data_on_device=VAL, **kw)
    e = _run(False, model, True, "bf16", qd, amp, attn=attn, mem=mem, epochs=epochs, steps=steps, validate=VAL, data_on_device=VAL, **kw)
    first = next((i for i, (a, b) in enumerate(zip(g["loss"], e["loss"])) if a.tobytes() != b.tobytes()), None)
    print(f"{attn} {mem} {model} amp={amp} {kw}: steps {len(g['loss'])}, replays {g['replays']}, first step whose loss differs {first}; "
          f"graph finite {bool(np.isfinite(g['loss']).all())} eager finite {bool(np.isfinite(e['loss']).all())}; last losses {g['loss'][-1]:.4f} / {e['loss'][-1]:.4f}", flush=True)
    if first is not None:
        lo = max(0, first - 2)
        print("    graph", [float(f"{v:.6f}") for v in g["loss"][lo:first + 4]], "\n    eager", [float(f"{v:.6f}") for v in e["loss"][lo:first + 4]], flush=True)
