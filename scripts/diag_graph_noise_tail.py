"""How far apart do N eager runs of the loop variants of tests/test_gpu_step_graph.py end up (the tail the tests' tolerances must clear)?
Prints, per variant, the largest pairwise |loss difference| relative to the tolerance form of the tests (|a - b| / (1 + |b|)), the largest
relative parting of the weight updates and of the queue / attention weights.   usage: python scripts/diag_graph_noise_tail.py [N]"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_gpu_step_graph import _run as _run_det, _run_impl

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
# DET=1: MIOpen's default algorithms instead of the searched ones (what the tests run under since round 6)
_run = _run_det if os.environ.get("DET") == "1" else _run_impl
WITH_GRAPH = os.environ.get("GRAPH") == "1"         # also one graph-served run per case, against the first eager run
# (attn, mem, prec, queue dtype, overlap, model, amp, extra keyword arguments of _run)
CASES = [("self", "MoCo", "bf16", "bf16", True, "resnet8", None, {}), ("self", "MoCo", "bf16", "fp32", True, "resnet8", None, {}),
         ("self", "MoCo", "fp32", "fp32", True, "resnet8", None, {}),
         ("self_nomix", "MoCo", "bf16", "bf16", True, "resnet8", None, {}), ("self_mix", "MoCo", "bf16", "bf16", True, "resnet8", None, {}),
         ("all", "MoCoAtt", "bf16", "fp32", True, "resnet8", None, {}), ("qk", "MoCoAtt", "bf16", "fp32", True, "resnet8", None, {}),
         ("dual", "MoCoAtt", "fp32", "fp32", True, "resnet8", None, {}),
         ("self", "MoCo", "bf16", "bf16", True, "effiB0", "bf16", dict(B=16, K=1024, d=128, size=64, lr=2e-4)),
         ("self", "MoCo", "bf16", "bf16", True, "effiB0", "bf16", dict(B=16, K=1024, d=128, size=64, lr=0.02)),
         ("self", "MoCo", "bf16", "bf16", True, "resnet8", "fp16", dict(scale0=2.0 ** 10, lr=2e-4)),
         ("self", "MoCo", "bf16", "bf16", True, "resnet8", "fp16", dict(scale0=2.0 ** 22, lr=0.02))]
for attn, mem, prec, qd, overlap, model, amp, kw in CASES:
    runs = [_run(False, model, overlap, prec, qd, amp, attn=attn, mem=mem, **kw) for _ in range(N)]
    worst = dict(loss=0.0, loss_kd=0.0, delta=0.0, memory=0.0, atts_q=0.0)
    for a, b in itertools.combinations(runs, 2):
        for key in ("loss", "loss_kd"):
            worst[key] = max(worst[key], float((np.abs(a[key] - b[key]) / (1.0 + np.abs(b[key]))).max()))
        worst["delta"] = max(worst["delta"], float(np.linalg.norm(a["delta"] - b["delta"]) / np.linalg.norm(b["delta"])))
        worst["memory"] = max(worst["memory"], float(np.abs(a["memory"] - b["memory"]).max()))
        worst["atts_q"] = max(worst["atts_q"], float(np.abs(a["atts_q"] - b["atts_q"]).max()))
    distinct = len({r["loss"].tobytes() for r in runs})
    if WITH_GRAPH:
        g = _run(True, model, overlap, prec, qd, amp, attn=attn, mem=mem, **kw)
        gl = max(float((np.abs(g[k] - runs[0][k]) / (1.0 + np.abs(runs[0][k]))).max()) for k in ("loss", "loss_kd"))
        gd = float(np.linalg.norm(g["delta"] - runs[0]["delta"]) / np.linalg.norm(runs[0]["delta"]))
        print(f"    graph vs eager[0]: losses {gl:.2e} delta {gd:.3f} replays {g['replays']}", flush=True)
    print(f"{attn:10s} {mem:8s} {prec}/{qd} {model} amp={amp} {kw.get('lr', 0.02):g}: {N} eager runs, {distinct} distinct loss traces | worst pair: loss {worst['loss']:.2e} loss_kd {worst['loss_kd']:.2e} "
          f"delta {worst['delta']:.3f} memory {worst['memory']:.2e} atts_q {worst['atts_q']:.2e}", flush=True)
