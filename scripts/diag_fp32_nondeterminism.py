"""Which part of the exact-fp32 loop is not bitwise repeatable run to run (with MIOpen's deterministic algorithms on)?
usage: python scripts/diag_fp32_nondeterminism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
torch.backends.cudnn.deterministic = True
from test_gpu_step_graph import _run

for label, kw in (("fp32 self overlap", dict(overlap=True, prec="fp32", qd="fp32")), ("fp32 self one stream", dict(overlap=False, prec="fp32", qd="fp32")),
                  ("fp32 K1/K2 over a bf16 queue, overlap", dict(overlap=True, prec="fp32", qd="bf16")),
                  ("bf16 K1/K2 over an fp32 queue, one stream", dict(overlap=False, prec="bf16", qd="fp32"))):
    runs = [_run(False, "resnet8", kw["overlap"], kw["prec"], kw["qd"], None) for _ in range(4)]
    first = None
    for i in range(len(runs[0]["loss"])):
        if len({r["loss"][i].tobytes() for r in runs}) > 1:
            first = i
            break
    kd_first = None
    for i in range(len(runs[0]["loss_kd"])):
        if len({r["loss_kd"][i].tobytes() for r in runs}) > 1:
            kd_first = i
            break
    print(f"{label}: distinct traces {len({r['loss'].tobytes() for r in runs})}, first step whose loss differs: {first}, loss_kd: {kd_first}", flush=True)
