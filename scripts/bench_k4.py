"""K4 (multi-tensor EMA) ALONE on one stream: EfficientNet-B0 parameter set (213 tensors, 4.01 M fp32), 12 B / parameter.
usage: python scripts/bench_k4.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd.backbones import model_dict
from moma_amd.learning.contrast_trainer import ContrastTrainer
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
a, b = model_dict["effiB0"](num_classes=4).cuda(), model_dict["effiB0"](num_classes=4).cuda()
P = sum(p.numel() for p in a.parameters())
for _ in range(10): ContrastTrainer.momentum_update(a, b, 0.999)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): ContrastTrainer.momentum_update(a, b, 0.999)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
print(f"K4 alone: {len(list(a.parameters()))} tensors, {P} params, {us:.1f} us per call back to back (host-paced), "
      f"{12 * P / us / 1e6:.2f} TB/s algorithmic (12 B/param)", flush=True)
