"""Micro-benchmark of K1 (batch-token attention fwd / fwd+bwd) through the C ABI."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd.MoMA.criterion_moco_att import Attention
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4
for prec in ("bf16", "fp32"):
    att = Attention(d, num_heads=H, qkv_bias=True, precision=prec).cuda()
    x = torch.nn.functional.normalize(torch.randn(N, d, device="cuda")).requires_grad_(True)
    for mode in ("fwd", "fwd+bwd"):
        def run():
            if mode == "fwd":
                with torch.no_grad():
                    att(x)
            else:
                y = att(x); y.sum().backward()
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        print(f"K1 N={N} d={d} H={H} {prec} {mode}: {e0.elapsed_time(e1)/20*1e3:.0f} us", flush=True)
