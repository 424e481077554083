"""Micro-benchmark of K1 (batch-token attention fwd / fwd+bwd / two modules grouped) through the C ABI."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd.MoMA.criterion_moco_att import Attention
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4
precs = sys.argv[4].split(",") if len(sys.argv) > 4 else ("bf16", "fp32")
for prec in precs:
    att = Attention(d, num_heads=H, qkv_bias=True, precision=prec).cuda()
    att2 = Attention(d, num_heads=H, qkv_bias=True, precision=prec).cuda()
    xdt = torch.bfloat16 if os.environ.get("K1_X_BF16") else torch.float32
    x = torch.nn.functional.normalize(torch.randn(N, d, device="cuda")).to(xdt).requires_grad_(True)
    x2 = torch.nn.functional.normalize(torch.randn(N, d, device="cuda")).to(xdt)
    for mode in ("fwd", "fwd+bwd", "fwd x2 grouped"):
        def run():
            if mode == "fwd":
                with torch.no_grad():
                    att(x)
            elif mode == "fwd+bwd":
                y = att(x); y.sum().backward()
            else:
                with torch.no_grad():
                    Attention.forward_group([att, att2], [x, x2])
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        print(f"K1 N={N} d={d} H={H} {prec} {mode}: {e0.elapsed_time(e1)/20*1e3:.0f} us", flush=True)
