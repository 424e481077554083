"""K2 under the exact-fp32 policy: one-pass kernel time (fp32 queue).  usage: python scripts/bench_k2_f32.py [B] [d] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
torch.manual_seed(0)
q = torch.nn.functional.normalize(torch.randn(B, d, device="cuda"))
k = torch.nn.functional.normalize(q + 0.3 * torch.randn(B, d, device="cuda"))
queue = torch.nn.functional.normalize(torch.randn(K, d, device="cuda"))
for grad in (True, False):
    qq = q.clone().requires_grad_(grad)
    for _ in range(3):
        ops.infonce_fused(qq, k, queue, 0.15, "fp32")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.infonce_fused(qq, k, queue, 0.15, "fp32")
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = (4.0 if grad else 2.0) * B * d * (K + 1)
    print(f"K2 fp32 policy B={B} d={d} K={K} dq={grad}: {ms*1e3:.0f} us/call = {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3*100:.1f}% of the 157 TF f32 matrix rate)", flush=True)
