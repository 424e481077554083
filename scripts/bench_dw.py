"""Micro-benchmark of the depthwise-conv kernels on the EfficientNet-B0 layer shapes (B=256, bf16):
HIP-event time per call and achieved GB/s on the algorithmic bytes (fwd: x + y; bwd-data: dy + dx; bwd-weight: x + dy)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd import ops
LAYERS = [(32, 112, 3, 1), (96, 112, 3, 2), (144, 56, 3, 1), (144, 56, 5, 2), (240, 28, 5, 1), (240, 28, 3, 2),
          (480, 14, 3, 1), (480, 14, 5, 1), (672, 14, 5, 1), (672, 14, 5, 2), (1152, 7, 5, 1), (1152, 7, 3, 1)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tot = [0.0, 0.0, 0.0]
for C, H, K, S in LAYERS:
    OH = math.ceil(H / S)
    ph = max((OH - 1) * S + K - H, 0)
    x = torch.randn(N, C, H, H, device="cuda").bfloat16().requires_grad_(True)
    w = torch.randn(C, 1, K, K, device="cuda").requires_grad_(True)
    dy = torch.randn(N, C, OH, OH, device="cuda").bfloat16()
    def t(fn, n=10):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tf = t(lambda: ops.dwconv(x, w, S, ph // 2, ph // 2, OH, OH))
    # needs_input_grad follows requires_grad of the inputs, so build one graph per gradient
    y1 = ops.dwconv(x, w.detach(), S, ph // 2, ph // 2, OH, OH)
    tdx = t(lambda: torch.autograd.grad(y1, x, dy, retain_graph=True))
    y2 = ops.dwconv(x.detach(), w, S, ph // 2, ph // 2, OH, OH)
    tdw = t(lambda: torch.autograd.grad(y2, w, dy, retain_graph=True))
    bx, by = x.numel() * 2, dy.numel() * 2
    print(f"C={C:5d} {H:3d}x{H:<3d} k{K} s{S}: fwd {tf:7.1f} us {((bx+by)/tf/1e3):6.0f} GB/s | bwd-data {tdx:7.1f} us "
          f"{((bx+by)/tdx/1e3):6.0f} GB/s | bwd-weight {tdw:7.1f} us {((bx+by)/tdw/1e3):6.0f} GB/s", flush=True)
    tot[0] += tf; tot[1] += tdx; tot[2] += tdw
print(f"sum over the 12 distinct layer shapes: fwd {tot[0]:.0f} us, bwd-data {tot[1]:.0f} us, bwd-weight {tot[2]:.0f} us")
