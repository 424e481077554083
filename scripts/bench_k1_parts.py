"""K1 fast path: which product of a grouped backward launch takes the time (run under rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd.MoMA.criterion_moco_att import Attention
N, d, H = 256, 512, 4
att = Attention(d, num_heads=H, qkv_bias=True, precision="bf16").cuda()
x0 = torch.nn.functional.normalize(torch.randn(N, d, device="cuda"))
for name, xg, wg in (("all", True, True), ("dx_only", True, False), ("dw_only", False, True)):
    for p in att.parameters():
        p.requires_grad_(wg)
    x = x0.clone().requires_grad_(xg)
    for _ in range(12):
        y = att(x)
        y.sum().backward()
    torch.cuda.synchronize()
    print(name, flush=True)
