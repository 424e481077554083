"""Forward of ops.mha (bf16 policy) against a torch fp32 restatement at a few shapes (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd import ops
torch.manual_seed(0)
for N, d, H in [(256, 512, 4), (8, 64, 4), (64, 128, 4), (32, 512, 4), (40, 256, 2)]:
    x = torch.randn(N, d, device="cuda") * 0.5
    wq = torch.randn(3 * d, d, device="cuda") / d ** 0.5; bq = torch.randn(3 * d, device="cuda") * 0.1
    wp = torch.eye(d, device="cuda"); bp = torch.zeros(d, device="cuda")
    y = ops.mha(x, wq, bq, wp, bp, H, "bf16")
    qkv = (x @ wq.t() + bq).reshape(N, 3, H, d // H).permute(1, 2, 0, 3)
    a = ((qkv[0] @ qkv[1].transpose(-2, -1)) * (d // H) ** -0.5).softmax(-1) @ qkv[2]
    ref = a.transpose(0, 1).reshape(N, d)
    nan = torch.isnan(y)
    print("  NaN count", int(nan.sum()), "rows with NaN", int(nan.any(1).sum()), "cols with NaN", int(nan.any(0).sum()))
    err = torch.nan_to_num(y - ref, nan=0.0).abs()
    hd = d // H
    percol = err.reshape(N, H, hd).amax(dim=(0, 1))
    print(N, d, H, "max err %.4f" % err.max().item(), "ref max %.3f" % ref.abs().max().item(),
          "bad cols(in head):", [int(i) for i in torch.nonzero(percol > 0.05).flatten()[:40]])
