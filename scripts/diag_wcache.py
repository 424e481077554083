"""Run-to-run spread of EfficientNet-B0 gradients under bf16 autocast, weight cache off / on (diagnostic for
tests/test_gpu_kernels.py::test_effnet_weight_cache_is_transparent)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd.backbones import efficientnet as E
torch.manual_seed(0)
net = E.efficientnet_b0(num_classes=3).cuda().train()
x = torch.randn(4, 3, 64, 64, device="cuda")
def run(cache):
    old, E._WCACHE = E._WCACHE, cache
    try:
        torch.manual_seed(1)
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            feats, logits = net(x, is_feat=True)
        logits.float().square().sum().backward()
        return logits.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        E._WCACHE = old
runs = [run(c) for c in (False, False, False, True, True, True)]
ref = runs[0][1]
worst = {}
for i, (y, g) in enumerate(runs[1:], 1):
    for n in ref:
        e = (g[n] - ref[n]).abs().max().item() / max(ref[n].abs().max().item(), 1e-12)
        if e > worst.get(n, (0, 0))[0]: worst[n] = (e, i)
top = sorted(worst.items(), key=lambda kv: -kv[1][0])[:12]
for n, (e, i) in top: print(f"{n:40s} spread {e:.3e} (run {i})  max|g| {ref[n].abs().max().item():.3e}")
print("largest gradient", max(g.abs().max().item() for g in ref.values()))
print("logit spread", max((r[0].float() - runs[0][0].float()).abs().max().item() for r in runs[1:]), "max|y|", runs[0][0].float().abs().max().item())
