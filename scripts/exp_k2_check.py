"""K2 against a plain torch fp32 restatement on the GPU at one shape (experiments on kernel variants selected by environment
switches).  usage: python scripts/exp_k2_check.py [B] [d] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd import ops
B, d, K = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 256), (2, 512), (3, 65536)))
T = 0.15
torch.manual_seed(0)
dev = "cuda"
q = torch.nn.functional.normalize(torch.randn(B, d, device=dev)).requires_grad_(True)
k = torch.nn.functional.normalize(q.detach() + 0.3 * torch.randn(B, d, device=dev))
queue = torch.nn.functional.normalize(torch.randn(K, d, device=dev)).to(torch.bfloat16)
out = ops.infonce_fused(q, k, queue, T, "bf16")
loss = out[0] if isinstance(out, (tuple, list)) else out
loss.sum().backward() if loss.dim() else loss.backward()
dq = q.grad.clone()
q2 = q.detach().double().requires_grad_(True)
logits = torch.cat([(q2 * k.double()).sum(1, keepdim=True), q2 @ queue.double().T], 1) / T
ref = torch.nn.functional.cross_entropy(logits, torch.zeros(B, dtype=torch.long, device=dev), reduction="none")
(ref.sum() if loss.dim() else ref.mean()).backward()
lv = loss if loss.dim() else loss
rv = ref if loss.dim() else ref.mean()
print("loss max rel err", float(((lv.detach().double() - rv.detach()).abs() / rv.detach().abs().clamp_min(1.0)).max()),
      " dq max err / max", float((dq.double() - q2.grad).abs().max() / q2.grad.abs().max()))
