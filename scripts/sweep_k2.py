"""Randomised sweep of K2 (bf16 policy) against a torch fp64 restatement on the GPU: ragged B and K, every width the one-pass and
the wide-row kernels take, peaked and flat logits.  Not a pytest (minutes of GPU time); prints one line per case and a summary.
usage: python scripts/sweep_k2.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moma_amd import ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dims = [128, 256, 384, 512, 640, 768, 896, 1024, 1280, 1536, 2048]
worst = 0.0
bad = 0
for ci in range(n_cases):
    d = int(rng.choice(dims))
    B = int(rng.choice([1, 7, 31, 32, 33, 100, 128, 129, 200, 256, 257, 300, 513]))
    K = int(rng.choice([17, 33, 500, 1000, 4096, 4097, 10000, 20011, 40000, 65536]))
    if B * K * d > 3.0e10:
        K = max(17, int(3.0e10 / (B * d)))
    T = float(rng.choice([0.07, 0.15, 0.5]))
    peaked = bool(rng.integers(0, 2))
    g = torch.Generator(device="cuda").manual_seed(1000 + ci)
    queue = torch.nn.functional.normalize(torch.randn(K, d, device="cuda", generator=g)).to(torch.bfloat16)
    q = torch.randn(B, d, device="cuda", generator=g) / d ** 0.5
    if peaked:
        idx = torch.randint(0, K, (B,), device="cuda", generator=g)
        q = q * 0.2 + queue[idx].float() * float(rng.uniform(1.0, 6.0))
    k = q * 0.7 + 0.3 * torch.randn(B, d, device="cuda", generator=g) / d ** 0.5
    tq = q.clone().requires_grad_(True)
    loss_rows, lse, top1 = ops.infonce_fused(tq, k, queue, T, "bf16")
    loss_rows.sum().backward()
    # fp64 restatement
    q64, k64, Q64 = q.double(), k.double(), queue.double()
    logits = torch.cat([(q64 * k64).sum(1, keepdim=True), q64 @ Q64.t()], 1) / T
    ref_lse = torch.logsumexp(logits, 1)
    p = torch.softmax(logits, 1)
    ref_dq = ((p[:, :1] - 1) * k64 + p[:, 1:] @ Q64) / T
    e_lse = (lse.double() - ref_lse).abs().max().item()
    # (p0 -> 1 makes dq a pure cancellation, ~1e-6: measure against the scale of its terms, |k| / T, as well)
    scale = max(ref_dq.abs().max().item(), 1e-2 * k64.abs().max().item() / T)
    e_dq = (tq.grad.double() - ref_dq).abs().max().item() / scale
    ok = e_lse < max(2e-2, 2e-3 * ref_lse.abs().max().item()) and e_dq < 4e-2 and bool(torch.isfinite(tq.grad).all())
    worst = max(worst, e_dq)
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} B={B:4d} d={d:4d} K={K:6d} T={T:.2f} peaked={int(peaked)}  |lse err| {e_lse:.2e}  dq err/max {e_dq:.2e}", flush=True)
print(f"{n_cases} cases, {bad} bad, worst dq error {worst:.2e}")
sys.exit(1 if bad else 0)
