"""Randomised sweep of K1 (ops.mha, bf16 policy: fused core where the head dim allows, staged path elsewhere) against torch
autograd in fp64: output and all five gradients.  usage: python scripts/sweep_k1.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moma_amd import ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for ci in range(n_cases):
    H = int(rng.choice([1, 2, 4, 8]))
    hd = int(rng.choice([16, 32, 48, 64, 96, 128, 160, 320]))
    d = H * hd
    if d > 1536:
        continue
    N = int(rng.choice([1, 5, 31, 32, 33, 64, 100, 255, 256, 257, 300, 520, 1000]))
    g = torch.Generator(device="cuda").manual_seed(ci)
    x = (torch.randn(N, d, device="cuda", generator=g) * 0.7).requires_grad_(True)
    ws = [torch.randn(3 * d, d, device="cuda", generator=g) / d ** 0.5, torch.randn(3 * d, device="cuda", generator=g) * 0.1,
          torch.randn(d, d, device="cuda", generator=g) / d ** 0.5, torch.randn(d, device="cuda", generator=g) * 0.1]
    ws = [w.requires_grad_(True) for w in ws]
    dy = torch.randn(N, d, device="cuda", generator=g)
    y = ops.mha(x, *ws, H, "bf16")
    grads = torch.autograd.grad(y, [x] + ws, dy)
    x64 = x.detach().double().requires_grad_(True)
    w64 = [w.detach().double().requires_grad_(True) for w in ws]
    qkv = (x64 @ w64[0].t() + w64[1]).reshape(N, 3, H, hd).permute(1, 2, 0, 3)
    a = ((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5).softmax(-1) @ qkv[2]
    y64 = a.transpose(0, 1).reshape(N, d) @ w64[2].t() + w64[3]
    g64 = torch.autograd.grad(y64, [x64] + w64, dy.double())
    errs = [((y.double() - y64).abs().max() / y64.abs().max()).item()]
    errs += [((a_.double() - b_).abs().max() / b_.abs().max().clamp_min(1e-9)).item() for a_, b_ in zip(grads, g64)]
    ok = max(errs) < 4e-2 and all(bool(torch.isfinite(t).all()) for t in grads)
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} N={N:4d} d={d:4d} H={H} hd={hd:3d}  y {errs[0]:.1e} dx {errs[1]:.1e} dWqkv {errs[2]:.1e} dbqkv {errs[3]:.1e} dWp {errs[4]:.1e} dbp {errs[5]:.1e}", flush=True)
print(f"{n_cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
