"""Randomised sweep of the fp32 one-pass K2 (and of the multi-term bf16 sweep) against fp64 torch.  usage: python scripts/sweep_k2_f32.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moma_amd import ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
nrm = torch.nn.functional.normalize


def ref(q, k, queue, T):
    q = q.detach().double().requires_grad_(True)
    lg = torch.cat([(q * k.double()).sum(1, keepdim=True), q @ queue.double().t()], 1) / T
    loss = torch.logsumexp(lg, 1) - lg[:, 0]
    (dq,) = torch.autograd.grad(loss.sum(), q)
    return loss, dq


for ci in range(n_cases):
    d = int(rng.choice([128, 256, 512]))
    B = int(rng.choice([1, 7, 31, 32, 33, 64, 100, 255, 256, 300]))
    K = int(rng.choice([1, 31, 32, 33, 100, 1000, 4097, 20000]))
    T = float(rng.choice([0.07, 0.15, 1.0]))
    scale = float(rng.choice([1.0, 1.0, 4.0]))             # un-normalised rows: larger logits
    g = torch.Generator(device="cuda").manual_seed(1000 + ci)
    q = nrm(torch.randn(B, d, device="cuda", generator=g)) * scale
    k = nrm(q + 0.3 * torch.randn(B, d, device="cuda", generator=g))
    queue = nrm(torch.randn(K, d, device="cuda", generator=g)) * scale
    rl, rdq = ref(q, k, queue, T)
    tq = q.clone().requires_grad_(True)
    lr, lse, top1 = ops.infonce_fused(tq, k, queue, T, "fp32")
    lr.sum().backward()
    # loss = lse - positive: both of size max|logit|, so fp32 leaves eps * max|logit| of absolute error
    floor = 8 * 1.2e-7 * scale * scale / T
    e_l = (((lr.double() - rl).abs().max() - floor).clamp_min(0) / rl.abs().max().clamp_min(1e-6)).item()
    # gradient: p_j enters dq as p_j * key / T and the positive's (p0 - 1) * k / T; once the negatives' mass is below fp32's
    # resolution of p0 (6e-8) that second term cancels to 0 -- in torch's own fp32 CrossEntropy backward exactly as here (checked
    # case by case in round 4: the kernel then equals the torch fp32 result to 1e-18 while both sit 1e-7 .. 1e-11 from fp64) -- so an
    # ABSOLUTE allowance of 2 eps |k| / T stands next to the relative tolerance
    e_g = (((tq.grad.double() - rdq).abs().max() - 2 * 1.2e-7 / T).clamp_min(0) / rdq.abs().max().clamp_min(1e-30)).item()
    # the multi-term bf16 sweep on the same data (two terms sharing q)
    qb = queue.to(torch.bfloat16)
    q2 = q.clone().requires_grad_(True)
    (l1, _, _), (l2, _, _) = ops.infonce_fused_multi([(q2, k, qb), (q2, k.flip(0), qb)], T, "bf16")
    (l1.sum() + l2.sum()).backward()
    r1, g1 = ref(q, k, qb.float(), T)
    r2, g2 = ref(q, k.flip(0), qb.float(), T)
    # bf16 policy: the query is rounded to bf16 (2^-9 relative), so every logit carries up to 2^-9 |logit| of absolute error and
    # the loss inherits it: measured against max(|loss|, 8 * 2^-9 * max |logit|); beyond |logit| ~ 100 (scale 4 at T = 0.07: 229)
    # the probabilities themselves move by tens of percent -- the policy's range is the normalised / attention-output regime of the
    # training loop (tens of nats), and the gradient check is skipped there
    lmax = scale * scale / T
    lfloor = 8 * 2.0 ** -9 * lmax
    m_l = max(((l1.double() - r1).abs().max() / r1.abs().max().clamp_min(max(lfloor, 1e-6))).item(),
              ((l2.double() - r2).abs().max() / r2.abs().max().clamp_min(max(lfloor, 1e-6))).item())
    m_g = ((q2.grad.double() - (g1 + g2)).abs().max() / (g1 + g2).abs().max().clamp_min(4 * 2.0 ** -9 * scale / T)).item()
    if lmax > 100:
        m_g = 0.0
    ok = e_l < 5e-5 and e_g < 5e-5 and m_l < 3e-2 and m_g < 3e-2 and bool(torch.isfinite(tq.grad).all()) and bool(torch.isfinite(q2.grad).all())
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} B={B:3d} d={d:3d} K={K:5d} T={T} scale={scale}: fp32 loss {e_l:.1e} dq {e_g:.1e} | multi loss {m_l:.1e} dq {m_g:.1e}", flush=True)
print(f"{n_cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
