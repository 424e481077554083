"""Randomised sweep of K3 (ring-buffer enqueue, fp32 / bf16 queues and the fp32 + bf16-mirror form) and K4 (multi-tensor EMA) against
the numpy oracle -- bit-exact, as the contract says (reference MoMA/mem_moco.py:14-27, learning/contrast_trainer.py:207-211): ragged
widths, n > K (duplicate slots: last writer wins), wraps, empty / one-element / block-edge tensors, m in {0, 1}.
usage: python scripts/sweep_k3_k4.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moma_amd import ops
from oracle import moma_oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
dev = "cuda"
for ci in range(n_cases):
    # ---- K3
    d = int(rng.choice([4, 8, 36, 64, 100, 128, 384, 512, 1000, 1280, 2048]))
    K = int(rng.choice([1, 2, 7, 64, 100, 1000, 4096, 65536 if d <= 512 else 8192]))
    n = int(rng.choice([1, 2, 3, K, K + 1, 2 * K + 3, max(1, K // 2), 256, 257]))
    n = min(n, 5000)
    index = int(rng.integers(0, K))
    form = str(rng.choice(["fp32", "bf16", "mirror"]))
    g = torch.Generator().manual_seed(5000 + ci)
    q0 = torch.randn(K, d, generator=g)
    rows = torch.randn(n, d, generator=g)
    want = q0.numpy().copy()
    O.update_memory(want, rows.numpy(), index)
    ok = True
    if form == "mirror":
        queue, mirror = q0.clone().to(dev), q0.to(torch.bfloat16).to(dev)
        ops.enqueue_mirror_(queue, mirror, rows.to(dev), index)
        ok = np.array_equal(queue.cpu().numpy(), want) and torch.equal(mirror.cpu(), torch.from_numpy(want).to(torch.bfloat16))
    else:
        dt = torch.float32 if form == "fp32" else torch.bfloat16
        queue = q0.to(dt).to(dev)
        ops.enqueue_(queue, rows.to(dev), index)
        want_q = q0.to(dt).numpy().copy() if form == "fp32" else None
        if form == "fp32":
            ok = np.array_equal(queue.cpu().numpy(), want)
        else:                      # untouched slots keep the queue's own bf16 values, written slots hold bf16(row)
            w16 = q0.to(torch.bfloat16).float().numpy().copy()
            O.update_memory(w16, rows.to(torch.bfloat16).float().numpy(), index)
            ok = np.array_equal(queue.float().cpu().numpy(), w16)
    if not ok:
        bad += 1
        print(f"BAD K3 form={form} K={K} d={d} n={n} index={index}", flush=True)
    # ---- K4
    nt = int(rng.integers(1, 12))
    sizes = [int(rng.choice([0, 1, 2, 3, 5, 4095, 4096, 4097, 8192, 12289, 100000, 1 << 20])) for _ in range(nt)]
    m = float(rng.choice([0.0, 0.5, 0.9, 0.999, 1.0]))
    ps = [torch.randn(s, generator=g) * float(rng.choice([1e-3, 1.0, 1e3])) for s in sizes]
    es = [torch.randn(s, generator=g) for s in sizes]
    p_np, e_np = [p.numpy().copy() for p in ps], [e.numpy().copy() for e in es]
    O.momentum_update(p_np, e_np, m)
    pd, ed = [p.to(dev) for p in ps], [e.to(dev) for e in es]
    if any(s > 0 for s in sizes):
        ops.ema_update_(ops.EmaTable(pd, ed), m)
    torch.cuda.synchronize()
    ok4 = all(np.array_equal(e.cpu().numpy(), w) for e, w in zip(ed, e_np)) and all(np.array_equal(p.cpu().numpy(), w.numpy()) for p, w in zip(pd, ps))
    if not ok4:
        bad += 1
        print(f"BAD K4 sizes={sizes} m={m}", flush=True)
print(f"{n_cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
