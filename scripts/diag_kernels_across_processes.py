"""Do this library's kernels give the same bits in every PROCESS (fresh allocator, fresh workspaces, uninitialised memory of another
history)?  One process = K1 forward + backward, K2 (+ the enqueue riding on it), the logits path, K4 on seeded inputs in both policies
-> one sha256 over every output; the driver starts N processes and compares.    usage: python scripts/diag_kernels_across_processes.py [N]"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np
    import torch
    from moma_amd import ops
    rng = np.random.default_rng(7)
    h = hashlib.sha256()
    def t(a, dt=torch.float32):
        return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    def feed(*xs):
        for x in xs:
            h.update(x.detach().float().cpu().numpy().tobytes())
    junk = torch.empty(int(rng.integers(1, 64)) << 20, device="cuda", dtype=torch.uint8).random_()      # (a different allocation history per run would go here)
    del junk
    for prec in ("bf16", "fp32"):
        for (N, d, H) in ((256, 512, 4), (64, 1280, 4), (40, 128, 8)):
            x = t(rng.standard_normal((N, d)).astype(np.float32) / np.sqrt(d)).requires_grad_(True)
            ws = [t(rng.uniform(-1, 1, s).astype(np.float32) / np.sqrt(d)).requires_grad_(True) for s in ((3 * d, d), (3 * d,), (d, d), (d,))]
            y = ops.mha(x, *ws, H, prec)
            (y * t(rng.standard_normal((N, d)).astype(np.float32))).sum().backward()
            feed(y, x.grad, *[w.grad for w in ws])
        for (B, d, K, qdt) in ((256, 512, 65536, torch.bfloat16), (64, 512, 16384, torch.bfloat16), (100, 1280, 8192, torch.bfloat16),
                               (256, 512, 8192, torch.float32), (33, 96, 1000, torch.float32), (70, 256, 3000, torch.float32)):
            q = t(rng.standard_normal((B, d)).astype(np.float32) / np.sqrt(d)).requires_grad_(True)
            k = t(rng.standard_normal((B, d)).astype(np.float32) / np.sqrt(d))
            queue = torch.nn.functional.normalize(t(rng.standard_normal((K, d)).astype(np.float32))).to(qdt)
            lr, lse, top1 = ops.infonce_fused(q, k, queue, 0.15, prec)
            lr.sum().backward()
            feed(lr, lse, top1, q.grad)
            q2 = q.detach().clone().requires_grad_(True)
            w = t(rng.standard_normal((B, K + 1)).astype(np.float32))
            (ops.infonce_logits(q2, k, queue, 0.15, prec) * w).sum().backward()
            feed(q2.grad)
            ops.enqueue_(queue, k, K - 7)
            feed(queue[:64], queue[-64:])
    torch.cuda.synchronize()
    print(h.hexdigest())
    sys.exit(0)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
digests = []
for r in range(N):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    if p.returncode != 0:
        print("FAILED", p.stderr[-1500:])
        sys.exit(1)
    digests.append(p.stdout.strip().splitlines()[-1])
print(f"{N} processes, {len(set(digests))} distinct digests of K1 fwd/bwd, K2 (one-pass, wide, staged, logits path), K3 in both policies")
for dg in digests:
    print("   ", dg)
