"""Full-size (BASELINE configs[1]: B=256, d=512, K=65536) checks through size-independent properties, where a
complete oracle evaluation per case would be slow: permutation invariance of the queue, ring-buffer round trip,
agreement of the one-pass kernel with the staged (reference-sequence) path, finite-difference check of dq, and
EMA fixed point.  All through the C ABI on the GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
B, D, K, T = 256, 512, 65536, 0.15


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd import ops
    torch.manual_seed(11)
    q = torch.nn.functional.normalize(torch.randn(B, D, device="cuda"))
    k = torch.nn.functional.normalize(q + 0.4 * torch.randn(B, D, device="cuda"))
    queue = torch.nn.functional.normalize(torch.randn(K, D, device="cuda"))
    return ops, q, k, queue


def test_queue_permutation_invariance(env):
    ops, q, k, queue = env
    qb = queue.to(torch.bfloat16)
    perm = torch.randperm(K, device="cuda")
    l0, lse0, t0 = ops.infonce_fused(q, k, qb, T, "bf16")
    l1, lse1, t1 = ops.infonce_fused(q, k, qb[perm].contiguous(), T, "bf16")
    # same multiset of keys -> same loss up to fp32 summation order; top-1 flags identical
    assert torch.allclose(lse0, lse1, rtol=0, atol=2e-5 * float(lse0.abs().max()))
    assert torch.equal(t0, t1)


def test_flash_matches_staged_reference_sequence(env):
    ops, q, k, queue = env
    qb = queue.to(torch.bfloat16)
    qa = q.clone().requires_grad_(True)
    qs = q.clone().requires_grad_(True)
    loss_rows, lse, top1 = ops.infonce_fused(qa, k, qb, T, "bf16")            # one-pass kernel
    loss_rows.mean().backward()
    logits = ops.infonce_logits(qs, k, qb, T, "bf16")                          # materialised [B,K+1] + torch CE
    loss2 = torch.nn.functional.cross_entropy(logits, torch.zeros(B, dtype=torch.long, device="cuda"))
    loss2.backward()
    assert abs(loss_rows.mean().item() - loss2.item()) < 1e-3                   # north-star loss tolerance
    acc2 = (logits.argmax(dim=1) == 0)
    assert (top1.bool() != acc2).sum().item() <= 1                             # ties/rounding at most one row
    num = (qa.grad - qs.grad).norm() / qs.grad.norm()
    assert num < 2e-2, float(num)


def test_dq_finite_difference(env):
    ops, q, k, queue = env
    qb = queue.to(torch.bfloat16)
    # fp32 policy on a K-subset for the numeric derivative (exact arithmetic), same rows
    sub = queue[:8192].contiguous()
    qa = q[:32].clone().requires_grad_(True)
    loss_rows, _, _ = ops.infonce_fused(qa, k[:32], sub, T, "fp32")
    loss_rows.sum().backward()
    g = qa.grad
    rng = np.random.default_rng(0)
    for _ in range(6):
        b, c = int(rng.integers(0, 32)), int(rng.integers(0, D))
        eps = 1e-2
        qp = q[:32].clone(); qp[b, c] += eps
        qm = q[:32].clone(); qm[b, c] -= eps
        lp = ops.infonce_fused(qp, k[:32], sub, T, "fp32")[0].double().sum()
        lm = ops.infonce_fused(qm, k[:32], sub, T, "fp32")[0].double().sum()
        fd = float((lp - lm) / (2 * eps))
        assert abs(fd - float(g[b, c])) < 2e-2 * max(1.0, abs(fd)), (b, c, fd, float(g[b, c]))


def test_ring_buffer_round_trip(env):
    ops, q, k, queue = env
    mem = queue.clone()
    rows = torch.randn(K, D, device="cuda")
    idx = 12345
    for s in range(0, K, 4096):                       # K rows in 16 enqueues, starting mid-ring -> wraps once
        ops.enqueue_(mem, rows[s:s + 4096].contiguous(), idx)
        idx = (idx + 4096) % K
    assert idx == 12345
    assert torch.equal(torch.roll(mem, -12345, dims=0), rows)          # every slot overwritten exactly once, in order


def test_ema_fixed_point_and_copy(env):
    ops, *_ = env
    p = [torch.randn(4_012_672 // 4, device="cuda") for _ in range(4)]  # EffNet-B0 sized parameter set
    e = [t.clone() for t in p]
    tab = ops.EmaTable(p, e)
    ops.ema_update_(tab, 0.999)
    for a, b in zip(p, e):
        assert torch.allclose(a, b, rtol=0, atol=2e-7 * 4)             # ema == p is a fixed point up to 1 ulp
    e2 = [torch.randn_like(t) for t in p]
    tab2 = ops.EmaTable(p, e2)
    ops.ema_update_(tab2, 0.0)                                          # m = 0 is the reference's "copy" use
    for a, b in zip(p, e2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("Bq,d,Kq,prec", [(130, 512, 4_300_000, "bf16"),      # one-pass kernel, 2.2e9 queue elements
                                          (40, 512, 4_300_000, "bf16"),       # small-batch kernel
                                          (40, 1280, 1_700_000, "bf16"),      # wide rows: two passes + the P scratch
                                          (4, 2048, 1_100_000, "bf16"),       # two register passes of Q
                                          (130, 512, 4_300_000, "fp32"),      # exact-fp32 one pass over an fp32 queue (8.8 GB)
                                          (20, 1280, 1_700_000, "fp32"),      # ... segment-streamed
                                          (130, 512, 4_300_000, "fp32/bf16")])  # exact fp32 over a bf16-STORED queue: widened in the workspace
def test_queue_beyond_2_31_elements(Bq, d, Kq, prec):
    """Maximum sizes: a queue of more than 2^31 elements (row offsets past 32 bits everywhere: LDS-DMA source addresses, the
    P scratch of the wide path, the enqueue's slot address).  K2's loss / dq against fp64 torch on the same bf16 queue values, then
    K3 at the wrap of that queue: the rows land in [K - 3, K) and [0, 4), bit for bit, nothing else changes."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd import ops
    assert Kq * d > 2 ** 31
    g = torch.Generator(device="cuda").manual_seed(Kq % 1000 + d)
    q = torch.nn.functional.normalize(torch.randn(Bq, d, device="cuda", generator=g)).requires_grad_(True)
    k = torch.nn.functional.normalize(q.detach() + 0.3 * torch.randn(Bq, d, device="cuda", generator=g))
    qdt = torch.bfloat16 if prec in ("bf16", "fp32/bf16") else torch.float32
    prec = prec.split("/")[0]
    queue = torch.empty(Kq, d, device="cuda", dtype=qdt)
    step = 1 << 18
    for i in range(0, Kq, step):                                # (normalised in pieces: no second copy of the whole queue)
        n = min(step, Kq - i)
        queue[i:i + n] = torch.nn.functional.normalize(torch.randn(n, d, device="cuda", generator=g)).to(qdt)
    loss_rows, lse, top1 = ops.infonce_fused(q, k, queue, T, prec)
    loss_rows.sum().backward()
    q64 = q.detach().double().requires_grad_(True)
    neg = torch.cat([q64 @ queue[i:i + step].double().T for i in range(0, Kq, step)], 1)
    logits = torch.cat([(q64 * k.double()).sum(1, keepdim=True), neg], 1) / T
    ref = torch.nn.functional.cross_entropy(logits, torch.zeros(Bq, dtype=torch.long, device="cuda"), reduction="none")
    ref.sum().backward()
    tol_l, tol_g = (1e-3, 2e-2) if prec == "bf16" else (2e-5, 1e-4)
    assert float(((loss_rows.detach().double() - ref.detach()).abs() / ref.detach().abs().clamp_min(1.0)).max()) < tol_l
    assert float((q.grad.double() - q64.grad).abs().max() / q64.grad.abs().max()) < tol_g
    del logits, neg
    # K3 across the end of the ring
    rows = torch.randn(7, d, device="cuda", generator=g)
    probe = torch.cat([torch.arange(0, 6, device="cuda"), torch.arange(Kq - 6, Kq, device="cuda"), torch.tensor([Kq // 2], device="cuda")])
    before = queue[probe].clone()
    ops.enqueue_(queue, rows, Kq - 3)
    want = before.clone()
    r16 = rows.to(qdt)
    want[9:12] = r16[0:3]                                        # slots K-3, K-2, K-1 (probe positions 9..11)
    want[0:4] = r16[3:7]                                         # slots 0..3
    assert torch.equal(queue[probe], want)
