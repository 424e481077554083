"""Out-of-bounds WRITES of the library's kernels, at ragged shapes: every output, workspace and saved tensor that moma_amd.ops
allocates for a call is carved out of a larger buffer filled with a canary pattern (4 KiB in front, 4 KiB + slack behind); after
the call the canaries must be untouched.  (GPU AddressSanitizer is not available on this pool; a kernel that stores its padded
rows -- B rounded up to 32 / 128, K to the tile, d to the segment -- past the end of a [B, d] result would corrupt whatever the
allocator placed next and no parity test would see it.)  The caller-owned targets of the in-place entry points (queue, EMA
tensors) get the same treatment.  The canary byte is 0xFF -- NaN as fp32 and as bf16 -- and the INPUTS sit in guarded buffers too:
a result that depended on a read past the end of an operand (a tail row, a tail key, a tail column) would come out NaN, and
every result is compared bit for bit with the same call on ordinary allocations."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GUARD = 4096
CANARY = 0xFF


class _Guarded:
    """stands in for the `torch` module inside moma_amd.ops: empty / zeros / *_like hand out guarded views"""

    def __init__(self):
        self.records = []

    def __getattr__(self, name):
        return getattr(torch, name)

    def _carve(self, shape, dtype, device, fill=None):
        shape = tuple(int(s) for s in shape)
        nbytes = int(math.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        pad = (-nbytes) % 256
        raw = torch.full((GUARD + nbytes + pad + GUARD,), CANARY, dtype=torch.uint8, device=device)
        view = raw[GUARD:GUARD + nbytes].view(dtype).view(shape)
        if fill is not None:
            view.fill_(fill)
        self.records.append((raw, nbytes))
        return view

    @staticmethod
    def _shape(size):
        return tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size

    def empty(self, *size, dtype=torch.float32, device=None, **kw):
        if device is None or torch.device(device).type != "cuda":
            return torch.empty(*size, dtype=dtype, device=device, **kw)
        return self._carve(self._shape(size), dtype, device)

    def zeros(self, *size, dtype=torch.float32, device=None, **kw):
        if device is None or torch.device(device).type != "cuda":
            return torch.zeros(*size, dtype=dtype, device=device, **kw)
        return self._carve(self._shape(size), dtype, device, fill=0)

    def empty_like(self, x, **kw):
        return self._carve(x.shape, kw.get("dtype", x.dtype), x.device) if x.is_cuda and x.is_contiguous() else torch.empty_like(x, **kw)

    def zeros_like(self, x, **kw):
        return self._carve(x.shape, kw.get("dtype", x.dtype), x.device, fill=0) if x.is_cuda and x.is_contiguous() else torch.zeros_like(x, **kw)

    def check(self, what):
        torch.cuda.synchronize()
        assert self.records, what
        for raw, nbytes in self.records:
            front, back = raw[:GUARD], raw[GUARD + nbytes:]
            assert bool((front == CANARY).all()) and bool((back == CANARY).all()), \
                f"{what}: a write outside a {nbytes}-byte buffer ({int((front != CANARY).sum())} bytes in front, {int((back != CANARY).sum())} behind)"
        n = len(self.records)
        self.records.clear()
        return n


@pytest.fixture()
def guard(monkeypatch):
    from moma_amd import ops
    g = _Guarded()
    monkeypatch.setattr(ops, "torch", g)
    return g


def _rand(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32)).cuda()


def _in(guard, t):
    """a copy of operand t between NaN guards (not recorded for the write check of the NEXT call: inputs are re-checked with it)"""
    g = guard.empty(*t.shape, device="cuda", dtype=t.dtype)
    g.copy_(t)
    return g


def test_the_guard_sees_an_overrun(guard):
    """the harness itself: one byte behind / in front of a guarded buffer is reported"""
    x = guard.empty(5, 3, device="cuda")
    x.fill_(1.0)
    assert guard.check("clean") == 1
    y = guard.empty(5, 3, device="cuda")
    raw = guard.records[0][0]
    raw[GUARD + 60] = 0
    with pytest.raises(AssertionError, match="1 behind"):
        guard.check("overrun")
    guard.records.clear()
    z = guard.empty(4, device="cuda", dtype=torch.bfloat16)
    guard.records[0][0][GUARD - 1] = 0
    with pytest.raises(AssertionError, match="1 bytes in front"):
        guard.check("underrun")
    del x, y, z


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("B,d,K,qdt", [(1, 128, 40, "bf16"), (33, 256, 777, "bf16"), (100, 512, 4097, "bf16"), (256, 512, 65536, "bf16"),
                                        (64, 512, 16384, "bf16"), (65, 384, 5000, "fp32"), (50, 1280, 1500, "bf16"), (256, 1280, 8192, "bf16"),
                                        (37, 2048, 3001, "bf16"), (7, 96, 333, "fp32"), (129, 768, 2049, "fp32"), (31, 1536, 1000, "fp32"),
                                        (200, 640, 9000, "bf16"), (5, 64, 31, "fp32"), (513, 512, 4097, "bf16"), (1000, 256, 2100, "bf16"),
                                        (700, 1280, 1500, "bf16"), (2050, 128, 1000, "fp32")])
def test_k2_writes_stay_inside(guard, prec, B, d, K, qdt, monkeypatch):
    """one-pass / small-batch / wide-row / exact-fp32 / staged K2, with and without dq, with the enqueue aboard, and the logits path"""
    from moma_amd import ops
    rng = np.random.default_rng(B * 7 + d + K)
    q0 = _rand(rng, B, d, scale=1 / np.sqrt(d))
    k0 = _rand(rng, B, d, scale=1 / np.sqrt(d))
    dt = torch.bfloat16 if qdt == "bf16" else torch.float32
    queue0 = torch.nn.functional.normalize(_rand(rng, K, d)).to(dt)
    w0 = _rand(rng, B, K + 1)
    n = min(B, K)

    def run(q, k, queue, w):
        fwd = ops.infonce_fused(q, k, queue, 0.15, prec)
        tq = q.clone().requires_grad_(True)
        lr, lse, top1 = ops.infonce_fused(tq, k, queue, 0.15, prec, enq=(k[:n].contiguous(), (K - 3) % K, None))
        lr.sum().backward()
        t2 = q.clone().requires_grad_(True)
        (ops.infonce_logits(t2, k, queue, 0.15, prec) * w).sum().backward()
        return [fwd[0], fwd[1], fwd[2], lr.detach(), lse, top1, tq.grad, t2.grad, queue.clone()]

    with monkeypatch.context() as m:
        m.setattr(ops, "torch", torch)                       # ordinary allocations
        ref = run(q0, k0, queue0.clone(), w0)
    got = run(_in(guard, q0), _in(guard, k0), _in(guard, queue0), _in(guard, w0))
    guard.check(f"K2 {prec} {(B, d, K, qdt)}")
    for a, b in zip(ref, got):
        assert torch.isfinite(b.float()).all() and torch.equal(a, b)


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("N,d,H", [(1, 64, 4), (33, 128, 8), (129, 256, 2), (256, 512, 4), (300, 512, 4), (1000, 512, 4), (77, 192, 4),
                                   (100, 1280, 4), (300, 1280, 4), (700, 1280, 4), (1024, 1280, 4), (64, 2048, 8), (40, 96, 3)])
def test_k1_writes_stay_inside(guard, prec, N, d, H, monkeypatch):
    """fast path (narrow / wide heads, 1-4 key tiles per wave, flash loop) and the staged exact-fp32 path, forward + backward"""
    from moma_amd import ops
    rng = np.random.default_rng(N + d + H)
    x0 = _rand(rng, N, d, scale=1 / np.sqrt(d))
    ws0 = [_rand(rng, *s, scale=1 / np.sqrt(d)) for s in ((3 * d, d), (3 * d,), (d, d), (d,))]
    dy0 = _rand(rng, N, d)

    def run(x, ws, dy):
        x = x.requires_grad_(True)
        ws = [w.requires_grad_(True) for w in ws]
        y = ops.mha(x, *ws, H, prec)
        (y * dy).sum().backward()
        with torch.no_grad():
            outs = ops.mha_group([(x.detach(), ws[0].detach(), ws[1].detach(), ws[2].detach(), ws[3].detach(), None)] * 2, H, prec)
        return [y.detach(), x.grad] + [w.grad for w in ws] + list(outs)

    with monkeypatch.context() as m:
        m.setattr(ops, "torch", torch)
        ref = run(x0.clone(), [w.clone() for w in ws0], dy0)
    got = run(_in(guard, x0), [_in(guard, w) for w in ws0], _in(guard, dy0))
    guard.check(f"K1 {prec} {(N, d, H)}")
    for a, b in zip(ref, got):
        assert torch.isfinite(b).all() and torch.equal(a, b)


@pytest.mark.parametrize("B,d,K", [(256, 512, 65536), (64, 512, 16384), (33, 256, 777), (100, 384, 4097), (1, 128, 40), (130, 512, 5000)])
def test_the_step_path_writes_stay_inside(guard, B, d, K, monkeypatch):
    """what the graph-served step does: atts_q writes q in K2's packed layout (QPack: pad rows up to 128), K2 runs into caller-owned
    K2Buffers with the enqueue aboard (bf16 mirror of an fp32 queue included), the dual-queue terms go through ONE multi-term call"""
    from moma_amd import ops
    rng = np.random.default_rng(B + d + K)
    H, T = 4, 0.15
    x0 = _rand(rng, B, d, scale=1 / np.sqrt(d))
    ws0 = [_rand(rng, *s, scale=1 / np.sqrt(d)) for s in ((3 * d, d), (3 * d,), (d, d), (d,))]
    k0 = _rand(rng, B, d, scale=1 / np.sqrt(d))
    q32_0 = torch.nn.functional.normalize(_rand(rng, K, d))
    n = min(B, K)

    def run(x, ws, k, q32):
        mirror = ops.torch.empty(K, d, device="cuda", dtype=torch.bfloat16)
        mirror.copy_(q32)
        qp = ops.QPack().prepare(B, d, T, x.device)
        with torch.no_grad():
            q = ops.mha(x, *ws, H, "bf16", None, qp)
        bufs = ops.K2Buffers(B, d, K, torch.bfloat16, "bf16", x.device)
        use = qp.buf if (qp is not None and qp.matches(q, T)) else None
        ops.infonce_fused_into(q, k, mirror, T, "bf16", use, bufs, enq=(k[:n].contiguous(), (K - 5) % K, q32))
        terms = ops.infonce_fused_multi([(q, k, mirror), (k, q, mirror)], T, "bf16")
        return [q, bufs.loss_rows, bufs.lse, bufs.top1, bufs.dq, mirror, q32] + [t for term in terms for t in term] + \
            ([] if qp is None else [qp.buf])

    with monkeypatch.context() as m:
        m.setattr(ops, "torch", torch)
        ref = run(x0.clone(), [w.clone() for w in ws0], k0.clone(), q32_0.clone())
    got = run(_in(guard, x0), [_in(guard, w) for w in ws0], _in(guard, k0), _in(guard, q32_0))
    guard.check(f"step path {(B, d, K)}")
    for a, b in zip(ref, got):
        assert torch.isfinite(b.float()).all() and torch.equal(a, b)
    assert torch.equal(got[5], got[6].to(torch.bfloat16))          # the mirror is the bf16 image of the fp32 queue, enqueued rows included


@pytest.mark.parametrize("K,d,n,index,qdt", [(1000, 36, 77, 990, "fp32"), (1000, 36, 77, 990, "bf16"), (64, 512, 256, 60, "bf16"), (4096, 1280, 256, 4000, "bf16"),
                                              (50, 24, 120, 49, "fp32"), (257, 130, 31, 256, "bf16")])
def test_k3_k4_writes_stay_inside(guard, K, d, n, index, qdt):
    """the in-place entry points on caller-owned storage: enqueue (wrap, n > K, odd widths), enqueue with mirror, multi-tensor EMA"""
    from moma_amd import ops
    rng = np.random.default_rng(K + d + n)
    dt = torch.bfloat16 if qdt == "bf16" else torch.float32
    queue = guard.empty(K, d, device="cuda", dtype=dt)
    queue.zero_()
    rows = _rand(rng, n, d)
    ops.enqueue_(queue, rows, index)
    guard.check(f"K3 {(K, d, n, index, qdt)}")
    q32, mirror = guard.empty(K, d, device="cuda"), guard.empty(K, d, device="cuda", dtype=torch.bfloat16)
    q32.zero_(); mirror.zero_()
    ops.enqueue_mirror_(q32, mirror, rows, index)
    guard.check(f"K3 mirror {(K, d, n, index)}")
    assert torch.equal(q32.to(torch.bfloat16), mirror)
    shapes = [(d,), (n, d), (3, 5, 7), (1,), (4097,), (K, 3)]
    ps = [_rand(rng, *s) for s in shapes]
    es = []
    for s in shapes:
        e = guard.empty(*s, device="cuda")
        e.normal_()
        es.append(e)
    ops.ema_update_(ops.EmaTable(ps, es), 0.99)
    guard.check(f"K4 {shapes}")


@pytest.mark.parametrize("shape,dtype", [((5, 3, 9, 11), torch.float32), ((8, 24, 28, 28), torch.bfloat16), ((3, 1152, 7, 7), torch.bfloat16)])
def test_backbone_helper_writes_stay_inside(guard, shape, dtype):
    """BN + activation (training / eval) of the EfficientNet modules: outputs, saved statistics and the workspace"""
    from moma_amd import ops
    rng = np.random.default_rng(sum(shape))
    C = shape[1]
    x = _rand(rng, *shape).to(dtype).requires_grad_(True)
    w, b = _rand(rng, C).requires_grad_(True), _rand(rng, C).requires_grad_(True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    for training in (True, False):
        y = ops._BNAct.apply(x, w, b, rm, rv, training, 0.1, 1e-5, ops.ACT_CODES["silu"], False)
        guard.check(f"bn forward {shape} {dtype} training={training}")
        y.float().sum().backward()
        guard.check(f"bn backward {shape} {dtype} training={training}")
        assert torch.isfinite(x.grad.float()).all()
