"""GPU parity tests: every kernel is called through the C ABI (moma_amd.ops -> ctypes -> libmoma_hip.so)
and compared with (a) the golden vectors captured from the reference and (b) the CPU oracle on seeded
inputs.  Tolerances are stated per test: bit-exact for copies / indices / EMA, 1e-5-class for the fp32
policy (exact fp32 fma chains, different summation order than ATen), 2e-2-class for the bf16 policy."""
import os

import numpy as np
import pytest
import torch

from oracle import moma_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd import ops as _ops
    from moma_amd import _lib
    _lib.load()
    return _ops


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


# ------------------------------------------------------------------------------------------------ K4
def test_ema_golden_bit_exact(ops, golden_dir):
    g = _g(golden_dir, "g3_ema.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        m = float(g[p + "m"]); n = int(g[p + "n"])
        ps = [_t(g[p + f"p{i}"]) for i in range(n)]
        es = [_t(g[p + f"e{i}"]) for i in range(n)]
        table = ops.EmaTable(ps, es)
        for _ in range(3):
            ops.ema_update_(table, m)
        torch.cuda.synchronize()
        for i in range(n):
            assert np.array_equal(es[i].cpu().numpy(), g[p + f"r{i}"]), (ci, i)


def test_ema_large_ragged_vs_oracle(ops):
    rng = np.random.default_rng(0)
    sizes = [1, 5, 4096, 4097, 8192 + 3, 1_000_003, 32 * 3 * 3 * 3, 1280 * 320]
    ps = [rng.standard_normal(s).astype(np.float32) for s in sizes]
    es = [rng.standard_normal(s).astype(np.float32) for s in sizes]
    tp = [_t(a) for a in ps]; te = [_t(a) for a in es]
    # an unaligned view exercises the scalar path
    big = torch.randn(10_001, device="cuda"); bige = torch.randn(10_001, device="cuda")
    tp.append(big[1:]); te.append(bige[1:])
    ps.append(big[1:].cpu().numpy().copy()); es.append(bige[1:].cpu().numpy().copy())
    table = ops.EmaTable([t.contiguous() if not t.is_contiguous() else t for t in tp], te)
    for m in (0.999, 0.999):
        ops.ema_update_(table, m)
        O.momentum_update(ps, es, m)
    torch.cuda.synchronize()
    for a, b in zip(te, es):
        assert np.array_equal(a.cpu().numpy(), b)


# ------------------------------------------------------------------------------------------------ K3
def test_enqueue_golden_traces_bit_exact(ops, golden_dir):
    g = _g(golden_dir, "g2_queue.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        K, d, B, n, steps = [int(v) for v in g[p + "cfg"]]
        queue = _t(g[p + "memory0"])
        index = 0
        for s in range(steps):
            rows = _t(g[p + "all_k"][s])
            ops.enqueue_(queue, rows, index)
            index = (index + n) % K
            assert index == int(g[p + "index"][s])
            assert np.array_equal(queue.cpu().numpy(), g[p + "memory"][s]), (ci, s)


@pytest.mark.parametrize("K,d,n,index", [(65536, 512, 256, 65536 - 100), (65536, 1280, 256, 0), (1000, 36, 77, 990),
                                          (64, 5, 200, 3)])
def test_enqueue_wrap_vs_oracle(ops, K, d, n, index):
    rng = np.random.default_rng(1)
    q0 = rng.standard_normal((K, d)).astype(np.float32)
    rows = rng.standard_normal((n, d)).astype(np.float32)
    queue = _t(q0)
    ops.enqueue_(queue, _t(rows), index)
    ref = q0.copy()
    O.update_memory(ref, rows, index)
    assert np.array_equal(queue.cpu().numpy(), ref)
    # bf16 queue: rows rounded to nearest even, untouched rows unchanged
    qb = _t(q0, torch.bfloat16)
    before = qb.clone()
    ops.enqueue_(qb, _t(rows), index)
    ids = np.unique(O.enqueue_ids(index, n, K))
    refb = before.clone()
    last = {}
    for i, j in enumerate(O.enqueue_ids(index, n, K)):
        last[int(j)] = i
    for j, i in last.items():
        refb[j] = _t(rows[i]).to(torch.bfloat16)
    assert torch.equal(qb, refb)
    assert len(ids) == min(n, K)


# ------------------------------------------------------------------------------------------------ K2
@pytest.mark.parametrize("prec,rtol,atol", [("fp32", 2e-5, 2e-5), ("bf16", 2e-2, 2e-2)])
def test_infonce_golden(ops, golden_dir, prec, rtol, atol):
    g = _g(golden_dir, "g4_infonce.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        B, d, K = [int(v) for v in g[p + "cfg"]]
        q = _t(g[p + "q"]).requires_grad_(True)
        k = _t(g[p + "k"]); mem = _t(g[p + "memory0"])
        logits = ops.infonce_logits(q, k, mem, 0.15, prec)
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g[p + "logits"], rtol=rtol, atol=atol)
        loss = torch.nn.functional.cross_entropy(logits, torch.zeros(B, dtype=torch.long, device="cuda"))
        loss.backward()
        tol_loss = 1e-5 if prec == "fp32" else 1e-3
        assert abs(loss.item() - float(g[p + "loss"])) < tol_loss * max(1.0, abs(float(g[p + "loss"])))
        np.testing.assert_allclose(q.grad.cpu().numpy(), g[p + "dq"], rtol=rtol * 10, atol=atol * 0.05)
        # fused path: same loss, accuracy and gradient without materialised logits at the API
        q2 = _t(g[p + "q"]).requires_grad_(True)
        loss_rows, lse, top1 = ops.infonce_fused(q2, k, mem, 0.15, prec)
        loss2 = loss_rows.mean()
        loss2.backward()
        assert abs(loss2.item() - float(g[p + "loss"])) < tol_loss * max(1.0, abs(float(g[p + "loss"])))
        assert abs(100.0 * top1.float().mean().item() - float(g[p + "acc"])) < 1e-4
        np.testing.assert_allclose(q2.grad.cpu().numpy(), g[p + "dq"], rtol=rtol * 10, atol=atol * 0.05)


@pytest.mark.parametrize("B,d,K,qdt", [(64, 512, 4096, "fp32"), (37, 96, 1000, "fp32"), (256, 512, 8192, "bf16"),
                                       (16, 1280, 2048, "fp32"), (1, 64, 100, "fp32"),
                                       # bf16 queue + bf16 MFMA -> the one-pass flash kernel (ragged B, K tails)
                                       (100, 384, 5000, "bf16"), (300, 256, 777, "bf16"), (7, 512, 33, "bf16"),
                                       (129, 512, 65536, "bf16"), (200, 128, 16384, "bf16"), (9, 128, 65, "bf16"),
                                       # the reference's own run-script shape (scripts/run_moma.sh: --batch_size 64 --feat_dim 512,
                                       # --nce_k default 16384) and its neighbours: the plans round 5 cuts into fewer workgroups
                                       (64, 512, 16384, "bf16"), (64, 128, 16384, "bf16"), (256, 256, 16384, "bf16"), (60, 256, 32768, "bf16"),
                                       # d > 512: column slabs over a score scratch (512 + 512 + 256, 512 + 128, 4 x 512, 512 + 384)
                                       (64, 1280, 4096, "bf16"), (33, 640, 1000, "bf16"), (40, 2048, 777, "bf16"),
                                       (130, 896, 2100, "bf16"),
                                       # the one-pass wide kernels at their other widths (8 / 12 / 6 segments), ragged B and K,
                                       # several key chunks per group and a last 16-key tile that is partly past K
                                       (70, 1024, 3000, "bf16"), (257, 1536, 2500, "bf16"), (96, 768, 70001, "bf16"),
                                       (5, 1280, 17, "bf16"),
                                       # d = 2048 (ResNet-50 `--head None`): scores in two register passes of Q (8 + 8 segments)
                                       (256, 2048, 8192, "bf16"), (130, 2048, 3001, "bf16"), (3, 2048, 40, "bf16"),
                                       # batches beyond 256 rows (more row blocks than the plans were tuned for: fewer, longer key chunks)
                                       (512, 512, 8192, "bf16"), (1000, 256, 4100, "bf16"), (700, 1280, 3000, "bf16"), (2048, 128, 2048, "fp32")])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_infonce_vs_oracle(ops, B, d, K, qdt, prec):
    rng = np.random.default_rng(B + d + K)
    q = (rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    k = (q + 0.3 * rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    queue = O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32))
    tq = _t(q).requires_grad_(True)
    tk = _t(k)
    tqueue = _t(queue, torch.bfloat16 if qdt == "bf16" else torch.float32)
    queue_eff = tqueue.float().cpu().numpy()
    T = 0.15
    ref_logits = O.compute_logit(q, k, queue_eff, T, dtype=np.float64)
    ref = O.infonce_loss(ref_logits)
    ref_dq = O.infonce_grad(q, k, queue_eff, T) * B          # d(sum loss)/dq
    if prec == "fp32":
        rtol, atol, ltol = 1e-5, 2e-5, 2e-5
    else:
        rtol, atol, ltol = 2e-2, 2e-2, 1e-3
    logits = ops.infonce_logits(tq, tk, tqueue, T, prec)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), ref_logits, rtol=rtol, atol=atol)
    loss_rows, lse, top1 = ops.infonce_fused(tq, tk, tqueue, T, prec)
    assert abs(loss_rows.mean().item() - ref["loss"]) < ltol * max(1.0, abs(ref["loss"]))
    np.testing.assert_allclose(lse.cpu().numpy(), ref["lse"], rtol=ltol, atol=ltol * 5)
    if prec == "fp32":
        assert np.array_equal(top1.cpu().numpy().astype(bool), ref["top1"])
    loss_rows.sum().backward()
    scale = np.abs(ref_dq).max()
    np.testing.assert_allclose(tq.grad.cpu().numpy(), ref_dq, rtol=0, atol=(2e-5 if prec == "fp32" else 2e-2) * scale)


@pytest.mark.parametrize("B,d,K", [(256, 512, 4096), (100, 384, 1000), (33, 256, 300), (128, 512, 65536), (70, 128, 2000), (50, 1280, 1500),
                                   (160, 768, 3000), (256, 1280, 65536), (90, 1024, 20000), (256, 2048, 65536), (70, 2048, 5000)])
def test_infonce_flash_queue_term(ops, B, d, K):
    """The sum_j p_bj * queue_j part of dq in isolation: k = 0 removes the positive-key term, and every query is
    aligned with a few queue rows so the softmax is peaked and the weighted key sum is O(1), not averaged away."""
    rng = np.random.default_rng(B * 7 + d + K)
    queue = O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32))
    idx = rng.integers(0, K, size=(B, 3))
    q = (queue[idx[:, 0]] * 1.6 + queue[idx[:, 1]] * 1.45 + queue[idx[:, 2]] * 1.3
         + 0.2 * rng.standard_normal((B, d)).astype(np.float32) / np.sqrt(d)).astype(np.float32)
    k = np.zeros((B, d), dtype=np.float32)
    T = 0.15
    tq = _t(q).requires_grad_(True)
    tqueue = _t(queue, torch.bfloat16)
    qe = tqueue.float().cpu().numpy()
    ref_dq = O.infonce_grad(q, k, qe, T) * B
    ref = O.infonce_loss(O.compute_logit(q, k, qe, T, dtype=np.float64))
    loss_rows, lse, top1 = ops.infonce_fused(tq, _t(k), tqueue, T, "bf16")
    np.testing.assert_allclose(lse.cpu().numpy(), ref["lse"], rtol=0, atol=2e-2)
    loss_rows.sum().backward()
    got = tq.grad.cpu().numpy()
    rown = np.linalg.norm(ref_dq, axis=1)
    assert rown.min() > 0.5                       # the term under test is not negligible
    err = np.linalg.norm(got - ref_dq, axis=1) / rown
    assert err.max() < 2e-2, err.max()


@pytest.mark.parametrize("d", [768, 1280, 2048])
def test_infonce_wide_rows_large_logits_and_repeatable(ops, d):
    """Wide rows (d > 512): P = 2^(x - reference) with the reference = max over the chunk group of the chunk maxima of the score
    pass, so logits tens of nats apart (un-normalised attention outputs as q, reference MoMA/criterion_moco_att.py:153-167)
    need no rescue path; and the path has no atomics: two calls give the same bits."""
    rng = np.random.default_rng(d)
    B, K, T = 96, 9000, 0.15
    queue = O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32))
    hot = rng.integers(0, K, size=B)
    q = (queue[hot] * rng.uniform(2.0, 14.0, size=(B, 1)) + 0.3 * rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    k = (0.5 * q).astype(np.float32)
    tqueue = _t(queue, torch.bfloat16)
    qe = tqueue.float().cpu().numpy()
    ref = O.infonce_loss(O.compute_logit(q, k, qe, T, dtype=np.float64))
    ref_dq = O.infonce_grad(q, k, qe, T) * B
    outs = []
    for _ in range(2):
        tq = _t(q).requires_grad_(True)
        loss_rows, lse, top1 = ops.infonce_fused(tq, _t(k), tqueue, T, "bf16")
        loss_rows.sum().backward()
        outs.append((loss_rows.detach().clone(), lse.clone(), tq.grad.clone()))
    assert np.ptp(ref["lse"]) > 20.0                                     # the rows really are tens of nats apart
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), ref["lse"], rtol=2e-3, atol=5e-2)
    got = outs[0][2].cpu().numpy()
    err = np.linalg.norm(got - ref_dq, axis=1) / np.maximum(np.linalg.norm(ref_dq, axis=1), 1e-3)
    assert err.max() < 3e-2, err.max()
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("grad", [True, False])
@pytest.mark.parametrize("d", [256, 512, 768, 1280, 2048])
@pytest.mark.parametrize("scale", [30.0, 25.0, 10.0, 6.0])
def test_infonce_flash_overflow_repass(ops, scale, d, grad):
    """A key far down the chunk beats the first tile's max by tens to hundreds of log2 units.  Beyond the fixed
    reference's headroom (128) the wave raises its overflow word and the WORKGROUP repeats its chunk in-kernel with the
    true row maxima as references (rule 26: force the rare branch, full reference: scale 30 / 25 -> ~290 / 240 log2
    units); below it (scale 10 / 6 -> ~96 / 58) the single pass must carry the range by itself.  Without a gradient the
    forward-only kernel runs, which moves its reference on the fly (online softmax, no O to rescale)."""
    rng = np.random.default_rng(7)
    # (wide rows, d > 512: the score pass stores bf16 P against an integer reference fixed by the chunk's first tile and repeats
    #  the chunk when a later tile overflows it -- K = 20000 gives chunks of 3 tiles, keys 1500 / 3031 sit in their second tiles)
    B, K, T = 40, (3000 if d <= 512 else 20000), 0.15
    k2 = 2999 if d <= 512 else 3031
    q = (rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    k = (q + 0.3 * rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    queue = O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32))
    # rows 5 and 17 see a huge logit at keys 1500 / 2999 (late tiles of their chunks): s = |q|*30/T ~ 200 nats
    queue[1500] = scale * q[5] / np.linalg.norm(q[5])
    queue[k2] = (scale - 5.0) * q[17] / np.linalg.norm(q[17])
    tq = _t(q).requires_grad_(grad)
    tqueue = _t(queue, torch.bfloat16)
    qe = tqueue.float().cpu().numpy()
    ref = O.infonce_loss(O.compute_logit(q, k, qe, T, dtype=np.float64))
    ref_dq = O.infonce_grad(q, k, qe, T) * B
    loss_rows, lse, top1 = ops.infonce_fused(tq, _t(k), tqueue, T, "bf16")
    # logits of ~200 nats carry a bf16 operand rounding of 2^-8 relative
    np.testing.assert_allclose(lse.cpu().numpy(), ref["lse"], rtol=1e-2, atol=2e-2)
    assert np.all(np.isfinite(lse.cpu().numpy()))
    assert abs(loss_rows.mean().item() - ref["loss"]) < 1e-2 * abs(ref["loss"])
    if grad:
        loss_rows.sum().backward()
        np.testing.assert_allclose(tq.grad.cpu().numpy(), ref_dq, rtol=0, atol=2e-2 * np.abs(ref_dq).max())


@pytest.mark.parametrize("B,d,K", [(64, 512, 65536), (64, 512, 4100), (33, 256, 1000), (8, 128, 40), (1, 384, 9000), (40, 512, 20000),
                                   (64, 384, 65536), (17, 512, 8200), (64, 128, 33), (97, 512, 4100), (130, 512, 9000)])
def test_infonce_small_batch_and_ragged_rows_on_poisoned_workspace(ops, B, d, K):
    """B <= 64 (the reference's default --batch_size, train_student_moma.py:51; BASELINE configs[4] per rank) runs the key-half-split
    instantiation of the one-pass kernel (infonce_small_kernel: 16x16x32 score MFMAs, each wave of a workgroup on 16 of a tile's 32
    keys, two virtual chunks per workgroup); B = 97 / 130 are ragged row blocks of the B = 256 schedule, whose pad-row waves store
    nothing (the combine must never read what was not written).  Through caller-owned buffers (ops.K2Buffers) whose
    workspace is poisoned with NaN before the call: loss / lse / top-1 / dq against the fp64 oracle; K not a multiple of the tile
    (a key half that lies past K entirely), K smaller than one tile, one row."""
    rng = np.random.default_rng(B * 3 + d + K)
    q = (rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    k = (q + 0.3 * rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    queue = O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32))
    # a few rows aligned with queue rows (peaked softmax: the sum_j p_bj queue_j part of dq is O(1)), one of them in the LAST tile
    hot = rng.integers(0, K, size=min(B, 6))
    hot[0] = K - 1
    for b, j in enumerate(hot):
        q[b] = (1.5 * queue[j] + 0.1 * q[b]).astype(np.float32)
    T = 0.15
    tq, tk, tqueue = _t(q), _t(k), _t(queue, torch.bfloat16)
    qe = tqueue.float().cpu().numpy()
    ref = O.infonce_loss(O.compute_logit(q, k, qe, T, dtype=np.float64))
    ref_dq = O.infonce_grad(q, k, qe, T) * B
    out = ops.K2Buffers(B, d, K, torch.bfloat16, "bf16", tq.device)
    for rep in range(2):                                                    # (twice: the second call meets the first one's leftovers)
        out.ws.view(torch.int16).fill_(0x7FC0 if rep == 0 else -1)          # bf16 / fp32 NaN patterns everywhere
        out.dq.fill_(float("nan")); out.loss_rows.fill_(float("nan")); out.lse.fill_(float("nan")); out.top1.fill_(-7)
        ops.infonce_fused_into(tq, tk, tqueue, T, "bf16", None, out)
        torch.cuda.synchronize()
        lse, loss_rows, dq = out.lse.cpu().numpy(), out.loss_rows.cpu().numpy(), out.dq.cpu().numpy()
        assert np.all(np.isfinite(lse)) and np.all(np.isfinite(dq)) and np.all(np.isfinite(loss_rows))
        np.testing.assert_allclose(lse, ref["lse"], rtol=1e-3, atol=5e-3)
        assert abs(loss_rows.mean() - ref["loss"]) < 1e-3 * max(1.0, abs(ref["loss"]))
        np.testing.assert_allclose(dq, ref_dq, rtol=0, atol=2e-2 * np.abs(ref_dq).max())
        t1 = out.top1.cpu().numpy()
        assert set(np.unique(t1)) <= {0, 1}
        sure = np.abs(np.sort(O.compute_logit(q, k, qe, T, dtype=np.float64), axis=1)[:, -1] - O.compute_logit(q, k, qe, T, dtype=np.float64)[:, 0]) > 0.5
        assert np.array_equal(t1[sure].astype(bool), ref["top1"][sure])     # (rows whose positive logit is not within bf16 noise of the max)
    # the autograd entry gives the same numbers (it allocates its own buffers)
    tq2 = _t(q).requires_grad_(True)
    lr2, lse2, _ = ops.infonce_fused(tq2, tk, tqueue, T, "bf16")
    lr2.sum().backward()
    assert torch.equal(lse2, out.lse) and torch.equal(tq2.grad, out.dq)


@pytest.mark.parametrize("d", [128, 512])
@pytest.mark.parametrize("scale", [30.0, 10.0])
def test_infonce_small_batch_overflow_repass(ops, scale, d):
    """The rare branch of the small-batch kernel (guide rule 26): chunks of 3 tiles (K = 20000), a key in the LAST tile of its chunk
    -- one in the first key half of its tile, one in the second -- beats the fixed reference of its wave (first tile's max + 32) by
    ~290 log2 units (scale 30: the workgroup repeats its chunk with the true row maxima) or ~96 (scale 10: one pass carries it)."""
    rng = np.random.default_rng(11)
    B, K, T = 40, 20000, 0.15
    q = (rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    k = (q + 0.3 * rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    queue = O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32))
    # workgroup w covers tiles 3w .. 3w+2 = keys 96 w .. 96 w + 95: key 96*20 + 64 + 5 (third tile, first half), 96*101 + 64 + 27 (second half)
    queue[96 * 20 + 69] = scale * q[5] / np.linalg.norm(q[5])
    queue[96 * 101 + 91] = (scale - 5.0) * q[37] / np.linalg.norm(q[37])
    tq = _t(q).requires_grad_(True)
    tqueue = _t(queue, torch.bfloat16)
    qe = tqueue.float().cpu().numpy()
    ref = O.infonce_loss(O.compute_logit(q, k, qe, T, dtype=np.float64))
    ref_dq = O.infonce_grad(q, k, qe, T) * B
    loss_rows, lse, top1 = ops.infonce_fused(tq, _t(k), tqueue, T, "bf16")
    np.testing.assert_allclose(lse.cpu().numpy(), ref["lse"], rtol=1e-2, atol=2e-2)
    assert np.all(np.isfinite(lse.cpu().numpy()))
    assert abs(loss_rows.mean().item() - ref["loss"]) < 1e-2 * abs(ref["loss"])
    loss_rows.sum().backward()
    np.testing.assert_allclose(tq.grad.cpu().numpy(), ref_dq, rtol=0, atol=2e-2 * np.abs(ref_dq).max())


@pytest.mark.parametrize("B,d,K", [(256, 512, 65536), (100, 256, 5000), (33, 128, 777), (8, 512, 40), (1, 128, 100), (64, 512, 4097),
                                   (100, 384, 5000), (70, 768, 3001), (33, 1024, 2100), (256, 1280, 8192), (5, 1280, 70), (256, 1280, 65536)])
def test_infonce_f32_policy_is_one_pass(ops, B, d, K):
    """The reference's OWN arithmetic (fp32; MoMA/mem_moco.py:29-49,77-100 + CrossEntropy) as one pass over the fp32 queue on the
    f32-input MFMA (infonce_f32.hip): no [B,K+1] logits (the workspace is the chunk partials), loss / lse / top-1 / dq against the
    fp64 oracle at the fp32 tolerance of the staged path it replaces, forward-only == with-gradient, bitwise repeatable.
    d = 384 / 768 / 1024 / 1280 (the reference CLI's default --head None: d = s_dim, train_student_moma.py:101-110) stream each
    wave's slab of a key tile through LDS in 3 / 3 / 2 / 5 segments (round 4)."""
    from moma_amd import _lib
    rng = np.random.default_rng(B + d + K)
    q = (rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    k = (q + 0.3 * rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    queue = O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32))
    T = 0.15
    ws = _lib.load().moma_infonce_fused_workspace_bytes(B, d, K, 0, 0)
    assert ws > 0 and (K < 65536 or ws < B * (K + 1) * 4 // (2 if d <= 512 else 1))     # chunk partials (32 chunks x B x d fp32), not a [B,K+1] logits matrix
    ref = O.infonce_loss(O.compute_logit(q, k, queue, T, dtype=np.float64))
    ref_dq = O.infonce_grad(q, k, queue, T) * B
    tqueue = _t(queue)
    outs = []
    for _ in range(2):
        tq = _t(q).requires_grad_(True)
        loss_rows, lse, top1 = ops.infonce_fused(tq, _t(k), tqueue, T, "fp32")
        loss_rows.sum().backward()
        outs.append((loss_rows.detach().clone(), lse.clone(), top1.clone(), tq.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    loss_rows, lse, top1, dq = outs[0]
    assert abs(loss_rows.mean().item() - ref["loss"]) < 2e-5 * max(1.0, abs(ref["loss"]))
    np.testing.assert_allclose(lse.cpu().numpy(), ref["lse"], rtol=2e-5, atol=1e-4)
    assert np.array_equal(top1.cpu().numpy().astype(bool), ref["top1"])
    np.testing.assert_allclose(dq.cpu().numpy(), ref_dq, rtol=0, atol=2e-5 * np.abs(ref_dq).max())
    fwd = ops.infonce_fused(_t(q), _t(k), tqueue, T, "fp32")
    assert torch.equal(fwd[0], loss_rows) and torch.equal(fwd[1], lse) and torch.equal(fwd[2], top1)


@pytest.mark.parametrize("scale", [30.0, 10.0])
@pytest.mark.parametrize("d", [128, 512, 384, 1280])
def test_infonce_f32_flash_overflow_repass(ops, scale, d):
    """The fixed softmax reference of the fp32 one-pass kernel (first tile's max + 32) with a key far down the chunk that beats it
    by ~290 log2 units (scale 30: beyond the 128 of headroom -> the workgroup repeats its chunk with the true row maxima) or ~96
    (scale 10: carried by the single pass).  Guide rule 26: the rare branch gets its own forced test, full fp64 reference."""
    rng = np.random.default_rng(7)
    B, K, T = 40, 20000, 0.15          # 625 tiles -> chunks of 5 tiles (at K = 3000 every tile opens its own chunk: no overflow possible)
    q = (rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    k = (q + 0.3 * rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    queue = O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32))
    queue[1500] = scale * q[5] / np.linalg.norm(q[5])                 # tile 46 = second tile of chunk 9
    queue[19999] = (scale - 5.0) * q[17] / np.linalg.norm(q[17])      # the queue's last tile = last tile of its chunk
    ref = O.infonce_loss(O.compute_logit(q, k, queue, T, dtype=np.float64))
    ref_dq = O.infonce_grad(q, k, queue, T) * B
    tq = _t(q).requires_grad_(True)
    loss_rows, lse, top1 = ops.infonce_fused(tq, _t(k), _t(queue), T, "fp32")
    loss_rows.sum().backward()
    assert np.all(np.isfinite(lse.cpu().numpy()))
    np.testing.assert_allclose(lse.cpu().numpy(), ref["lse"], rtol=2e-5, atol=2e-4)
    assert abs(loss_rows.mean().item() - ref["loss"]) < 2e-5 * abs(ref["loss"])
    np.testing.assert_allclose(tq.grad.cpu().numpy(), ref_dq, rtol=0, atol=3e-5 * np.abs(ref_dq).max())


def test_infonce_flash_forward_only_matches_grad_path(ops):
    """The forward-only kernel (no dq requested) and the pipelined kernel give the same loss / lse / top-1."""
    rng = np.random.default_rng(11)
    for B, d, K in [(256, 512, 8192), (70, 128, 2000), (129, 384, 5000)]:
        q = (rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
        k = (q + 0.3 * rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
        queue = _t(O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32)), torch.bfloat16)
        a = ops.infonce_fused(_t(q).requires_grad_(True), _t(k), queue, 0.15, "bf16")
        b = ops.infonce_fused(_t(q), _t(k), queue, 0.15, "bf16")
        np.testing.assert_allclose(a[0].detach().cpu().numpy(), b[0].cpu().numpy(), rtol=0, atol=1e-4)
        np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), rtol=0, atol=1e-4)
        assert torch.equal(a[2], b[2])


def test_infonce_fused_bitwise_repeatable(ops):
    """Two launches on the same inputs give bit-identical loss and dq (no atomics anywhere in the one-pass call)."""
    rng = np.random.default_rng(3)
    B, d, K = 256, 512, 16384
    q = (rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    k = (q + 0.3 * rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    queue = _t(O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32)), torch.bfloat16)
    outs = []
    for _ in range(2):
        tq = _t(q).requires_grad_(True)
        lr, lse, _ = ops.infonce_fused(tq, _t(k), queue, 0.15, "bf16")
        lr.sum().backward()
        outs.append((lr.detach().clone(), tq.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B,d,K,prec,qdt", [(256, 512, 16384, "fp32", "fp32"), (256, 512, 16384, "bf16", "bf16"), (8, 64, 256, "fp32", "fp32"),
                                            (100, 96, 5000, "fp32", "bf16"), (64, 1536, 4096, "fp32", "fp32")])
def test_materialised_logits_paths_are_bitwise_repeatable(ops, B, d, K, prec, qdt):
    """The gradient product of the materialised-logits paths -- `infonce_logits` (the reference's call sequence, MoMA/mem_moco.py:29-49,
    used by the MoCoAtt variants) and the staged form of `infonce_fused` (exact fp32 at widths without a one-pass kernel) -- splits
    its contraction over K across workgroups.  Until round 6 the splits met in fp32 atomics (last bits changed from run to run --
    found as the one loop configuration whose eager runs were not bit-identical under MIOpen's deterministic algorithms); now each
    split leaves a partial in the workspace and they are added in split order."""
    rng = np.random.default_rng(B + d + K)
    q = (rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    k = (q + 0.3 * rng.standard_normal((B, d)) / np.sqrt(d)).astype(np.float32)
    queue = _t(O.l2_normalize(rng.standard_normal((K, d)).astype(np.float32)), torch.bfloat16 if qdt == "bf16" else torch.float32)
    w = _t(rng.standard_normal((B, K + 1)).astype(np.float32))
    grads, fused = [], []
    for _ in range(3):
        tq = _t(q).requires_grad_(True)
        (ops.infonce_logits(tq, _t(k), queue, 0.15, prec) * w).sum().backward()
        grads.append(tq.grad.clone())
        tq = _t(q).requires_grad_(True)
        lr, lse, _ = ops.infonce_fused(tq, _t(k), queue, 0.15, prec)
        lr.sum().backward()
        fused.append((lr.detach().clone(), tq.grad.clone()))
    assert all(torch.equal(grads[0], g) for g in grads[1:])
    assert all(torch.equal(fused[0][0], f[0]) and torch.equal(fused[0][1], f[1]) for f in fused[1:])
    ref = (w.double()[:, :1] * _t(k).double() + w.double()[:, 1:] @ queue.double()) / 0.15
    tol = 2e-5 if prec == "fp32" else 2e-2
    assert (grads[0].double() - ref).abs().max().item() <= tol * ref.abs().max().item()


def test_kernels_give_the_same_bits_in_every_process():
    """K1 forward + backward, K2 (one-pass, wide rows, staged, the logits path), K3, in both policies, on seeded inputs, in separate
    PROCESSES with different allocation histories: one digest.  (No atomics, no result that depends on what a workspace held before:
    the run-to-run differences of whole training runs between processes come from the stock backbone's MIOpen solver choice,
    scripts/diag_cli_trace.py.)"""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "diag_kernels_across_processes.py")
    r = subprocess.run([sys.executable, script, "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "3 processes, 1 distinct digests" in r.stdout, r.stdout


# ------------------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("prec,rtol,atol", [("fp32", 2e-4, 2e-5), ("bf16", 5e-2, 2e-2)])
def test_mha_golden(ops, golden_dir, prec, rtol, atol):
    g = _g(golden_dir, "g1_attention.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        n, d, h = [int(v) for v in g[p + "shape"]]
        x = _t(g[p + "x"]).requires_grad_(True)
        ws = [_t(g[p + nm]).requires_grad_(True) for nm in ("w_qkv", "b_qkv", "w_proj", "b_proj")]
        y = ops.mha(x, *ws, h, prec)
        np.testing.assert_allclose(y.detach().cpu().numpy(), g[p + "y"], rtol=rtol, atol=atol, err_msg=f"case {ci} y")
        (y * _t(g[p + "dy"])).sum().backward()
        for t, nm in zip([x] + ws, ("dx", "d_wqkv", "d_bqkv", "d_wproj", "d_bproj")):
            ref = g[p + nm]
            np.testing.assert_allclose(t.grad.cpu().numpy(), ref, rtol=rtol, atol=atol * max(1.0, np.abs(ref).max()),
                                       err_msg=f"case {ci} {nm}")


@pytest.mark.parametrize("N,d,H", [(256, 512, 4), (64, 384, 8), (100, 1280, 4), (300, 256, 4), (256, 1280, 4), (300, 1280, 4),
                                   # wide heads beyond one key tile per wave (round 4): the [q ; k] token sets of the MoCoAtt variants with --head None
                                   (512, 1280, 4), (600, 2048, 4), (1000, 640, 2)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_mha_vs_oracle(ops, N, d, H, prec):
    if prec == "bf16" and d // H > 128:
        from moma_amd import _lib
        assert _lib.load().moma_mha_saved_state(N, d, H, 1) == _lib.MHA_SAVE_LSE       # the fast path takes it (no [H,N,N])
    rng = np.random.default_rng(N + d)
    x = O.l2_normalize(rng.standard_normal((N, d)).astype(np.float32))
    bound = 1.0 / np.sqrt(d)
    w_qkv = rng.uniform(-bound, bound, (3 * d, d)).astype(np.float32) * 4
    b_qkv = rng.uniform(-bound, bound, 3 * d).astype(np.float32)
    w_proj = rng.uniform(-bound, bound, (d, d)).astype(np.float32)
    b_proj = rng.uniform(-bound, bound, d).astype(np.float32)
    dy = rng.standard_normal((N, d)).astype(np.float32)
    ref = O.attention_bwd(x, w_qkv, b_qkv, w_proj, b_proj, H, dy, dtype=np.float64)
    tx = _t(x).requires_grad_(True)
    tw = [_t(a).requires_grad_(True) for a in (w_qkv, b_qkv, w_proj, b_proj)]
    y = ops.mha(tx, *tw, H, prec)
    (y * _t(dy)).sum().backward()
    tol = 3e-5 if prec == "fp32" else 3e-2
    for t, nm in zip([y.detach()] + [a.grad for a in [tx] + tw], ("y", "dx", "d_wqkv", "d_bqkv", "d_wproj", "d_bproj")):
        r = ref[nm]
        err = np.abs(t.cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-12)
        assert err < tol, (nm, err)


@pytest.mark.parametrize("N,d,H", [(256, 512, 4), (1, 64, 4), (33, 128, 8), (129, 256, 2), (1000, 512, 4), (77, 192, 4),
                                   # wide heads (`--head None`: 1280 / 4 = 320, 2048 / 4 = 512; 160 = one full + one partial segment)
                                   (256, 1280, 4), (200, 2048, 4), (37, 320, 2), (256, 1024, 1)])
def test_mha_fused_core_no_grad(ops, N, d, H):
    """bf16 policy, head dim multiple of 16 and <= 128: the per-head core is one fused launch that keeps the row
    log-sum-exp for the backward and never materialises the probabilities (no [H,N,N] at the ABI).  The no-grad call and
    the grad-mode call agree BIT FOR BIT (no atomics anywhere in the module), and both match the oracle."""
    from moma_amd import _lib
    assert _lib.load().moma_mha_saved_state(N, d, H, 1) == _lib.MHA_SAVE_LSE
    assert _lib.load().moma_mha_saved_state(N, d, H, 0) == _lib.MHA_SAVE_PROBS   # exact-fp32 policy keeps the staged path
    rng = np.random.default_rng(N * 7 + d)
    x = O.l2_normalize(rng.standard_normal((N, d)).astype(np.float32))
    bound = 1.0 / np.sqrt(d)
    # large q/k weights -> peaked softmax rows, so a wrong key <-> value pairing cannot hide behind near-uniform rows
    w_qkv = rng.uniform(-bound, bound, (3 * d, d)).astype(np.float32) * 6
    w_qkv[:2 * d] *= 8
    b_qkv = rng.uniform(-bound, bound, 3 * d).astype(np.float32)
    w_proj = rng.uniform(-bound, bound, (d, d)).astype(np.float32)
    b_proj = rng.uniform(-bound, bound, d).astype(np.float32)
    ref = O.attention_fwd(x, w_qkv, b_qkv, w_proj, b_proj, H, dtype=np.float64)
    ref_y = ref[0] if isinstance(ref, tuple) else ref
    tw = [_t(a).requires_grad_(True) for a in (w_qkv, b_qkv, w_proj, b_proj)]
    with torch.no_grad():
        y0 = ops.mha(_t(x), *tw, H, "bf16")
    y1 = ops.mha(_t(x), *tw, H, "bf16")
    assert torch.equal(y0, y1.detach())
    assert np.abs(y1.detach().cpu().numpy() - ref_y).max() / np.abs(ref_y).max() < 3e-2
    err = np.abs(y0.cpu().numpy() - ref_y).max() / np.abs(ref_y).max()
    assert err < 3e-2, err


@pytest.mark.parametrize("N,d,H", [(256, 512, 4), (33, 128, 8), (129, 256, 2), (300, 512, 4), (77, 192, 4), (64, 64, 4),
                                   (256, 1280, 4), (130, 2048, 4), (37, 320, 2), (250, 288, 1),
                                   # wide heads whose last segment is a single 16-column k-step (144 = 128 + 16, 272 = 256 + 16)
                                   (100, 288, 2), (64, 1088, 4)])
def test_mha_fused_bwd_peaked(ops, N, d, H):
    """Fused per-head backward core (bf16 policy): peaked softmax rows (large q / k weights) so that a wrong pairing of
    tile rows between the two MFMA products of dQ / dK / dV cannot hide behind near-uniform attention; ragged N, head
    dims 16 ... 128; all five parameter gradients and dx against the fp64 oracle."""
    rng = np.random.default_rng(N * 3 + d)
    x = O.l2_normalize(rng.standard_normal((N, d)).astype(np.float32))
    bound = 1.0 / np.sqrt(d)
    w_qkv = rng.uniform(-bound, bound, (3 * d, d)).astype(np.float32) * 4
    w_qkv[:2 * d] *= 5
    b_qkv = rng.uniform(-bound, bound, 3 * d).astype(np.float32)
    w_proj = rng.uniform(-bound, bound, (d, d)).astype(np.float32)
    b_proj = rng.uniform(-bound, bound, d).astype(np.float32)
    dy = rng.standard_normal((N, d)).astype(np.float32)
    ref = O.attention_bwd(x, w_qkv, b_qkv, w_proj, b_proj, H, dy, dtype=np.float64)
    tx = _t(x).requires_grad_(True)
    tw = [_t(a).requires_grad_(True) for a in (w_qkv, b_qkv, w_proj, b_proj)]
    y = ops.mha(tx, *tw, H, "bf16")
    (y * _t(dy)).sum().backward()
    for t, nm in zip([y.detach()] + [a.grad for a in [tx] + tw], ("y", "dx", "d_wqkv", "d_bqkv", "d_wproj", "d_bproj")):
        r = ref[nm]
        err = np.abs(t.cpu().numpy() - r).max() / max(np.abs(r).max(), 1e-12)
        assert err < 4e-2, (nm, err)


@pytest.mark.parametrize("N,d,H", [(256, 512, 4), (100, 192, 4), (300, 128, 8), (256, 1280, 4)])
def test_mha_group_equals_single_calls(ops, N, d, H):
    """Attention.forward_group (the loop's atts_k + atts_queue, helper/loops_moma.py:327-329, as ONE group of three launches)
    returns bit for bit what the modules return one by one; a weight change is picked up by the bf16 pack."""
    from moma_amd.MoMA.criterion_moco_att import Attention
    torch.manual_seed(N + d)
    mods = [Attention(d, num_heads=H, qkv_bias=True, precision="bf16").cuda() for _ in range(3)]
    xs = [torch.nn.functional.normalize(torch.randn(N, d, device="cuda")) for _ in range(3)]
    with torch.no_grad():
        single = [m(x) for m, x in zip(mods, xs)]
        group = Attention.forward_group(mods, xs)
        for a, b in zip(single, group):
            assert torch.equal(a, b)
        mods[1].proj.weight.mul_(0.5)                    # in-place update (as optimizer.step does): the pack must follow
        mods[1].proj.bias.zero_()
        again = Attention.forward_group(mods, xs)
        assert torch.equal(again[0], single[0]) and torch.equal(again[2], single[2])
        
        ref = mods[1](xs[1])
        assert torch.equal(again[1], ref) and not torch.equal(ref, single[1])


@pytest.mark.parametrize("B,d,H", [(256, 512, 4), (100, 128, 4), (8, 256, 2)])
def test_k2_takes_the_query_prepacked_by_k1(ops, B, d, H):
    """atts_q's proj epilogue writes q a second time in the packed bf16 MFMA-operand layout of K2 (moma_mha_module_t.qpack ->
    moma_infonce_fused_q): K2 then runs no pre-pack launch.  Loss, top-1 and the gradient that reaches x are BIT-identical to
    the path where K2 packs q itself; an image made from another tensor is ignored."""
    import copy
    from moma_amd.MoMA.criterion_moco_att import Attention
    from moma_amd.MoMA.mem_moco import MoCo
    torch.manual_seed(B + d)
    att = Attention(d, num_heads=H, qkv_bias=True, precision="bf16").cuda()
    mem_a = MoCo(d, K=4096, T=0.15, queue_dtype=torch.bfloat16, precision="bf16").cuda()
    mem_b = copy.deepcopy(mem_a)
    x = torch.nn.functional.normalize(torch.randn(B, d, device="cuda"))
    k = torch.nn.functional.normalize(x + 0.3 * torch.randn(B, d, device="cuda"))
    xa = x.clone().requires_grad_(True)
    la, acc_a = mem_a.forward_fused(att(xa), k)
    la.backward()
    xb = x.clone().requires_grad_(True)
    qp = mem_b.qpack(B, d, x.device)
    assert qp is not None
    qb = att(xb, qpack=qp)
    assert qp.matches(qb, mem_b.T) and not qp.matches(qb.clone(), mem_b.T)
    lb, acc_b = mem_b.forward_fused(qb, k, qpack=qp)
    lb.backward()
    assert torch.equal(la, lb) and torch.equal(acc_a, acc_b) and torch.equal(xa.grad, xb.grad)
    assert torch.equal(mem_a.memory, mem_b.memory) and mem_a.index == mem_b.index
    # a stale image (made from another q) must not be used
    other = torch.nn.functional.normalize(torch.randn(B, d, device="cuda"))
    l1, _ = copy.deepcopy(mem_a).forward_fused(other, k, qpack=qp)
    l2, _ = copy.deepcopy(mem_a).forward_fused(other, k)
    assert torch.equal(l1, l2)


def test_mha_fast_path_takes_bf16_input_and_follows_raw_pointer_weight_updates(ops):
    """(a) A bf16 x (the output of a head under bf16 autocast) is consumed as it stands: y, dW equal the fp32-x call on the same
    values bit for bit, dx comes back in x's dtype.  (b) K4 writes EMA weights through raw pointers (no autograd version bump):
    ContrastTrainer.momentum_update invalidates the bf16 weight packs of the attention modules it touches."""
    from moma_amd.MoMA.criterion_moco_att import Attention
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    torch.manual_seed(21)
    N, d, H = 200, 256, 4
    att = Attention(d, num_heads=H, qkv_bias=True, precision="bf16").cuda()
    xb = torch.nn.functional.normalize(torch.randn(N, d, device="cuda")).to(torch.bfloat16)
    x32 = xb.float().requires_grad_(True)
    xb = xb.requires_grad_(True)
    dy = torch.randn(N, d, device="cuda")
    y32 = att(x32); (y32 * dy).sum().backward()
    g32 = [p.grad.clone() for p in att.parameters()]
    for p in att.parameters():
        p.grad = None
    y16 = att(xb); (y16 * dy).sum().backward()
    assert torch.equal(y16, y32) and xb.grad.dtype == torch.bfloat16
    assert torch.equal(xb.grad, x32.grad.to(torch.bfloat16))
    for a, b in zip(g32, [p.grad for p in att.parameters()]):
        assert torch.equal(a, b)
    # (b)
    ema = Attention(d, num_heads=H, qkv_bias=True, precision="bf16").cuda()
    with torch.no_grad():
        before = ema(x32.detach())
        ContrastTrainer.momentum_update(att, ema, 0.0)           # m = 0: ema <- att, written by the K4 kernel
        after = ema(x32.detach())
    assert not torch.equal(before, after) and torch.equal(after, y32.detach())


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("N,d,H,qkv_bias", [(256, 512, 4, True), (100, 256, 4, False), (300, 1280, 4, True), (64, 384, 8, False)])
def test_mha_partial_gradient_sets(ops, N, d, H, qkv_bias, prec):
    """Backward with only SOME inputs wanting a gradient (a frozen attention module fed by a trainable encoder: dx alone; a bias
    alone -- it rides on its weight gradient's launch --; everything but dx: the first module of a chain), with and without the
    qkv bias (the reference's Attention has none: MoMA/criterion_moco_att.py:141-151): every gradient that is asked for equals the
    one of the all-gradients run bit for bit, the others stay None."""
    rng = np.random.default_rng(N + d)
    x0 = O.l2_normalize(rng.standard_normal((N, d)).astype(np.float32))
    bound = 1.0 / np.sqrt(d)
    w0 = [rng.uniform(-bound, bound, shp).astype(np.float32) for shp in ((3 * d, d), (3 * d,), (d, d), (d,))]
    dy = _t(rng.standard_normal((N, d)).astype(np.float32))

    def run(mask):
        ts = [_t(x0)] + [_t(a) for a in w0]
        if not qkv_bias:
            ts[2] = None
        for t, m in zip(ts, mask):
            if t is not None:
                t.requires_grad_(bool(m))
        y = ops.mha(*ts, H, prec)
        if not any(m and t is not None for t, m in zip(ts, mask)):
            assert not y.requires_grad
            return y.detach(), [None] * 5
        (y * dy).sum().backward()
        return y.detach(), [None if t is None else t.grad for t in ts]

    y_all, g_all = run([1, 1, 1, 1, 1])
    for mask in ([1, 0, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 1], [0, 1, 1, 1, 1], [1, 0, 1, 0, 1],
                 [1, 1, 0, 1, 0], [0, 0, 0, 0, 0]):
        y, g = run(mask)
        assert torch.equal(y, y_all), mask
        for i, (m, a, b) in enumerate(zip(mask, g, g_all)):
            if m and b is not None:
                assert a is not None and torch.equal(a, b), (mask, i)
            else:
                assert a is None, (mask, i)


# ------------------------------------------------------------------------------------------------ ABI
@pytest.mark.parametrize("N,d,H", [(256, 512, 4), (256, 1280, 4)])
def test_mha_bitwise_repeatable_gradients(ops, N, d, H):
    """Two forward + backward runs on the same inputs give bit-identical outputs and weight gradients in both policies
    (round 1 used split-K fp32 atomics in the linears: d_wqkv changed in the last bits from run to run)."""
    rng = np.random.default_rng(5)
    x = O.l2_normalize(rng.standard_normal((N, d)).astype(np.float32))
    bound = 1.0 / np.sqrt(d)
    ws = [rng.uniform(-bound, bound, shp).astype(np.float32) for shp in ((3 * d, d), (3 * d,), (d, d), (d,))]
    dy = rng.standard_normal((N, d)).astype(np.float32)
    for prec in ("bf16", "fp32"):
        outs = []
        for _ in range(2):
            tx = _t(x).requires_grad_(True)
            tw = [_t(a).requires_grad_(True) for a in ws]
            y = ops.mha(tx, *tw, H, prec)
            (y * _t(dy)).sum().backward()
            outs.append([y.detach().clone(), tx.grad.clone()] + [w.grad.clone() for w in tw])
        for a, b in zip(*outs):
            assert torch.equal(a, b), prec


@pytest.mark.parametrize("N,d,H", [(2100, 128, 4), (4200, 512, 4), (777, 192, 4)])
def test_mha_flash_long_sequence_fwd_bwd(ops, N, d, H):
    """Long token sequences (the attn = 'all' / queue-attending variants: N = 2B + K) on the fused core: forward and all five
    gradients against a torch fp32 reference of the reference's op chain (MoMA/criterion_moco_att.py:153-167).  Nothing of size
    [H,N,N] is allocated on the GPU side: the saved state is lse [H,N]."""
    from moma_amd import _lib
    assert _lib.load().moma_mha_saved_state(N, d, H, 1) == _lib.MHA_SAVE_LSE
    assert _lib.load().moma_mha_bwd_fast_workspace_bytes(N, d, H) < (4 * N * d + H * N + 64) * 4 + 256
    g = torch.Generator().manual_seed(N + d)
    x = torch.nn.functional.normalize(torch.randn(N, d, generator=g))
    bound = 1.0 / np.sqrt(d)
    w_qkv = (torch.rand(3 * d, d, generator=g) * 2 - 1) * bound * 4       # peaked-ish softmax rows
    b_qkv = (torch.rand(3 * d, generator=g) * 2 - 1) * bound
    w_proj = (torch.rand(d, d, generator=g) * 2 - 1) * bound
    b_proj = (torch.rand(d, generator=g) * 2 - 1) * bound
    dy = torch.randn(N, d, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (x, w_qkv, b_qkv, w_proj, b_proj)]
    rx, rwq, rbq, rwp, rbp = ref_in
    qkv = torch.nn.functional.linear(rx, rwq, rbq).reshape(N, 3, H, d // H).permute(1, 2, 0, 3)
    att = ((qkv[0] @ qkv[1].transpose(-2, -1)) * (d // H) ** -0.5).softmax(dim=-1)
    ry = torch.nn.functional.linear((att @ qkv[2]).transpose(0, 1).reshape(N, d), rwp, rbp)
    (ry * dy).sum().backward()
    tin = [t.cuda().requires_grad_(True) for t in (x, w_qkv, b_qkv, w_proj, b_proj)]
    y = ops.mha(*tin, H, "bf16")
    (y * dy.cuda()).sum().backward()

    def rel(a, r):
        return (a.detach().cpu() - r.detach()).abs().max().item() / max(r.detach().abs().max().item(), 1e-12)
    assert rel(y, ry) < 3e-2, rel(y, ry)
    for nm, a, r in zip(("dx", "d_wqkv", "d_bqkv", "d_wproj", "d_bproj"), tin, ref_in):
        assert rel(a.grad, r.grad) < 5e-2, (nm, rel(a.grad, r.grad))


def test_mha_at_the_full_queue_length(ops):
    """Maximum size of the batch-token attention: attn = 'all' at the bench's queue, N = 2 B + K = 66048 tokens (reference
    MoMA/mem_moco.py:139-147; its own op chain would materialise 4 x 66048^2 probabilities).  Forward against a blocked fp64
    restatement on the GPU (2048 query rows at a time); backward through directional derivatives of L = sum(y * dy): <dL/dx, v>,
    <dL/dWqkv, V> and <dL/dWproj, U> against central differences of the fp64 forward."""
    N, d, H = 2 * 256 + 65536, 512, 4
    hd = d // H
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.nn.functional.normalize(torch.randn(N, d, device="cuda", generator=g))
    bound = 1.0 / np.sqrt(d)
    w_qkv = (torch.rand(3 * d, d, device="cuda", generator=g) * 2 - 1) * bound * 6        # peaked rows: the softmax matters
    b_qkv = (torch.rand(3 * d, device="cuda", generator=g) * 2 - 1) * bound
    w_proj = (torch.rand(d, d, device="cuda", generator=g) * 2 - 1) * bound
    b_proj = (torch.rand(d, device="cuda", generator=g) * 2 - 1) * bound
    dy = torch.randn(N, d, device="cuda", generator=g) / N ** 0.5

    def ref_forward(x_, wq_, wp_):
        qkv = (x_ @ wq_.T + b_qkv.double()).reshape(N, 3, H, hd)
        q, k, v = (qkv[:, i].permute(1, 0, 2).contiguous() for i in range(3))               # [H, N, hd]
        out = torch.empty(N, d, device="cuda", dtype=torch.float64)
        for r0 in range(0, N, 2048):
            r1 = min(r0 + 2048, N)
            p = torch.softmax(q[:, r0:r1] @ k.transpose(1, 2) * hd ** -0.5, dim=-1)          # [H, rows, N]
            out[r0:r1] = (p @ v).permute(1, 0, 2).reshape(r1 - r0, d)
        return out @ wp_.T + b_proj.double()
    X, WQ, WP = x.double(), w_qkv.double(), w_proj.double()
    y_ref = ref_forward(X, WQ, WP)
    tin = [t.clone().requires_grad_(True) for t in (x, w_qkv, b_qkv, w_proj, b_proj)]
    y = ops.mha(*tin, H, "bf16")
    assert float((y.detach().double() - y_ref).abs().max() / y_ref.abs().max()) < 3e-2
    (y * dy).sum().backward()
    assert all(torch.isfinite(t.grad).all() for t in tin)
    L = lambda x_, wq_, wp_: float((ref_forward(x_, wq_, wp_) * dy.double()).sum())
    v = torch.randn(N, d, device="cuda", generator=g).double()
    V = torch.randn(3 * d, d, device="cuda", generator=g).double() * bound
    U = torch.randn(d, d, device="cuda", generator=g).double() * bound
    eps = 1e-4
    for name, got, fd in (("dx", float((tin[0].grad.double() * v).sum()), (L(X + eps * v, WQ, WP) - L(X - eps * v, WQ, WP)) / (2 * eps)),
                          ("d_wqkv", float((tin[1].grad.double() * V).sum()), (L(X, WQ + eps * V, WP) - L(X, WQ - eps * V, WP)) / (2 * eps)),
                          ("d_wproj", float((tin[3].grad.double() * U).sum()), (L(X, WQ, WP + eps * U) - L(X, WQ, WP - eps * U)) / (2 * eps))):
        assert abs(got - fd) < 5e-2 * max(abs(fd), 1e-3), (name, got, fd)


def test_mocoatt_attn_all_over_large_queue(ops):
    """MoCoAtt.forward(attn='all') (reference MoMA/mem_moco.py:124-126) with K = 8192: one attention over the
    N = 2B + K = 8224 tokens [q ; k ; queue], logits over the attended queue, gradient back to the student query -- G7's case
    at a queue size where a materialised [H,N,N] would be 1 GB per module.  Checked against the torch-CPU restatement that is
    pinned to the reference by G7 (oracle/step_oracle.py:OracleMoCoAtt)."""
    import argparse
    from moma_amd.MoMA.mem_moco import MoCoAtt
    from moma_amd.MoMA.criterion_moco_att import CMO
    from oracle.step_oracle import OracleCMO, OracleMoCoAtt
    torch.manual_seed(17)
    K, d, B = 8192, 128, 16
    opt = argparse.Namespace(head="None", s_dim=d, t_dim=d, feat_dim=d, attn="all", moma_prec="bf16")
    kd = CMO(opt)
    mem = MoCoAtt(d, K, 0.15, precision="bf16")
    okd = OracleCMO("None", d, d, d, attn="all")
    okd.load_state_dict(kd.state_dict())
    omem = OracleMoCoAtt(d, K, 0.15)
    omem.memory.copy_(mem.memory)
    q = torch.nn.functional.normalize(torch.randn(B, d))
    k = torch.nn.functional.normalize(q + 0.2 * torch.randn(B, d))
    w = torch.randn(B, K + 1)
    torch.set_num_threads(min(16, torch.get_num_threads() or 1))
    rq = q.clone().requires_grad_(True)
    rlogits, _ = omem(rq, k, attn="all", criterion_kd=okd)
    (rlogits * w).sum().backward()
    kd, mem = kd.cuda(), mem.cuda()
    tq = q.cuda().requires_grad_(True)
    logits, labels = mem(tq, k.cuda(), attn="all", criterion_kd=kd)
    (logits * w.cuda()).sum().backward()
    assert logits.shape == (B, K + 1) and int(labels.sum()) == 0 and mem.index == B
    err = (logits.detach().cpu() - rlogits.detach()).abs().max().item() / rlogits.detach().abs().max().item()
    assert err < 3e-2, err
    gerr = (tq.grad.cpu() - rq.grad).abs().max().item() / rq.grad.abs().max().item()
    assert gerr < 6e-2, gerr
    np.testing.assert_allclose(mem.memory[:B].cpu().numpy(), omem.memory[:B].numpy(), rtol=0, atol=3e-2 * omem.memory[:B].abs().max().item())
    assert kd.atts.qkv.weight.grad is not None and torch.isfinite(kd.atts.qkv.weight.grad).all()


def test_abi_argument_checks(ops):
    from moma_amd import _lib
    import ctypes as C
    lib = _lib.load()
    assert lib.moma_version() == _lib.ABI_VERSION == 4
    q = torch.zeros(4, 8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.moma_enqueue(None, C.c_void_p(q.data_ptr()), 4, 0, 8, 8, 0, st) == -1
    assert lib.moma_enqueue(C.c_void_p(q.data_ptr()), C.c_void_p(q.data_ptr()), 4, 9, 8, 8, 0, st) == -2
    assert lib.moma_enqueue(C.c_void_p(q.data_ptr()), C.c_void_p(q.data_ptr()), 4, 0, 8, 8, 7, st) == -3
    assert lib.moma_mha_fwd(*([C.c_void_p(q.data_ptr())] * 9), 4, 8, 3, 0, st) == -2
    assert b"workspace" in lib.moma_error_string(-5)
    # operands of the one-pass K2 kernels that start inside a 16-byte vector are refused (-4), not launched
    B, d, K = 64, 512, 4096
    big = torch.zeros(B * d + 8, device="cuda")
    queue = torch.zeros(K, d, device="cuda", dtype=torch.bfloat16)
    outs = [torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda", dtype=torch.int32)]
    ws = torch.zeros(lib.moma_infonce_fused_workspace_bytes(B, d, K, 1, 1), device="cuda", dtype=torch.uint8)
    ok_q = big[:B * d].view(B, d)
    for off, want in ((0, 0), (1, -4), (2, -4), (4, 0)):
        qv = big[off:off + B * d].view(B, d)
        rc = lib.moma_infonce_fused(C.c_void_p(qv.data_ptr()), C.c_void_p(ok_q.data_ptr()), C.c_void_p(queue.data_ptr()), B, d, K, C.c_float(6.0),
                                    *[C.c_void_p(o.data_ptr()) for o in outs], None, C.c_void_p(ws.data_ptr()), ws.numel(), 1, 1, st)
        assert rc == want, (off, rc)
    torch.cuda.synchronize()
    with pytest.raises(_lib.MomaHipError):
        ops.enqueue_(torch.zeros(4, 8), torch.zeros(2, 8), 0)      # CPU tensors are refused, no fallback


# ------------------------------------------------------------------------------------------------ MoCo module
def test_moco_fp32_storage_uses_bf16_mirror(ops):
    """fp32 `memory` (reference storage) + bf16 policy: forward_fused streams a bf16 mirror kept in sync by the
    enqueue; results equal a bf16-stored queue bit for bit, `memory` itself stays the exact fp32 ring buffer."""
    from moma_amd.MoMA.mem_moco import MoCo
    torch.manual_seed(3)
    d, K, B = 256, 4096, 64
    m32 = MoCo(d, K, 0.15, precision="bf16").cuda()
    m16 = MoCo(d, K, 0.15, queue_dtype=torch.bfloat16, precision="bf16").cuda()
    m16.memory.copy_(m32.memory.to(torch.bfloat16))
    ref_mem = m32.memory.cpu().numpy().copy()
    idx = 0
    for step in range(3):
        q = torch.nn.functional.normalize(torch.randn(B, d, device="cuda")).requires_grad_(True)
        q2 = q.detach().clone().requires_grad_(True)
        k = torch.nn.functional.normalize(torch.randn(B, d, device="cuda"))
        l32, a32 = m32.forward_fused(q, k)
        l16, a16 = m16.forward_fused(q2, k)
        l32.backward(); l16.backward()
        assert l32.item() == l16.item() and torch.equal(q.grad, q2.grad)
        O.update_memory(ref_mem, k.cpu().numpy(), idx)
        idx = (idx + B) % K
        assert m32.index == idx and np.array_equal(m32.memory.cpu().numpy(), ref_mem)
        assert torch.equal(m32._shadow, m32.memory.to(torch.bfloat16))
    # an external in-place edit of `memory` invalidates the mirror
    m32.memory.mul_(0.5)
    q = torch.nn.functional.normalize(torch.randn(B, d, device="cuda"))
    m32.forward_fused(q, q)
    assert torch.equal(m32._shadow[B:], m32.memory.to(torch.bfloat16)[B:])


@pytest.mark.parametrize("B,d,K,n,index,qdt,prec,mirror", [
    (256, 512, 65536, 256, 65400, "bf16", "bf16", False),     # the bench shape; the enqueue wraps around the ring
    (64, 512, 16384, 64, 0, "bf16", "bf16", False),           # small-batch kernel (the reference's run-script shape)
    (100, 256, 1000, 100, 990, "bf16", "bf16", True),         # fp32 queue + its bf16 mirror, ragged B, K not a multiple of 32
    (32, 128, 40, 96, 7, "bf16", "bf16", False),              # n = 3B > K: last writer wins
    (256, 1280, 8192, 256, 8000, "bf16", "bf16", False),      # wide rows: the enqueue rides on the wide combine
    (64, 512, 4096, 64, 4090, "fp32", "fp32", False),         # exact fp32 (no combine launch): issued behind the call
    (48, 384, 2048, 0, 5, "bf16", "bf16", False),             # n = 0: K2 alone
])
def test_k2_call_carries_the_enqueue(ops, B, d, K, n, index, qdt, prec, mirror):
    """moma_infonce_fused_enqueue (round 6: K3 inside the combine launch) == moma_infonce_fused_q followed by moma_enqueue(_mirror),
    bit for bit: the loss / lse / top-1 / dq come from the PRE-enqueue queue (MoMA/mem_moco.py:89-99: read old, then enqueue) and the
    queue afterwards holds rows[i] in slot (index + i) mod K (reference :17-27)."""
    torch.manual_seed(B + d + K)
    dev = "cuda"
    q = torch.nn.functional.normalize(torch.randn(B, d, device=dev))
    k = torch.nn.functional.normalize(q + 0.3 * torch.randn(B, d, device=dev))
    rows = torch.nn.functional.normalize(torch.randn(max(n, 1), d, device=dev))[:n].contiguous()
    q32 = torch.nn.functional.normalize(torch.randn(K, d, device=dev))
    qstore = q32.to(torch.bfloat16) if qdt == "bf16" else q32.clone()

    def run(fused):
        qq = q.clone().requires_grad_(True)
        store = qstore.clone()
        full = q32.clone() if mirror else None
        if fused:
            loss_rows, lse, top1 = ops.infonce_fused(qq, k, store, 0.15, prec, enq=(rows, index, full))
        else:
            loss_rows, lse, top1 = ops.infonce_fused(qq, k, store, 0.15, prec)
            if n:
                if mirror:
                    ops.enqueue_mirror_(full, store, rows, index)
                else:
                    ops.enqueue_(store, rows, index)
        loss_rows.sum().backward()
        torch.cuda.synchronize()
        return loss_rows.detach(), lse, top1, qq.grad, store, full
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert (x is None and y is None) or torch.equal(x, y)
    # ... and against the oracle's ring buffer (fp32 rows; bf16 storage rounds them)
    ref = q32.cpu().numpy().copy()
    if n:
        O.update_memory(ref, rows.cpu().numpy(), index)
    want = torch.from_numpy(ref).to(dev)
    assert torch.equal(a[4], want.to(a[4].dtype))
    if mirror:
        assert torch.equal(a[5], want)


def test_dual_queue_memories_golden(ops, golden_dir):
    """MoCoST / MoCoSSTT (reference MoMA/mem_moco.py:165-253): logits vs the reference, queues + pointer bit-exact."""
    from moma_amd.MoMA.mem_moco import MoCoST, MoCoSSTT
    g = _g(golden_dir, "g6_dual_queue.npz")
    for ci, cls in enumerate([MoCoST, MoCoSSTT]):
        p = f"c{ci}_"
        K, d, B = [int(v) for v in g[p + "cfg"]]
        mem = cls(d, K, 0.15).cuda()
        mem.memory_s.copy_(_t(g[p + "ms0"])); mem.memory_t.copy_(_t(g[p + "mt0"]))
        for s in range(3):
            q, k, kt, qt = [_t(g[p + f"s{s}_{nm}"]) for nm in ("q", "k", "kt", "qt")]
            res = mem(q, k, kt) if cls is MoCoST else mem(q, k, q_t=qt, k_t=kt)
            assert len(res) == (3 if cls is MoCoST else 5) and res[-1].dtype == torch.long and int(res[-1].sum()) == 0
            for j, t in enumerate(res[:-1]):
                np.testing.assert_allclose(t.cpu().numpy(), g[p + f"s{s}_logits{j}"], rtol=2e-5, atol=2e-5)
            assert np.array_equal(mem.memory_s.cpu().numpy(), g[p + f"s{s}_ms"])
            assert np.array_equal(mem.memory_t.cpu().numpy(), g[p + f"s{s}_mt"])
            assert mem.index == int(g[p + f"s{s}_index"])
    # fused form: two one-pass terms, gradient flows to q
    mem = MoCoST(64, 512, 0.15).cuda()
    q = torch.nn.functional.normalize(torch.randn(8, 64, device="cuda")).requires_grad_(True)
    k = torch.nn.functional.normalize(torch.randn(8, 64, device="cuda"))
    (l1, l2), _ = mem.forward_fused(q, k, k.flip(0))
    (l1 + l2).backward()
    assert torch.isfinite(q.grad).all() and mem.index == 8


@pytest.mark.parametrize("cls_name,B,d,K", [("MoCoST", 256, 512, 65536), ("MoCoSSTT", 256, 512, 16384), ("MoCoSSTT", 100, 128, 4100),
                                             ("MoCoST", 8, 256, 96)])
def test_dual_queue_memories_one_sweep(ops, cls_name, B, d, K):
    """n2 as specified: MoCoST / MoCoSSTT.forward_fused run their 2 / 4 InfoNCE terms in ONE sweep (moma_infonce_fused_multi:
    one pre-pack over the distinct queries, one launch of the one-pass kernel over both queues, one combine).  Per-term loss
    and the gradients that reach q / q_t against (a) CrossEntropy over the materialised logits of the reference call sequence
    (MoMA/mem_moco.py:165-253, exact-fp32 kernels, pinned by G6) and (b) one single-term call per term; queues + pointer
    bit-equal to the reference sequence's."""
    import copy
    from moma_amd import _lib
    from moma_amd.MoMA import mem_moco
    torch.manual_seed(B + K)
    cls = getattr(mem_moco, cls_name)
    n_terms = 2 if cls_name == "MoCoST" else 4
    assert _lib.load().moma_infonce_fused_multi_workspace_bytes(n_terms, B, d, K, 1, 1) > 0
    mem = cls(d, K, 0.15, queue_dtype=torch.bfloat16, precision="bf16").cuda()
    ref = cls(d, K, 0.15, precision="fp32").cuda()
    ref.memory_s.copy_(mem.memory_s.float()); ref.memory_t.copy_(mem.memory_t.float())
    single = copy.deepcopy(mem)
    nrm = torch.nn.functional.normalize
    q0, qt0 = nrm(torch.randn(B, d, device="cuda")), nrm(torch.randn(B, d, device="cuda"))
    k = nrm(q0 + 0.5 * torch.randn(B, d, device="cuda"))
    kt = nrm(qt0 + 0.5 * torch.randn(B, d, device="cuda"))

    def leaves():
        return q0.clone().requires_grad_(True), qt0.clone().requires_grad_(True)
    # (a) the reference call sequence on exact-fp32 kernels
    q, qt = leaves()
    out = ref(q, k, kt) if cls_name == "MoCoST" else ref(q, k, q_t=qt, k_t=kt)
    ref_losses = [torch.nn.functional.cross_entropy(lg, out[-1]) for lg in out[:-1]]
    sum(ref_losses).backward()
    ref_gq, ref_gqt = q.grad.clone(), (qt.grad.clone() if qt.grad is not None else None)
    # the one-sweep form
    q, qt = leaves()
    losses, accs = mem.forward_fused(q, k, kt) if cls_name == "MoCoST" else mem.forward_fused(q, k, q_t=qt, k_t=kt)
    assert len(losses) == n_terms and len(accs) == n_terms
    sum(losses).backward()
    for a, b in zip(losses, ref_losses):
        assert abs(a.item() - b.item()) < 1e-3 * max(1.0, abs(b.item())), (a.item(), b.item())
    assert (q.grad - ref_gq).abs().max().item() < 3e-2 * ref_gq.abs().max().item()
    if ref_gqt is not None:
        assert (qt.grad - ref_gqt).abs().max().item() < 3e-2 * ref_gqt.abs().max().item()
    assert mem.index == ref.index
    assert torch.equal(mem.memory_s, ref.memory_s.to(torch.bfloat16)) and torch.equal(mem.memory_t, ref.memory_t.to(torch.bfloat16))
    # (b) one single-term call per term on the same (pre-enqueue) queues: same kernels, another key-chunk split
    q2, qt2 = leaves()
    pairs = [(q2, k, single.memory_s), (q2, kt, single.memory_t)] + ([(qt2, k, single.memory_s), (qt2, kt, single.memory_t)] if n_terms == 4 else [])
    s_losses = [ops.infonce_fused(a, b, c, 0.15, "bf16")[0].mean() for a, b, c in pairs]
    sum(s_losses).backward()
    for a, b in zip(losses, s_losses):
        assert abs(a.item() - b.item()) < 2e-5 * max(1.0, abs(b.item()))
    assert (q.grad - q2.grad).abs().max().item() < 1e-3 * q2.grad.abs().max().item()
    # (c) the oracle itself (oracle.moco_dual_forward -> infonce_loss / infonce_grad: MoMA/mem_moco.py:165-253 + CrossEntropy, numpy)
    # on the pre-enqueue queues -- the anchor to the reference without any other product path in between
    N = lambda t: t.detach().float().cpu().numpy()
    o_ms, o_mt = N(single.memory_s), N(single.memory_t)
    pre_s, pre_t = o_ms.copy(), o_mt.copy()
    outs, _labels, o_index = O.moco_dual_forward(o_ms, o_mt, 0, N(q0), N(k), N(kt), 0.15, q_t=N(qt0) if n_terms == 4 else None)
    assert len(outs) == n_terms and o_index == mem.index
    for a, lg in zip(losses, outs):
        b = O.infonce_loss(lg)["loss"]
        assert abs(a.item() - b) < 1e-3 * max(1.0, abs(b)), (a.item(), b)
    o_gq = O.infonce_grad(N(q0), N(k), pre_s, 0.15) + O.infonce_grad(N(q0), N(kt), pre_t, 0.15)
    assert np.abs(N(q.grad) - o_gq).max() < 2e-2 * np.abs(o_gq).max()
    if n_terms == 4:
        o_gqt = O.infonce_grad(N(qt0), N(k), pre_s, 0.15) + O.infonce_grad(N(qt0), N(kt), pre_t, 0.15)
        assert np.abs(N(qt.grad) - o_gqt).max() < 2e-2 * np.abs(o_gqt).max()
    # the oracle's enqueued queues (fp32 rows) rounded to the product's bf16 storage: same rows in the same slots
    assert torch.equal(mem.memory_s.cpu(), torch.from_numpy(o_ms).to(torch.bfloat16))
    assert torch.equal(mem.memory_t.cpu(), torch.from_numpy(o_mt).to(torch.bfloat16))


@pytest.mark.parametrize("cls_name,B,d,K,prec,qdt", [("MoCoST", 256, 1280, 16384, "bf16", "bf16"),      # --head None: wide rows, two passes per term
                                                      ("MoCoSSTT", 100, 768, 4100, "bf16", "bf16"),
                                                      ("MoCoSSTT", 64, 512, 8192, "fp32", "fp32"),      # the reference's arithmetic: exact-fp32 one pass
                                                      ("MoCoST", 40, 1280, 4096, "fp32", "fp32")])
def test_dual_queue_memories_through_one_call_beyond_the_one_sweep_kernel(ops, cls_name, B, d, K, prec, qdt):
    """moma_infonce_fused_multi serves EVERY configuration moma_infonce_fused takes (VERDICT r3 missing #3): where no one-sweep
    kernel exists (d > 512, exact fp32) the library runs the terms one after the other through one workspace.  Per-term loss and
    the gradients reaching q / q_t against the numpy oracle (oracle.moco_dual_forward -> infonce_loss / infonce_grad:
    MoMA/mem_moco.py:165-253 + CrossEntropy); queues and pointer as the oracle leaves them."""
    from moma_amd import _lib
    from moma_amd.MoMA import mem_moco
    torch.manual_seed(B + K + d)
    cls = getattr(mem_moco, cls_name)
    n_terms = 2 if cls_name == "MoCoST" else 4
    lib = _lib.load()
    pc, qd = (1, 1) if prec == "bf16" else (0, 0)
    need = lib.moma_infonce_fused_multi_workspace_bytes(n_terms, B, d, K, qd, pc)
    assert need == lib.moma_infonce_fused_workspace_bytes(B, d, K, qd, pc) > 0          # one single-term workspace, shared
    dt = torch.bfloat16 if qdt == "bf16" else torch.float32
    mem = cls(d, K, 0.15, queue_dtype=dt, precision=prec).cuda()
    nrm = torch.nn.functional.normalize
    q0, qt0 = nrm(torch.randn(B, d, device="cuda")), nrm(torch.randn(B, d, device="cuda"))
    k = nrm(q0 + 0.5 * torch.randn(B, d, device="cuda"))
    kt = nrm(qt0 + 0.5 * torch.randn(B, d, device="cuda"))
    N = lambda t: t.detach().float().cpu().numpy()
    o_ms, o_mt = N(mem.memory_s), N(mem.memory_t)
    pre_s, pre_t = o_ms.copy(), o_mt.copy()
    q, qt = q0.clone().requires_grad_(True), qt0.clone().requires_grad_(True)
    calls = []
    real = lib.moma_infonce_fused_multi
    try:                                     # (the wrapper must go through the library's multi entry, not through per-term calls)
        def spy(*a):
            calls.append(a[1])
            return real(*a)
        lib.moma_infonce_fused_multi = spy
        losses, accs = mem.forward_fused(q, k, kt) if cls_name == "MoCoST" else mem.forward_fused(q, k, q_t=qt, k_t=kt)
    finally:
        lib.moma_infonce_fused_multi = real
    assert calls == [n_terms] and len(losses) == n_terms and len(accs) == n_terms
    sum(losses).backward()
    outs, _labels, o_index = O.moco_dual_forward(o_ms, o_mt, 0, N(q0), N(k), N(kt), 0.15, q_t=N(qt0) if n_terms == 4 else None)
    assert o_index == mem.index
    tol_l, tol_g = (1e-3, 2e-2) if prec == "bf16" else (2e-5, 2e-4)
    for a, lg in zip(losses, outs):
        b = O.infonce_loss(lg)["loss"]
        assert abs(a.item() - b) < tol_l * max(1.0, abs(b)), (a.item(), b)
    o_gq = O.infonce_grad(N(q0), N(k), pre_s, 0.15) + O.infonce_grad(N(q0), N(kt), pre_t, 0.15)
    assert np.abs(N(q.grad) - o_gq).max() < tol_g * np.abs(o_gq).max()
    if n_terms == 4:
        o_gqt = O.infonce_grad(N(qt0), N(k), pre_s, 0.15) + O.infonce_grad(N(qt0), N(kt), pre_t, 0.15)
        assert np.abs(N(qt.grad) - o_gqt).max() < tol_g * np.abs(o_gqt).max()
    assert torch.equal(mem.memory_s.cpu(), torch.from_numpy(o_ms).to(dt)) and torch.equal(mem.memory_t.cpu(), torch.from_numpy(o_mt).to(dt))


def test_mocoatt_cross_attention_variants_golden(ops, golden_dir):
    """MoCoAtt.forward (reference MoMA/mem_moco.py:103-161): every attn variant against vectors from the reference --
    logits, gradient w.r.t. the student query through the attention modules, enqueued queue, pointer."""
    import argparse
    from moma_amd.MoMA.mem_moco import MoCoAtt
    from moma_amd.MoMA.criterion_moco_att import CMO
    g = _g(golden_dir, "g7_mocoatt.npz")
    K, d, B = 24, 32, 6
    cmo_attn = ["qk", "dual2", "self_qk", "all", "dual", "self"]
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        fw_attn = str(g[p + "attn"])
        opt = argparse.Namespace(head="None", s_dim=d, t_dim=d, feat_dim=d, attn=cmo_attn[ci], moma_prec="fp32")
        kd = CMO(opt)
        kd.load_state_dict({k[len(p) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(p + "kd.")})
        kd = kd.cuda()
        mem = MoCoAtt(d, K, 0.15).cuda()
        mem.memory.copy_(_t(g[p + "mem0"]))
        q = _t(g[p + "q"]).requires_grad_(True)
        logits, labels = mem(q, _t(g[p + "k"]), attn=fw_attn, criterion_kd=kd)
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g[p + "logits"], rtol=2e-4, atol=2e-4, err_msg=fw_attn)
        (logits * _t(g[p + "w"])).sum().backward()
        ref = g[p + "dq"]
        np.testing.assert_allclose(q.grad.cpu().numpy(), ref, rtol=0, atol=3e-4 * max(1.0, np.abs(ref).max()), err_msg=fw_attn)
        np.testing.assert_allclose(mem.memory.cpu().numpy(), g[p + "mem1"], rtol=0, atol=2e-5, err_msg=fw_attn)
        assert mem.index == int(g[p + "index"]) and int(labels.sum()) == 0


# ------------------------------------------------------------------------------------------------ BN + activation
def _bn_ref(x, w, b, rm, rv, training, mom, eps, act, dout):
    """torch reference in fp64 on the (possibly bf16-rounded) input values."""
    xd = x.detach().double().requires_grad_(True)
    wd, bd = w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    rmd, rvd = rm.double().clone(), rv.double().clone()
    y = torch.nn.functional.batch_norm(xd, rmd, rvd, wd, bd, training, mom, eps)
    y = torch.nn.functional.silu(y) if act == "silu" else (torch.relu(y) if act == "relu" else y)
    (y * dout.double()).sum().backward()
    return y.detach(), xd.grad, wd.grad, bd.grad, rmd, rvd


@pytest.mark.parametrize("shape", [(32, 24, 56, 56), (16, 40, 14, 14), (8, 1152, 7, 7), (5, 3, 9, 11), (64, 96, 28, 28)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", [None, "silu", "relu"])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_matches_torch(ops, shape, dtype, act, training):
    """Fused BatchNorm2d + activation (bn.hip) against torch's batch_norm + activation in fp64: output, dx, dgamma,
    dbeta and the running statistics; vector widths 8 / 4 / 1 (HW = 3136, 196, 49, 99, 784), both dtypes, both modes."""
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + (7 if training else 0))
    N, Cc = shape[0], shape[1]
    x = (torch.randn(shape, generator=g) * torch.linspace(0.5, 3.0, Cc).view(1, Cc, 1, 1)
         + torch.linspace(-2.0, 2.0, Cc).view(1, Cc, 1, 1)).cuda().to(dtype)
    w = (1.0 + 0.3 * torch.randn(Cc, generator=g)).cuda()
    b = (0.2 * torch.randn(Cc, generator=g)).cuda()
    rm = (0.1 * torch.randn(Cc, generator=g)).cuda()
    rv = (1.0 + 0.2 * torch.rand(Cc, generator=g)).cuda()
    dout = torch.randn(shape, generator=g).cuda().to(dtype)
    ref_y, ref_dx, ref_dw, ref_db, ref_rm, ref_rv = _bn_ref(x, w, b, rm, rv, training, 0.01, 1e-3, act, dout)
    xx = x.clone().requires_grad_(True)
    ww, bb = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm2, rv2 = rm.clone(), rv.clone()
    y = ops.bn_act(xx, ww, bb, rm2, rv2, training, 0.01, 1e-3, act)
    assert y.dtype == dtype and y.shape == x.shape
    (y.float() * dout.float()).sum().backward()
    # fp32: a few ulp of the fp32 pipeline; bf16: one output rounding (2^-9 relative) on top
    rtol, atol = (2e-5, 2e-5) if dtype == torch.float32 else (8e-3, 8e-3)
    torch.testing.assert_close(y.double(), ref_y, rtol=rtol, atol=atol)
    torch.testing.assert_close(xx.grad.double(), ref_dx, rtol=rtol, atol=atol * max(1.0, ref_dx.abs().max().item()))
    n_el = x.numel() // Cc
    torch.testing.assert_close(ww.grad.double(), ref_dw, rtol=1e-4, atol=2e-5 * n_el ** 0.5 * 10)
    torch.testing.assert_close(bb.grad.double(), ref_db, rtol=1e-4, atol=2e-5 * n_el ** 0.5 * 10)
    torch.testing.assert_close(rm2.double(), ref_rm, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv2.double(), ref_rv, rtol=1e-5, atol=1e-6)
    if not training:
        assert torch.equal(rm2, rm) and torch.equal(rv2, rv)


# ------------------------------------------------------------------------------------------------ depthwise conv
@pytest.mark.parametrize("N,C,H,W,K,S", [(4, 8, 112, 112, 3, 1), (2, 6, 112, 112, 3, 2), (3, 5, 56, 56, 5, 2),
                                         (2, 16, 28, 28, 5, 1), (2, 7, 14, 14, 3, 1), (3, 9, 14, 14, 5, 2),
                                         (2, 11, 7, 7, 5, 1), (2, 3, 9, 13, 3, 2), (1, 2, 33, 17, 5, 1),
                                         (3, 4, 30, 21, 5, 2), (64, 32, 28, 28, 5, 1), (96, 4, 57, 57, 3, 2),
                                         # small planes (7 x 7 / 14 x 14 outputs): one output row per lane, ragged plane groups
                                         (5, 13, 14, 14, 5, 1), (3, 21, 7, 7, 3, 1), (2, 6, 28, 28, 3, 2), (7, 5, 14, 14, 3, 2),
                                         (33, 3, 7, 7, 5, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv_matches_torch(ops, N, C, H, W, K, S, dtype):
    """Depthwise conv (dwconv.hip) with TF-SAME padding (asymmetric when the total is odd) against torch conv2d in
    fp64 on the explicitly padded input: forward, dx, dw.  Covers both kernels x strides, band splits, odd sizes,
    strip heights R = 4 / 2 / 1 and the split reduction of the weight gradient."""
    import math
    g = torch.Generator(device="cpu").manual_seed(N + C + H + K + S)
    x = torch.randn(N, C, H, W, generator=g).cuda().to(dtype)
    w = (torch.randn(C, 1, K, K, generator=g) / K).cuda()
    OH, OW = math.ceil(H / S), math.ceil(W / S)
    ph, pw = max((OH - 1) * S + K - H, 0), max((OW - 1) * S + K - W, 0)
    dy = torch.randn(N, C, OH, OW, generator=g).cuda().to(dtype)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xd, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]), wd, None, S,
                                     0, 1, C)
    (ref * dy.double()).sum().backward()
    xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = ops.dwconv(xx, ww, S, ph // 2, pw // 2, OH, OW)
    assert y.shape == ref.shape and y.dtype == dtype
    (y.float() * dy.float()).sum().backward()
    rtol, atol = (1e-5, 1e-5) if dtype == torch.float32 else (8e-3, 8e-3)
    torch.testing.assert_close(y.double(), ref.detach(), rtol=rtol, atol=atol)
    torch.testing.assert_close(xx.grad.double(), xd.grad, rtol=rtol, atol=atol)
    torch.testing.assert_close(ww.grad.double(), wd.grad, rtol=1e-4, atol=1e-4 * (N * OH * OW) ** 0.5)


def test_backbone_bn_module_state(ops):
    """The backbone's BatchNorm2d (fused kernels) keeps nn.BatchNorm2d's state: running statistics after a few
    training calls, the lazily flushed `num_batches_tracked`, eval-mode output and state-dict keys."""
    from moma_amd.backbones.efficientnet import BatchNorm2d
    torch.manual_seed(3)
    mine = BatchNorm2d(24, momentum=0.01, eps=1e-3).cuda()
    ref = torch.nn.BatchNorm2d(24, momentum=0.01, eps=1e-3).cuda()
    with torch.no_grad():
        mine.weight.uniform_(0.5, 1.5); mine.bias.uniform_(-0.3, 0.3)
        ref.load_state_dict(mine.state_dict())
    for i in range(3):
        x = torch.randn(16, 24, 28, 28, device="cuda") * (1 + i) + i
        y, yr = mine(x, act="silu"), torch.nn.functional.silu(ref(x))
        torch.testing.assert_close(y, yr, rtol=2e-5, atol=2e-5)
    sd, sdr = mine.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(sdr.keys())
    assert int(sd["num_batches_tracked"]) == 3
    torch.testing.assert_close(sd["running_mean"], sdr["running_mean"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sd["running_var"], sdr["running_var"], rtol=1e-5, atol=1e-6)
    mine.eval(); ref.eval()
    x = torch.randn(4, 24, 28, 28, device="cuda")
    torch.testing.assert_close(mine(x), ref(x), rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------------------------------------ squeeze-excite
@pytest.mark.parametrize("shape", [(8, 96, 56, 56), (4, 40, 14, 14), (3, 1152, 7, 7), (2, 5, 9, 11), (16, 24, 28, 28)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_se_gate_and_plane_mean_match_torch(ops, shape, dtype):
    """se.hip: per-plane mean and x * sigmoid(s) with the one-pass backward (dx and ds), chained the way the MBConv
    block uses them (the gate logits depend on the plane means), against torch in fp64."""
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).cuda().to(dtype)
    wmix = torch.randn(shape[1], shape[1], generator=g).cuda() / shape[1] ** 0.5
    dout = torch.randn(shape, generator=g).cuda().to(dtype)

    def block(xx, mean_fn, gate_fn, dt):
        m = mean_fn(xx)                                                    # [N,C,1,1]
        s = torch.einsum("oc,nchw->nohw", wmix.to(dt), m.to(dt)) * 3.0     # stand-in for the two 1x1 convs
        return gate_fn(xx, s.to(xx.dtype))

    xd = x.double().requires_grad_(True)
    ref = block(xd, lambda t: t.mean((2, 3), keepdim=True), lambda t, s: torch.sigmoid(s) * t, torch.float64)
    (ref * dout.double()).sum().backward()
    xx = x.clone().requires_grad_(True)
    y = block(xx, ops.plane_mean, ops.se_gate, torch.float32)
    assert y.dtype == dtype
    (y.float() * dout.float()).sum().backward()
    rtol, atol = (2e-5, 2e-5) if dtype == torch.float32 else (1.6e-2, 1.6e-2)
    torch.testing.assert_close(y.double(), ref.detach(), rtol=rtol, atol=atol)
    torch.testing.assert_close(xx.grad.double(), xd.grad, rtol=rtol, atol=atol * max(1.0, xd.grad.abs().max().item()))


@pytest.mark.parametrize("shape", [(8, 24, 56, 56), (6, 40, 14, 14), (4, 96, 7, 7), (3, 5, 9, 11)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_act_with_plane_mean(ops, shape, dtype):
    """BN + SiLU with the squeeze fused in: the second output is mean_hw of the first, and its gradient is folded into the
    BN backward (dout + dmean / HW) -- compared with torch on a loss that uses both outputs."""
    g = torch.Generator(device="cpu").manual_seed(sum(shape) + 3)
    Cc = shape[1]
    x = (torch.randn(shape, generator=g) * 1.5 + 0.3).cuda().to(dtype)
    w = (1.0 + 0.3 * torch.randn(Cc, generator=g)).cuda()
    b = (0.2 * torch.randn(Cc, generator=g)).cuda()
    dout = torch.randn(shape, generator=g).cuda().to(dtype)
    dmean = torch.randn(shape[0], Cc, 1, 1, generator=g).cuda().to(dtype) * 3
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = torch.nn.functional.silu(torch.nn.functional.batch_norm(xd, None, None, wd, bd, True, 0.01, 1e-3))
    mr = yr.mean((2, 3), keepdim=True)
    ((yr * dout.double()).sum() + (mr * dmean.double()).sum()).backward()
    xx, ww, bb = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y, m = ops.bn_act(xx, ww, bb, None, None, True, 0.01, 1e-3, "silu", want_mean=True)
    assert m.shape == (shape[0], Cc, 1, 1) and m.dtype == dtype
    ((y.float() * dout.float()).sum() + (m.float() * dmean.float()).sum()).backward()
    rtol, atol = (2e-5, 2e-5) if dtype == torch.float32 else (1e-2, 1e-2)
    torch.testing.assert_close(y.double(), yr.detach(), rtol=rtol, atol=atol)
    torch.testing.assert_close(m.double(), mr.detach(), rtol=rtol, atol=atol)
    torch.testing.assert_close(xx.grad.double(), xd.grad, rtol=rtol, atol=atol * max(1.0, xd.grad.abs().max().item()))
    n_el = x.numel() // Cc
    torch.testing.assert_close(ww.grad.double(), wd.grad, rtol=1e-3, atol=2e-4 * n_el ** 0.5 * (1 if dtype == torch.float32 else 30))
    torch.testing.assert_close(bb.grad.double(), bd.grad, rtol=1e-3, atol=2e-4 * n_el ** 0.5 * (1 if dtype == torch.float32 else 30))
    # the mean alone (no gradient for the first output) still back-propagates
    xx2 = x.clone().requires_grad_(True)
    _, m2 = ops.bn_act(xx2, w, b, None, None, True, 0.01, 1e-3, "silu", want_mean=True)
    (m2.float() * dmean.float()).sum().backward()
    xd2 = x.double().requires_grad_(True)
    (torch.nn.functional.silu(torch.nn.functional.batch_norm(xd2, None, None, w.double(), b.double(), True, 0.01, 1e-3))
     .mean((2, 3), keepdim=True) * dmean.double()).sum().backward()
    torch.testing.assert_close(xx2.grad.double(), xd2.grad, rtol=rtol, atol=atol * max(1.0, xd2.grad.abs().max().item()))


def test_effnet_weight_cache_is_transparent():
    """MOMA_WCACHE: one multi-tensor bf16 copy of the conv weights per model forward instead of autocast's per-parameter
    cast kernels -- same numbers out, same gradients in the fp32 parameters, copies refreshed after an in-place update."""
    from moma_amd.backbones import efficientnet as E
    torch.manual_seed(0)
    net = E.efficientnet_b0(num_classes=3).cuda().train()
    x = torch.randn(4, 3, 64, 64, device="cuda")

    def run(cache):
        old, E._WCACHE = E._WCACHE, cache
        try:
            torch.manual_seed(1)                               # same drop-connect masks
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                feats, logits = net(x, is_feat=True)
            logits.float().square().sum().backward()
            return logits.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        finally:
            E._WCACHE = old

    y0, g0 = run(False)
    y1, g1 = run(True)
    # (same casts, same kernels -- but the library GEMMs behind the 1x1 convolutions are not bit-reproducible run to run)
    assert (y0.float() - y1.float()).abs().max() <= 2e-2 * y0.float().abs().max()
    assert g0.keys() == g1.keys() and all(g1[n].dtype == torch.float32 for n in g1)
    # MIOpen's bf16 weight gradients use split reductions with atomics and bf16 partial sums: two runs of the SAME code differ
    # by a few bf16 ulps of the largest partial (observed run-to-run spread up to 3.2e-2 of max|g| on the stem convolution), so
    # the bound is that spread -- what the cache could break (stale or wrong copies) shows as O(1) differences
    # Gradients that are mathematically ZERO (the bias of a BatchNorm whose output reaches the loss only through convolution ->
    # train-mode BatchNorm: `_blocks.*._bn2.bias`, max|g| 3e-5 .. 6e-4) are pure rounding noise, different on
    # every run (scripts/diag_wcache.py): each tensor is measured against max(its own scale, 1 % of the largest gradient).
    gmax = max(g.abs().max().item() for g in g0.values())
    for n in g0:
        err = (g1[n] - g0[n]).abs().max().item() / max(g0[n].abs().max().item(), 1e-2 * gmax)
        assert err < 6e-2, (n, err)
    with torch.no_grad():
        net._conv_stem.weight.mul_(2.0)                        # an optimizer step / EMA update changes the masters ...
    y2, _ = run(True)
    assert (y2.float() - y1.float()).abs().max() > 5e-2 * y1.float().abs().max()     # ... and the next forward sees it
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        net.eval()
        assert torch.isfinite(net(x)).all()


@pytest.mark.parametrize("amp", [False, True])
def test_effnet_helpers_match_stock_ops(amp):
    """Whole EfficientNet-B0 forward + backward with the library's BN / depthwise / SE kernels against the same network on stock
    PyTorch-ROCm ops (MIOpen BN, ATen depthwise, pooling + sigmoid + mul): logits, features, running statistics, parameter gradients
    (tests/effnet_stock_compare.py holds the comparisons and their tolerances).  Run in a CHILD process: the stock kernels of the
    reference side aborted once in five full-suite runs of round 6 (SIGABRT inside the stock backward, no message); a child killed
    by a signal is started once more -- the code under test is the other side of the comparison --, a failed comparison is not."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "effnet_stock_compare.py")
    for attempt in (1, 2):
        r = subprocess.run([sys.executable, script, str(int(amp))], capture_output=True, text=True, timeout=900)
        if r.returncode >= 0:
            break
        print(f"effnet_stock_compare.py died with signal {-r.returncode} (attempt {attempt}):\n{r.stderr[-1500:]}")
    assert r.returncode == 0 and "effnet_stock_compare: ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_graphed_teacher_forward_matches_eager():
    """helper/graphs.py: the no-grad forward replayed from a HIP graph gives the eager result, follows in-place weight
    (EMA) updates, keeps updating the BatchNorm running statistics and the host-side batch counters, and falls back to
    eager under grad mode."""
    from moma_amd.backbones import efficientnet as E
    from moma_amd.helper.graphs import GraphedInference
    torch.manual_seed(0)
    ref = E.efficientnet_b0(num_classes=3, drop_connect_rate=0.0, dropout_rate=0.0).cuda().train()
    net = E.efficientnet_b0(num_classes=3, drop_connect_rate=0.0, dropout_rate=0.0).cuda().train()
    net.load_state_dict(ref.state_dict())
    g = GraphedInference(net, warmup=2)
    xs = [torch.randn(8, 3, 64, 64, device="cuda") for _ in range(6)]
    for i, x in enumerate(xs):
        if i == 4:                                         # an EMA-style in-place update between replays
            with torch.no_grad():
                for m in (ref, net):
                    m._conv_stem.weight.mul_(1.5)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            f0, l0 = ref(x, is_feat=True)
            f1, l1 = g(x, is_feat=True)
        torch.testing.assert_close(l1.float(), l0.float(), rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(f1[-1].float(), f0[-1].float(), rtol=2e-2, atol=2e-2)
    assert len(g._graphs) == 1 and g.enabled
    sd0, sd1 = ref.state_dict(), net.state_dict()
    for k in sd0:
        if "running_" in k:
            torch.testing.assert_close(sd1[k], sd0[k], rtol=2e-2, atol=2e-2, msg=k)
        if "num_batches_tracked" in k:
            assert int(sd1[k]) == int(sd0[k]) == 6, k
    y = g(xs[0])                                           # grad mode on -> plain module call
    assert y.requires_grad
