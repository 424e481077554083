"""CPU oracle for the MoMA contrastive-distillation hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the reference algorithm (trinhvg/MoMA).  It exists so that
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check / time the HIP path against
it.  Nothing under moma_amd/ may import it: the product path is the HIP library and fails loudly when
that library is missing.

Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py against
fixtures in tests/golden/*.npz that were produced by importing the reference's own modules
(tests/golden/make_golden.py, run in the build container where /root/reference is mounted).

Each function cites the reference file:line it restates (paths relative to the reference root).
All arithmetic is fp32 unless `dtype=np.float64` is requested (used to bound fp32 rounding in tests).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------------------
# MoMA/criterion_moco_att.py:12-18  Normalize  (F.normalize(x, p=2, dim=1), eps=1e-12)
# --------------------------------------------------------------------------------------------------
def l2_normalize(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(x.dtype)
    return x / np.maximum(n, np.asarray(eps, dtype=x.dtype))


# --------------------------------------------------------------------------------------------------
# MoMA/criterion_moco_att.py:141-167  Attention.forward  (batch-token multi-head self attention)
#   x [N,C] -> unsqueeze(0) -> qkv = x Wqkv^T + b -> [3,1,H,N,C/H] -> softmax(q k^T * hd^-1/2) v
#   -> [N,C] -> proj
# --------------------------------------------------------------------------------------------------
def attention_fwd(x, w_qkv, b_qkv, w_proj, b_proj, num_heads, return_cache=False, dtype=F32):
    x = x.astype(dtype)
    w_qkv = w_qkv.astype(dtype)
    w_proj = w_proj.astype(dtype)
    n, c = x.shape
    h = num_heads
    hd = c // h
    scale = dtype(hd ** -0.5)                                   # :146
    qkv = x @ w_qkv.T                                           # :148,156
    if b_qkv is not None:
        qkv = qkv + b_qkv.astype(dtype)
    # reshape(B,N,3,H,hd).permute(2,0,3,1,4) with B=1  -> q,k,v each [H,N,hd]   (:156-157)
    qkv = qkv.reshape(n, 3, h, hd).transpose(1, 2, 0, 3)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(0, 2, 1)) * scale                      # :159
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(axis=-1, keepdims=True)                       # :160 softmax(dim=-1)
    a = (p @ v).transpose(1, 0, 2).reshape(n, c)                # :163
    y = a @ w_proj.T + b_proj.astype(dtype)                     # :164
    if return_cache:
        return y, dict(q=q, k=k, v=v, p=p, a=a, scale=scale)
    return y


def attention_bwd(x, w_qkv, b_qkv, w_proj, b_proj, num_heads, dy, dtype=F32):
    """Gradient of sum(y * dy) w.r.t. x and the four parameters (what autograd produces for
    MoMA/criterion_moco_att.py:153-167; only atts_q ever receives it -- SURVEY Q6)."""
    y, cch = attention_fwd(x, w_qkv, b_qkv, w_proj, b_proj, num_heads, True, dtype)
    x = x.astype(dtype)
    dy = dy.astype(dtype)
    n, c = x.shape
    h = num_heads
    hd = c // h
    q, k, v, p, a, scale = cch["q"], cch["k"], cch["v"], cch["p"], cch["a"], cch["scale"]
    d_wproj = dy.T @ a
    d_bproj = dy.sum(axis=0)
    da = dy @ w_proj.astype(dtype)                              # [N,C]
    da = da.reshape(n, h, hd).transpose(1, 0, 2)                # [H,N,hd]
    dv = p.transpose(0, 2, 1) @ da
    dp = da @ v.transpose(0, 2, 1)
    ds = p * (dp - (dp * p).sum(axis=-1, keepdims=True))
    ds = ds * scale
    dq = ds @ k
    dk = ds.transpose(0, 2, 1) @ q
    dqkv = np.stack([dq, dk, dv], axis=0)                       # [3,H,N,hd]
    dqkv = dqkv.transpose(2, 0, 1, 3).reshape(n, 3 * c)
    d_wqkv = dqkv.T @ x
    d_bqkv = dqkv.sum(axis=0)
    dx = dqkv @ w_qkv.astype(dtype)
    return dict(y=y, dx=dx, d_wqkv=d_wqkv, d_bqkv=d_bqkv, d_wproj=d_wproj, d_bproj=d_bproj)


# --------------------------------------------------------------------------------------------------
# MoMA/mem_moco.py:29-49  BaseMoCo._compute_logit
#   out[b,0] = <q_b,k_b>/T ; out[b,1+j] = <queue_j,q_b>/T     -> [B,K+1]
# (the reference's trailing .squeeze() turns B==1 into a 1-D tensor -- latent bug, not replicated)
# --------------------------------------------------------------------------------------------------
def compute_logit(q, k, queue, T, dtype=F32):
    q = q.astype(dtype)
    k = k.astype(dtype)
    queue = queue.astype(dtype)
    pos = (q * k).sum(axis=1, keepdims=True)                    # :37-38 bmm
    neg = (queue @ q.T).T                                       # :41-42
    out = np.concatenate([pos, neg], axis=1)                    # :44
    return out / dtype(T)                                       # :45


# --------------------------------------------------------------------------------------------------
# helper/loops_moma.py:322,332-335 + learning/contrast_trainer.py:189-205 + learning/util.py:25-41
#   loss_kd = CrossEntropy(logits, zeros) = mean_b(lse_b - logits[b,0]); top-1 accuracy in percent
# --------------------------------------------------------------------------------------------------
def infonce_loss(logits):
    lg = logits.astype(np.float64)
    m = lg.max(axis=1, keepdims=True)
    lse = (m + np.log(np.exp(lg - m).sum(axis=1, keepdims=True)))[:, 0]
    loss_rows = lse - lg[:, 0]
    # torch.topk(1) picks the first maximal index; label 0 is "correct" iff argmax == 0
    top1 = (lg.argmax(axis=1) == 0)
    return dict(loss=loss_rows.mean(), loss_rows=loss_rows, lse=lse,
                top1=top1, acc=100.0 * top1.mean())


def infonce_grad(q, k, queue, T, dtype=np.float64):
    """d loss_kd / d q   (what autograd gives through CE -> div -> cat -> mm/bmm,
    MoMA/mem_moco.py:29-49 with k detached at :86)."""
    lg = compute_logit(q, k, queue, T, dtype=dtype)
    m = lg.max(axis=1, keepdims=True)
    p = np.exp(lg - m)
    p = p / p.sum(axis=1, keepdims=True)
    b = q.shape[0]
    dlog = p.copy()
    dlog[:, 0] -= 1.0
    dlog /= b                                                   # mean reduction
    dq = (dlog[:, :1] * k.astype(dtype) + dlog[:, 1:] @ queue.astype(dtype)) / dtype(T)
    return dq


def logits_bwd(dlogits, k, queue, T, dtype=F32):
    """dq for an arbitrary upstream gradient on the [B,K+1] logits (compat path backward)."""
    dl = dlogits.astype(dtype)
    return (dl[:, :1] * k.astype(dtype) + dl[:, 1:] @ queue.astype(dtype)) / dtype(T)


# --------------------------------------------------------------------------------------------------
# MoMA/mem_moco.py:17-27  BaseMoCo._update_memory   queue[(index+i) mod K] = k[i]
# MoMA/mem_moco.py:14-15  BaseMoCo._update_pointer  index = (index+n) mod K
# index_copy_ with duplicate indices (n > K) is last-writer-wins on CPU; restated as a serial loop.
# --------------------------------------------------------------------------------------------------
def enqueue_ids(index: int, n: int, K: int) -> np.ndarray:
    return np.fmod(np.arange(n, dtype=np.int64) + index, K).astype(np.int64)


def update_memory(queue: np.ndarray, k: np.ndarray, index: int) -> np.ndarray:
    K = queue.shape[0]
    ids = enqueue_ids(index, k.shape[0], K)
    for i, j in enumerate(ids):                                 # serial == last writer wins
        queue[j] = k[i]
    return ids


def update_pointer(index: int, n: int, K: int) -> int:
    return (index + n) % K


# --------------------------------------------------------------------------------------------------
# MoMA/mem_moco.py:77-100  MoCo.forward: logits from the PRE-enqueue queue, labels zeros, then enqueue
# --------------------------------------------------------------------------------------------------
def moco_forward(queue, index, q, k, all_k, T):
    K = queue.shape[0]
    logits = compute_logit(q, k, queue.copy(), T)               # :89 clone().detach(), :90
    labels = np.zeros(q.shape[0], dtype=np.int64)               # :94
    all_k = k if all_k is None else all_k                       # :97
    update_memory(queue, all_k, index)                          # :98
    index = update_pointer(index, all_k.shape[0], K)            # :99
    return logits, labels, index


# --------------------------------------------------------------------------------------------------
# learning/contrast_trainer.py:207-211  momentum_update
#   p2.mul_(m).add_(p1, alpha=1-m) in fp32.  (1-m) is formed in Python double and cast to fp32 by the
#   scalar->tensor-dtype conversion; ATen's add kernel evaluates a + alpha*b as ONE fused multiply-add
#   (checked bit-for-bit against the reference's output, tests/golden/g3_ema.npz), i.e.
#       p2 = fma(fl32(1-m), p1, fl32(p2 * fl32(m)))
#   The fma is emulated exactly here: a product of two fp32 is exact in fp64.
# --------------------------------------------------------------------------------------------------
def momentum_update(params, params_ema, m: float):
    mf = F32(m)
    omf = np.float64(F32(1.0 - m))
    for p, pe in zip(params, params_ema):
        t = (pe.astype(F32) * mf).astype(F32)
        pe[...] = (t.astype(np.float64) + omf * p.astype(np.float64)).astype(F32)


# --------------------------------------------------------------------------------------------------
# distiller_zoo/KD.py:13-17  DistillKL  (KL(log_softmax(ys/T) || softmax(yt/T)) * T^2, batchmean)
# --------------------------------------------------------------------------------------------------
def distill_kl(y_s, y_t, T):
    ys = y_s.astype(np.float64) / T
    yt = y_t.astype(np.float64) / T
    ls = ys - ys.max(1, keepdims=True)
    ls = ls - np.log(np.exp(ls).sum(1, keepdims=True))
    lt = yt - yt.max(1, keepdims=True)
    lt = lt - np.log(np.exp(lt).sum(1, keepdims=True))
    pt = np.exp(lt)
    return float((pt * (lt - ls)).sum() / y_s.shape[0] * T * T)


# --------------------------------------------------------------------------------------------------
# MoMA/mem_moco.py:165-253  MoCoST / MoCoSSTT: two queues sharing one pointer; logits of q (and optionally q_t)
# against (k, memory_s) and (k_t, memory_t) from the PRE-enqueue queues, then both queues are enqueued.
# --------------------------------------------------------------------------------------------------
def moco_dual_forward(mem_s, mem_t, index, q, k, k_t, T, q_t=None, all_k=None, all_k_t=None):
    K = mem_s.shape[0]
    qs, qt = mem_s.copy(), mem_t.copy()
    outs = [compute_logit(q, k, qs, T), compute_logit(q, k_t, qt, T)]
    if q_t is not None:
        outs += [compute_logit(q_t, k, qs, T), compute_logit(q_t, k_t, qt, T)]
    all_k = k if all_k is None else all_k
    all_k_t = k_t if all_k_t is None else all_k_t
    update_memory(mem_s, all_k, index)
    update_memory(mem_t, all_k_t, index)
    return outs, np.zeros(q.shape[0], dtype=np.int64), update_pointer(index, all_k.shape[0], K)
