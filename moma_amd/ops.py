"""torch-facing wrappers of the C ABI (include/moma_hip.h): device pointers and the current HIP stream
are handed to libmoma_hip.so; autograd Functions wire the hand-written backward kernels in.

PyTorch is plumbing here (memory, streams, autograd graph); all arithmetic of the hot path happens in
the HIP library.  Every wrapper refuses CPU tensors: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import torch

from . import _lib
from ._lib import DT_BF16, DT_F32, PREC_BF16, PREC_F32, EMA_BLOCK_ELEMS, MomaHipError, check

_PREC = {"fp32": PREC_F32, "f32": PREC_F32, "float32": PREC_F32, PREC_F32: PREC_F32,
         "bf16": PREC_BF16, "bfloat16": PREC_BF16, PREC_BF16: PREC_BF16}


def prec_code(p) -> int:
    try:
        return _PREC[p]
    except KeyError:
        raise ValueError(f"unknown precision {p!r}; use 'fp32' or 'bf16'")


def _qdtype(queue: torch.Tensor) -> int:
    if queue.dtype == torch.float32:
        return DT_F32
    if queue.dtype == torch.bfloat16:
        return DT_BF16
    raise TypeError(f"queue dtype must be float32 or bfloat16, got {queue.dtype}")


def _dev(t: torch.Tensor, name: str, dtype=torch.float32, contiguous=True) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise MomaHipError(f"{name}: the MoMA hot path runs only on the GPU (HIP library); got a "
                           f"{'CPU tensor' if isinstance(t, torch.Tensor) else type(t)}. No CPU fallback exists.")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _dense_pair(p: torch.Tensor, e: torch.Tensor) -> None:
    """Element-wise kernels walk the raw storage: both tensors must be dense with identical strides
    (row-major or channels_last -- the memory format only permutes the walk order)."""
    for t, nm in ((p, "param"), (e, "param_ema")):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise MomaHipError(f"{nm}: the MoMA hot path runs only on the GPU (HIP library). No CPU fallback exists.")
        if t.dtype != torch.float32:
            raise TypeError(f"{nm}: expected float32, got {t.dtype}")
    dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
    if not dense or p.stride() != e.stride():
        raise ValueError("momentum_update: parameter pair must be dense with identical strides "
                         f"(got {p.stride()} vs {e.stride()})")


# Optional instrumentation (bench.py): a callable (name) -> context manager recording HIP events on the
# current stream around a C-ABI call.  None in normal operation.
_EVENT_RECORDER = None


def set_event_recorder(rec) -> None:
    global _EVENT_RECORDER
    _EVENT_RECORDER = rec


# Optional (bench.py): a callable returning a (hipEvent_t, hipEvent_t) pair of raw handles that the library records
# around the dominant kernel of the next moma_infonce_fused call (see moma_infonce_fused_ex).
_KERNEL_EVENTS = None


def set_kernel_event_provider(fn) -> None:
    global _KERNEL_EVENTS
    _KERNEL_EVENTS = fn


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _timed(name):
    return _EVENT_RECORDER(name) if _EVENT_RECORDER is not None else _NullCtx()


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


# ------------------------------------------------------------------------------------------------
# K4 multi-tensor EMA   (learning/contrast_trainer.py:207-211)
# ------------------------------------------------------------------------------------------------
class EmaTable:
    """Device-side pointer table for one (model, model_ema) pair; rebuilt if any tensor moves."""

    def __init__(self, params: Sequence[torch.Tensor], params_ema: Sequence[torch.Tensor]):
        params, params_ema = list(params), list(params_ema)
        # zip() semantics of the reference: stops at the shorter list; shapes must match pairwise
        n = min(len(params), len(params_ema))
        rows, first = [], 0
        self.keys = []
        for p, e in zip(params[:n], params_ema[:n]):
            if p.shape != e.shape:
                raise RuntimeError(f"The size of tensor a {tuple(e.shape)} must match the size of tensor b "
                                   f"{tuple(p.shape)} (momentum_update needs identical architectures)")
            _dense_pair(p, e)
            if e.numel() == 0:
                continue
            rows.append((e.data_ptr(), p.data_ptr(), e.numel(), first))
            first += (e.numel() + EMA_BLOCK_ELEMS - 1) // EMA_BLOCK_ELEMS
            self.keys.append((e.data_ptr(), p.data_ptr(), e.numel()))
        self.n = len(rows)
        self.total_blocks = first
        dev = params[0].device if n else torch.device("cuda")
        self.table = (torch.tensor(rows, dtype=torch.int64).reshape(-1, 4).to(dev) if rows
                      else torch.zeros(0, 4, dtype=torch.int64, device=dev))

    def matches(self, params, params_ema) -> bool:
        keys = [(e.data_ptr(), p.data_ptr(), e.numel()) for p, e in zip(params, params_ema) if e.numel()]
        return keys == self.keys


def ema_update_(table: EmaTable, m: float) -> None:
    """ema <- fma(fl32(1-m), p, ema*fl32(m)) for every tensor of the table, one launch."""
    lib = _lib.load()
    if table.n == 0:
        return
    with _timed("moma_ema_multi"):
        check(lib.moma_ema_multi(_ptr(table.table), table.n, table.total_blocks, float(m), float(1.0 - m), _stream()),
              "moma_ema_multi")


# ------------------------------------------------------------------------------------------------
# K3 ring-buffer enqueue   (MoMA/mem_moco.py:17-27)
# ------------------------------------------------------------------------------------------------
def enqueue_(queue: torch.Tensor, rows: torch.Tensor, index: int) -> None:
    lib = _lib.load()
    _dev(queue, "queue", None); _dev(rows, "rows")
    K, d = queue.shape
    if rows.dim() != 2 or rows.shape[1] != d:
        raise ValueError(f"rows must be [n,{d}], got {tuple(rows.shape)}")
    check(lib.moma_enqueue(_ptr(queue), _ptr(rows), rows.shape[0], int(index), K, d, _qdtype(queue), _stream()),
          "moma_enqueue")


def queue_prefetch(queue: torch.Tensor, stream=None) -> None:
    """Cache hint (moma_queue_prefetch): sweep the queue into the Infinity Cache on `stream` (default: current)."""
    lib = _lib.load()
    _dev(queue, "queue", None)
    st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
    check(lib.moma_queue_prefetch(_ptr(queue), queue.numel() * queue.element_size(), st), "moma_queue_prefetch")


def enqueue_mirror_(queue: torch.Tensor, mirror: torch.Tensor, rows: torch.Tensor, index: int) -> None:
    """fp32 queue + its bf16 mirror, one launch (moma_enqueue_mirror)."""
    lib = _lib.load()
    _dev(queue, "queue"); _dev(mirror, "mirror", torch.bfloat16); _dev(rows, "rows")
    K, d = queue.shape
    if mirror.shape != queue.shape or rows.dim() != 2 or rows.shape[1] != d:
        raise ValueError(f"rows must be [n,{d}] and mirror {tuple(queue.shape)}, got {tuple(rows.shape)} / {tuple(mirror.shape)}")
    check(lib.moma_enqueue_mirror(_ptr(queue), _ptr(mirror), _ptr(rows), rows.shape[0], int(index), K, d, _stream()),
          "moma_enqueue_mirror")


# ------------------------------------------------------------------------------------------------
# K2 InfoNCE
# ------------------------------------------------------------------------------------------------
def _check_qk(q, k, queue):
    _dev(q, "q"); _dev(k, "k"); _dev(queue, "queue", None)
    if q.dim() != 2 or q.shape != k.shape or queue.dim() != 2 or queue.shape[1] != q.shape[1]:
        raise ValueError(f"shape mismatch: q {tuple(q.shape)} k {tuple(k.shape)} queue {tuple(queue.shape)}")


class _InfoNCELogits(torch.autograd.Function):
    """[B,K+1] logits of BaseMoCo._compute_logit (MoMA/mem_moco.py:29-49) with the backward w.r.t. q."""

    @staticmethod
    def forward(ctx, q, k, queue, T, prec):
        lib = _lib.load()
        q = q.contiguous(); k = k.contiguous()
        _check_qk(q, k, queue)
        B, d = q.shape
        K = queue.shape[0]
        out = torch.empty(B, K + 1, device=q.device, dtype=torch.float32)
        check(lib.moma_infonce_logits(_ptr(q), _ptr(k), _ptr(queue), _ptr(out), B, d, K, float(1.0 / T),
                                      _qdtype(queue), prec, _stream()), "moma_infonce_logits")
        ctx.save_for_backward(q, k, queue)
        ctx.T, ctx.prec = T, prec
        return out

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        q, k, queue = ctx.saved_tensors
        dlogits = dlogits.contiguous()
        B, d = k.shape
        K = queue.shape[0]
        dq = dk = dqueue = None
        if ctx.needs_input_grad[0]:
            dq = torch.empty(B, d, device=k.device, dtype=torch.float32)
            check(lib.moma_infonce_logits_bwd(_ptr(dlogits), _ptr(k), _ptr(queue), _ptr(dq), B, d, K,
                                              float(1.0 / ctx.T), _qdtype(queue), ctx.prec, _stream()),
                  "moma_infonce_logits_bwd")
        # k / queue carry gradient only in the MoCoAtt cross-attention variants (they are attention outputs there)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            if ctx.needs_input_grad[1]:
                dk = torch.empty(B, d, device=k.device, dtype=torch.float32)
            if ctx.needs_input_grad[2]:
                if queue.dtype != torch.float32:
                    raise TypeError("a queue that requires grad must be float32")
                dqueue = torch.empty(K, d, device=k.device, dtype=torch.float32)
            check(lib.moma_infonce_logits_bwd_kq(_ptr(dlogits), _ptr(q), _ptr(dk), _ptr(dqueue), B, d, K,
                                                 float(1.0 / ctx.T), ctx.prec, _stream()), "moma_infonce_logits_bwd_kq")
        return dq, dk, dqueue, None, None


def infonce_logits(q, k, queue, T: float, prec="fp32") -> torch.Tensor:
    return _InfoNCELogits.apply(q, k, queue, float(T), prec_code(prec))


class _InfoNCEFused(torch.autograd.Function):
    """One pass over the queue: per-row CE(label 0) loss, lse, top-1 flag and d(sum loss)/dq."""

    @staticmethod
    def forward(ctx, q, k, queue, T, prec):
        lib = _lib.load()
        q = q.contiguous(); k = k.contiguous()
        _check_qk(q, k, queue)
        B, d = q.shape
        K = queue.shape[0]
        dev = q.device
        need_grad = ctx.needs_input_grad[0]
        loss_rows = torch.empty(B, device=dev, dtype=torch.float32)
        lse = torch.empty(B, device=dev, dtype=torch.float32)
        top1 = torch.empty(B, device=dev, dtype=torch.int32)
        dq = torch.empty(B, d, device=dev, dtype=torch.float32) if need_grad else None
        qd = _qdtype(queue)
        ws_bytes = lib.moma_infonce_fused_workspace_bytes(B, d, K, qd, prec)
        ws = torch.empty(max(ws_bytes, 16), device=dev, dtype=torch.uint8)
        ev0, ev1 = _KERNEL_EVENTS() if _KERNEL_EVENTS is not None else (None, None)
        with _timed("moma_infonce_fused"):
            check(lib.moma_infonce_fused_ex(_ptr(q), _ptr(k), _ptr(queue), B, d, K, float(1.0 / T), _ptr(loss_rows),
                                            _ptr(lse), _ptr(top1), _ptr(dq), _ptr(ws), ws.numel(), qd, prec, _stream(),
                                            C.c_void_p(ev0), C.c_void_p(ev1)),
                  "moma_infonce_fused")
        if need_grad:
            ctx.save_for_backward(dq)
        ctx.mark_non_differentiable(lse, top1)
        return loss_rows, lse, top1

    @staticmethod
    def backward(ctx, g_loss, g_lse, g_top1):
        (dq,) = ctx.saved_tensors
        return dq * g_loss.unsqueeze(1), None, None, None, None


def infonce_fused(q, k, queue, T: float, prec="fp32") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (loss_rows [B], lse [B], top1 [B] int32).  loss_kd = loss_rows.mean()."""
    return _InfoNCEFused.apply(q, k, queue, float(T), prec_code(prec))


# ------------------------------------------------------------------------------------------------
# K1 batch-token multi-head attention   (MoMA/criterion_moco_att.py:153-167)
# ------------------------------------------------------------------------------------------------
class _MHA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_qkv, b_qkv, w_proj, b_proj, H, prec, grad_mode):
        lib = _lib.load()
        x = x.contiguous()
        for t, nm in ((x, "x"), (w_qkv, "qkv.weight"), (w_proj, "proj.weight"), (b_proj, "proj.bias")):
            _dev(t, nm)
        if b_qkv is not None:
            _dev(b_qkv, "qkv.bias")
        N, d = x.shape
        if w_qkv.shape != (3 * d, d) or w_proj.shape != (d, d) or d % H:
            raise ValueError(f"bad attention shapes: x {tuple(x.shape)} Wqkv {tuple(w_qkv.shape)} H={H}")
        dev = x.device
        y = torch.empty(N, d, device=dev, dtype=torch.float32)
        qkv = torch.empty(N, 3 * d, device=dev, dtype=torch.float32)
        need_bwd = grad_mode and any(ctx.needs_input_grad)      # (grad mode is always off inside forward)
        # what the backward needs besides qkv / attn_out: the fused per-head core keeps the row log-sum-exp [H,N] (it recomputes
        # P per tile); the staged path (exact fp32, odd head dims) the probabilities [H,N,N] -- which it also computes through
        save_lse = lib.moma_mha_saved_state(N, d, H, prec) == _lib.MHA_SAVE_LSE
        probs = None if save_lse else torch.empty(H, N, N, device=dev, dtype=torch.float32)
        lse = torch.empty(H, N, device=dev, dtype=torch.float32) if (save_lse and need_bwd) else None
        attn_out = torch.empty(N, d, device=dev, dtype=torch.float32)
        with _timed("moma_mha_fwd"):
            check(lib.moma_mha_fwd(_ptr(x), _ptr(w_qkv), _ptr(b_qkv), _ptr(w_proj), _ptr(b_proj), _ptr(y), _ptr(qkv),
                                   _ptr(probs), _ptr(lse), _ptr(attn_out), N, d, H, prec, _stream()), "moma_mha_fwd")
        if need_bwd:
            ctx.save_for_backward(x, w_qkv, w_proj, qkv, lse if save_lse else probs, attn_out)
        ctx.H, ctx.prec, ctx.has_bqkv, ctx.save_lse = H, prec, b_qkv is not None, save_lse
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_qkv, w_proj, qkv, state, attn_out = ctx.saved_tensors
        probs, lse = (None, state) if ctx.save_lse else (state, None)
        dy = dy.contiguous()
        N, d = x.shape
        H = ctx.H
        dev = x.device
        need = ctx.needs_input_grad
        dx = torch.empty_like(x) if need[0] else None
        dw_qkv = torch.empty_like(w_qkv) if need[1] else None
        db_qkv = torch.empty(3 * d, device=dev, dtype=torch.float32) if (need[2] and ctx.has_bqkv) else None
        dw_proj = torch.empty_like(w_proj) if need[3] else None
        db_proj = torch.empty(d, device=dev, dtype=torch.float32) if need[4] else None
        ws_bytes = lib.moma_mha_bwd_workspace_bytes(N, d, H, ctx.prec)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        with _timed("moma_mha_bwd"):
            check(lib.moma_mha_bwd(_ptr(x), _ptr(w_qkv), _ptr(w_proj), _ptr(qkv), _ptr(probs), _ptr(lse), _ptr(attn_out),
                                   _ptr(dy), _ptr(dx), _ptr(dw_qkv), _ptr(db_qkv), _ptr(dw_proj), _ptr(db_proj), _ptr(ws),
                                   ws.numel(), N, d, H, ctx.prec, _stream()), "moma_mha_bwd")
        return dx, dw_qkv, db_qkv, dw_proj, db_proj, None, None, None


def mha(x, w_qkv, b_qkv, w_proj, b_proj, num_heads: int, prec="fp32") -> torch.Tensor:
    return _MHA.apply(x, w_qkv, b_qkv, w_proj, b_proj, int(num_heads), prec_code(prec), torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------
# BatchNorm2d + fused activation on NCHW activations (backbone helper, include/moma_hip.h "BN")
# ------------------------------------------------------------------------------------------------
ACT_CODES = {None: 0, "none": 0, "silu": 1, "relu": 2}
_DT_CODES = {torch.float32: 0, torch.bfloat16: 1}


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, act, want_mean):
        lib = _lib.load()
        _dev(x, "x", dtype=None, contiguous=False)
        if x.dim() < 2 or x.dtype not in _DT_CODES:
            raise ValueError(f"bn_act: expects [N, C, ...] float32 / bfloat16, got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        N, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (N * Cc)
        dev = x.device
        out = torch.empty_like(x)
        save_mean = torch.empty(Cc, device=dev, dtype=torch.float32)
        save_invstd = torch.empty(Cc, device=dev, dtype=torch.float32)
        ws = torch.empty(lib.moma_bn_workspace_bytes(Cc), device=dev, dtype=torch.uint8)
        pmean = torch.empty(N, Cc, 1, 1, device=dev, dtype=x.dtype) if want_mean else None
        check(lib.moma_bn_fwd(_ptr(x), _ptr(out), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                              _ptr(save_mean), _ptr(save_invstd), _ptr(ws), ws.numel(), N, Cc, HW, _DT_CODES[x.dtype],
                              act, int(training), float(momentum), float(eps), _ptr(pmean), _stream()), "moma_bn_fwd")
        ctx.save_for_backward(x, weight, bias, save_mean, save_invstd)
        ctx.cfg = (N, Cc, HW, act, int(training))
        ctx.set_materialize_grads(False)
        return (out, pmean) if want_mean else out

    @staticmethod
    def backward(ctx, dout, dmean=None):
        lib = _lib.load()
        x, weight, bias, save_mean, save_invstd = ctx.saved_tensors
        N, Cc, HW, act, training = ctx.cfg
        if dout is None:
            dout = torch.zeros_like(x)
        dout = dout.contiguous()
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        if dmean is not None:
            dmean = dmean.to(x.dtype).contiguous()
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dgamma = torch.empty(Cc, device=dev, dtype=torch.float32) if (weight is not None and ctx.needs_input_grad[1]) else None
        dbeta = torch.empty(Cc, device=dev, dtype=torch.float32) if (bias is not None and ctx.needs_input_grad[2]) else None
        ws = torch.empty(lib.moma_bn_workspace_bytes(Cc), device=dev, dtype=torch.uint8)
        check(lib.moma_bn_bwd(_ptr(x), _ptr(dout), _ptr(weight), _ptr(bias), _ptr(save_mean), _ptr(save_invstd), _ptr(dx),
                              _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel(), N, Cc, HW, _DT_CODES[x.dtype], act,
                              training, _ptr(dmean), _stream()), "moma_bn_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


def bn_act(x, weight, bias, running_mean, running_var, training: bool, momentum: float, eps: float, act=None,
           want_mean: bool = False):
    """act(batch_norm(x)) on a contiguous NCHW tensor; running statistics are updated in place when training.
    want_mean: also return the [N,C,1,1] per-plane mean of the result (the squeeze of a squeeze-excite block)."""
    return _BNAct.apply(x, weight, bias, running_mean, running_var, bool(training), momentum, eps, ACT_CODES[act],
                        bool(want_mean))


# ------------------------------------------------------------------------------------------------
# Depthwise convolution on NCHW activations (backbone helper, include/moma_hip.h "DW")
# ------------------------------------------------------------------------------------------------
class _DWConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad_top, pad_left, out_h, out_w):
        lib = _lib.load()
        _dev(x, "x", dtype=None, contiguous=False)
        _dev(w, "weight")
        if x.dim() != 4 or x.dtype not in _DT_CODES or w.dim() != 4 or w.shape[1] != 1 or w.shape[0] != x.shape[1] \
                or w.shape[2] != w.shape[3]:
            raise ValueError(f"dwconv: x {tuple(x.shape)} {x.dtype}, weight {tuple(w.shape)}")
        x = x.contiguous()
        w = w.contiguous()
        N, Cc, H, W = x.shape
        K = w.shape[2]
        y = torch.empty(N, Cc, out_h, out_w, device=x.device, dtype=x.dtype)
        check(lib.moma_dwconv_fwd(_ptr(x), _ptr(w), _ptr(y), N, Cc, H, W, out_h, out_w, K, stride, pad_top, pad_left,
                                  _DT_CODES[x.dtype], _stream()), "moma_dwconv_fwd")
        ctx.save_for_backward(x, w)
        ctx.cfg = (N, Cc, H, W, out_h, out_w, K, stride, pad_top, pad_left)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        N, Cc, H, W, OH, OW, K, stride, pt, pl = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib.moma_dwconv_bwd_data(_ptr(dy), _ptr(w), _ptr(dx), N, Cc, H, W, OH, OW, K, stride, pt, pl,
                                           _DT_CODES[x.dtype], _stream()), "moma_dwconv_bwd_data")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            ws = torch.empty(lib.moma_dwconv_workspace_bytes(Cc, K), device=x.device, dtype=torch.uint8)
            check(lib.moma_dwconv_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), ws.numel(), N, Cc, H, W, OH, OW, K,
                                             stride, pt, pl, _DT_CODES[x.dtype], _stream()), "moma_dwconv_bwd_weight")
        return dx, dw, None, None, None, None, None


def dwconv_supported(kernel: int, stride: int) -> bool:
    return kernel in (3, 5) and stride in (1, 2)


def dwconv(x, weight, stride: int, pad_top: int, pad_left: int, out_h: int, out_w: int):
    """Depthwise conv2d, weight [C,1,K,K] fp32, explicit (possibly asymmetric) zero padding given by the top/left pad
    and the output size."""
    return _DWConv.apply(x, weight, int(stride), int(pad_top), int(pad_left), int(out_h), int(out_w))


# ------------------------------------------------------------------------------------------------
# Squeeze-excite helpers (include/moma_hip.h "SE")
# ------------------------------------------------------------------------------------------------
class _PlaneMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        _dev(x, "x", dtype=None, contiguous=False)
        if x.dim() != 4 or x.dtype not in _DT_CODES:
            raise ValueError(f"plane_mean: expects [N,C,H,W] float32 / bfloat16, got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        N, Cc, H, W = x.shape
        out = torch.empty(N, Cc, 1, 1, device=x.device, dtype=x.dtype)
        check(lib.moma_plane_mean(_ptr(x), _ptr(out), N * Cc, H * W, _DT_CODES[x.dtype], _stream()), "moma_plane_mean")
        ctx.shape = (N, Cc, H, W)
        return out

    @staticmethod
    def backward(ctx, dmean):
        N, Cc, H, W = ctx.shape
        return (dmean / (H * W)).expand(N, Cc, H, W)          # a stride-0 view: nothing is materialised here


class _SEGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        lib = _lib.load()
        _dev(x, "x", dtype=None, contiguous=False)
        _dev(s, "s", dtype=None, contiguous=False)
        if x.dim() != 4 or x.dtype not in _DT_CODES or s.numel() != x.shape[0] * x.shape[1]:
            raise ValueError(f"se_gate: x {tuple(x.shape)} {x.dtype}, s {tuple(s.shape)}")
        x = x.contiguous()
        s = s.to(x.dtype).contiguous()
        N, Cc, H, W = x.shape
        out = torch.empty_like(x)
        check(lib.moma_se_gate_fwd(_ptr(x), _ptr(s), _ptr(out), N * Cc, H * W, _DT_CODES[x.dtype], _stream()), "moma_se_gate_fwd")
        ctx.save_for_backward(x, s)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, s = ctx.saved_tensors
        N, Cc, H, W = x.shape
        dout = dout.contiguous()
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        dx = torch.empty_like(x)
        ds = torch.empty_like(s)
        check(lib.moma_se_gate_bwd(_ptr(x), _ptr(s), _ptr(dout), _ptr(dx), _ptr(ds), N * Cc, H * W, _DT_CODES[x.dtype],
                                   _stream()), "moma_se_gate_bwd")
        return dx, ds


def plane_mean(x):
    """[N,C,H,W] -> [N,C,1,1] mean over each plane (F.adaptive_avg_pool2d(x, 1))."""
    return _PlaneMean.apply(x)


def se_gate(x, s):
    """x * sigmoid(s) with s [N,C,1,1]."""
    return _SEGate.apply(x, s)
